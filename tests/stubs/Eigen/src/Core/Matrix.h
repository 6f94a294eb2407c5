#pragma once
#include <Eigen/Dense>
