#pragma once
#include <Eigen/Geometry>
