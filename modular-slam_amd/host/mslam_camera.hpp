// mslam_camera.hpp — Vector2 and CameraParameters as the reference declares them (types/basic_types.hpp: Eigen::Vector2d;
// sensors/camera_parameters.hpp:7-12), shared by the interface mirror and the TUM file tools.
#pragma once
#ifdef MSLAM_USE_REFERENCE_HEADERS
#include "modular_slam/sensors/camera_parameters.hpp"
#else
namespace mslam
{
// Eigen::Vector2d stand-in: 16-byte aligned pair of doubles with x()/y() accessors
struct alignas(16) Vector2
{
    double v[2]{0, 0};
    Vector2() = default;
    Vector2(double x, double y) : v{x, y} {}
    double& x() { return v[0]; }
    double& y() { return v[1]; }
    double x() const { return v[0]; }
    double y() const { return v[1]; }
};

struct CameraParameters
{
    Vector2 principalPoint;
    Vector2 focal;
    float factor;
};
} // namespace mslam
#endif
