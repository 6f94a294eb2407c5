// mslam_interfaces.hpp — host-side mirror of the reference's plugin interfaces for the feature path.
//
// Same names, template parameters, argument meaning and ownership as the reference:
//   Keypoint / KeypointDescriptor / DescriptorMatch   frontend/feature/feature_interface.hpp:18-41
//   IFeatureDetector / IFeatureMatcher                feature_interface.hpp:50-70
//   RgbFrame / Size                                   types/rgb_frame.hpp:12-16, types/basic_types.hpp:23-27
//   IRelocalizer                                      relocalizer.hpp:11-20
//   ILoopDetector                                     loop_detection.hpp:10-15
//   IOrbFeatureDetector / IOrbMatcher / OrbKeypoint   orb_feature.hpp:15-17
//
// The reference headers pull in Eigen, OpenCV and Boost, none of which exist in this build image, so
// this file declares the same shapes standalone.  When compiled inside the reference tree, define
// MSLAM_USE_REFERENCE_HEADERS and the real headers are used instead (the adapter source is the same).
#pragma once

#ifdef MSLAM_USE_REFERENCE_HEADERS
#include "modular_slam/loop_detection.hpp"
#include "modular_slam/orb_feature.hpp"
#include "modular_slam/pnp.hpp"
#include "modular_slam/relocalizer.hpp"
#include "modular_slam/types/slam3d_types.hpp"
#else

#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <optional>
#include <vector>

#include "mslam_camera.hpp"

namespace mslam
{
using Id = std::uint64_t;

struct Size
{
    int width;
    int height;
};

struct RgbFrame
{
    std::vector<std::uint8_t> data; // interleaved 3-channel bytes; B,G,R as both providers deliver them
    Size size;
};

struct Keypoint
{
    Id id;
    Vector2 coordinates;
};

template <typename DescriptorType, int Length = 32>
struct KeypointDescriptor
{
    Keypoint keypoint;
    std::array<DescriptorType, Length> descriptor;
};

struct DescriptorMatch
{
    std::size_t fromIndex;
    std::size_t toIndex;
};

template <typename SensorData, typename DescriptorType, int Length>
class IFeatureDetector
{
  public:
    virtual std::vector<KeypointDescriptor<DescriptorType, Length>> detect(const SensorData& sensorData) = 0;
    virtual ~IFeatureDetector() {}
};

template <typename DescriptorType, int Length>
class IFeatureMatcher
{
  public:
    virtual std::vector<DescriptorMatch>
    match(const std::vector<KeypointDescriptor<DescriptorType, Length>>& firstDescriptors,
          const std::vector<KeypointDescriptor<DescriptorType, Length>>& secondDescriptors) = 0;
    virtual ~IFeatureMatcher() {}
};

// types/keyframe.hpp: only what this path touches
template <typename StateType>
struct Keyframe
{
    Id id;
    StateType state;
};

// Eigen::Vector3d / Eigen::Quaterniond stand-ins: only what the adapters touch (types/basic_types.hpp)
struct Vector3
{
    double v[3]{0, 0, 0};
    Vector3() = default;
    Vector3(double x, double y, double z) : v{x, y, z} {}
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
    Vector3 operator-() const { return Vector3(-v[0], -v[1], -v[2]); }
};
struct Quaternion
{
    double q[4]{1, 0, 0, 0}; // w, x, y, z
    Quaternion() = default;
    Quaternion(double w, double x, double y, double z) : q{w, x, y, z} {}
    double w() const { return q[0]; }
    double x() const { return q[1]; }
    double y() const { return q[2]; }
    double z() const { return q[3]; }
    Quaternion inverse() const // unit quaternion
    {
        return Quaternion(q[0], -q[1], -q[2], -q[3]);
    }
    Vector3 operator*(const Vector3& p) const // rotate
    {
        const double w = q[0], x = q[1], y = q[2], z = q[3];
        const double tx = 2 * (y * p.z() - z * p.y()), ty = 2 * (z * p.x() - x * p.z()), tz = 2 * (x * p.y() - y * p.x());
        return Vector3(p.x() + w * tx + (y * tz - z * ty), p.y() + w * ty + (z * tx - x * tz), p.z() + w * tz + (x * ty - y * tx));
    }
};

// types/state.hpp, types/landmark.hpp, sensors/camera_parameters.hpp
template <typename PositionType, typename OrientationType>
struct State
{
    PositionType position;
    OrientationType orientation;
};
template <typename StateType>
struct Landmark
{
    Id id;
    StateType state;
};
namespace slam3d
{
using SensorState = State<Vector3, Quaternion>;
} // namespace slam3d

// pnp.hpp:14-36 (PnpResult::inliers is a boost::dynamic_bitset there)
template <typename SensorStateType, typename LandmarkStateType>
class IPnpAlgorithm
{
  public:
    struct PnpResult
    {
        SensorStateType pose;
        std::vector<bool> inliers;
    };
    virtual std::optional<PnpResult> solvePnp(const std::vector<std::shared_ptr<Landmark<LandmarkStateType>>>& landmarks,
                                              const std::vector<Vector2>& imgPoints,
                                              const SensorStateType& initial = SensorStateType()) = 0;
    void setCameraParameters(const CameraParameters& newParameters) { cameraParams = newParameters; }
    [[nodiscard]] const CameraParameters& cameraParameters() const { return cameraParams; }
    virtual ~IPnpAlgorithm() = default;

  protected:
    CameraParameters cameraParams;
};

template <typename StateType, typename DescriptorType, int DescriptorLength>
class IRelocalizer
{
  public:
    virtual std::vector<std::shared_ptr<Keyframe<StateType>>>
    relocalize(const std::vector<KeypointDescriptor<DescriptorType>>& keypoints) = 0;
    virtual void addKeyframe(std::shared_ptr<Keyframe<StateType>> keyframe,
                             const std::vector<KeypointDescriptor<DescriptorType>>& keypoints) = 0;
    virtual void removeKeyframe(std::shared_ptr<Keyframe<StateType>> keyframe) = 0;
};

template <typename StateType>
class ILoopDetector
{
  public:
    virtual std::shared_ptr<Keyframe<StateType>> detectLoop() = 0;
};

using IOrbFeatureDetector = IFeatureDetector<RgbFrame, std::uint8_t, 32>;
using IOrbMatcher = IFeatureMatcher<std::uint8_t, 32>;
using OrbKeypoint = KeypointDescriptor<std::uint8_t, 32>;

} // namespace mslam
#endif // MSLAM_USE_REFERENCE_HEADERS

namespace mslam
{
// the matcher relies on sizeof(OrbKeypoint) as the descriptor row stride (orb_feature.cpp:88-91)
static_assert(sizeof(OrbKeypoint) == 64, "OrbKeypoint is expected to be a 64-byte record");
using IOrbRelocalizer = IRelocalizer<slam3d::SensorState, std::uint8_t, 32>;
using IOrbLoopDetector = ILoopDetector<slam3d::SensorState>;
using ISlam3dPnp = IPnpAlgorithm<slam3d::SensorState, Vector3>;
} // namespace mslam
