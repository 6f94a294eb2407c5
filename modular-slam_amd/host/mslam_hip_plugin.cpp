// mslam_hip_plugin.cpp — adapters from the reference's plugin interfaces to the C ABI (mslam_hip.h),
// and the factory aliases the reference's loader imports.
//
//   HipOrbDetector    : IFeatureDetector<RgbFrame,u8,32>   drop-in for DistributedOrbOpenCvDetector
//                                                           (distributed_cv_feature.cpp:1181-1222) and, created by
//                                                           hipCvOrbDetectorFactory, for OrbOpenCvDetector
//                                                           (orb_feature.cpp:25,33-65: the cv::ORB mode)
//   HipOrbMatcher     : IFeatureMatcher<u8,32>             drop-in for OrbOpenCvMatcher (orb_feature.cpp:84-130)
//   HipOrbRelocalizer : IRelocalizer                       what OrbRelocalizer is wired for
//                                                           (orb_relocalizer.cpp:26-50, rgbd_feature_frontend.cpp:153,176)
//   HipRansacPnp      : IPnpAlgorithm<SensorState,Vector3>  drop-in for OpenCvRansacPnp (cv_ransac_pnp.cpp:14-85)
//   HipLoopDetector   : ILoopDetector                      (loop_detection.hpp:10-15, rgbd_feature_frontend.cpp:202);
//                                                           both sit on ONE shared BoW database
//
// Errors: the reference interfaces have no status channel, so a non-zero C-ABI status becomes a
// std::runtime_error carrying mslam_hip_last_error() (the reference itself lets OpenCV/DBoW3 throw,
// orb_relocalizer.cpp:28).  No CPU fallback exists.
#include "mslam_interfaces.hpp"
#include "plugin_loader.hpp"

#include "../../include/mslam_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

namespace mslam
{
namespace
{
[[noreturn]] void raise(mslam_hip_ctx* ctx, const char* what, int rc)
{
    std::string msg = std::string("mslam_hip: ") + what + " failed (" + std::to_string(rc) + "): " + mslam_hip_last_error(ctx);
    std::fprintf(stderr, "[error] %s\n", msg.c_str());
    throw std::runtime_error(msg);
}

struct Ctx
{
    mslam_hip_ctx* h = nullptr;
    int width = -1, height = -1, capacity = 0;
    // width = height = 0: a context without detector buffers (the matcher and the BoW database need none)
    int detector = MSLAM_HIP_DETECTOR_DISTRIBUTED;
    void ensure(int w, int h_, int max_keypoints = 0)
    {
        if(h && w == width && h_ == height && (max_keypoints == 0 || max_keypoints == capacity))
            return;
        if(h)
            mslam_hip_destroy(h);
        h = nullptr;
        mslam_hip_params p;
        mslam_hip_default_params(&p); // the reference's hard-coded operating point
        p.width = w;
        p.height = h_;
        p.detector = detector;
        if(max_keypoints > 0)
        {
            p.max_keypoints = max_keypoints;
            p.max_candidates = std::max(p.max_candidates, 4 * max_keypoints);
        }
        const int rc = mslam_hip_create(&p, &h);
        if(rc != MSLAM_HIP_OK)
            raise(nullptr, "mslam_hip_create", rc);
        width = w;
        height = h_;
        capacity = p.max_keypoints;
    }
    ~Ctx()
    {
        if(h)
            mslam_hip_destroy(h);
    }
};

void gather_descriptors(const std::vector<OrbKeypoint>& kps, std::vector<std::uint8_t>& out)
{
    out.resize(kps.size() * 32);
    for(std::size_t i = 0; i < kps.size(); ++i)
        std::memcpy(&out[i * 32], kps[i].descriptor.data(), 32);
}
} // namespace

class HipOrbDetector : public IOrbFeatureDetector
{
  public:
    explicit HipOrbDetector(int detector = MSLAM_HIP_DETECTOR_DISTRIBUTED) { ctx.detector = detector; }
    std::vector<OrbKeypoint> detect(const RgbFrame& sensorData) override
    {
        std::vector<OrbKeypoint> result;
        if(sensorData.data.empty()) // distributed_cv_feature.cpp:724-727: empty image => empty result
            return result;
        const int w = sensorData.size.width, h = sensorData.size.height;
        // The reference's output is unbounded (a std::vector).  Start from a capacity sized for the frame
        // area (8192 at 640x480) and, if a frame yields more, rebuild the context with twice the room and
        // run the frame again: the capacity only ever grows.
        if(ctx.width != w || ctx.height != h)
        {
            const long long area = (long long)w * h;
            wanted = (int)std::min<long long>(65535, std::max<long long>(wanted, 8192 * ((area + 307199) / 307200)));
        }
        int n = 0;
        for(;;)
        {
            ctx.ensure(w, h, wanted);
            xy.resize(2 * (std::size_t)ctx.capacity);
            desc.resize(32 * (std::size_t)ctx.capacity);
            const int rc = mslam_hip_detect(ctx.h, sensorData.data.data(), w, h, ctx.capacity, xy.data(), desc.data(),
                                            nullptr, nullptr, nullptr, &n);
            if(rc == MSLAM_HIP_OK)
                break;
            if(rc != MSLAM_HIP_E_CAPACITY || wanted >= 65535)
                raise(ctx.h, "mslam_hip_detect", rc);
            wanted = std::min(65535, 2 * wanted);
        }
        result.resize(static_cast<std::size_t>(n));
        for(int i = 0; i < n; ++i)
        {
            // distributed_cv_feature.cpp:1203-1213: id = running index, float coordinates widened to double
            result[i].keypoint.id = static_cast<Id>(i);
            result[i].keypoint.coordinates.x() = xy[2 * i];
            result[i].keypoint.coordinates.y() = xy[2 * i + 1];
            std::memcpy(result[i].descriptor.data(), &desc[32 * i], 32);
        }
        return result;
    }

  private:
    Ctx ctx;
    int wanted = 0;
    std::vector<float> xy;
    std::vector<std::uint8_t> desc;
};

class HipOrbMatcher : public IOrbMatcher
{
  public:
    std::vector<DescriptorMatch> match(const std::vector<OrbKeypoint>& fromDescriptors,
                                       const std::vector<OrbKeypoint>& toDescriptors) override
    {
        ctx.ensure(0, 0); // the matcher needs no detector buffers
        gather_descriptors(fromDescriptors, from);
        gather_descriptors(toDescriptors, to);
        fi.resize(toDescriptors.size() + 1);
        ti.resize(toDescriptors.size() + 1);
        int n = 0;
        const int rc = mslam_hip_match(ctx.h, from.data(), static_cast<int>(fromDescriptors.size()), to.data(),
                                       static_cast<int>(toDescriptors.size()), 0.7 /* orb_feature.cpp:101 */, fi.data(),
                                       ti.data(), &n);
        if(rc != MSLAM_HIP_OK)
            raise(ctx.h, "mslam_hip_match", rc);
        std::vector<DescriptorMatch> matches(static_cast<std::size_t>(n));
        for(int i = 0; i < n; ++i)
            matches[i] = DescriptorMatch{static_cast<std::size_t>(fi[i]), static_cast<std::size_t>(ti[i])};
        return matches;
    }

  private:
    Ctx ctx;
    std::vector<std::uint8_t> from, to;
    std::vector<std::int32_t> fi, ti;
};

// The BoW object both interfaces sit on.  The reference frontend holds the relocalizer and the loop detector as
// two separate members (rgbd_feature_frontend.cpp:153 and the loopDetector it is constructed with); keyframes are
// only ever fed through IRelocalizer::addKeyframe (:176), and ILoopDetector::detectLoop() takes no arguments
// (loop_detection.hpp:13).  So the two adapters returned by the two factories share ONE database: what
// addKeyframe feeds is what detectLoop() reports on.
class BowDatabase
{
  public:
    using KeyframePtr = std::shared_ptr<Keyframe<slam3d::SensorState>>;

    // like OrbRelocalizer (orb_relocalizer.cpp:26-30): loads "orbvoc.dbow3" from the working directory and throws
    // when it is missing.  MSLAM_ORB_VOCABULARY overrides the path (the reference hard-codes it).
    BowDatabase()
    {
        const char* env = std::getenv("MSLAM_ORB_VOCABULARY");
        const std::string path = env ? env : "orbvoc.dbow3";
        std::ifstream f(path, std::ios::binary);
        if(!f)
            throw std::runtime_error("HipOrbRelocalizer: could not open vocabulary " + path);
        std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        ctx.ensure(0, 0);
        const int rc = mslam_hip_bow_load(ctx.h, blob.data(), blob.size());
        if(rc != MSLAM_HIP_OK)
            raise(ctx.h, "mslam_hip_bow_load", rc);
    }

    static std::shared_ptr<BowDatabase> shared()
    {
        static std::mutex m;
        static std::weak_ptr<BowDatabase> live;
        std::lock_guard<std::mutex> lock(m);
        auto p = live.lock();
        if(!p)
        {
            p = std::make_shared<BowDatabase>();
            live = p;
        }
        return p;
    }

    std::vector<KeyframePtr> relocalize(const std::vector<OrbKeypoint>& keypoints, std::vector<double>* scoresOut = nullptr)
    {
        gather_descriptors(keypoints, desc);
        std::int32_t ids[64];
        double scores[64];
        int n = 0;
        const int rc = mslam_hip_bow_db_query(ctx.h, desc.data(), static_cast<int>(keypoints.size()), 64, ids, scores, &n);
        if(rc != MSLAM_HIP_OK)
            raise(ctx.h, "mslam_hip_bow_db_query", rc);
        std::vector<KeyframePtr> out;
        for(int i = 0; i < n && out.size() < 4; ++i)
        {
            auto it = entryToKeyframe.find(ids[i]);
            if(it != entryToKeyframe.end())
            {
                out.push_back(it->second);
                if(scoresOut)
                    scoresOut->push_back(scores[i]);
            }
        }
        return out;
    }

    void addKeyframe(KeyframePtr keyframe, const std::vector<OrbKeypoint>& keypoints)
    {
        if(keypoints.empty()) // the reference asserts non-empty (orb_relocalizer.cpp:42)
            return;
        // loop candidate = best earlier keyframe for the one being added
        const auto candidates = relocalize(keypoints);
        lastLoop = candidates.empty() ? nullptr : candidates.front();
        int entry = -1;
        const int rc = mslam_hip_bow_db_add(ctx.h, desc.data(), static_cast<int>(keypoints.size()), &entry);
        if(rc != MSLAM_HIP_OK)
            raise(ctx.h, "mslam_hip_bow_db_add", rc);
        entryToKeyframe[entry] = std::move(keyframe);
    }

    void removeKeyframe(const KeyframePtr& keyframe)
    {
        for(auto it = entryToKeyframe.begin(); it != entryToKeyframe.end();)
        {
            if(it->second == keyframe)
            {
                const int rc = mslam_hip_bow_db_remove(ctx.h, it->first); // never scored again
                if(rc != MSLAM_HIP_OK)
                    raise(ctx.h, "mslam_hip_bow_db_remove", rc);
                it = entryToKeyframe.erase(it);
            }
            else
                ++it;
        }
        if(lastLoop == keyframe)
            lastLoop = nullptr;
    }

    KeyframePtr detectLoop() const { return lastLoop; }

  private:
    Ctx ctx;
    std::vector<std::uint8_t> desc;
    std::map<int, KeyframePtr> entryToKeyframe;
    KeyframePtr lastLoop;
};

class HipOrbRelocalizer : public IOrbRelocalizer
{
  public:
    HipOrbRelocalizer() : db(BowDatabase::shared()) {}
    std::vector<BowDatabase::KeyframePtr> relocalize(const std::vector<OrbKeypoint>& keypoints) override
    {
        return db->relocalize(keypoints);
    }
    void addKeyframe(BowDatabase::KeyframePtr keyframe, const std::vector<OrbKeypoint>& keypoints) override
    {
        db->addKeyframe(std::move(keyframe), keypoints);
    }
    void removeKeyframe(BowDatabase::KeyframePtr keyframe) override { db->removeKeyframe(keyframe); }

  private:
    std::shared_ptr<BowDatabase> db;
};

class HipLoopDetector : public IOrbLoopDetector
{
  public:
    HipLoopDetector() : db(BowDatabase::shared()) {}
    BowDatabase::KeyframePtr detectLoop() override { return db->detectLoop(); }

  private:
    std::shared_ptr<BowDatabase> db;
};

// drop-in for OpenCvRansacPnp (cv_ransac_pnp.cpp:14-85): the same conversions around the solver call — landmark states and
// image points cast to float (:22-40), the initial sensor pose turned into the world -> camera transform and its rotation
// into a Rodrigues vector (:42-50), cv::solvePnPRansac's arguments (useExtrinsicGuess, 100 iterations, 5 px; :56-57), and
// the result inverted back into a sensor pose (:65-78).
class HipRansacPnp : public ISlam3dPnp
{
  public:
    std::optional<PnpResult> solvePnp(const std::vector<std::shared_ptr<Landmark<Vector3>>>& landmarks,
                                      const std::vector<Vector2>& sensorPoints, const slam3d::SensorState& initial) override
    {
        if(landmarks.size() != sensorPoints.size() || landmarks.size() < 4) // cv::solvePnPRansac asserts npoints >= 4
            return std::nullopt;
        const std::size_t n = landmarks.size();
        obj.resize(3 * n);
        img.resize(2 * n);
        for(std::size_t i = 0; i < n; ++i)
        {
            obj[3 * i] = static_cast<float>(landmarks[i]->state.x());
            obj[3 * i + 1] = static_cast<float>(landmarks[i]->state.y());
            obj[3 * i + 2] = static_cast<float>(landmarks[i]->state.z());
            img[2 * i] = static_cast<float>(sensorPoints[i].x());
            img[2 * i + 1] = static_cast<float>(sensorPoints[i].y());
        }
        // toCameraCoordinateSystemProjection (projection.cpp:19-28)
        const auto inverse = initial.orientation.inverse();
        const Vector3 t0 = -(inverse * initial.position);
        double rvec[3], tvec[3] = {t0.x(), t0.y(), t0.z()};
        toRodrigues(inverse.w(), inverse.x(), inverse.y(), inverse.z(), rvec);
        mask.assign(n, 0);
        int nInliers = 0;
        ctx.ensure(0, 0);
        const int rc = mslam_hip_pnp_ransac(ctx.h, obj.data(), img.data(), static_cast<int>(n), cameraParams.focal.x(),
                                            cameraParams.focal.y(), cameraParams.principalPoint.x(),
                                            cameraParams.principalPoint.y(), 1, 100, 5.0, 0, rvec, tvec, mask.data(), &nInliers);
        if(rc == MSLAM_HIP_E_NO_MODEL)
        {
            std::fprintf(stderr, "[error] Didnt find pnp solution\n"); // cv_ransac_pnp.cpp:61
            return std::nullopt;
        }
        if(rc != MSLAM_HIP_OK)
            raise(ctx.h, "mslam_hip_pnp_ransac", rc);
        // :65-78: the camera rotation as angle-axis, inverted = the sensor orientation; position = -(orientation * t)
        const double angle = std::sqrt(rvec[0] * rvec[0] + rvec[1] * rvec[1] + rvec[2] * rvec[2]);
        double w = 1, x = 0, y = 0, z = 0;
        if(angle > 0)
        {
            const double s = std::sin(0.5 * angle) / angle;
            w = std::cos(0.5 * angle), x = rvec[0] * s, y = rvec[1] * s, z = rvec[2] * s;
        }
        PnpResult result;
        result.pose.orientation = Quaternion(w, x, y, z).inverse();
        result.pose.position = -(result.pose.orientation * Vector3(tvec[0], tvec[1], tvec[2]));
        result.inliers.resize(n, false);
        for(std::size_t i = 0; i < n; ++i)
            result.inliers[i] = mask[i] != 0;
        return result;
    }

  private:
    static void toRodrigues(double w, double x, double y, double z, double r[3])
    {
        // Eigen::AngleAxisd(q) (cv_ransac_pnp.cpp:44-48): angle = 2 atan2(|v|, |w|) from the POSITIVE norm, then the axis is
        // v / |v| for w >= 0 and -v / |v| for w < 0 (q and -q are the same rotation)
        const double nv = std::sqrt(x * x + y * y + z * z);
        if(nv < 1e-300)
        {
            r[0] = r[1] = r[2] = 0;
            return;
        }
        const double angle = 2.0 * std::atan2(nv, std::fabs(w));
        const double d = w < 0 ? -nv : nv;
        r[0] = x / d * angle, r[1] = y / d * angle, r[2] = z / d * angle;
    }
    Ctx ctx;
    std::vector<float> obj, img;
    std::vector<std::uint8_t> mask;
};

// ---- factories + aliases (what loadFactoryMethod<T>(lib, name) imports) -------------------------------
std::unique_ptr<IOrbFeatureDetector> createHipOrbDetector() { return std::make_unique<HipOrbDetector>(); }
// drop-in for OrbOpenCvDetector (orb_feature.cpp:25,33-65; wired by src/app/slam/rgbd_slam.cpp:74-76).  The reference leaves
// Keypoint::id uninitialised there (orb_feature.cpp:54-61); the running index is used instead.
std::unique_ptr<IOrbFeatureDetector> createHipCvOrbDetector()
{
    return std::make_unique<HipOrbDetector>(MSLAM_HIP_DETECTOR_CV_ORB);
}
std::unique_ptr<IOrbMatcher> createHipOrbMatcher() { return std::make_unique<HipOrbMatcher>(); }
std::unique_ptr<IOrbRelocalizer> createHipOrbRelocalizer() { return std::make_unique<HipOrbRelocalizer>(); }
std::unique_ptr<IOrbLoopDetector> createHipLoopDetector() { return std::make_unique<HipLoopDetector>(); }
std::unique_ptr<ISlam3dPnp> createHipRansacPnp() { return std::make_unique<HipRansacPnp>(); }

} // namespace mslam

MSLAM_DLL_ALIAS(mslam::createHipOrbDetector, hipOrbDetectorFactory)
MSLAM_DLL_ALIAS(mslam::createHipCvOrbDetector, hipCvOrbDetectorFactory)
MSLAM_DLL_ALIAS(mslam::createHipOrbMatcher, hipOrbMatcherFactory)
MSLAM_DLL_ALIAS(mslam::createHipOrbRelocalizer, hipOrbRelocalizerFactory)
MSLAM_DLL_ALIAS(mslam::createHipLoopDetector, loopDetection) // key used by test/plugin_config.json
MSLAM_DLL_ALIAS(mslam::createHipRansacPnp, hipRansacPnpFactory)
