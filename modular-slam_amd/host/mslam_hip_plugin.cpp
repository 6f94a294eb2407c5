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

} // namespace mslam

MSLAM_DLL_ALIAS(mslam::createHipOrbDetector, hipOrbDetectorFactory)
MSLAM_DLL_ALIAS(mslam::createHipCvOrbDetector, hipCvOrbDetectorFactory)
MSLAM_DLL_ALIAS(mslam::createHipOrbMatcher, hipOrbMatcherFactory)
MSLAM_DLL_ALIAS(mslam::createHipOrbRelocalizer, hipOrbRelocalizerFactory)
MSLAM_DLL_ALIAS(mslam::createHipLoopDetector, loopDetection) // key used by test/plugin_config.json
