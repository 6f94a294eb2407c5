// harness.cpp — a minimal C++ host that uses the plugin exactly as the reference would:
// loadFactoryMethod<T>(library, alias) (plugin_loader.hpp:20-24, as in test/plugin_loader_test.cpp:17-21),
// then the call order of RgbdFeatureFrontend::processSensorData: detect(frame_t) ->
// match(from = keypoints_t, to = keypoints_{t-1}) (rgbd_feature_frontend.cpp:187,237).
//
// usage: mslam_harness <plugin.so> <width> <height> <frame0.bgr> <frame1.bgr>     raw B,G,R frames
//        mslam_harness <plugin.so> --tum <associations.txt>                       a TUM RGB-D sequence, read the way
//                                                                                 the reference's RgbdFileProvider does
//        mslam_harness <plugin.so> --bow <vocabulary.dbow3> <width> <height> <frame.bgr>...
//                                    the BoW boundary: relocalizer + loop detector factories (one shared database),
//                                    fed in the frontend's order: detect -> addKeyframe (rgbd_feature_frontend.cpp:176)
//                                    -> detectLoop (:202); then relocalize / removeKeyframe / relocalize
// prints one line per frame/match with an FNV-1a checksum the parity test compares with the oracle's.
#include "mslam_interfaces.hpp"
#include "plugin_loader.hpp"
#include "tum_io.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

static std::uint32_t fnv(const void* p, std::size_t n, std::uint32_t h = 0x811C9DC5u)
{
    const auto* b = static_cast<const unsigned char*>(p);
    for(std::size_t i = 0; i < n; ++i)
        h = (h ^ b[i]) * 0x01000193u;
    return h;
}

int main(int argc, char** argv)
{
    if(argc < 2)
    {
        std::fprintf(stderr, "usage: %s <plugin> [<width> <height> <frame0.bgr> <frame1.bgr>]\n", argv[0]);
        return 2;
    }
    try
    {
        // MSLAM_HARNESS_DETECTOR=cvorb: the factory of the OrbOpenCvDetector drop-in instead of the in-tree extractor's
        const char* det = std::getenv("MSLAM_HARNESS_DETECTOR");
        const bool cvorb = det && std::strcmp(det, "cvorb") == 0;
        auto makeDetector = mslam::loadFactoryMethod<mslam::IOrbFeatureDetector>(
            argv[1], cvorb ? "hipCvOrbDetectorFactory" : "hipOrbDetectorFactory");
        auto makeMatcher = mslam::loadFactoryMethod<mslam::IOrbMatcher>(argv[1], "hipOrbMatcherFactory");
        if(!makeDetector || !makeMatcher)
            return 3;
        std::unique_ptr<mslam::IOrbFeatureDetector> detector = makeDetector();
        std::unique_ptr<mslam::IOrbMatcher> matcher = makeMatcher();
        std::printf("loaded %s\n", detector && matcher ? "ok" : "null");
        if(argc == 4 && std::strcmp(argv[2], "--pnp") == 0)
        {
            // scene file: u32 n, then n x (3 f64 landmark, 2 f64 image point), then the initial sensor pose (3 f64 position,
            // 4 f64 quaternion w x y z) and fx fy cx cy
            std::ifstream in(argv[3], std::ios::binary);
            std::uint32_t n = 0;
            in.read(reinterpret_cast<char*>(&n), 4);
            std::vector<std::shared_ptr<mslam::Landmark<mslam::Vector3>>> landmarks;
            std::vector<mslam::Vector2> points;
            for(std::uint32_t i = 0; i < n; ++i)
            {
                double v[5];
                in.read(reinterpret_cast<char*>(v), sizeof(v));
                auto lm = std::make_shared<mslam::Landmark<mslam::Vector3>>();
                lm->id = i;
                lm->state = mslam::Vector3(v[0], v[1], v[2]);
                landmarks.push_back(lm);
                points.emplace_back(v[3], v[4]);
            }
            double pose[7], cam[4];
            in.read(reinterpret_cast<char*>(pose), sizeof(pose));
            if(!in.read(reinterpret_cast<char*>(cam), sizeof(cam)))
            {
                std::fprintf(stderr, "cannot read %s\n", argv[3]);
                return 5;
            }
            auto makePnp = mslam::loadFactoryMethod<mslam::ISlam3dPnp>(argv[1], "hipRansacPnpFactory");
            std::unique_ptr<mslam::ISlam3dPnp> pnp = makePnp();
            mslam::CameraParameters cp;
            cp.focal = mslam::Vector2(cam[0], cam[1]);
            cp.principalPoint = mslam::Vector2(cam[2], cam[3]);
            cp.factor = 1.0f / 5000.0f;
            pnp->setCameraParameters(cp);
            mslam::slam3d::SensorState initial;
            initial.position = mslam::Vector3(pose[0], pose[1], pose[2]);
            initial.orientation = mslam::Quaternion(pose[3], pose[4], pose[5], pose[6]);
            const auto result = pnp->solvePnp(landmarks, points, initial);
            if(!result)
            {
                std::printf("pnp none\n");
                return 0;
            }
            std::size_t inl = 0;
            for(bool b : result->inliers)
                inl += b ? 1 : 0;
            std::printf("pnp position %.17g %.17g %.17g orientation %.17g %.17g %.17g %.17g inliers %zu\n",
                        result->pose.position.x(), result->pose.position.y(), result->pose.position.z(),
                        result->pose.orientation.w(), result->pose.orientation.x(), result->pose.orientation.y(),
                        result->pose.orientation.z(), inl);
            return 0;
        }
        if(argc >= 7 && std::strcmp(argv[2], "--bow") == 0)
        {
            setenv("MSLAM_ORB_VOCABULARY", argv[3], 1); // the reference hard-codes "orbvoc.dbow3" in the working directory
            auto makeReloc = mslam::loadFactoryMethod<mslam::IOrbRelocalizer>(argv[1], "hipOrbRelocalizerFactory");
            auto makeLoop = mslam::loadFactoryMethod<mslam::IOrbLoopDetector>(argv[1], "loopDetection");
            std::unique_ptr<mslam::IOrbRelocalizer> relocalizer = makeReloc();
            std::unique_ptr<mslam::IOrbLoopDetector> loopDetector = makeLoop();
            const int w = std::atoi(argv[4]), h = std::atoi(argv[5]);
            using Kf = mslam::Keyframe<mslam::slam3d::SensorState>;
            std::vector<std::shared_ptr<Kf>> keyframes;
            std::vector<std::vector<mslam::OrbKeypoint>> all;
            for(int f = 0; f + 6 < argc; ++f)
            {
                mslam::RgbFrame frame;
                frame.size = {w, h};
                frame.data.resize(static_cast<std::size_t>(w) * h * 3);
                std::ifstream in(argv[6 + f], std::ios::binary);
                if(!in.read(reinterpret_cast<char*>(frame.data.data()), static_cast<std::streamsize>(frame.data.size())))
                {
                    std::fprintf(stderr, "cannot read %s\n", argv[6 + f]);
                    return 5;
                }
                auto kps = detector->detect(frame);
                auto kf = std::make_shared<Kf>();
                kf->id = 100 + static_cast<mslam::Id>(f);
                relocalizer->addKeyframe(kf, kps);
                const auto loop = loopDetector->detectLoop();
                std::printf("keyframe %d keypoints %zu loop %lld\n", f, kps.size(), loop ? (long long)loop->id : -1LL);
                keyframes.push_back(kf);
                all.push_back(std::move(kps));
            }
            for(std::size_t f = 0; f < all.size(); ++f)
            {
                std::printf("relocalize %zu:", f);
                for(const auto& k : relocalizer->relocalize(all[f]))
                    std::printf(" %llu", (unsigned long long)k->id);
                std::printf("\n");
            }
            relocalizer->removeKeyframe(keyframes[0]);
            std::printf("after remove 0:");
            for(const auto& k : relocalizer->relocalize(all[0]))
                std::printf(" %llu", (unsigned long long)k->id);
            std::printf("\n");
            return 0;
        }
        const bool tum = argc == 4 && std::strcmp(argv[2], "--tum") == 0;
        if(argc < 6 && !tum)
            return detector && matcher ? 0 : 4;
        const int w = tum ? 0 : std::atoi(argv[2]), h = tum ? 0 : std::atoi(argv[3]);
        mslam::RgbdFileProvider provider(tum ? mslam::readTumRgbdDataset(argv[3]) : mslam::RgbdFilePaths{},
                                         mslam::tumRgbdCameraParams());
        if(tum && !provider.init())
        {
            std::fprintf(stderr, "no frames listed in %s\n", argv[3]);
            return 5;
        }
        std::vector<mslam::OrbKeypoint> prev;
        for(int f = 0; tum ? provider.fetch() : f < 2; ++f)
        {
            mslam::RgbFrame frame;
            if(tum)
            {
                const auto data = provider.recentData();
                frame.size = {data->width, data->height};
                frame.data = data->rgb;
            }
            else
            {
                frame.size = {w, h};
                frame.data.resize(static_cast<std::size_t>(w) * h * 3);
                std::ifstream in(argv[4 + f], std::ios::binary);
                if(!in.read(reinterpret_cast<char*>(frame.data.data()), static_cast<std::streamsize>(frame.data.size())))
                {
                    std::fprintf(stderr, "cannot read %s\n", argv[4 + f]);
                    return 5;
                }
            }
            auto kps = detector->detect(frame);
            std::uint32_t hc = 0x811C9DC5u;
            for(const auto& k : kps)
            {
                const double xy[2] = {k.keypoint.coordinates.x(), k.keypoint.coordinates.y()};
                hc = fnv(&k.keypoint.id, 8, hc);
                hc = fnv(xy, 16, hc);
                hc = fnv(k.descriptor.data(), 32, hc);
            }
            std::printf("frame %d keypoints %zu fnv %08x\n", f, kps.size(), hc);
            if(f > 0)
            {
                auto m = matcher->match(kps, prev);
                std::uint32_t hm = 0x811C9DC5u;
                for(const auto& d : m)
                {
                    const std::uint64_t ft[2] = {d.fromIndex, d.toIndex};
                    hm = fnv(ft, 16, hm);
                }
                std::printf("match %d pairs %zu fnv %08x\n", f, m.size(), hm);
            }
            prev = std::move(kps);
        }
    }
    catch(const std::exception& e)
    {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
