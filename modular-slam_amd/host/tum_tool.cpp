// tum_tool — command-line face of tum_io.hpp, used by tests/test_tum_io.py (no GPU involved).
//   tum_tool list <associations.txt>                 -> "timestamp rgbPath depthPath" per parsed line
//   tum_tool decode <file.png> <out.raw>             -> writes what cv::imread would deliver; prints "w h channels bytes"
//   tum_tool frames <associations.txt> <out_prefix>  -> RgbdFileProvider loop: <prefix>NNNN.bgr / .depth16, prints one line per frame
//   tum_tool associate <a.txt> <b.txt> [offset] [max_difference]   -> associate.py's default output
//   tum_tool traj <poses.txt> <out.kitti> <out.tum>  -> poses.txt: "timestamp tx ty tz qx qy qz qw" per line
#include "tum_io.hpp"
#include <cstdio>
#include <iostream>

using namespace mslam;

static bool write_file(const std::string& path, const void* data, size_t n)
{
    std::ofstream f(path, std::ios::binary);
    f.write(static_cast<const char*>(data), (std::streamsize)n);
    return (bool)f;
}

int main(int argc, char** argv)
{
    const std::string cmd = argc > 1 ? argv[1] : "";
    if(cmd == "list" && argc == 3)
    {
        const auto p = readTumRgbdDataset(argv[2]);
        for(size_t i = 0; i < p.rgbPaths.size(); ++i)
            std::printf("%.6f %s %s\n", p.timestamps[i], p.rgbPaths[i].c_str(), p.depthPaths[i].c_str());
        return 0;
    }
    if(cmd == "decode" && argc == 4)
    {
        const auto img = decodePng(argv[2]);
        if(img.empty())
        {
            std::printf("empty\n");
            return 0;
        }
        if(!write_file(argv[3], img.data.data(), img.data.size()))
            return 2;
        std::printf("%d %d %d %d\n", img.width, img.height, img.channels, img.bytesPerSample);
        return 0;
    }
    if(cmd == "frames" && argc == 4)
    {
        RgbdFileProvider prov(readTumRgbdDataset(argv[2]), tumRgbdCameraParams());
        if(!prov.init())
        {
            std::printf("init failed\n");
            return 0;
        }
        int i = 0;
        while(prov.fetch())
        {
            const auto f = prov.recentData();
            char name[512];
            std::snprintf(name, sizeof name, "%s%04d", argv[3], i);
            write_file(std::string(name) + ".bgr", f->rgb.data(), f->rgb.size());
            write_file(std::string(name) + ".depth16", f->depth.data(), f->depth.size() * 2);
            std::printf("%.6f %d %d %g %g %g %g %.9g\n", f->timestamp, f->width, f->height, f->cameraParameters.focal.x(),
                        f->cameraParameters.focal.y(), f->cameraParameters.principalPoint.x(),
                        f->cameraParameters.principalPoint.y(), f->cameraParameters.factor);
            ++i;
        }
        std::printf("end after %d frames, recentData %s\n", i, prov.recentData() ? "set" : "null");
        return 0;
    }
    if(cmd == "associate" && argc >= 4)
    {
        const double offset = argc > 4 ? std::atof(argv[4]) : 0.0, maxd = argc > 5 ? std::atof(argv[5]) : 0.02;
        const auto A = readFileList(argv[2]), B = readFileList(argv[3]);
        std::vector<double> ka, kb;
        for(const auto& kv : A)
            ka.push_back(kv.first);
        for(const auto& kv : B)
            kb.push_back(kv.first);
        auto join = [](const std::vector<std::string>& v) {
            std::string s;
            for(size_t i = 0; i < v.size(); ++i)
                s += (i ? " " : "") + v[i];
            return s;
        };
        for(const auto& m : associate(ka, kb, offset, maxd)) // associate.py:124-126
            std::printf("%f %s %f %s\n", m.first, join(A.at(m.first)).c_str(), m.second - offset, join(B.at(m.second)).c_str());
        return 0;
    }
    if(cmd == "traj" && argc == 5)
    {
        std::ifstream in(argv[2]);
        KittiLocalizationDumper kitti(argv[3]);
        TumLocalizationDumper tum(argv[4]);
        double t;
        Pose p;
        while(in >> t >> p.position[0] >> p.position[1] >> p.position[2] >> p.qx >> p.qy >> p.qz >> p.qw)
        {
            kitti(p);
            tum(t, p);
        }
        return 0;
    }
    std::fprintf(stderr, "usage: tum_tool list|decode|frames|associate|traj ...\n");
    return 1;
}
