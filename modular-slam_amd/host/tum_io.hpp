// tum_io.hpp — the data formats on either side of the feature path, standalone C++17 + zlib:
//
//   readTumRgbdDataset / tumRgbdCameraParams / RgbdFileProvider   rgbd_file_provider.cpp:41-149
//       (the reference decodes with cv::imread; OpenCV is not available here, so decodePng below reads the
//        non-interlaced 8-bit RGB(A)/gray and 16-bit gray PNGs TUM RGB-D consists of, and delivers what
//        imread delivers: B,G,R bytes, and native-endian u16 for IMREAD_ANYDEPTH)
//   associate                                                     utils/tools/py/associate.py:49-101
//   KittiLocalizationDumper / TumLocalizationDumper               src/app/viewer/viewer.cpp:105-164
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <tuple>
#include <utility>
#include <vector>
#include <zlib.h>

#include "mslam_camera.hpp"

namespace mslam
{

struct RgbdFilePaths // sensors/rgbd_file_provider.hpp
{
    std::vector<std::string> rgbPaths;
    std::vector<std::string> depthPaths;
    std::vector<double> timestamps;
};

// rgbd_file_provider.cpp:109-134: "timestamp rgbPath <ignored> depthPath" per line, paths relative to the
// file's directory; lines that do not parse are skipped
inline RgbdFilePaths readTumRgbdDataset(const std::filesystem::path& tumFile)
{
    std::ifstream ifss{tumFile};
    RgbdFilePaths paths;
    const auto rootDir = tumFile.parent_path();
    for(std::string line; std::getline(ifss, line);)
    {
        std::istringstream fields{line};
        double stamp;
        std::string rgb, depth_stamp_ignored, depth;
        if(fields >> stamp >> rgb >> depth_stamp_ignored >> depth) // all four fields present, the first a number
        {
            paths.timestamps.push_back(stamp);
            paths.rgbPaths.push_back((rootDir / rgb).string());
            paths.depthPaths.push_back((rootDir / depth).string());
        }
    }
    return paths;
}

// rgbd_file_provider.cpp:136-147
inline CameraParameters tumRgbdCameraParams()
{
    CameraParameters p; // rgbd_file_provider.cpp:136-145
    p.focal = Vector2(525.0, 525.0);
    p.principalPoint = Vector2(319.5, 239.5);
    p.factor = 1.f / 5000.f;
    return p;
}

struct DecodedImage
{
    int width = 0, height = 0, channels = 0, bytesPerSample = 0;
    std::vector<std::uint8_t> data; // 8-bit: interleaved, colour as B,G,R(,A dropped); 16-bit: native-endian u16
    bool empty() const { return data.empty(); }
};

// Non-interlaced PNG, colour types 0 (gray 8/16), 2 (RGB 8), 6 (RGBA 8, alpha dropped as imread does by default).
// Returns an empty image on anything else or on a damaged file (cv::imread returns an empty Mat there).
inline DecodedImage decodePng(const std::string& path)
{
    DecodedImage out;
    std::ifstream f(path, std::ios::binary);
    if(!f)
        return out;
    std::vector<std::uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const std::uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if(file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0)
        return out;
    auto be32 = [&](size_t o) { return ((uint32_t)file[o] << 24) | ((uint32_t)file[o + 1] << 16) | ((uint32_t)file[o + 2] << 8) | file[o + 3]; };
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<std::uint8_t> idat;
    for(size_t o = 8; o + 12 <= file.size();)
    {
        const uint32_t len = be32(o);
        if(o + 12 + (size_t)len > file.size())
            return out;
        const char* type = reinterpret_cast<const char*>(&file[o + 4]);
        if(std::memcmp(type, "IHDR", 4) == 0 && len >= 13)
        {
            w = be32(o + 8);
            h = be32(o + 12);
            depth = file[o + 16];
            ctype = file[o + 17];
            interlace = file[o + 20];
        }
        else if(std::memcmp(type, "IDAT", 4) == 0)
            idat.insert(idat.end(), file.begin() + o + 8, file.begin() + o + 8 + len);
        else if(std::memcmp(type, "IEND", 4) == 0)
            break;
        o += 12 + (size_t)len;
    }
    int samples = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
    if(!w || !h || interlace || !samples || !((depth == 8) || (depth == 16 && ctype == 0)))
        return out;
    const size_t bps = depth / 8, bpp = bps * samples, stride = (size_t)w * bpp;
    std::vector<std::uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    if(uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size())
        return out;
    // undo the per-row filters in place (PNG specification §9)
    std::vector<std::uint8_t> img(stride * h);
    for(uint32_t y = 0; y < h; ++y)
    {
        const std::uint8_t ft = raw[(stride + 1) * y];
        const std::uint8_t* src = &raw[(stride + 1) * y + 1];
        std::uint8_t* cur = &img[stride * y];
        const std::uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        for(size_t i = 0; i < stride; ++i)
        {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred = 0;
            switch(ft)
            {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4:
            {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: return out;
            }
            cur[i] = (std::uint8_t)(src[i] + pred);
        }
    }
    out.width = (int)w;
    out.height = (int)h;
    out.bytesPerSample = (int)bps;
    if(depth == 16)
    {
        out.channels = 1;
        out.data.resize((size_t)w * h * 2);
        for(size_t i = 0; i < (size_t)w * h; ++i)
        {
            const std::uint16_t v = (std::uint16_t)((img[2 * i] << 8) | img[2 * i + 1]); // PNG is big-endian
            std::memcpy(&out.data[2 * i], &v, 2);
        }
    }
    else if(samples == 1)
    {
        out.channels = 1;
        out.data = std::move(img);
    }
    else
    {
        out.channels = 3;
        out.data.resize((size_t)w * h * 3);
        for(size_t i = 0; i < (size_t)w * h; ++i)
        {
            out.data[3 * i] = img[samples * i + 2]; // B
            out.data[3 * i + 1] = img[samples * i + 1];
            out.data[3 * i + 2] = img[samples * i];
        }
    }
    return out;
}

struct RgbdFrameData // types/rgbd_frame.hpp, flattened
{
    double timestamp = 0;
    int width = 0, height = 0;
    std::vector<std::uint8_t> rgb;    // B,G,R interleaved
    std::vector<std::uint16_t> depth; // raw sensor units
    CameraParameters cameraParameters{};
};

// rgbd_file_provider.cpp:41-107
class RgbdFileProvider
{
  public:
    RgbdFileProvider(RgbdFilePaths paths, const CameraParameters& params) : filePaths(std::move(paths)), cameraParameters(params) {}
    bool init() { return !filePaths.rgbPaths.empty() && filePaths.rgbPaths.size() == filePaths.depthPaths.size(); }
    bool fetch()
    {
        if(currentIndex >= filePaths.rgbPaths.size())
        {
            recentFrame = nullptr;
            return false;
        }
        const DecodedImage rgb = decodePng(filePaths.rgbPaths[currentIndex]);
        const DecodedImage depth = decodePng(filePaths.depthPaths[currentIndex]);
        if(rgb.empty() || depth.empty() || rgb.channels != 3 || depth.bytesPerSample != 2)
        {
            recentFrame = nullptr;
            return false;
        }
        auto frame = std::make_shared<RgbdFrameData>();
        frame->timestamp = filePaths.timestamps[currentIndex];
        frame->width = rgb.width;
        frame->height = rgb.height;
        frame->rgb = rgb.data;
        frame->depth.resize((size_t)depth.width * depth.height);
        std::memcpy(frame->depth.data(), depth.data.data(), depth.data.size());
        frame->cameraParameters = cameraParameters;
        currentIndex += 1;
        recentFrame = std::move(frame);
        return true;
    }
    std::shared_ptr<RgbdFrameData> recentData() const { return recentFrame; }

  private:
    RgbdFilePaths filePaths;
    CameraParameters cameraParameters;
    std::size_t currentIndex = 0;
    std::shared_ptr<RgbdFrameData> recentFrame;
};

// associate.py:49-70: "stamp d1 d2 ..." lines; commas and tabs are blanks; lines starting with '#', empty lines
// and lines with nothing after the stamp are skipped; a repeated stamp keeps its last line
inline std::map<double, std::vector<std::string>> readFileList(const std::string& filename)
{
    std::ifstream f(filename);
    std::map<double, std::vector<std::string>> out;
    std::string line;
    while(std::getline(f, line))
    {
        if(line.empty() || line[0] == '#')
            continue;
        for(char& ch : line)
            if(ch == ',' || ch == '\t')
                ch = ' ';
        std::istringstream iss(line);
        std::vector<std::string> tok;
        for(std::string t; iss >> t;)
            tok.push_back(t);
        if(tok.size() > 1)
            out[std::stod(tok[0])] = std::vector<std::string>(tok.begin() + 1, tok.end());
    }
    return out;
}

// associate.py:72-101: every pair closer than max_difference is a candidate; candidates are taken in order of
// (difference, first stamp, second stamp), each stamp at most once; the result is sorted by first stamp
inline std::vector<std::pair<double, double>> associate(const std::vector<double>& first, const std::vector<double>& second,
                                                        double offset, double max_difference)
{
    std::vector<std::tuple<double, double, double>> cand;
    for(double a : first)
        for(double b : second)
            if(std::fabs(a - (b + offset)) < max_difference)
                cand.emplace_back(std::fabs(a - (b + offset)), a, b);
    std::sort(cand.begin(), cand.end());
    std::map<double, bool> fa, fb;
    for(double a : first)
        fa[a] = true;
    for(double b : second)
        fb[b] = true;
    std::vector<std::pair<double, double>> matches;
    for(const auto& [d, a, b] : cand)
    {
        (void)d;
        if(fa[a] && fb[b])
        {
            fa[a] = fb[b] = false;
            matches.emplace_back(a, b);
        }
    }
    std::sort(matches.begin(), matches.end());
    return matches;
}

struct Pose // slam3d::SensorState: position + unit quaternion (x, y, z, w)
{
    double position[3];
    double qx, qy, qz, qw;
};

// viewer.cpp:105-137: the 3x4 [R | t] row by row, default ostream formatting, a blank after every number
class KittiLocalizationDumper
{
  public:
    explicit KittiLocalizationDumper(const std::string& path) : output(path) {}
    void operator()(const Pose& p)
    {
        // Eigen::Quaternion::toRotationMatrix
        const double tx = 2 * p.qx, ty = 2 * p.qy, tz = 2 * p.qz;
        const double twx = tx * p.qw, twy = ty * p.qw, twz = tz * p.qw;
        const double txx = tx * p.qx, txy = ty * p.qx, txz = tz * p.qx;
        const double tyy = ty * p.qy, tyz = tz * p.qy, tzz = tz * p.qz;
        const double R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy},
                                {txy + twz, 1 - (txx + tzz), tyz - twx},
                                {txz - twy, tyz + twx, 1 - (txx + tyy)}};
        for(int i = 0; i < 3; ++i)
        {
            for(int j = 0; j < 3; ++j)
                output << R[i][j] << " ";
            output << p.position[i] << " ";
        }
        output << "\n";
    }

  private:
    std::ofstream output;
};

// viewer.cpp:139-164: "timestamp tx ty tz qx qy qz qw", fixed, 6 decimals
class TumLocalizationDumper
{
  public:
    explicit TumLocalizationDumper(const std::string& path) : output(path)
    {
        output.setf(std::ios::fixed);
        output.precision(6);
    }
    void operator()(double timestamp, const Pose& p)
    {
        output << timestamp << " ";
        output << p.position[0] << " " << p.position[1] << " " << p.position[2] << " ";
        output << p.qx << " " << p.qy << " " << p.qz << " " << p.qw;
        output << "\n";
    }

  private:
    std::ofstream output;
};

} // namespace mslam
