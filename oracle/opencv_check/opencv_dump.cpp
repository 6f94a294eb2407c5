// opencv_dump.cpp — runs the OpenCV primitives the reference's feature path calls, on the repository's two bundled
// frames, and writes every result to a flat binary file that compare.py diffs against the oracle.
//
// Call sites reproduced (paths in the reference repo, src/lib/modular_slam/):
//   cv::resize INTER_LINEAR        distributed_cv_feature.cpp:839      (chained pyramid, sizes as :836-837)
//   cv::FAST(sub, 20 / 7, true)    distributed_cv_feature.cpp:918,:924 (here: on the 70x70 cell tiles of level 0 and on
//                                                                        the whole level)
//   cv::GaussianBlur 7x7 sigma 2   distributed_cv_feature.cpp:797-798
//   cv::fastAtan2                  distributed_cv_feature.cpp:569
//   BFMatcher(HAMMING).knnMatch    orb_feature.cpp:96
//   cv::ORB::create(1000)->detectAndCompute   orb_feature.cpp:25,40     (plus resize INTER_LINEAR_EXACT, its pyramid)
// This file is never compiled in the build image (no OpenCV there); it is the one route by which the oracle can be
// pinned.  Record layout: u32 name length, name, u32 dtype code (0 u8, 1 i32, 2 f32), u32 ndim, u32 dims[], raw data.
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/imgproc.hpp>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

static std::ofstream g_out;

static void put(const std::string& name, uint32_t dtype, const std::vector<uint32_t>& dims, const void* data, size_t bytes)
{
    const uint32_t n = (uint32_t)name.size(), nd = (uint32_t)dims.size();
    g_out.write((const char*)&n, 4);
    g_out.write(name.data(), n);
    g_out.write((const char*)&dtype, 4);
    g_out.write((const char*)&nd, 4);
    g_out.write((const char*)dims.data(), 4 * nd);
    g_out.write((const char*)data, (std::streamsize)bytes);
}
static void put_mat_u8(const std::string& name, const cv::Mat& m)
{
    cv::Mat c = m.isContinuous() ? m : m.clone();
    put(name, 0, {(uint32_t)c.rows, (uint32_t)c.cols}, c.data, (size_t)c.rows * c.cols);
}
static void put_keypoints(const std::string& name, const std::vector<cv::KeyPoint>& k)
{
    std::vector<float> v;
    for(const auto& p : k)
    {
        v.push_back(p.pt.x), v.push_back(p.pt.y), v.push_back(p.response), v.push_back(p.angle), v.push_back((float)p.octave);
    }
    put(name, 2, {(uint32_t)k.size(), 5}, v.data(), v.size() * 4);
}

// frame.cpp:6-27 (toGrayScale): min(255.f, 0.299f*c0 + 0.587f*c1 + 0.114f*c2), truncated
static cv::Mat to_gray(const std::vector<uint8_t>& bgr, int w, int h)
{
    cv::Mat g(h, w, CV_8UC1);
    for(int i = 0; i < w * h; ++i)
    {
        const float v = 0.299f * bgr[3 * i] + 0.587f * bgr[3 * i + 1] + 0.114f * bgr[3 * i + 2];
        g.data[i] = (uint8_t)std::min(255.f, v);
    }
    return g;
}

int main(int argc, char** argv)
{
    if(argc != 6)
    {
        std::fprintf(stderr, "usage: %s frame0.bgr frame1.bgr width height out.bin\n", argv[0]);
        return 2;
    }
    const int W = std::atoi(argv[3]), H = std::atoi(argv[4]);
    g_out.open(argv[5], std::ios::binary);
    std::vector<cv::Mat> descs;
    for(int f = 0; f < 2; ++f)
    {
        std::vector<uint8_t> bgr((size_t)W * H * 3);
        std::ifstream in(argv[1 + f], std::ios::binary);
        if(!in.read((char*)bgr.data(), (std::streamsize)bgr.size()))
        {
            std::fprintf(stderr, "cannot read %s\n", argv[1 + f]);
            return 3;
        }
        const std::string F = "f" + std::to_string(f) + "_";
        const cv::Mat gray = to_gray(bgr, W, H);
        put_mat_u8(F + "gray", gray);

        // in-tree pyramid: float32 scale chain, sizes round(dim / scale) in double, chained INTER_LINEAR (:411-420, :830-841)
        std::vector<cv::Mat> pyr{gray};
        float scale = 1.0f;
        for(int l = 1; l < 8; ++l)
        {
            scale = 1.2f * scale;
            const cv::Size sz((int)std::round(W * 1.0 / (double)scale), (int)std::round(H * 1.0 / (double)scale));
            cv::Mat d;
            cv::resize(pyr.back(), d, sz, 0, 0, cv::INTER_LINEAR);
            pyr.push_back(d);
            put_mat_u8(F + "linear_L" + std::to_string(l), d);
        }
        // cv::ORB's pyramid: INTER_LINEAR_EXACT, sizes cvRound(dim * (1.0f / (float)pow(1.2f as double, l)))
        std::vector<cv::Mat> epyr{gray};
        for(int l = 1; l < 8; ++l)
        {
            const float s = (float)std::pow((double)1.2f, (double)l), inv = 1.0f / s;
            cv::Mat d;
            cv::resize(epyr.back(), d, cv::Size(cvRound(W * inv), cvRound(H * inv)), 0, 0, cv::INTER_LINEAR_EXACT);
            epyr.push_back(d);
            put_mat_u8(F + "exact_L" + std::to_string(l), d);
        }
        for(int l = 0; l < 8; l += 3)
        {
            cv::Mat b;
            cv::GaussianBlur(pyr[l], b, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
            put_mat_u8(F + "blur_L" + std::to_string(l), b);
            std::vector<cv::KeyPoint> k;
            cv::FAST(pyr[l], k, 20, true);
            put_keypoints(F + "fast20_L" + std::to_string(l), k);
        }
        // FAST per 70x70 cell (origin 19 + 64 j), thresholds 20 and 7, first two cell rows of level 0
        for(int i = 0; i < 2; ++i)
            for(int j = 0; j < 9; ++j)
                for(int thr : {20, 7})
                {
                    std::vector<cv::KeyPoint> k;
                    cv::FAST(gray(cv::Rect(19 + 64 * j, 19 + 64 * i, 70, 70)), k, thr, true);
                    put_keypoints(F + "cell_" + std::to_string(i) + "_" + std::to_string(j) + "_t" + std::to_string(thr), k);
                }
        // the whole cv::ORB detector (orb_feature.cpp:25,40)
        std::vector<cv::KeyPoint> kps;
        cv::Mat desc;
        cv::ORB::create(1000)->detectAndCompute(gray, cv::noArray(), kps, desc);
        put_keypoints(F + "orb_keypoints", kps);
        put_mat_u8(F + "orb_descriptors", desc);
        descs.push_back(desc);
    }
    // fastAtan2 on a grid of (y, x) including the axes and the diagonals
    {
        std::vector<float> v;
        for(int y = -40; y <= 40; ++y)
            for(int x = -40; x <= 40; ++x)
                v.push_back(cv::fastAtan2((float)(y * 977), (float)(x * 1013)));
        put("fast_atan2", 2, {81, 81}, v.data(), v.size() * 4);
    }
    // knnMatch(query = frame 0's descriptors, train = frame 1's, k = 2) (orb_feature.cpp:96)
    {
        std::vector<std::vector<cv::DMatch>> m;
        cv::DescriptorMatcher::create(cv::DescriptorMatcher::BRUTEFORCE_HAMMING)->knnMatch(descs[0], descs[1], m, 2);
        std::vector<int32_t> v;
        for(const auto& q : m)
            for(int k = 0; k < 2; ++k)
            {
                v.push_back(k < (int)q.size() ? q[k].trainIdx : -1);
                v.push_back(k < (int)q.size() ? (int32_t)q[k].distance : -1);
            }
        put("knn2", 1, {(uint32_t)m.size(), 4}, v.data(), v.size() * 4);
    }
    std::printf("wrote %s (OpenCV %s)\n", argv[5], CV_VERSION);
    return 0;
}
