// std_order.cpp — what the REAL C++ library of this image does in KeyPointsFilter::retainBest (OpenCV features2d,
// keypoint.cpp): std::nth_element + std::partition on a vector of 28-byte records compared by a float member.  Test
// infrastructure: tests/test_oracle_std_order.py builds it with g++ and compares oracle/mslam_oracle.c's restatement of
// libstdc++'s algorithms with it.  Nothing of the reference or of OpenCV is in here: the call sequence is the published one.
#include <algorithm>
#include <cstdint>
#include <vector>

namespace
{
struct KeyPoint // layout of cv::KeyPoint: pt (2 floats), size, angle, response, octave, class_id
{
    float x, y, size, angle, response;
    int octave, class_id;
};
struct ResponseGreater
{
    bool operator()(const KeyPoint& a, const KeyPoint& b) const { return a.response > b.response; }
};
struct ResponseGreaterOrEqual
{
    float value;
    bool operator()(const KeyPoint& k) const { return k.response >= value; }
};
} // namespace

extern "C" int real_std_retain_best_order(const float* response, int n, int n_points, int32_t* order)
{
    std::vector<KeyPoint> kp((size_t)n);
    for(int i = 0; i < n; ++i)
        kp[(size_t)i] = KeyPoint{0.f, 0.f, 7.f, -1.f, response[i], 0, i};
    if(n_points >= 0 && kp.size() > (size_t)n_points)
    {
        if(n_points == 0)
            kp.clear();
        else
        {
            std::nth_element(kp.begin(), kp.begin() + n_points - 1, kp.end(), ResponseGreater());
            const float ambiguous = kp[(size_t)n_points - 1].response;
            std::vector<KeyPoint>::const_iterator new_end =
                std::partition(kp.begin() + n_points, kp.end(), ResponseGreaterOrEqual{ambiguous});
            kp.resize((size_t)(new_end - kp.begin()));
        }
    }
    for(size_t i = 0; i < kp.size(); ++i)
        order[i] = kp[i].class_id;
    return (int)kp.size();
}
