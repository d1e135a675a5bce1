// mi355_gms.hpp -- header-only C++ face of the C ABI (include/gms.h) with the reference's exact signature:
//
//   cv::xfeatures2d::matchGMS(const Size&, const Size&, const std::vector<KeyPoint>&, const std::vector<KeyPoint>&,
//                             const std::vector<DMatch>&, std::vector<DMatch>&, bool = false, bool = false, double = 6.0)
//
// (reference call sites: SfM-GMS/SfM-GMS/FeatureMatchUtil.cpp:69, DisparityUtil.cpp:149, :299).
// With OpenCV headers present the cv:: types are used directly (cv::KeyPoint and cv::DMatch are
// layout-identical to gms_keypoint / gms_dmatch); without them the same-shaped PODs below stand in, so the
// call sites compile unchanged apart from the namespace. Link with libgms_hip.so.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "gms.h"

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define MI355_GMS_HAVE_OPENCV 1
#endif
#endif

namespace mi355 {

#ifdef MI355_GMS_HAVE_OPENCV
using Size = cv::Size;
using KeyPoint = cv::KeyPoint;
using DMatch = cv::DMatch;
#else
struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
};
struct Point2f {
    float x = 0, y = 0;
};
struct KeyPoint {  // cv::KeyPoint
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
struct DMatch {  // cv::DMatch
    int queryIdx = -1, trainIdx = -1, imgIdx = -1;
    float distance = 0;
};
#endif

static_assert(sizeof(KeyPoint) == sizeof(gms_keypoint), "KeyPoint must be 28 bytes like cv::KeyPoint");
static_assert(sizeof(DMatch) == sizeof(gms_dmatch), "DMatch must be 16 bytes like cv::DMatch");

// Same arguments, same output contract (matchesGMS is cleared, then receives the surviving matches verbatim and in
// input order). Where the reference has undefined behaviour (bad indices, points outside the image) or where no
// GPU is available this throws instead; there is no CPU fallback.
inline void matchGMS(const Size& size1, const Size& size2, const std::vector<KeyPoint>& keypoints1,
                     const std::vector<KeyPoint>& keypoints2, const std::vector<DMatch>& matches1to2,
                     std::vector<DMatch>& matchesGMS, const bool withRotation = false, const bool withScale = false,
                     const double thresholdFactor = 6.0)
{
    std::vector<DMatch> out(matches1to2.size());
    int n_out = 0;
    const int rc = gms_match(reinterpret_cast<const gms_keypoint*>(keypoints1.data()), (int)keypoints1.size(), size1.width,
                             size1.height, reinterpret_cast<const gms_keypoint*>(keypoints2.data()),
                             (int)keypoints2.size(), size2.width, size2.height,
                             reinterpret_cast<const gms_dmatch*>(matches1to2.data()), (int)matches1to2.size(),
                             withRotation ? 1 : 0, withScale ? 1 : 0, thresholdFactor,
                             reinterpret_cast<gms_dmatch*>(out.data()), &n_out);
    if (rc != GMS_OK) throw std::runtime_error(std::string("mi355::matchGMS: ") + gms_error_string(rc));
    out.resize((size_t)n_out);
    matchesGMS.swap(out);
}

}  // namespace mi355
