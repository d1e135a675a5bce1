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
#include <algorithm>
#include <cstdint>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
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

// The same filter for a whole sequence in one call: what a caller looping `matchGMS` over image pairs (FeatureMatchUtil.cpp:66-69
// once per pair) switches to for throughput. keypoints[f] / sizes[f] describe frame f; pair p filters matches1to2[p] between frames
// pairs[p].first (query side) and pairs[p].second (train side); matchesGMS[p] receives the survivors. Host vectors in, host vectors
// out: the library stages them through pinned memory on two streams (gms_filter_host_batch) and keeps the frames resident on the GPU
// for the duration of the call. Pairs the reference has undefined behaviour on come back empty with ok[p] = false (if given).
inline void matchGMSBatch(const std::vector<Size>& sizes, const std::vector<std::vector<KeyPoint>>& keypoints,
                          const std::vector<std::pair<int, int>>& pairs, const std::vector<std::vector<DMatch>>& matches1to2,
                          std::vector<std::vector<DMatch>>& matchesGMS, const bool withRotation = false, const bool withScale = false,
                          const double thresholdFactor = 6.0, std::vector<bool>* ok = nullptr)
{
    if (sizes.size() != keypoints.size() || pairs.size() != matches1to2.size()) throw std::invalid_argument("mi355::matchGMSBatch: sizes");
    static gms_ctx* ctx = nullptr;  // one context per process, created on first use (by exactly one of the threads that race here)
    static int ctx_rc = GMS_OK;
    static std::once_flag ctx_once;
    std::call_once(ctx_once, [] { ctx_rc = gms_ctx_create(0, &ctx); });
    if (ctx_rc != GMS_OK || !ctx) throw std::runtime_error(std::string("mi355::matchGMSBatch: ") + gms_error_string(ctx_rc));
    std::vector<int64_t> frame_off(keypoints.size() + 1, 0);
    std::vector<int32_t> wh(2 * keypoints.size());
    for (size_t f = 0; f < keypoints.size(); ++f) {
        frame_off[f + 1] = frame_off[f] + (int64_t)keypoints[f].size();
        wh[2 * f] = sizes[f].width;
        wh[2 * f + 1] = sizes[f].height;
    }
    std::vector<KeyPoint> kp_all((size_t)frame_off.back());
    for (size_t f = 0; f < keypoints.size(); ++f) std::copy(keypoints[f].begin(), keypoints[f].end(), kp_all.begin() + frame_off[f]);
    std::vector<gms_pair> prs(pairs.size());
    int64_t total = 0;
    for (size_t p = 0; p < pairs.size(); ++p) {
        prs[p] = gms_pair{pairs[p].first, pairs[p].second, (int32_t)matches1to2[p].size(), 0, total};
        total += (int64_t)matches1to2[p].size();
    }
    std::vector<DMatch> m_all((size_t)total), out_all((size_t)total);
    for (size_t p = 0; p < pairs.size(); ++p) std::copy(matches1to2[p].begin(), matches1to2[p].end(), m_all.begin() + prs[p].match_off);
    std::vector<gms_pair_result> res(pairs.size());
    const int rc = gms_filter_host_batch(ctx, reinterpret_cast<const gms_keypoint*>(kp_all.data()), frame_off.data(), wh.data(),
                                         (int)keypoints.size(), prs.data(), (int)prs.size(), reinterpret_cast<const gms_dmatch*>(m_all.data()),
                                         withRotation ? 1 : 0, withScale ? 1 : 0, thresholdFactor,
                                         reinterpret_cast<gms_dmatch*>(out_all.data()), res.data());
    if (rc != GMS_OK) throw std::runtime_error(std::string("mi355::matchGMSBatch: ") + gms_error_string(rc));
    matchesGMS.assign(pairs.size(), std::vector<DMatch>());
    if (ok) ok->assign(pairs.size(), true);
    for (size_t p = 0; p < pairs.size(); ++p) {
        if (res[p].status != GMS_OK) {
            if (ok) (*ok)[p] = false;
            continue;
        }
        matchesGMS[p].assign(out_all.begin() + prs[p].match_off, out_all.begin() + prs[p].match_off + res[p].n_inliers);
    }
}

}  // namespace mi355
