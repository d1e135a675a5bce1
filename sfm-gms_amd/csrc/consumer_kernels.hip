// consumer_kernels.hip -- the two consumers of the filtered matches, on the device-resident output of gms_filter_device:
//
//   match-based disparity + RMS    DisparityUtil.cpp:179-201 (matchBasedDispCalculate): a W x H byte map, 255 = "no match"; for
//       every surviving match in order, map(y, x) = (uchar)|x - x1| with x, y, x1 the integer parts of the matched keypoints'
//       coordinates (a later match overwrites an earlier one on the same pixel); then over the ground-truth image the count,
//       the sum of squares and the maximum of a = |map - gt / disp_ratio| wherever map != 255, rms = sqrt(sum / count).
//       All integer work: the scatter resolves same-pixel matches by match index (one atomicMax of (index + 1) << 8 | value,
//       so the LAST match wins as in the reference's loop), the statistics are integer sums.
//   matched-point gather           SfMUtil.cpp:25-35 (structureFromMotion): coords1[i] = kpts1[queryIdx].pt, coords2[i] =
//       kpts2[trainIdx].pt for the surviving matches -- the two Point2f arrays findEssentialMat / recoverPose /
//       undistortPoints consume (SfMUtil.cpp:39,45,78-79).
// The number of surviving matches is known on the device only (gms_pair_result::n_inliers): both take a device pointer to it
// and a host-known upper bound for the grid.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gms_kernels.h"
#include "twoview_core.h"

namespace gms {
namespace {

__global__ void __launch_bounds__(256)
disparity_scatter_kernel(const gms_keypoint* __restrict__ kp1, int n1, const gms_keypoint* __restrict__ kp2, int n2,
                         const gms_dmatch* __restrict__ matches, const int32_t* __restrict__ n_matches, int cap, int w, int h,
                         uint32_t* __restrict__ work, gms_disparity_stats* __restrict__ stats)
{
    const int n = min(*n_matches, cap);
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n) return;
    const gms_dmatch m = matches[i];
    bool bad = (uint32_t)m.queryIdx >= (uint32_t)n1 || (uint32_t)m.trainIdx >= (uint32_t)n2;
    if (!bad) {
        // int x = pt.x: float -> int truncation (cvttss2si); coordinates are finite and inside the image on the parity domain
        const float fx = kp1[m.queryIdx].x, fy = kp1[m.queryIdx].y, fx1 = kp2[m.trainIdx].x;
        const bool finite = fabsf(fx) < 1e9f && fabsf(fy) < 1e9f && fabsf(fx1) < 1e9f;
        const int x = finite ? (int)fx : -1, y = finite ? (int)fy : -1, x1 = finite ? (int)fx1 : 0;
        bad = !finite || (uint32_t)x >= (uint32_t)w || (uint32_t)y >= (uint32_t)h;  // Mat::at outside the image: UB in the reference
        if (!bad) {
            const uint32_t v = (uint32_t)abs(x - x1) & 255u;  // int -> uchar keeps the low byte
            atomicMax(&work[(size_t)y * w + x], ((uint32_t)(i + 1) << 8) | v);
        }
    }
    if (bad) stats->status = GMS_ERR_DOMAIN;  // benign race: every writer stores the same value
}

__global__ void __launch_bounds__(256)
disparity_finish_kernel(const uint32_t* __restrict__ work, const uint8_t* __restrict__ gt, int disp_ratio, int64_t n_pix,
                        uint8_t* __restrict__ disparity, gms_disparity_stats* __restrict__ stats)
{
    __shared__ unsigned long long s_sum[4], s_cnt[4];
    __shared__ uint32_t s_max[4];
    unsigned long long sum = 0, cnt = 0;
    uint32_t mx = 0;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_pix; p += (int64_t)gridDim.x * 256) {
        const uint32_t wv = work[p];
        const uint32_t d = wv ? (wv & 255u) : 255u;
        disparity[p] = (uint8_t)d;
        if (d != 255u && gt != nullptr) {
            const uint32_t a = (uint32_t)abs((int)d - (int)gt[p] / disp_ratio);
            sum += (unsigned long long)a * a;
            cnt += 1;
            mx = max(mx, a);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        sum += __shfl_xor(sum, o);
        cnt += __shfl_xor(cnt, o);
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_sum[wave] = sum;
        s_cnt[wave] = cnt;
        s_max[wave] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        sum = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        mx = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        if (cnt) {
            atomicAdd(reinterpret_cast<unsigned long long*>(&stats->sum_sq), sum);
            atomicAdd(reinterpret_cast<unsigned long long*>(&stats->count), cnt);
            atomicMax(reinterpret_cast<uint32_t*>(&stats->max_abs), mx);
        }
    }
}

__global__ void __launch_bounds__(256)
gather_points_kernel(const gms_keypoint* __restrict__ kp1, int n1, const gms_keypoint* __restrict__ kp2, int n2,
                     const gms_dmatch* __restrict__ matches, const int32_t* __restrict__ n_matches, int cap,
                     float2* __restrict__ coords1, float2* __restrict__ coords2, int32_t* __restrict__ status)
{
    const int n = min(*n_matches, cap);
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n) return;
    const gms_dmatch m = matches[i];
    if ((uint32_t)m.queryIdx >= (uint32_t)n1 || (uint32_t)m.trainIdx >= (uint32_t)n2) {
        *status = GMS_ERR_DOMAIN;
        return;
    }
    coords1[i] = make_float2(kp1[m.queryIdx].x, kp1[m.queryIdx].y);
    coords2[i] = make_float2(kp2[m.trainIdx].x, kp2[m.trainIdx].y);
}

// ---- two-view triangulation + reprojection error (SfMUtil.cpp:76-82,128-143) --------------------------------------------------------
// Per surviving match: cv::undistortPoints with the camera matrix and (k1, k2, p1, p2, k3) -- the published five fixed-point
// iterations -- gives normalised coordinates; cv::triangulatePoints' homogeneous DLT (rows x P[2] - P[0], y P[2] - P[1] of both
// cameras, the right singular vector of the smallest singular value) gives X; SfMUtil.cpp:134-139 divides by X[3]. The arithmetic
// (undistort_point, dlt_point) is twoview_core.h. Floating point, not bit-exact against OpenCV's SVD: the tests state the tolerance.
struct CameraModel {
    tv::Camera c;
    double P1[12], P2[12];
};

__global__ void __launch_bounds__(256)
triangulate_kernel(CameraModel cam, const float2* __restrict__ coords1, const float2* __restrict__ coords2,
                   const int32_t* __restrict__ n_matches, int cap, double* __restrict__ points3d, gms_triangulation_stats* __restrict__ stats)
{
    const int n = min(*n_matches, cap);
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    double e1 = 0.0, e2 = 0.0;
    unsigned long long cnt = 0, behind = 0;
    if (i < n) {
        double x1, y1, x2, y2;
        tv::undistort_point(cam.c, (double)coords1[i].x, (double)coords1[i].y, x1, y1);
        tv::undistort_point(cam.c, (double)coords2[i].x, (double)coords2[i].y, x2, y2);
        double X[4];
        tv::dlt_point(cam.P1, cam.P2, x1, y1, x2, y2, X);
        const double px = X[0] / X[3], py = X[1] / X[3], pz = X[2] / X[3];  // SfMUtil.cpp:134-137
        points3d[3 * (size_t)i] = px;
        points3d[3 * (size_t)i + 1] = py;
        points3d[3 * (size_t)i + 2] = pz;
        // reprojection through both cameras, in normalised image coordinates
        const double w1 = cam.P1[8] * px + cam.P1[9] * py + cam.P1[10] * pz + cam.P1[11];
        const double w2 = cam.P2[8] * px + cam.P2[9] * py + cam.P2[10] * pz + cam.P2[11];
        const double u1 = (cam.P1[0] * px + cam.P1[1] * py + cam.P1[2] * pz + cam.P1[3]) / w1 - x1;
        const double v1 = (cam.P1[4] * px + cam.P1[5] * py + cam.P1[6] * pz + cam.P1[7]) / w1 - y1;
        const double u2 = (cam.P2[0] * px + cam.P2[1] * py + cam.P2[2] * pz + cam.P2[3]) / w2 - x2;
        const double v2 = (cam.P2[4] * px + cam.P2[5] * py + cam.P2[6] * pz + cam.P2[7]) / w2 - y2;
        const bool finite = isfinite(px) && isfinite(py) && isfinite(pz) && isfinite(u1) && isfinite(v1) && isfinite(u2) && isfinite(v2);
        if (finite) {
            e1 = u1 * u1 + v1 * v1;
            e2 = u2 * u2 + v2 * v2;
            cnt = 1;
            behind = (w1 <= 0.0 || w2 <= 0.0) ? 1 : 0;  // fails the cheirality test of recoverPose (SfMUtil.cpp:45)
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        e1 += __shfl_xor(e1, o);
        e2 += __shfl_xor(e2, o);
        cnt += __shfl_xor(cnt, o);
        behind += __shfl_xor(behind, o);
    }
    if ((threadIdx.x & 63) == 0 && cnt) {
        atomicAdd(&stats->sum_sq_err1, e1);
        atomicAdd(&stats->sum_sq_err2, e2);
        atomicAdd(reinterpret_cast<unsigned long long*>(&stats->count), cnt);
        atomicAdd(reinterpret_cast<unsigned long long*>(&stats->behind), behind);
    }
}

// ---- cv::recoverPose (SfMUtil.cpp:45) -----------------------------------------------------------------------------------------------
// The four (R, t) an essential matrix decomposes into (the host does the 3 x 3 decomposition: gms_capi.cpp) are tried on every
// correspondence as OpenCV 4.5.2 does: triangulate with P0 = [I|0] and P = [R|t] in normalised coordinates ((x - cx) / fx, no
// distortion model), keep the point if its depth is positive and below the distance threshold in both cameras. One byte of four
// vote bits per correspondence, four counters; pose_pick_kernel then takes the first hypothesis with the most votes in the order
// (R1, t), (R2, t), (R1, -t), (R2, -t) and writes the mask (255 / 0, as cv::Mat comparisons do).
struct PoseModel {
    double fx, fy, cx, cy, dist_thresh;
    double P[4][12];
};

__global__ void __launch_bounds__(256)
pose_votes_kernel(PoseModel pm, const float2* __restrict__ coords1, const float2* __restrict__ coords2, const int32_t* __restrict__ n_matches,
                  int cap, const uint8_t* __restrict__ in_mask, uint8_t* __restrict__ votes, unsigned long long* __restrict__ counts)
{
    const int n = min(*n_matches, cap);
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    uint32_t bits = 0;
    if (i < n && (in_mask == nullptr || in_mask[i] != 0)) {
        const double x1 = ((double)coords1[i].x - pm.cx) / pm.fx, y1 = ((double)coords1[i].y - pm.cy) / pm.fy;
        const double x2 = ((double)coords2[i].x - pm.cx) / pm.fx, y2 = ((double)coords2[i].y - pm.cy) / pm.fy;
        bits = tv::pose_votes(pm.P, pm.dist_thresh, x1, y1, x2, y2);
    }
    if (i < n) votes[i] = (uint8_t)bits;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const unsigned long long c = (unsigned long long)__popcll(__ballot((bits >> h) & 1u));
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts[h], c);
    }
}

__global__ void __launch_bounds__(256)
pose_pick_kernel(PoseModel pm, const int32_t* __restrict__ n_matches, int cap, const uint8_t* __restrict__ votes,
                 const unsigned long long* __restrict__ counts, const uint8_t* in_mask, gms_pose* __restrict__ pose, uint8_t* out_mask)
{
    const unsigned long long g0 = counts[0], g1 = counts[1], g2 = counts[2], g3 = counts[3];
    int w;  // recoverPose's chain of comparisons
    if (g0 >= g1 && g0 >= g2 && g0 >= g3) w = 0;
    else if (g1 >= g0 && g1 >= g2 && g1 >= g3) w = 1;
    else if (g2 >= g0 && g2 >= g1 && g2 >= g3) w = 2;
    else w = 3;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) pose->R[3 * r + c] = pm.P[w][4 * r + c];
            pose->t[r] = pm.P[w][4 * r + 3];
        }
        pose->n_good = (int32_t)(w == 0 ? g0 : w == 1 ? g1 : w == 2 ? g2 : g3);
        pose->which = w;
    }
    if (out_mask != nullptr) {
        const int n = min(*n_matches, cap);
        const int i = (int)(blockIdx.x * 256u + threadIdx.x);
        // cv::recoverPose: bitwise_and(mask, hypothesis mask) -- the caller's byte where the point passes (255 without an input mask)
        if (i < n) out_mask[i] = ((votes[i] >> w) & 1u) ? (in_mask ? in_mask[i] : (uint8_t)255) : (uint8_t)0;
    }
}

}  // namespace

// d_work: max_matches vote bytes, then (8-byte aligned) four 64-bit counters
hipError_t launch_recover_pose(const double* camera, const double P[4][12], double dist_thresh, const float* d_coords1, const float* d_coords2,
                               const int32_t* d_n_matches, int max_matches, const uint8_t* d_in_mask, gms_pose* d_pose, uint8_t* d_out_mask,
                               void* d_work, hipStream_t stream)
{
    PoseModel pm;
    pm.fx = camera[0]; pm.fy = camera[1]; pm.cx = camera[2]; pm.cy = camera[3];
    pm.dist_thresh = dist_thresh;
    for (int h = 0; h < 4; ++h)
        for (int k = 0; k < 12; ++k) pm.P[h][k] = P[h][k];
    uint8_t* votes = reinterpret_cast<uint8_t*>(d_work);
    unsigned long long* counts = reinterpret_cast<unsigned long long*>(votes + (((size_t)max_matches + 7) & ~(size_t)7));
    hipError_t e = hipMemsetAsync(counts, 0, 4 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)((max_matches + 255) / 256);
    if (max_matches > 0)
        hipLaunchKernelGGL(pose_votes_kernel, dim3(blocks), dim3(256), 0, stream, pm, reinterpret_cast<const float2*>(d_coords1),
                           reinterpret_cast<const float2*>(d_coords2), d_n_matches, max_matches, d_in_mask, votes, counts);
    hipLaunchKernelGGL(pose_pick_kernel, dim3(blocks ? blocks : 1u), dim3(256), 0, stream, pm, d_n_matches, max_matches, votes, counts, d_in_mask, d_pose,
                       max_matches > 0 ? d_out_mask : nullptr);
    return hipGetLastError();
}

hipError_t launch_triangulate(const double* camera, const double* dist, const double* P1, const double* P2, const float* d_coords1,
                              const float* d_coords2, const int32_t* d_n_matches, int max_matches, double* d_points3d,
                              gms_triangulation_stats* d_stats, hipStream_t stream)
{
    CameraModel cam;
    cam.c.fx = camera[0]; cam.c.fy = camera[1]; cam.c.cx = camera[2]; cam.c.cy = camera[3];
    cam.c.k1 = dist ? dist[0] : 0.0; cam.c.k2 = dist ? dist[1] : 0.0; cam.c.p1 = dist ? dist[2] : 0.0; cam.c.p2 = dist ? dist[3] : 0.0;
    cam.c.k3 = dist ? dist[4] : 0.0;
    for (int k = 0; k < 12; ++k) {
        cam.P1[k] = P1[k];
        cam.P2[k] = P2[k];
    }
    hipError_t e = hipMemsetAsync(d_stats, 0, sizeof(gms_triangulation_stats), stream);
    if (e != hipSuccess) return e;
    if (max_matches > 0)
        hipLaunchKernelGGL(triangulate_kernel, dim3((unsigned)((max_matches + 255) / 256)), dim3(256), 0, stream, cam,
                           reinterpret_cast<const float2*>(d_coords1), reinterpret_cast<const float2*>(d_coords2), d_n_matches, max_matches,
                           d_points3d, d_stats);
    return hipGetLastError();
}

hipError_t launch_disparity(const gms_keypoint* d_kp1, int n1, const gms_keypoint* d_kp2, int n2, const gms_dmatch* d_matches,
                            const int32_t* d_n_matches, int max_matches, int w, int h, const uint8_t* d_gt, int disp_ratio,
                            uint8_t* d_disparity, uint32_t* d_work, gms_disparity_stats* d_stats, hipStream_t stream)
{
    const int64_t n_pix = (int64_t)w * h;
    hipError_t e = hipMemsetAsync(d_work, 0, (size_t)n_pix * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_stats, 0, sizeof(gms_disparity_stats), stream);
    if (e != hipSuccess) return e;
    if (max_matches > 0)
        hipLaunchKernelGGL(disparity_scatter_kernel, dim3((unsigned)((max_matches + 255) / 256)), dim3(256), 0, stream, d_kp1, n1, d_kp2, n2,
                           d_matches, d_n_matches, max_matches, w, h, d_work, d_stats);
    int64_t blocks = (n_pix + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(disparity_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_work, d_gt, disp_ratio, n_pix, d_disparity,
                       d_stats);
    return hipGetLastError();
}

hipError_t launch_gather_points(const gms_keypoint* d_kp1, int n1, const gms_keypoint* d_kp2, int n2, const gms_dmatch* d_matches,
                                const int32_t* d_n_matches, int max_matches, float* d_coords1, float* d_coords2, int32_t* d_status,
                                hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_status, 0, 4, stream);
    if (e != hipSuccess) return e;
    if (max_matches > 0)
        hipLaunchKernelGGL(gather_points_kernel, dim3((unsigned)((max_matches + 255) / 256)), dim3(256), 0, stream, d_kp1, n1, d_kp2, n2,
                           d_matches, d_n_matches, max_matches, reinterpret_cast<float2*>(d_coords1), reinterpret_cast<float2*>(d_coords2),
                           d_status);
    return hipGetLastError();
}

}  // namespace gms
