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

}  // namespace

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
