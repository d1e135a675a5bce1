// twoview_kernels.hip -- the consumers of the filtered matches for a whole BATCH of pairs, on the same pair table and at the same
// offsets as gms_filter_device: what structureFromMotion does with the survivors of one pair (SfMUtil.cpp:25-82), per launch for
// every pair of the batch.
//
//   gather_batch_kernel        SfMUtil.cpp:25-35   coords1[i] = kpts1[queryIdx].pt, coords2[i] = kpts2[trainIdx].pt
//   find_essential_kernel      SfMUtil.cpp:39      cv::findEssentialMat(coords1, coords2, cameraMatrix, RANSAC, prob, threshold, inliers)
//   recover_pose_batch_kernel  SfMUtil.cpp:45      cv::recoverPose(E, coords1, coords2, cameraMatrix, R, t, inliers)
//   triangulate_batch_kernel   SfMUtil.cpp:65-82, 128-143   the inliers compacted in order, cv::undistortPoints, cv::triangulatePoints,
//                                                  division by the fourth coordinate (+ the reprojection error sums of BASELINE config 5)
//   disparity_batch_*          DisparityUtil.cpp:170-201   the match-based disparity map + RMS statistics, one map per pair
//
// One workgroup per pair for the three geometry kernels (a pair's RANSAC is a sequential decision process over parallel work:
// samples are drawn by one lane exactly in cv::RNG's order, solved sixteen at a time by sixteen lanes -- the five-point solver of
// twoview_core.h with its matrices in LDS --, scored against every correspondence by the whole workgroup, and the reference's
// "first model with a strictly larger inlier count wins, then the iteration bound shrinks" is replayed over the round's models in
// order by one lane). fp64 throughout; the arithmetic itself is twoview_core.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gms_kernels.h"
#include "twoview_core.h"

namespace gms {
namespace {

constexpr int kTvThreads = 256;
constexpr int kSolvers = 16;  // samples solved per round

// per-lane view of an LDS array whose element i of solver lane s sits at base[i * kSolvers + s]: the sixteen lanes of a solver
// wave touch sixteen consecutive doubles whatever (run-time) element each of them indexes in lockstep
template <int STRIDE>
struct LdsLaneT {
    double* base;
    __device__ __forceinline__ double& operator()(int i) const { return base[i * STRIDE]; }
};
using LdsLane = LdsLaneT<kSolvers>;

__device__ __forceinline__ int pair_count(const gms_pair& pr, const gms_two_view& t)
{
    return max(0, min(t.n_points, pr.m));
}

// ---- SfMUtil.cpp:25-35 -----------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
gather_batch_kernel(const gms_keypoint* __restrict__ kp, const int64_t* __restrict__ frame_off, int n_frames,
                    const gms_pair* __restrict__ pairs, const gms_dmatch* __restrict__ filtered, const gms_pair_result* __restrict__ results,
                    float2* __restrict__ coords1, float2* __restrict__ coords2, gms_two_view* __restrict__ tv)
{
    const int p = (int)blockIdx.y;
    const gms_pair pr = pairs[p];
    const gms_pair_result res = results[p];
    const bool pair_ok = res.status == GMS_OK && pr.frame_a >= 0 && pr.frame_a < n_frames && pr.frame_b >= 0 && pr.frame_b < n_frames;
    const int n = pair_ok ? max(0, min(res.n_inliers, pr.m)) : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        tv[p].n_points = n;
        if (!pair_ok) tv[p].status = res.status != GMS_OK ? res.status : GMS_ERR_DOMAIN;
    }
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n) return;
    const int64_t oa = frame_off[pr.frame_a], ob = frame_off[pr.frame_b];
    const int na = (int)(frame_off[pr.frame_a + 1] - oa), nb = (int)(frame_off[pr.frame_b + 1] - ob);
    const gms_dmatch m = filtered[pr.match_off + i];
    if ((uint32_t)m.queryIdx >= (uint32_t)na || (uint32_t)m.trainIdx >= (uint32_t)nb) {
        tv[p].status = GMS_ERR_DOMAIN;  // benign race: every writer stores the same value
        coords1[pr.match_off + i] = make_float2(0.0f, 0.0f);
        coords2[pr.match_off + i] = make_float2(0.0f, 0.0f);
        return;
    }
    const gms_keypoint a = kp[oa + m.queryIdx], b = kp[ob + m.trainIdx];
    coords1[pr.match_off + i] = make_float2(a.x, a.y);
    coords2[pr.match_off + i] = make_float2(b.x, b.y);
}

// ---- cv::findEssentialMat(..., RANSAC, prob, threshold, mask) (SfMUtil.cpp:39) ------------------------------------------------------------
struct EssentialParams {
    double fx, fy, cx, cy;
    double prob, threshold;  // threshold in pixels, as the caller gives it
    int max_iters;
};

// THREADS x SOLVERS. The minimal solver is ONE wave's latency chain (256 vector registers: two waves per SIMD, eight per CU) however
// many of its lanes hold a sample; the scoring of a round's models is throughput work for all the waves of the workgroup. What a CU can
// overlap is workgroups, and their number is set by the LDS of the solver lanes (3.1 KB per sample). Round 3: 256 threads and sixteen
// samples, two pairs per CU (14.3 -> 10.9 ms per 1024 pairs for the stage against one). Round 4 swept the shape (tools/poses_bench.py
// with GMS_TV_GEOM, two-view stage per 1024 pairs): 256 x 16 6.50 ms, 256 x 8 6.57, 192 x 12 6.89, 128 x 16 7.37 and 128 x 14 7.42
// (three workgroups fit), 128 x 10 5.82, 128 x 8 6.16, 64 x 12 5.75, 64 x 6 7.43, 64 x 4 7.90, **128 x 12 5.4** -- four pairs per
// CU, 37 KB each, one solver wave per SIMD and a second wave per pair for the scoring. At SfMUtil.cpp:39's confidence of 0.7 the mean
// iteration count is 3.2: twelve samples are one round for almost every pair.
template <int THREADS, int SOLVERS>
__global__ void __launch_bounds__(THREADS, 2)  // (two waves per SIMD: the 256-register budget -- eight waves per CU)
find_essential_kernel(EssentialParams prm, const gms_pair* __restrict__ pairs, const float2* __restrict__ coords1,
                      const float2* __restrict__ coords2, uint8_t* __restrict__ mask, gms_two_view* __restrict__ tv)
{
    __shared__ double s_A[200 * SOLVERS];
    __shared__ double s_basis[36 * SOLVERS];
    __shared__ double s_work[60 * SOLVERS];
    __shared__ double s_models[SOLVERS * 10 * 9];  // [solver][model][9]
    __shared__ double s_best[9];
    __shared__ int s_samples[SOLVERS * 5];
    __shared__ int s_nmodels[SOLVERS];
    __shared__ unsigned s_counts[SOLVERS * 10];
    __shared__ int s_ctl[4];  // [0] iteration bound, [1] iterations done, [2] best inlier count
    static_assert(SOLVERS <= 64, "the solver lanes in one wave");

    const int p = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63;
    const gms_pair pr = pairs[p];
    const int n = tv[p].status == GMS_OK ? pair_count(pr, tv[p]) : 0;
    const float2* __restrict__ c1 = coords1 + pr.match_off;
    const float2* __restrict__ c2 = coords2 + pr.match_off;
    uint8_t* __restrict__ mk = mask + pr.match_off;
    // findEssentialMat: threshold /= (fx + fy) / 2; findInliers: float t = (float)(thresh * thresh)
    const double thr = prm.threshold / ((prm.fx + prm.fy) / 2.0);
    const float t = (float)(thr * thr);
    auto norm1 = [&](int i, double& x, double& y) {
        const float2 v = c1[i];
        x = ((double)v.x - prm.cx) / prm.fx;
        y = ((double)v.y - prm.cy) / prm.fy;
    };
    auto norm2 = [&](int i, double& x, double& y) {
        const float2 v = c2[i];
        x = ((double)v.x - prm.cx) / prm.fx;
        y = ((double)v.y - prm.cy) / prm.fy;
    };
    auto solve = [&](int k) {  // lane k of the solver wave: the sample s_samples[5 k ..] -> s_models[k], s_nmodels[k]
        double x1[5], y1[5], x2[5], y2[5];
        for (int j = 0; j < 5; ++j) {
            const int idx = s_samples[5 * k + j];
            norm1(idx, x1[j], y1[j]);
            norm2(idx, x2[j], y2[j]);
        }
        using Lane = LdsLaneT<SOLVERS>;
        tv::FivePointMem<Lane> mem{Lane{s_A + k}, Lane{s_basis + k}, Lane{s_work + k}};
        struct Out {
            double* base;
            __device__ __forceinline__ double& operator()(int i) const { return base[i]; }
        } out{s_models + k * 90};
        s_nmodels[k] = tv::five_point(x1, y1, x2, y2, mem, out);
    };

    if (tid < 9) s_best[tid] = 0.0;
    if (tid == 0) {
        s_ctl[0] = max(prm.max_iters, 1);
        s_ctl[1] = 0;
        s_ctl[2] = 0;
    }
    __syncthreads();

    if (n == 5) {
        // RANSACPointSetRegistrator::run, count == modelPoints: the sample is the whole set; its model, every point an inlier
        // (with several real roots the reference hands back all of them stacked, which its callers cannot use: the first is kept)
        if (tid == 0) {
            for (int j = 0; j < 5; ++j) s_samples[j] = j;
            solve(0);
            if (s_nmodels[0] > 0) {
                for (int k = 0; k < 9; ++k) s_best[k] = s_models[k];
                s_ctl[2] = 5;
            }
        }
        __syncthreads();
    } else if (n > 5) {
        tv::CvRng rng;
        rng.seed(0xFFFFFFFFFFFFFFFFull);  // RNG rng((uint64)-1)
        // (A round solves sixteen samples in lockstep. Measured and dropped: a first round of four -- the lanes of a round run as many
        //  sweeps of the root finder as the slowest of them -- costs more in second rounds, each a whole solve's latency, than it saves:
        //  4.2 -> 5.0 ms per 1024 pairs at SfMUtil.cpp:39's confidence of 0.7, where the mean iteration count is 3.2.)
        for (int it0 = 0, width = SOLVERS;; it0 += width) {
            // (1) the samples of iterations it0 .. it0 + width - 1, in cv::RNG's order
            if (tid == 0)
                for (int k = 0; k < width; ++k) rng.sample5(n, s_samples + 5 * k);
            for (int i = tid; i < SOLVERS * 10; i += THREADS) s_counts[i] = 0u;
            __syncthreads();
            const int bound = s_ctl[0];
            const int live = min(width, bound - it0);  // iterations of this round that can still be reached
            // (2) sixteen minimal solves, one lane each
            if (tid < SOLVERS) {
                if (tid < live) solve(tid);
                else s_nmodels[tid] = 0;
            }
            __syncthreads();
            // (3) every model of the round against every correspondence
            for (int base = 0; base < n; base += THREADS) {
                const int i = base + tid;
                double x1 = 0.0, y1 = 0.0, x2 = 0.0, y2 = 0.0;
                if (i < n) {
                    norm1(i, x1, y1);
                    norm2(i, x2, y2);
                }
                for (int k = 0; k < live; ++k) {
                    const int nm = s_nmodels[k];
                    for (int j = 0; j < nm; ++j) {
                        const float err = tv::sampson_error(s_models + (k * 10 + j) * 9, x1, y1, x2, y2);
                        const unsigned long long in = __ballot(i < n && err <= t);
                        if (lane == 0 && in) atomicAdd(&s_counts[k * 10 + j], (unsigned)__popcll(in));
                    }
                }
            }
            __syncthreads();
            // (4) the reference's loop over the round's iterations, in order
            if (tid == 0) {
                int niters = s_ctl[0], best = s_ctl[2], it = it0;
                for (int k = 0; k < live && it < niters; ++k, ++it) {
                    for (int j = 0; j < s_nmodels[k]; ++j) {
                        const int good = (int)s_counts[k * 10 + j];
                        if (good > max(best, 4)) {  // goodCount > MAX(maxGoodCount, modelPoints - 1)
                            best = good;
                            for (int e = 0; e < 9; ++e) s_best[e] = s_models[(k * 10 + j) * 9 + e];
                            niters = tv::ransac_update_num_iters(prm.prob, (double)(n - good) / n, 5, niters);
                        }
                    }
                }
                s_ctl[0] = niters;
                s_ctl[1] = it;
                s_ctl[2] = best;
            }
            __syncthreads();
            if (s_ctl[1] >= s_ctl[0]) break;
        }
    }

    // the winner's inlier mask (findInliers' 1 / 0 bytes) and the record
    const int best_count = s_ctl[2];
    double E[9];
    for (int k = 0; k < 9; ++k) E[k] = s_best[k];
    for (int i = tid; i < n; i += THREADS) {
        uint8_t in = 0;
        if (best_count > 0) {
            if (n == 5) {
                in = 1;
            } else {
                double x1, y1, x2, y2;
                norm1(i, x1, y1);
                norm2(i, x2, y2);
                in = tv::sampson_error(E, x1, y1, x2, y2) <= t ? 1 : 0;
            }
        }
        mk[i] = in;
    }
    if (tid == 0) {
        gms_two_view& o = tv[p];
        if (best_count > 0) tv::canonical_sign(E);
        for (int k = 0; k < 9; ++k) o.E[k] = best_count > 0 ? E[k] : 0.0;
        o.n_ransac = best_count;
        o.ransac_iters = s_ctl[1];
        if (o.status == GMS_OK && best_count == 0) o.status = GMS_ERR_NO_MODEL;
    }
}

// Test hook: the minimal solver alone, as the RANSAC kernel runs it (sixteen lanes of a wave, matrices in LDS), on caller-given samples of
// normalised points: pts[20 s ..] = x1[5], y1[5], x2[5], y2[5] of sample s -> models[90 s ..], counts[s].
__global__ void __launch_bounds__(64)
five_point_selftest_kernel(const double* __restrict__ pts, int n_samples, double* __restrict__ models, int* __restrict__ counts)
{
    __shared__ double s_A[200 * kSolvers];
    __shared__ double s_basis[36 * kSolvers];
    __shared__ double s_work[60 * kSolvers];
    __shared__ double s_models[kSolvers * 90];
    const int k = (int)threadIdx.x, s = (int)blockIdx.x * kSolvers + k;
    if (k >= kSolvers || s >= n_samples) return;
    double x1[5], y1[5], x2[5], y2[5];
    for (int j = 0; j < 5; ++j) {
        x1[j] = pts[20 * s + j];
        y1[j] = pts[20 * s + 5 + j];
        x2[j] = pts[20 * s + 10 + j];
        y2[j] = pts[20 * s + 15 + j];
    }
    tv::FivePointMem<LdsLane> mem{LdsLane{s_A + k}, LdsLane{s_basis + k}, LdsLane{s_work + k}};
    struct Out {
        double* base;
        __device__ __forceinline__ double& operator()(int i) const { return base[i]; }
    } out{s_models + k * 90};
    const int n = tv::five_point(x1, y1, x2, y2, mem, out);
    counts[s] = n;
    for (int i = 0; i < 90; ++i) models[90 * (size_t)s + i] = i < 9 * n ? s_models[k * 90 + i] : 0.0;
}

// ---- cv::recoverPose(E, points1, points2, cameraMatrix, R, t, mask) (SfMUtil.cpp:45), OpenCV 4.5.2: distance threshold 50 -------------------
__global__ void __launch_bounds__(kTvThreads)
recover_pose_batch_kernel(double fx, double fy, double cx, double cy, double dist_thresh, int use_in_mask, const gms_pair* __restrict__ pairs,
                          const float2* __restrict__ coords1, const float2* __restrict__ coords2, uint8_t* __restrict__ mask,
                          gms_two_view* __restrict__ tv)
{
    __shared__ double s_P[4][12];
    __shared__ unsigned s_votes[4];
    __shared__ int s_ok, s_winner;
    // the four votes of every correspondence, kept from the counting pass for the mask pass (pairs up to kVoteCache points; beyond
    // that the winner's test is worked out again): the mask then IS the set that was counted, and the second triangulation is saved
    constexpr int kVoteCache = 16384;
    __shared__ uint8_t s_bits[kVoteCache];
    const int p = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63;
    const gms_pair pr = pairs[p];
    const bool have_e = tv[p].status == GMS_OK;
    const int n = have_e ? pair_count(pr, tv[p]) : 0;
    const float2* __restrict__ c1 = coords1 + pr.match_off;
    const float2* __restrict__ c2 = coords2 + pr.match_off;
    uint8_t* __restrict__ mk = mask + pr.match_off;
    if (tid < 4) s_votes[tid] = 0u;
    if (tid == 0) {
        double R1[9], R2[9], t[3], E[9];
        for (int k = 0; k < 9; ++k) E[k] = tv[p].E[k];
        const bool ok = have_e && tv::decompose_essential(E, R1, R2, t);
        s_ok = ok ? 1 : 0;
        if (ok)
            for (int h = 0; h < 4; ++h) {  // (R1, t), (R2, t), (R1, -t), (R2, -t)
                const double* R = (h & 1) ? R2 : R1;
                const double sg = h < 2 ? 1.0 : -1.0;
                for (int r = 0; r < 3; ++r) {
                    for (int k = 0; k < 3; ++k) s_P[h][4 * r + k] = R[3 * r + k];
                    s_P[h][4 * r + 3] = sg * t[r];
                }
            }
    }
    __syncthreads();
    const bool ok = s_ok != 0;
    for (int base = 0; base < n && ok; base += kTvThreads) {
        const int i = base + tid;
        unsigned bits = 0;
        if (i < n && (!use_in_mask || mk[i] != 0)) {
            const double x1 = ((double)c1[i].x - cx) / fx, y1 = ((double)c1[i].y - cy) / fy;
            const double x2 = ((double)c2[i].x - cx) / fx, y2 = ((double)c2[i].y - cy) / fy;
            bits = tv::pose_votes(s_P, dist_thresh, x1, y1, x2, y2);
        }
        if (i < min(n, kVoteCache)) s_bits[i] = (uint8_t)bits;
        for (int h = 0; h < 4; ++h) {
            const unsigned long long b = __ballot((bits >> h) & 1u);
            if (lane == 0 && b) atomicAdd(&s_votes[h], (unsigned)__popcll(b));
        }
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned g0 = s_votes[0], g1 = s_votes[1], g2 = s_votes[2], g3 = s_votes[3];
        int w;  // recoverPose's chain of comparisons
        if (g0 >= g1 && g0 >= g2 && g0 >= g3) w = 0;
        else if (g1 >= g0 && g1 >= g2 && g1 >= g3) w = 1;
        else if (g2 >= g0 && g2 >= g1 && g2 >= g3) w = 2;
        else w = 3;
        s_winner = w;
        gms_two_view& o = tv[p];
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < 3; ++k) o.R[3 * r + k] = ok ? s_P[w][4 * r + k] : 0.0;
            o.t[r] = ok ? s_P[w][4 * r + 3] : 0.0;
        }
        o.n_pose = ok ? (int)(w == 0 ? g0 : w == 1 ? g1 : w == 2 ? g2 : g3) : 0;
        o.pose_which = ok ? w : -1;
        if (o.status == GMS_OK && !ok) o.status = GMS_ERR_NO_MODEL;
    }
    __syncthreads();
    const int w = s_winner;
    // the mask after recoverPose: bitwise_and(mask, hypothesis mask) -- the caller's bytes where the point passes, 0 elsewhere
    // (255 / 0 when no mask came in)
    for (int i = tid; i < n; i += kTvThreads) {
        uint8_t out = 0;
        const uint8_t in = use_in_mask ? mk[i] : (uint8_t)255;
        if (ok && in != 0) {
            if (i < kVoteCache) {
                out = ((s_bits[i] >> w) & 1u) ? in : (uint8_t)0;
            } else {
                const double x1 = ((double)c1[i].x - cx) / fx, y1 = ((double)c1[i].y - cy) / fy;
                const double x2 = ((double)c2[i].x - cx) / fx, y2 = ((double)c2[i].y - cy) / fy;
                out = tv::pose_vote_one(s_P, w, dist_thresh, x1, y1, x2, y2) ? in : (uint8_t)0;
            }
        }
        mk[i] = out;
    }
}

// ---- SfMUtil.cpp:65-82, 128-143: inliers compacted in order, undistortPoints, triangulatePoints, / w; reprojection error sums --------------
__global__ void __launch_bounds__(kTvThreads)
triangulate_batch_kernel(tv::Camera cam, const gms_pair* __restrict__ pairs, const float2* __restrict__ coords1,
                         const float2* __restrict__ coords2, const uint8_t* __restrict__ mask, double* __restrict__ points3d,
                         gms_two_view* __restrict__ tv)
{
    __shared__ unsigned s_wave[kTvThreads / 64];
    __shared__ double s_e1[kTvThreads], s_e2[kTvThreads];
    __shared__ unsigned s_cnt[kTvThreads], s_behind[kTvThreads];
    const int p = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const gms_pair pr = pairs[p];
    const bool ok = tv[p].status == GMS_OK;
    const int n = ok ? pair_count(pr, tv[p]) : 0;
    const float2* __restrict__ c1 = coords1 + pr.match_off;
    const float2* __restrict__ c2 = coords2 + pr.match_off;
    const uint8_t* __restrict__ mk = mask ? mask + pr.match_off : nullptr;
    double* __restrict__ out = points3d + 3 * pr.match_off;
    const double P1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};  // SfMUtil.cpp:53-55
    double P2[12];                                               // SfMUtil.cpp:57-59: [R | t] of recoverPose
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) P2[4 * r + k] = tv[p].R[3 * r + k];
        P2[4 * r + 3] = tv[p].t[r];
    }
    double e1 = 0.0, e2 = 0.0;
    unsigned cnt = 0, behind = 0, kept_before = 0;
    for (int base = 0; base < n; base += kTvThreads) {
        const int i = base + tid;
        const bool keep = i < n && (mk == nullptr || mk[i] != 0);  // SfMUtil.cpp:70
        const unsigned long long b = __ballot(keep);
        if (lane == 0) s_wave[wave] = (unsigned)__popcll(b);
        __syncthreads();
        unsigned before = kept_before, total = 0;
        for (int w = 0; w < kTvThreads / 64; ++w) {
            before += w < wave ? s_wave[w] : 0u;
            total += s_wave[w];
        }
        __syncthreads();
        if (keep) {
            const unsigned pos = before + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
            double x1, y1, x2, y2, X[4];
            tv::undistort_point(cam, (double)c1[i].x, (double)c1[i].y, x1, y1);  // SfMUtil.cpp:78-79
            tv::undistort_point(cam, (double)c2[i].x, (double)c2[i].y, x2, y2);
            tv::dlt_point(P1, P2, x1, y1, x2, y2, X);
            const double px = X[0] / X[3], py = X[1] / X[3], pz = X[2] / X[3];  // SfMUtil.cpp:134-137
            out[3 * (size_t)pos] = px;
            out[3 * (size_t)pos + 1] = py;
            out[3 * (size_t)pos + 2] = pz;
            const double w1 = pz, w2 = P2[8] * px + P2[9] * py + P2[10] * pz + P2[11];
            const double u1 = px / w1 - x1, v1 = py / w1 - y1;
            const double u2 = (P2[0] * px + P2[1] * py + P2[2] * pz + P2[3]) / w2 - x2;
            const double v2 = (P2[4] * px + P2[5] * py + P2[6] * pz + P2[7]) / w2 - y2;
            if (isfinite(px) && isfinite(py) && isfinite(pz) && isfinite(u1) && isfinite(v1) && isfinite(u2) && isfinite(v2)) {
                e1 += u1 * u1 + v1 * v1;
                e2 += u2 * u2 + v2 * v2;
                cnt += 1;
                behind += (w1 <= 0.0 || w2 <= 0.0) ? 1u : 0u;
            }
        }
        kept_before += total;
    }
    // reduction in a fixed order (thread partial sums, then a tree): the same bits run to run
    s_e1[tid] = e1;
    s_e2[tid] = e2;
    s_cnt[tid] = cnt;
    s_behind[tid] = behind;
    __syncthreads();
    for (int o = kTvThreads / 2; o >= 1; o >>= 1) {
        if (tid < o) {
            s_e1[tid] += s_e1[tid + o];
            s_e2[tid] += s_e2[tid + o];
            s_cnt[tid] += s_cnt[tid + o];
            s_behind[tid] += s_behind[tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        gms_two_view& o = tv[p];
        o.n_triangulated = (int)kept_before;
        o.n_finite = (int64_t)s_cnt[0];
        o.n_behind = (int64_t)s_behind[0];
        o.sum_sq_err1 = s_e1[0];
        o.sum_sq_err2 = s_e2[0];
    }
}

// ---- DisparityUtil.cpp:170-201 for a batch: one map per pair --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
disparity_batch_scatter_kernel(const gms_keypoint* __restrict__ kp, const int64_t* __restrict__ frame_off, const int32_t* __restrict__ wh,
                               int n_frames, const gms_pair* __restrict__ pairs, const gms_dmatch* __restrict__ filtered,
                               const gms_pair_result* __restrict__ results, int64_t map_stride, uint32_t* __restrict__ work,
                               gms_disparity_stats* __restrict__ stats)
{
    const int p = (int)blockIdx.y;
    const gms_pair pr = pairs[p];
    const gms_pair_result res = results[p];
    const bool pair_ok = res.status == GMS_OK && pr.frame_a >= 0 && pr.frame_a < n_frames && pr.frame_b >= 0 && pr.frame_b < n_frames;
    if (!pair_ok) {
        if (blockIdx.x == 0 && threadIdx.x == 0) stats[p].status = res.status != GMS_OK ? res.status : GMS_ERR_DOMAIN;
        return;
    }
    const int n = max(0, min(res.n_inliers, pr.m));
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n) return;
    const int w = wh[2 * pr.frame_a], h = wh[2 * pr.frame_a + 1];
    const int64_t oa = frame_off[pr.frame_a], ob = frame_off[pr.frame_b];
    const int na = (int)(frame_off[pr.frame_a + 1] - oa), nb = (int)(frame_off[pr.frame_b + 1] - ob);
    const gms_dmatch m = filtered[pr.match_off + i];
    bool bad = (uint32_t)m.queryIdx >= (uint32_t)na || (uint32_t)m.trainIdx >= (uint32_t)nb || (int64_t)w * h > map_stride || w <= 0 || h <= 0;
    if (!bad) {
        const float fx = kp[oa + m.queryIdx].x, fy = kp[oa + m.queryIdx].y, fx1 = kp[ob + m.trainIdx].x;
        const bool finite = fabsf(fx) < 1e9f && fabsf(fy) < 1e9f && fabsf(fx1) < 1e9f;
        const int x = finite ? (int)fx : -1, y = finite ? (int)fy : -1, x1 = finite ? (int)fx1 : 0;  // DisparityUtil.cpp:181-183
        bad = !finite || (uint32_t)x >= (uint32_t)w || (uint32_t)y >= (uint32_t)h;
        if (!bad) {
            const uint32_t v = (uint32_t)abs(x - x1) & 255u;  // int -> uchar keeps the low byte (DisparityUtil.cpp:184)
            atomicMax(&work[(size_t)p * map_stride + (size_t)y * w + x], ((uint32_t)(i + 1) << 8) | v);  // the LAST match on a pixel wins
        }
    }
    if (bad) stats[p].status = GMS_ERR_DOMAIN;
}

__global__ void __launch_bounds__(256)
disparity_batch_finish_kernel(const int32_t* __restrict__ wh, int n_frames, const gms_pair* __restrict__ pairs, int64_t map_stride,
                              const uint32_t* __restrict__ work, const uint8_t* __restrict__ gt, int64_t gt_stride, int disp_ratio,
                              uint8_t* __restrict__ disparity, gms_disparity_stats* __restrict__ stats)
{
    __shared__ unsigned long long s_sum[4], s_cnt[4];
    __shared__ uint32_t s_max[4];
    const int p = (int)blockIdx.y;
    const gms_pair pr = pairs[p];
    if (pr.frame_a < 0 || pr.frame_a >= n_frames) return;
    const int64_t n_pix = min((int64_t)wh[2 * pr.frame_a] * wh[2 * pr.frame_a + 1], map_stride);
    const uint32_t* __restrict__ wk = work + (size_t)p * map_stride;
    uint8_t* __restrict__ dm = disparity + (size_t)p * map_stride;
    const uint8_t* __restrict__ g = gt ? gt + (size_t)p * gt_stride : nullptr;
    unsigned long long sum = 0, cnt = 0;
    uint32_t mx = 0;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n_pix; q += (int64_t)gridDim.x * 256) {
        const uint32_t wv = wk[q];
        const uint32_t d = wv ? (wv & 255u) : 255u;  // 255 = "no match" (DisparityUtil.cpp:170)
        dm[q] = (uint8_t)d;
        if (d != 255u && g != nullptr) {
            const uint32_t a = (uint32_t)abs((int)d - (int)g[q] / disp_ratio);  // DisparityUtil.cpp:193
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
            atomicAdd(reinterpret_cast<unsigned long long*>(&stats[p].sum_sq), sum);
            atomicAdd(reinterpret_cast<unsigned long long*>(&stats[p].count), cnt);
            atomicMax(reinterpret_cast<uint32_t*>(&stats[p].max_abs), mx);
        }
    }
}

}  // namespace

hipError_t launch_gather_batch(const gms_keypoint* d_kp, const int64_t* d_frame_off, int n_frames, const gms_pair* d_pairs, int n_pairs, int max_m,
                               const gms_dmatch* d_filtered, const gms_pair_result* d_results, float* d_coords1, float* d_coords2,
                               gms_two_view* d_tv, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_tv, 0, (size_t)n_pairs * sizeof(gms_two_view), stream);
    if (e != hipSuccess) return e;
    const unsigned bx = (unsigned)((max_m + 255) / 256);
    hipLaunchKernelGGL(gather_batch_kernel, dim3(bx ? bx : 1u, (unsigned)n_pairs), dim3(256), 0, stream, d_kp, d_frame_off, n_frames, d_pairs,
                       d_filtered, d_results, reinterpret_cast<float2*>(d_coords1), reinterpret_cast<float2*>(d_coords2), d_tv);
    return hipGetLastError();
}

hipError_t launch_find_essential_batch(const gms_camera& cam, double prob, double threshold, int max_iters, const gms_pair* d_pairs, int n_pairs,
                                       const float* d_coords1, const float* d_coords2, uint8_t* d_mask, gms_two_view* d_tv, hipStream_t stream)
{
    EssentialParams prm{cam.fx, cam.fy, cam.cx, cam.cy, prob, threshold, max_iters};
    // GMS_TV_GEOM=0: round 3's workgroup shape (256 threads, sixteen samples per round), for A/B runs
    static const bool wide = [] { const char* e = getenv("GMS_TV_GEOM"); return e && atoi(e) == 0; }();
    if (wide)
        hipLaunchKernelGGL((find_essential_kernel<kTvThreads, kSolvers>), dim3((unsigned)n_pairs), dim3(kTvThreads), 0, stream, prm, d_pairs,
                           reinterpret_cast<const float2*>(d_coords1), reinterpret_cast<const float2*>(d_coords2), d_mask, d_tv);
    else
        hipLaunchKernelGGL((find_essential_kernel<128, 12>), dim3((unsigned)n_pairs), dim3(128), 0, stream, prm, d_pairs,
                           reinterpret_cast<const float2*>(d_coords1), reinterpret_cast<const float2*>(d_coords2), d_mask, d_tv);
    return hipGetLastError();
}

hipError_t launch_five_point_selftest(const double* d_pts, int n_samples, double* d_models, int* d_counts, hipStream_t stream)
{
    if (n_samples <= 0) return hipSuccess;
    hipLaunchKernelGGL(five_point_selftest_kernel, dim3((unsigned)((n_samples + kSolvers - 1) / kSolvers)), dim3(64), 0, stream, d_pts, n_samples,
                       d_models, d_counts);
    return hipGetLastError();
}

hipError_t launch_recover_pose_batch(const gms_camera& cam, double dist_thresh, int use_in_mask, const gms_pair* d_pairs, int n_pairs,
                                     const float* d_coords1, const float* d_coords2, uint8_t* d_mask, gms_two_view* d_tv, hipStream_t stream)
{
    hipLaunchKernelGGL(recover_pose_batch_kernel, dim3((unsigned)n_pairs), dim3(kTvThreads), 0, stream, cam.fx, cam.fy, cam.cx, cam.cy, dist_thresh,
                       use_in_mask, d_pairs, reinterpret_cast<const float2*>(d_coords1), reinterpret_cast<const float2*>(d_coords2), d_mask, d_tv);
    return hipGetLastError();
}

hipError_t launch_triangulate_batch(const gms_camera& cam, const gms_pair* d_pairs, int n_pairs, const float* d_coords1, const float* d_coords2,
                                    const uint8_t* d_mask, double* d_points3d, gms_two_view* d_tv, hipStream_t stream)
{
    tv::Camera c{cam.fx, cam.fy, cam.cx, cam.cy, cam.k1, cam.k2, cam.p1, cam.p2, cam.k3};
    hipLaunchKernelGGL(triangulate_batch_kernel, dim3((unsigned)n_pairs), dim3(kTvThreads), 0, stream, c, d_pairs,
                       reinterpret_cast<const float2*>(d_coords1), reinterpret_cast<const float2*>(d_coords2), d_mask, d_points3d, d_tv);
    return hipGetLastError();
}

hipError_t launch_disparity_batch(const gms_keypoint* d_kp, const int64_t* d_frame_off, const int32_t* d_wh, int n_frames, const gms_pair* d_pairs,
                                  int n_pairs, int max_m, const gms_dmatch* d_filtered, const gms_pair_result* d_results, const uint8_t* d_gt,
                                  int64_t gt_stride, int disp_ratio, uint8_t* d_disparity, int64_t map_stride, uint32_t* d_work,
                                  gms_disparity_stats* d_stats, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_work, 0, (size_t)n_pairs * (size_t)map_stride * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_stats, 0, (size_t)n_pairs * sizeof(gms_disparity_stats), stream);
    if (e != hipSuccess) return e;
    const unsigned bx = (unsigned)((max_m + 255) / 256);
    if (bx)
        hipLaunchKernelGGL(disparity_batch_scatter_kernel, dim3(bx, (unsigned)n_pairs), dim3(256), 0, stream, d_kp, d_frame_off, d_wh, n_frames,
                           d_pairs, d_filtered, d_results, map_stride, d_work, d_stats);
    int64_t blocks = (map_stride + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(disparity_batch_finish_kernel, dim3((unsigned)(blocks ? blocks : 1), (unsigned)n_pairs), dim3(256), 0, stream, d_wh, n_frames,
                       d_pairs, map_stride, d_work, d_gt, gt_stride, disp_ratio, d_disparity, d_stats);
    return hipGetLastError();
}

}  // namespace gms
