// gms_kernels.hip -- hand-written HIP kernels for gfx950 (CDNA4): the GMS match filter.
//
// What the reference does per pair (cv::xfeatures2d::matchGMS, opencv_xfeatures2d452.dll; SURVEY.md
// section 8a) is a dense 400 x N_right int32 "motion" matrix that is zeroed, filled and scanned 4 times
// per hypothesis. That matrix has at most M non-zeros, so here it is never materialised: one workgroup
// owns one image pair and keeps the pair's whole state in the CU's 160 KB LDS:
//
//   code[m]     one dword per match: right cell, unshifted left cell (x, y), the two half-cell shift
//               bits that derive grid types 2..4, a valid bit, and 8 per-rotation inlier bits
//   keys/cnt    open-addressing hash table  (left cell, right cell) -> count, i.e. the non-zeros of
//               the motion matrix of the current (scale, grid type); built with LDS atomics
//   nleft/best/accept  per-left-cell: match count, packed arg-max (count, lowest right cell),
//               and the 8-bit "passes the threshold under rotation r" set
//
// Rotation only changes which neighbour counts are summed, so one table build serves all 8 rotations;
// the right cell only depends on the scale, the left cell only on the grid type. HBM sees each
// match once on the way in (16 B, coalesced) plus the two 8-B keypoint gathers, and 16 B per survivor
// on the way out. No MFMA: this is integer histogramming.
//
// Bit-exactness notes (vs the DLL): fp32 multiply then floor for unshifted axes; widen the fp32
// product to fp64, add 0.5, floor for shifted axes (DLL@0x180047bc0); fp64 div -> sqrt -> mul -> '>'
// for the threshold (DLL@0x180049171); arg-max keeps the LOWEST right cell among maxima; hypotheses
// are compared scale-outer / rotation-inner with strict '>' (DLL@0x180047dc0). Built with
// -ffp-contract=off and HIP's default correctly rounded fp32 divide.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gms_kernels.h"

namespace gms {

// mRotationPatterns - 1 (DLL .rdata 0x18012f520).
__constant__ int8_t c_rot[8][9] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8}, {3, 0, 1, 6, 4, 2, 7, 8, 5}, {6, 3, 0, 7, 4, 1, 8, 5, 2},
    {7, 6, 3, 8, 4, 0, 5, 2, 1}, {8, 7, 6, 5, 4, 3, 2, 1, 0}, {5, 8, 7, 2, 4, 6, 1, 0, 3},
    {2, 5, 8, 1, 4, 7, 0, 3, 6}, {1, 2, 5, 0, 4, 8, 3, 6, 7}};

constexpr uint32_t kEmpty = 0xFFFFFFFFu;

// code word layout
constexpr uint32_t kRMask = 0x7FFu;       // bits 0..10  right cell (< 1600)
constexpr int kLxShift = 11;              // bits 11..15 floor(20*nx), clamped to 31
constexpr int kLyShift = 16;              // bits 16..20 floor(20*ny), clamped to 31
constexpr uint32_t kSxBit = 1u << 21;     // floor(20*nx + 0.5) - floor(20*nx)
constexpr uint32_t kSyBit = 1u << 22;
constexpr uint32_t kValidBit = 1u << 23;
constexpr int kAccShift = 24;             // bits 24..31 inlier-under-rotation bits (OR over grid types)

__device__ __forceinline__ int floor_f32(float v)
{
    int i = (int)v;
    return i - ((float)i > v);
}
__device__ __forceinline__ int floor_f64(double v)
{
    int i = (int)v;
    return i - ((double)i > v);
}

// Left cell of grid type g (0..3; bit0 = x shifted by half a cell, bit1 = y shifted), or -1.
__device__ __forceinline__ int left_cell(uint32_t c, int g)
{
    int x = (int)((c >> kLxShift) & 31u) + ((g & 1) ? (int)((c >> 21) & 1u) : 0);
    int y = (int)((c >> kLyShift) & 31u) + ((g & 2) ? (int)((c >> 22) & 1u) : 0);
    if (!(c & kValidBit) || x >= kLeftW || y >= kLeftH) return -1;
    return x + y * kLeftW;
}

__device__ __forceinline__ uint32_t hash_slot(uint32_t key, uint32_t S)
{
    return __umulhi(key * 0x9E3779B1u, S);
}

// motion[l][r]++ on the sparse table. Every lane terminates: S > number of distinct keys.
__device__ __forceinline__ void table_insert(uint32_t* keys, uint32_t* cnt, uint32_t S, uint32_t key)
{
    uint32_t h = hash_slot(key, S);
    for (uint32_t probe = 0; probe < S; ++probe) {
        uint32_t prev = atomicCAS(&keys[h], kEmpty, key);
        if (prev == kEmpty || prev == key) {
            atomicAdd(&cnt[h >> 1], 1u << ((h & 1u) << 4));
            return;
        }
        if (++h == S) h = 0;
    }
}

// motion[l][r]
__device__ __forceinline__ uint32_t table_lookup(const uint32_t* keys, const uint32_t* cnt, uint32_t S,
                                                 uint32_t key)
{
    uint32_t h = hash_slot(key, S);
    for (uint32_t probe = 0; probe < S; ++probe) {
        uint32_t k = keys[h];
        if (k == key) return (cnt[h >> 1] >> ((h & 1u) << 4)) & 0xFFFFu;
        if (k == kEmpty) return 0;
        if (++h == S) h = 0;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// normalizePoints (DLL@0x180048420): one thread per keypoint; frame found by binary search.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
normalize_kernel(const gms_keypoint* __restrict__ kp, const int64_t* __restrict__ frame_off,
                 const int32_t* __restrict__ wh, int n_frames, int64_t total, float2* __restrict__ pts)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        int lo = 0, hi = n_frames - 1;  // last frame f with frame_off[f] <= i
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (frame_off[mid] <= i) lo = mid; else hi = mid - 1;
        }
        float w = (float)wh[2 * lo], h = (float)wh[2 * lo + 1];
        const float* p = reinterpret_cast<const float*>(kp + i);
        float2 o;
        o.x = p[0] / w;  // IEEE fp32 divide (divss)
        o.y = p[1] / h;
        pts[i] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// The filter: one workgroup per pair.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
filter_kernel(FilterParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    const gms_pair pr = p.pairs[blockIdx.x];
    const int m = pr.m;
    const gms_dmatch* __restrict__ matches = p.matches + pr.match_off;

    const int mcap = p.mcap;  // multiple of 64, >= every m
    const uint32_t S = p.table_slots;
    uint32_t* code = smem;
    uint32_t* keys = code + mcap;
    uint32_t* cnt = keys + S;                    // S/2 dwords, two 16-bit counters each
    uint32_t* bestmask = cnt + (S >> 1);         // mcap/32 dwords
    uint32_t* nleft = bestmask + (mcap >> 5);    // 400
    uint32_t* best = nleft + kLeftN;             // 400
    uint32_t* accept = best + kLeftN;            // 400
    uint32_t* chunk_base = accept + kLeftN;      // mcap/64 (+1)
    uint32_t* misc = chunk_base + (mcap >> 6) + 1;  // [0..7] rotation counts, [8] error, [9..] scan scratch

    if (tid < 16 + kThreads / 64) misc[tid] = 0;
    for (int i = tid; i < (mcap >> 5); i += kThreads) bestmask[i] = 0;

    const bool bad_pair = m < 0 || m > mcap || pr.frame_a < 0 || pr.frame_a >= p.n_frames ||
                          pr.frame_b < 0 || pr.frame_b >= p.n_frames;
    int64_t offA = 0, offB = 0;
    int nA = 0, nB = 0;
    if (!bad_pair) {
        offA = p.frame_off[pr.frame_a];
        offB = p.frame_off[pr.frame_b];
        nA = (int)(p.frame_off[pr.frame_a + 1] - offA);
        nB = (int)(p.frame_off[pr.frame_b + 1] - offB);
    }
    const float2* __restrict__ ptsA = p.pts + offA;
    const float2* __restrict__ ptsB = p.pts + offB;
    const int mm = bad_pair ? 0 : m;

    const int n_scales = p.with_scale ? 5 : 1;
    const int n_rot = p.with_rotation ? 8 : 1;
    uint32_t best_count = 0;
    int best_scale = -1, best_rot = -1;
    __syncthreads();
    if (bad_pair && tid == 0) misc[8] = 1;

    for (int s = 0; s < n_scales; ++s) {
        const int wr = p.right_w[s], hr = p.right_h[s];
        const int nr = wr * hr;
        const float fwr = (float)wr, fhr = (float)hr;

        // ---- bin every match: left cells of the 4 grid types, right cell of this scale --------------
        for (int i = tid; i < mm; i += kThreads) {
            const int2 qt = *reinterpret_cast<const int2*>(&matches[i]);  // queryIdx, trainIdx
            uint32_t c = 0;
            bool ok = qt.x >= 0 && qt.x < nA && qt.y >= 0 && qt.y < nB;
            if (ok) {
                const float2 a = ptsA[qt.x];
                const float2 b = ptsB[qt.y];
                // parity domain: finite, non-negative, < 2^20 (NaN fails every compare)
                ok = a.x >= 0.f && a.x < 1048576.f && a.y >= 0.f && a.y < 1048576.f &&
                     b.x >= 0.f && b.x < 1048576.f && b.y >= 0.f && b.y < 1048576.f;
                if (ok) {
                    const float fx = 20.0f * a.x, fy = 20.0f * a.y;       // mulss, rounded to fp32
                    const int lx = floor_f32(fx), ly = floor_f32(fy);
                    const int lx2 = floor_f64((double)fx + 0.5), ly2 = floor_f64((double)fy + 0.5);
                    const int rx = floor_f32(fwr * b.x), ry = floor_f32(fhr * b.y);
                    const int r = rx + ry * wr;                             // no bounds test in the reference
                    ok = r >= 0 && r < nr;
                    c = (uint32_t)r | ((uint32_t)min(lx, 31) << kLxShift) | ((uint32_t)min(ly, 31) << kLyShift) |
                        ((lx2 > lx) ? kSxBit : 0u) | ((ly2 > ly) ? kSyBit : 0u) | kValidBit;
                }
            }
            if (!ok) {
                c = 0;
                misc[8] = 1;  // benign race: every writer stores 1
            }
            code[i] = c;
        }

        for (int g = 0; g < 4; ++g) {
            // ---- motion.setTo(0), nLeft = 0, cellPairs = -1 ------------------------------------------
            for (uint32_t i = tid; i < S; i += kThreads) keys[i] = kEmpty;
            for (uint32_t i = tid; i < (S >> 1); i += kThreads) cnt[i] = 0;
            for (int i = tid; i < kLeftN; i += kThreads) {
                nleft[i] = 0;
                best[i] = 0;
                accept[i] = 0;
            }
            __syncthreads();

            // ---- assignMatchPairs: sparse motion[l][r]++, nLeft[l]++ ---------------------------------
            for (int i = tid; i < mm; i += kThreads) {
                const uint32_t c = code[i];
                const int l = left_cell(c, g);
                if (l >= 0) {
                    table_insert(keys, cnt, S, ((uint32_t)l << 11) | (c & kRMask));
                    atomicAdd(&nleft[l], 1u);
                }
            }
            __syncthreads();

            // ---- arg-max over each left cell's row: max count, lowest right cell on ties -------------
            for (uint32_t h = tid; h < S; h += kThreads) {
                const uint32_t k = keys[h];
                if (k != kEmpty) {
                    const uint32_t c = (cnt[h >> 1] >> ((h & 1u) << 4)) & 0xFFFFu;
                    atomicMax(&best[k >> 11], (c << 11) | (2047u - (k & kRMask)));
                }
            }
            __syncthreads();

            // ---- verifyCellPairs: one thread per (left cell, rotation) -----------------------------------
            for (int item = tid; item < kLeftN * n_rot; item += kThreads) {
                const int i = item / n_rot, rot = item - i * n_rot;
                if (nleft[i] == 0) continue;
                const int j = 2047 - (int)(best[i] & kRMask);
                const int jx = j % wr, jy = j / wr;
                const int ix = i % kLeftW, iy = i / kLeftW;
                int score = 0, tsum = 0, numpair = 0;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int lx = ix + (k % 3) - 1, ly = iy + (k / 3) - 1;
                    const int q = c_rot[rot][k];
                    const int rx = jx + (q % 3) - 1, ry = jy + (q / 3) - 1;
                    if (lx < 0 || lx >= kLeftW || ly < 0 || ly >= kLeftH) continue;  // ll == -1
                    if (rx < 0 || rx >= wr || ry < 0 || ry >= hr) continue;          // rr == -1
                    const int ll = lx + ly * kLeftW, rr = rx + ry * wr;
                    score += (int)table_lookup(keys, cnt, S, ((uint32_t)ll << 11) | (uint32_t)rr);
                    tsum += (int)nleft[ll];
                    numpair++;
                }
                // divsd, sqrtsd, mulsd, comisd: reject iff thresh > score
                const double thresh = sqrt((double)tsum / (double)numpair) * p.threshold_factor;
                if (!(thresh > (double)score)) atomicOr(&accept[i], 1u << rot);
            }
            __syncthreads();

            // ---- mark inliers: cellPairs[l] == r, for all rotations at once ------------------------------
            for (int i = tid; i < mm; i += kThreads) {
                const uint32_t c = code[i];
                const int l = left_cell(c, g);
                if (l >= 0) {
                    const uint32_t j = 2047u - (best[l] & kRMask);
                    if (j == (c & kRMask)) code[i] = c | (accept[l] << kAccShift);
                }
            }
            __syncthreads();
        }

        // ---- run() return value for each rotation of this scale ---------------------------------------
        for (int i0 = tid - lane; i0 < mm; i0 += kThreads) {
            const int i = i0 + lane;
            const uint32_t bits = (i < mm) ? (code[i] >> kAccShift) : 0u;
            for (int rot = 0; rot < n_rot; ++rot) {
                const unsigned long long b = __ballot((bits >> rot) & 1u);
                if (lane == 0 && b) atomicAdd(&misc[rot], (uint32_t)__popcll(b));
            }
        }
        __syncthreads();

        // ---- getInlierMask: keep on strict '>' (scale outer, rotation inner) ---------------------------
        int winner = -1;
        for (int rot = 0; rot < n_rot; ++rot) {
            const uint32_t c = misc[rot];
            if (c > best_count) {
                best_count = c;
                best_scale = s;
                best_rot = rot + 1;
                winner = rot;
            }
        }
        if (winner >= 0) {
            for (int i0 = tid - lane; i0 < mm; i0 += kThreads) {
                const int i = i0 + lane;
                const uint32_t bit = (i < mm) ? ((code[i] >> (kAccShift + winner)) & 1u) : 0u;
                const unsigned long long b = __ballot(bit);
                if (lane == 0) {
                    bestmask[(i0 >> 5)] = (uint32_t)b;
                    bestmask[(i0 >> 5) + 1] = (uint32_t)(b >> 32);
                }
            }
        }
        __syncthreads();
        if (tid < 8) misc[tid] = 0;
        __syncthreads();
    }

    // ---- copy-out: surviving DMatch verbatim, in input order (DLL@0x180048340) -----------------------
    const bool failed = misc[8] != 0;
    const int n_chunks = (mm + 63) >> 6;
    {
        // exclusive scan of per-chunk popcounts; THREADS chunks per round, carry in misc[9]
        uint32_t* wave_tot = misc + 16;
        if (tid == 0) misc[9] = 0;
        __syncthreads();
        for (int base = 0; base < n_chunks; base += kThreads) {
            const int c = base + tid;
            const uint32_t v = (c < n_chunks && !failed) ? __popc(bestmask[2 * c]) + __popc(bestmask[2 * c + 1]) : 0u;
            uint32_t incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d);
                if (lane >= d) incl += t;
            }
            if (lane == 63) wave_tot[tid >> 6] = incl;
            __syncthreads();
            uint32_t wave_off = misc[9];
            for (int w = 0; w < (tid >> 6); ++w) wave_off += wave_tot[w];
            if (c < n_chunks) chunk_base[c] = wave_off + incl - v;
            __syncthreads();
            if (tid == kThreads - 1) misc[9] = wave_off + incl;
            __syncthreads();
        }
    }
    const uint32_t total = misc[9];

    gms_dmatch* __restrict__ out = p.out + pr.match_off;
    uint8_t* mask_out = p.mask ? p.mask + pr.match_off : nullptr;
    for (int i0 = tid - lane; i0 < mm; i0 += kThreads) {
        const int i = i0 + lane;
        const int c = i0 >> 6;
        const unsigned long long bits =
            failed ? 0ull : ((unsigned long long)bestmask[2 * c] | ((unsigned long long)bestmask[2 * c + 1] << 32));
        const bool in = (bits >> lane) & 1ull;
        if (i < mm) {
            if (mask_out) mask_out[i] = in ? 1 : 0;
            if (in) {
                const uint32_t pos = chunk_base[c] + (uint32_t)__popcll(bits & ((1ull << lane) - 1ull));
                const uint4 v = *reinterpret_cast<const uint4*>(&matches[i]);
                *reinterpret_cast<uint4*>(&out[pos]) = v;
            }
        }
    }
    if (tid == 0) {
        gms_pair_result r;
        r.n_inliers = failed ? 0 : (int)total;
        r.best_scale = failed ? -1 : best_scale;
        r.best_rot = failed ? -1 : best_rot;
        r.status = failed ? GMS_ERR_DOMAIN : GMS_OK;
        p.results[blockIdx.x] = r;
    }
}

// Test hook: the threshold comparison in device fp64.
__global__ void threshold_kernel(const int32_t* T, const int32_t* n, const int32_t* score, double factor,
                                 int count, uint8_t* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        const double thresh = sqrt((double)T[i] / (double)n[i]) * factor;
        out[i] = thresh > (double)score[i] ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// launch helpers (called from gms_capi.cpp through gms_kernels.h)
// ------------------------------------------------------------------------------------------------
size_t filter_lds_bytes(int mcap, uint32_t table_slots)
{
    size_t dwords = (size_t)mcap + table_slots + (table_slots >> 1) + (mcap >> 5) + 3 * kLeftN +
                    (mcap >> 6) + 1 + 16 + kThreads / 64 + 16;
    return dwords * 4;
}

hipError_t launch_normalize(const gms_keypoint* d_kp, const int64_t* d_frame_off, const int32_t* d_wh,
                            int n_frames, int64_t total_kp, float* d_pts, hipStream_t stream)
{
    if (total_kp <= 0) return hipSuccess;
    int64_t blocks = (total_kp + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_kp, d_frame_off,
                       d_wh, n_frames, total_kp, reinterpret_cast<float2*>(d_pts));
    return hipGetLastError();
}

hipError_t launch_filter(const FilterParams& p, int n_pairs, size_t lds_bytes, hipStream_t stream)
{
    if (n_pairs <= 0) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(filter_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(filter_kernel, dim3((unsigned)n_pairs), dim3(kThreads), lds_bytes, stream, p);
    return hipGetLastError();
}

hipError_t launch_threshold(const int32_t* d_T, const int32_t* d_n, const int32_t* d_score, double factor,
                            int count, uint8_t* d_out, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(threshold_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, d_T, d_n, d_score,
                       factor, count, d_out);
    return hipGetLastError();
}

}  // namespace gms
