// bf_kernels.hip -- brute-force descriptor matcher for gfx950: the producer of the match array the GMS filter consumes.
//
// What the reference runs in front of matchGMS (FeatureMatchUtil.cpp:66-68; DisparityUtil.cpp:104-109,143):
//     BFMatcher::create()->match(descriptors1, descriptors2, matches)        NORM_L2, no cross-check
// -- one DMatch per query row i: {queryIdx = i, trainIdx = arg min_j dist(i, j) (first minimum), imgIdx = 0, distance}.
// Here it works on the same resident frame table as the filter: descriptor i of a frame belongs to keypoint i, pair
// (frame_a, frame_b) gets its n(frame_a) matches written at d_matches[match_off ...] -- straight into the batch's match
// array, so the matches never cross PCIe.
//
//   NORM_HAMMING, 256-bit rows (ORB)   bf_hamming_kernel: integer VALU. A lane keeps Q query rows in registers; the train
//       rows are wave-uniform, so they arrive through the SCALAR cache (s_load_dwordx8) and feed v_xor / v_bcnt directly
//       as SGPR operands: no LDS, no vector loads in the loop. 18 VALU instructions per (query, train): 8 xor, 8 popcount-
//       accumulate, one shift-or that packs (distance, trainIdx), one unsigned min (lowest trainIdx wins ties for free).
//   NORM_L2, 128 floats (SIFT)         bf_l2_mfma_kernel: d^2 = |a|^2 + |b|^2 - 2 a.b with the cross term on the matrix cores
//       (v_mfma_f32_32x32x16_bf16, train rows x query columns, K = 128). SIFT descriptors are integers 0..255 stored as
//       floats: exact in bf16 (8 significant bits), every product and partial sum below 2^24 is exact in the fp32
//       accumulator, so d^2 is the exact integer the reference's fp32 loop produces, in any summation order -- the
//       arg-min and sqrtf(d^2) are bit-identical. gms_bf_prepare_device checks that property per frame while it builds the
//       bf16 table and the norms; a pair with a frame that fails it takes
//   bf_l2_loop_kernel                  the reference's own arithmetic, sum_k (a_k - b_k)^2 in fp32 in index order (no FMA
//       contraction), one lane per query, train rows broadcast from LDS. Slow and exact; also any dimension other than 128.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gms_kernels.h"

namespace gms {
namespace {

constexpr uint32_t kHamShift = 22;  // key = distance << 22 | trainIdx (distance <= 256, trainIdx < 2^22)

struct FramePair {
    int64_t offA, offB;
    int nA, nB, m;
    bool ok;
};

__device__ __forceinline__ FramePair frame_pair(const gms_pair& pr, const int64_t* __restrict__ frame_off, int n_frames)
{
    FramePair f;
    f.ok = pr.frame_a >= 0 && pr.frame_a < n_frames && pr.frame_b >= 0 && pr.frame_b < n_frames && pr.m >= 0;
    f.offA = f.offB = 0;
    f.nA = f.nB = f.m = 0;
    if (f.ok) {
        f.offA = frame_off[pr.frame_a];
        f.offB = frame_off[pr.frame_b];
        f.nA = (int)(frame_off[pr.frame_a + 1] - f.offA);
        f.nB = (int)(frame_off[pr.frame_b + 1] - f.offB);
        f.m = pr.m < f.nA ? pr.m : f.nA;  // one match per query row: M = N1 (FeatureMatchUtil.cpp:66-68)
    }
    return f;
}

// Blocks b and b + 8 share an XCD (round-robin dispatch: MI355X_MICROARCH.md). Consecutive tasks -- the query tiles of one
// pair, which all stream the same train frame -- are dealt to the same XCD so that frame comes out of one L2. Bijective for any
// task count; placement only changes speed.
__device__ __forceinline__ uint32_t xcd_task(uint32_t b, uint32_t n)
{
    const uint32_t q = n >> 3, r = n & 7u, x = b & 7u;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + (b >> 3);
}

// ---- NORM_HAMMING, 32-byte rows ---------------------------------------------------------------------------------------------
template <int Q>
__global__ void __launch_bounds__(256)
bf_hamming_kernel(const uint32_t* __restrict__ desc, const int64_t* __restrict__ frame_off, int n_frames,
                  const gms_pair* __restrict__ pairs, int tiles_per_pair, uint32_t n_tasks, gms_dmatch* __restrict__ matches)
{
    const uint32_t task = xcd_task(blockIdx.x, n_tasks);
    const int pair_idx = (int)(task / (uint32_t)tiles_per_pair), tile = (int)(task % (uint32_t)tiles_per_pair);
    const gms_pair pr = pairs[pair_idx];
    const FramePair f = frame_pair(pr, frame_off, n_frames);
    const int q0 = tile * 256 * Q;
    if (!f.ok || q0 >= f.m) return;  // workgroup-uniform
    const int tid = (int)threadIdx.x;

    uint32_t qv[Q][8], best[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int q = min(q0 + k * 256 + tid, f.m - 1);
        const uint4* src = reinterpret_cast<const uint4*>(desc + (size_t)(f.offA + q) * 8);
        const uint4 lo = src[0], hi = src[1];
        qv[k][0] = lo.x; qv[k][1] = lo.y; qv[k][2] = lo.z; qv[k][3] = lo.w;
        qv[k][4] = hi.x; qv[k][5] = hi.y; qv[k][6] = hi.z; qv[k][7] = hi.w;
        best[k] = 0xFFFFFFFFu;
    }
    // train rows: the address is wave-uniform and the memory read-only for the kernel -> scalar loads, SGPR operands. Two rows
    // per step, the next step's two requested before this step's are used (a scalar load's latency is a whole row's worth of VALU).
    const uint32_t* __restrict__ tr = desc + (size_t)f.offB * 8;
    auto load_row = [&](int j, uint32_t (&t)[8]) {
        const uint32_t* src = tr + (size_t)min(j, f.nB - 1) * 8;
#pragma unroll
        for (int w = 0; w < 8; ++w) t[w] = src[w];
    };
    auto use_row = [&](int j, const uint32_t (&t)[8]) {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            uint32_t d = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) d += (uint32_t)__builtin_popcount(qv[k][w] ^ t[w]);
            best[k] = min(best[k], (d << kHamShift) | (uint32_t)j);  // strict minimum, lowest trainIdx on ties
        }
    };
    if (f.nB > 0) {
        uint32_t t0[8], t1[8];
        load_row(0, t0);
        load_row(1, t1);
        for (int j = 0; j < f.nB; j += 2) {
            uint32_t n0[8], n1[8];
            load_row(j + 2, n0);
            load_row(j + 3, n1);
            use_row(j, t0);
            // a row past the end repeats the last one: same distance, higher index, never the minimum
            use_row(min(j + 1, f.nB - 1) == j + 1 ? j + 1 : (1 << kHamShift) - 1, t1);
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                t0[w] = n0[w];
                t1[w] = n1[w];
            }
        }
    }
    gms_dmatch* __restrict__ out = matches + pr.match_off;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int q = q0 + k * 256 + tid;
        if (q < f.m) {
            gms_dmatch r;
            r.queryIdx = q;
            r.trainIdx = f.nB > 0 ? (int)(best[k] & ((1u << kHamShift) - 1u)) : -1;
            r.imgIdx = 0;
            r.distance = f.nB > 0 ? (float)(best[k] >> kHamShift) : 3.402823466e+38f;
            *reinterpret_cast<uint4*>(&out[q]) = *reinterpret_cast<const uint4*>(&r);
        }
    }
}

// ---- NORM_L2: the per-frame tables --------------------------------------------------------------------------------------------
// One 32-lane group per 128-float row: 16 bytes per lane. Writes the row as bf16, |row|^2, and flags the row's frame when a value
// is not an integer in [0, 255] (then bf16 and the fp32 sums below 2^24 are no longer exact and the frame's pairs take the loop).
__global__ void __launch_bounds__(256)
bf_l2_prepare_kernel(const float4* __restrict__ desc, const int64_t* __restrict__ frame_off, int n_frames, int64_t total,
                     uint2* __restrict__ rows_bf16, float* __restrict__ norms, uint32_t* __restrict__ frame_bad)
{
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
    const int part = (int)(threadIdx.x & 31);
    if (row >= total) return;
    const float4 v = desc[row * 32 + part];
    auto is_u8 = [](float x) { return x >= 0.0f && x <= 255.0f && x == floorf(x); };
    const bool bad = !(is_u8(v.x) && is_u8(v.y) && is_u8(v.z) && is_u8(v.w));
    auto bf = [](float x) { return (uint32_t)(__float_as_uint(x) >> 16); };  // exact for 8-bit integers (the only case it is used for)
    rows_bf16[row * 32 + part] = make_uint2(bf(v.x) | (bf(v.y) << 16), bf(v.z) | (bf(v.w) << 16));
    float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) s += __shfl_xor(s, d, 32);
    if (part == 0) norms[row] = s;
    if (__ballot(bad) != 0ull && bad) {
        int lo = 0, hi = n_frames - 1;  // last frame f with frame_off[f] <= row
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (frame_off[mid] <= row) lo = mid; else hi = mid - 1;
        }
        frame_bad[lo] = 1u;  // benign race: every writer stores 1
    }
}

// ---- NORM_L2 on the matrix cores ----------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kL2Dim = 128;
constexpr int kTileRows = 64;                    // train rows per LDS tile
constexpr uint32_t kRowPitch = 256u + 16u;       // bf16 row + 16 B: the 32 rows a ds_read_b128 touches fall on different banks
constexpr uint32_t kTileBytes = kTileRows * kRowPitch;               // 17 408
constexpr uint32_t kNormOff = 2u * kTileBytes;                       // two tiles, then two norm tiles
constexpr uint32_t kL2LdsBytes = kNormOff + 2u * kTileRows * 4u;     // 35 328
constexpr int kQueriesPerBlock = 256;            // 4 waves x 64 query columns

__global__ void __launch_bounds__(256)
bf_l2_mfma_kernel(const uint4* __restrict__ rows_bf16, const float* __restrict__ norms, const uint32_t* __restrict__ frame_bad,
                  const int64_t* __restrict__ frame_off, int n_frames, const gms_pair* __restrict__ pairs, int tiles_per_pair,
                  uint32_t n_tasks, gms_dmatch* __restrict__ matches)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kL2LdsBytes];
    const uint32_t task = xcd_task(blockIdx.x, n_tasks);
    const int pair_idx = (int)(task / (uint32_t)tiles_per_pair), tile = (int)(task % (uint32_t)tiles_per_pair);
    const gms_pair pr = pairs[pair_idx];
    const FramePair f = frame_pair(pr, frame_off, n_frames);
    const int q0 = tile * kQueriesPerBlock;
    if (!f.ok || q0 >= f.m) return;                                     // workgroup-uniform
    if (frame_bad[pr.frame_a] | frame_bad[pr.frame_b]) return;          // the loop kernel owns this pair
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;

    // ---- this wave's 64 query columns as B operands, scaled by -2 (exact), resident for the whole kernel:
    //      bq[c][s] = -2 * Q[32 c + col][16 s + 8 half .. + 8)
    bf16x8 bq[2][8];
    int qrow[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        qrow[c] = q0 + wave * 64 + c * 32 + col;
        const uint4* src = rows_bf16 + (size_t)(f.offA + min(qrow[c], f.m - 1)) * 16;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const uint4 raw = src[2 * s + half];
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float lo = __uint_as_float(w[i] << 16) * -2.0f, hi = __uint_as_float(w[i] & 0xFFFF0000u) * -2.0f;
                o[i] = (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xFFFF0000u);
            }
            bq[c][s] = __builtin_bit_cast(bf16x8, make_uint4(o[0], o[1], o[2], o[3]));
        }
    }
    float bestv[2] = {3.402823466e+38f, 3.402823466e+38f};
    int besti[2] = {-1, -1};

    const uint4* __restrict__ trB = rows_bf16 + (size_t)f.offB * 16;
    const float* __restrict__ nrmB = norms + f.offB;
    const int n_tiles = (f.nB + kTileRows - 1) / kTileRows;
    // stage tile t into buffer t & 1: 64 rows x 256 B = 1024 16-byte pieces, four per thread, rows beyond nB repeat the last row
    // (their norm is +inf, so they never win)
    auto stage_load = [&](int t, uint4 (&v)[4], float& nv) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = i * 256 + tid, r = piece >> 4, c16 = piece & 15;
            v[i] = trB[(size_t)min(t * kTileRows + r, f.nB - 1) * 16 + c16];
        }
        const int r = t * kTileRows + (tid & 63);
        nv = (tid < kTileRows) ? (r < f.nB ? nrmB[r] : __builtin_inff()) : 0.0f;
    };
    auto stage_store = [&](int t, const uint4 (&v)[4], float nv) {
        unsigned char* base = lds + (uint32_t)(t & 1) * kTileBytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = i * 256 + tid, r = piece >> 4, c16 = piece & 15;
            *reinterpret_cast<uint4*>(base + (uint32_t)r * kRowPitch + (uint32_t)c16 * 16u) = v[i];
        }
        if (tid < kTileRows) reinterpret_cast<float*>(lds + kNormOff)[(t & 1) * kTileRows + tid] = nv;
    };
    uint4 sv[4];
    float snv;
    stage_load(0, sv, snv);
    stage_store(0, sv, snv);
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
        const bool more = t + 1 < n_tiles;
        if (more) stage_load(t + 1, sv, snv);  // in flight during this tile's MFMAs
        const unsigned char* tb = lds + (uint32_t)(t & 1) * kTileBytes;
        const float* nb = reinterpret_cast<const float*>(lds + kNormOff) + (t & 1) * kTileRows;
        f32x16 acc[2][2];
        // C-in = |b|^2 of the accumulator's row: rows (reg & 3) + 8 (reg >> 2) + 4 half of the 32-row block
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 n4 = *reinterpret_cast<const float4*>(nb + rb * 32 + 8 * g + 4 * half);
                acc[rb][0][4 * g + 0] = n4.x; acc[rb][0][4 * g + 1] = n4.y; acc[rb][0][4 * g + 2] = n4.z; acc[rb][0][4 * g + 3] = n4.w;
            }
            acc[rb][1] = acc[rb][0];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            // A operand: train row (32 rb + col), k = 16 s + 8 half .. + 8
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(tb + (uint32_t)col * kRowPitch + (uint32_t)(32 * s + 16 * half));
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(tb + (uint32_t)(32 + col) * kRowPitch + (uint32_t)(32 * s + 16 * half));
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[0][s], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[1][s], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[0][s], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[1][s], acc[1][1], 0, 0, 0);
        }
        // acc = |b|^2 - 2 a.b for (train row, query column): first strict minimum, rows ascending
        const int t0 = t * kTileRows + 4 * half;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float v = acc[rb][c][reg];
                    const bool lt = v < bestv[c];
                    bestv[c] = lt ? v : bestv[c];
                    besti[c] = lt ? t0 + rb * 32 + (reg & 3) + 8 * (reg >> 2) : besti[c];
                }
        if (more) stage_store(t + 1, sv, snv);  // the other buffer: last read one barrier ago
        __syncthreads();
    }
    // the two halves of the wave hold disjoint row sets of the same query: lower value, then lower row
    gms_dmatch* __restrict__ out = matches + pr.match_off;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float ov = __shfl_xor(bestv[c], 32);
        const int oi = __shfl_xor(besti[c], 32);
        const bool take = ov < bestv[c] || (ov == bestv[c] && (uint32_t)oi < (uint32_t)besti[c]);
        const float v = take ? ov : bestv[c];
        const int bi = take ? oi : besti[c];
        if (half == 0 && qrow[c] < f.m) {
            const float na = norms[f.offA + qrow[c]];
            gms_dmatch r;
            r.queryIdx = qrow[c];
            r.trainIdx = bi;
            r.imgIdx = 0;
            r.distance = bi >= 0 ? sqrtf(fmaxf(v + na, 0.0f)) : 3.402823466e+38f;
            *reinterpret_cast<uint4*>(&out[qrow[c]]) = *reinterpret_cast<const uint4*>(&r);
        }
    }
}

// ---- NORM_L2, the reference's arithmetic literally: sum_k (a_k - b_k)^2 in fp32, k ascending ---------------------------------------
// One lane per query (its row in registers, DIM <= 128), 32 train rows per LDS tile read as broadcasts. only_flagged: skip the pairs
// the MFMA kernel has taken.
template <int DIM>
__global__ void __launch_bounds__(256)
bf_l2_loop_kernel(const float* __restrict__ desc, const uint32_t* __restrict__ frame_bad, const int64_t* __restrict__ frame_off,
                  int n_frames, const gms_pair* __restrict__ pairs, int tiles_per_pair, gms_dmatch* __restrict__ matches)
{
    constexpr int kRows = 32;
    __shared__ __attribute__((aligned(16))) float tile[kRows * DIM];
    const int pair_idx = (int)(blockIdx.x / (uint32_t)tiles_per_pair), tl = (int)(blockIdx.x % (uint32_t)tiles_per_pair);
    const gms_pair pr = pairs[pair_idx];
    const FramePair f = frame_pair(pr, frame_off, n_frames);
    const int q0 = tl * 256;
    if (!f.ok || q0 >= f.m) return;
    if (frame_bad != nullptr && !(frame_bad[pr.frame_a] | frame_bad[pr.frame_b])) return;
    const int tid = (int)threadIdx.x;
    const int q = q0 + tid;
    float a[DIM];
    {
        const float4* src = reinterpret_cast<const float4*>(desc + (size_t)(f.offA + min(q, f.m - 1)) * DIM);
#pragma unroll
        for (int k = 0; k < DIM / 4; ++k) {
            const float4 v = src[k];
            a[4 * k] = v.x; a[4 * k + 1] = v.y; a[4 * k + 2] = v.z; a[4 * k + 3] = v.w;
        }
    }
    float best = 3.402823466e+38f;
    int bi = -1;
    const float4* __restrict__ trB = reinterpret_cast<const float4*>(desc + (size_t)f.offB * DIM);
    for (int j0 = 0; j0 < f.nB; j0 += kRows) {
        const int rows = min(kRows, f.nB - j0);
        __syncthreads();
        for (int i = tid; i < rows * (DIM / 4); i += 256) reinterpret_cast<float4*>(tile)[i] = trB[(size_t)j0 * (DIM / 4) + i];
        __syncthreads();
        for (int j = 0; j < rows; ++j) {
            const float4* b4 = reinterpret_cast<const float4*>(tile + j * DIM);
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < DIM / 4; ++k) {
                const float4 b = b4[k];
                float d;
                d = a[4 * k] - b.x;     s = s + d * d;
                d = a[4 * k + 1] - b.y; s = s + d * d;
                d = a[4 * k + 2] - b.z; s = s + d * d;
                d = a[4 * k + 3] - b.w; s = s + d * d;
            }
            if (s < best) {
                best = s;
                bi = j0 + j;
            }
        }
    }
    if (q < f.m) {
        gms_dmatch r;
        r.queryIdx = q;
        r.trainIdx = bi;
        r.imgIdx = 0;
        r.distance = bi >= 0 ? sqrtf(best) : 3.402823466e+38f;
        *reinterpret_cast<uint4*>(&matches[pr.match_off + q]) = *reinterpret_cast<const uint4*>(&r);
    }
}

}  // namespace

// ---- launch helpers ------------------------------------------------------------------------------------------------------------
// prepared block of the L2 path: [total][128] bf16 | [total] float norms | [n_frames] u32 "not SIFT-like" flags
size_t bf_prepared_bytes(int kind, int64_t total, int n_frames)
{
    if (kind != GMS_DESC_L2_F32X128 || total < 0 || n_frames < 0) return 0;
    return ((size_t)total * 256 + (size_t)total * 4 + (size_t)n_frames * 4 + 15) & ~(size_t)15;
}

hipError_t launch_bf_prepare(const void* d_desc, const int64_t* d_frame_off, int n_frames, int64_t total, void* d_prep, hipStream_t stream)
{
    if (total <= 0) return hipSuccess;
    char* base = reinterpret_cast<char*>(d_prep);
    uint32_t* bad = reinterpret_cast<uint32_t*>(base + (size_t)total * 260);
    hipError_t e = hipMemsetAsync(bad, 0, (size_t)n_frames * 4, stream);
    if (e != hipSuccess) return e;
    const int64_t blocks = (total * 32 + 255) / 256;
    hipLaunchKernelGGL(bf_l2_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const float4*>(d_desc),
                       d_frame_off, n_frames, total, reinterpret_cast<uint2*>(base), reinterpret_cast<float*>(base + (size_t)total * 256), bad);
    return hipGetLastError();
}

hipError_t launch_bf_match(int kind, const void* d_desc, const void* d_prep, int64_t total, const int64_t* d_frame_off, int n_frames,
                           const gms_pair* d_pairs, int n_pairs, int max_query, gms_dmatch* d_matches, hipStream_t stream)
{
    if (n_pairs <= 0 || max_query <= 0) return hipSuccess;
    if (kind == GMS_DESC_HAMMING256) {
        // four query rows per lane when the launch fills the chip anyway, one when it does not
        const int tiles4 = (max_query + 1023) / 1024, tiles1 = (max_query + 255) / 256;
        if ((int64_t)tiles4 * n_pairs >= 1024) {
            const uint32_t n = (uint32_t)tiles4 * (uint32_t)n_pairs;
            hipLaunchKernelGGL(bf_hamming_kernel<4>, dim3(n), dim3(256), 0, stream, reinterpret_cast<const uint32_t*>(d_desc), d_frame_off,
                               n_frames, d_pairs, tiles4, n, d_matches);
        } else {
            const uint32_t n = (uint32_t)tiles1 * (uint32_t)n_pairs;
            hipLaunchKernelGGL(bf_hamming_kernel<1>, dim3(n), dim3(256), 0, stream, reinterpret_cast<const uint32_t*>(d_desc), d_frame_off,
                               n_frames, d_pairs, tiles1, n, d_matches);
        }
        return hipGetLastError();
    }
    if (kind == GMS_DESC_L2_F32X128) {
        const char* base = reinterpret_cast<const char*>(d_prep);
        const float* norms = reinterpret_cast<const float*>(base + (size_t)total * 256);
        const uint32_t* bad = reinterpret_cast<const uint32_t*>(base + (size_t)total * 260);
        const int tiles = (max_query + kQueriesPerBlock - 1) / kQueriesPerBlock;
        const uint32_t n = (uint32_t)tiles * (uint32_t)n_pairs;
        hipLaunchKernelGGL(bf_l2_mfma_kernel, dim3(n), dim3(256), 0, stream, reinterpret_cast<const uint4*>(base), norms, bad, d_frame_off,
                           n_frames, d_pairs, tiles, n, d_matches);
        // pairs with a frame that is not SIFT-like (every block of the other pairs returns at once)
        hipLaunchKernelGGL(bf_l2_loop_kernel<kL2Dim>, dim3(n), dim3(256), 0, stream, reinterpret_cast<const float*>(d_desc), bad, d_frame_off,
                           n_frames, d_pairs, tiles, d_matches);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

}  // namespace gms
