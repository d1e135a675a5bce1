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

#include <type_traits>

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

// ---- NORM_HAMMING: the per-frame tables of the matrix-core path ------------------------------------------------------------------
// hamming(a, b) = |a| + |b| - 2 a.b over the 256 bits as 0/1 bytes: each row is expanded to 256 int8 (the A / B operands of
// v_mfma_i32_32x32x32_i8) and its popcount kept beside it. One thread per (row, 16 bits).
__global__ void __launch_bounds__(256)
bf_ham_prepare_kernel(const uint16_t* __restrict__ desc, int64_t total, uint4* __restrict__ rows_i8, int32_t* __restrict__ norms)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // 16 pieces of 16 bits per row
    if (i >= total * 16) return;
    const uint32_t bits = desc[i];
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t n = (bits >> (4 * k)) & 15u;
        w[k] = (n & 1u) | ((n & 2u) << 7) | ((n & 4u) << 14) | ((n & 8u) << 21);  // bit j of the nibble -> byte j
    }
    rows_i8[i] = make_uint4(w[0], w[1], w[2], w[3]);
    uint32_t pc = (uint32_t)__builtin_popcount(bits);
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) pc += (uint32_t)__shfl_xor((int)pc, d, 16);
    if ((threadIdx.x & 15) == 0) norms[i >> 4] = (int32_t)pc;
}

// ---- both norms on the matrix cores ------------------------------------------------------------------------------------------------
// One 256-thread workgroup = 256 query rows of one pair (four waves x 64 query columns, the B operands, resident in registers and
// pre-multiplied by -2) against all train rows of the pair's other frame, streamed through LDS in tiles of 64 rows (the A operands).
// An accumulator starts from |b|^2 of its row, so after the K loop it holds |b|^2 - 2 a.b: the distance minus the query's own
// norm. Per tile a lane only keeps the MINIMUM of its 32 values per query column (one v_min3 per two values) and the tile it came
// from -- tracking the arg-min per value would cost more vector instructions than the tile's MFMAs take. The winning TILE of a
// query is then searched once more at the end, with the norm's own exact arithmetic, for the first minimal row.
//   L2:       rows = bf16 x 128 (256 B), v_mfma_f32_32x32x16_bf16, fp32 accumulators; every value an exact integer (see the file header)
//   Hamming:  rows = int8 x 256 (256 B), v_mfma_i32_32x32x32_i8, int32 accumulators
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

constexpr int kL2Dim = 128;
constexpr int kTileRows = 64;                    // train rows per LDS tile
constexpr uint32_t kRowPitch = 256u + 16u;       // a 256-byte row + 16 B: the 32 rows a ds_read_b128 touches fall on different banks
constexpr uint32_t kTileBytes = kTileRows * kRowPitch;               // 17 408
constexpr int kSub = 2;                          // tiles per staging step (one workgroup barrier per step)
constexpr uint32_t kStepBytes = kSub * kTileBytes;                   // 34 816
constexpr uint32_t kNormOff = 2u * kStepBytes;                       // two steps' tiles, then two steps' norms
constexpr uint32_t kMfmaLdsBytes = kNormOff + 2u * kSub * kTileRows * 4u;   // 70 656: two workgroups per CU
constexpr int kQueriesPerBlock = 256;            // 4 waves x 64 query columns

template <bool HAM>
__global__ void __launch_bounds__(256)
bf_mfma_kernel(const uint4* __restrict__ rows, const uint32_t* __restrict__ norms, const uint32_t* __restrict__ frame_bad,
               const void* __restrict__ raw, const int64_t* __restrict__ frame_off, int n_frames, const gms_pair* __restrict__ pairs,
               int tiles_per_pair, uint32_t n_tasks, gms_dmatch* __restrict__ matches)
{
    using Acc = typename std::conditional<HAM, i32x16, f32x16>::type;
    using Val = typename std::conditional<HAM, int32_t, float>::type;   // an accumulator element / a norm
    __shared__ __attribute__((aligned(16))) unsigned char lds[kMfmaLdsBytes];
    const uint32_t task = xcd_task(blockIdx.x, n_tasks);
    const int pair_idx = (int)(task / (uint32_t)tiles_per_pair), tile = (int)(task % (uint32_t)tiles_per_pair);
    const gms_pair pr = pairs[pair_idx];
    const FramePair f = frame_pair(pr, frame_off, n_frames);
    const int q0 = tile * kQueriesPerBlock;
    if (!f.ok || q0 >= f.m) return;                                                  // workgroup-uniform
    if (!HAM && (frame_bad[pr.frame_a] | frame_bad[pr.frame_b])) return;             // the loop kernel owns this pair
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const Val kBig = HAM ? (Val)0x3FFFFFFF : (Val)3.0e38f;
    if (f.nB <= 0) {  // nothing to match against: the reference's matcher returns no match for the row
        const int q = q0 + tid;
        if (q < f.m) {
            gms_dmatch m;
            m.queryIdx = q;
            m.trainIdx = -1;
            m.imgIdx = 0;
            m.distance = 3.402823466e+38f;
            *reinterpret_cast<uint4*>(&matches[pr.match_off + q]) = *reinterpret_cast<const uint4*>(&m);
        }
        return;
    }

    // ---- this wave's 64 query columns as B operands, scaled by -2 (exact), resident for the whole kernel:
    //      bq[c][s] = -2 * Q[32 c + col][bytes 32 s + 16 half .. + 16)
    uint4 bq[2][8];
    int qrow[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        qrow[c] = q0 + wave * 64 + c * 32 + col;
        const uint4* src = rows + (size_t)(f.offA + min(qrow[c], f.m - 1)) * 16;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const uint4 r4 = src[2 * s + half];
            const uint32_t w[4] = {r4.x, r4.y, r4.z, r4.w};
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (HAM) {
                    o[i] = w[i] * 0xFEu;  // bytes 0 / 1 -> 0 / -2 (0xFE): no carries between bytes
                } else {
                    const float lo = __uint_as_float(w[i] << 16) * -2.0f, hi = __uint_as_float(w[i] & 0xFFFF0000u) * -2.0f;
                    o[i] = (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xFFFF0000u);
                }
            }
            bq[c][s] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    Val bestv[2] = {kBig, kBig};
    int bestt[2] = {0, 0};

    const uint4* __restrict__ trB = rows + (size_t)f.offB * 16;
    const uint32_t* __restrict__ nrmB = norms + f.offB;
    const int n_tiles = (f.nB + kTileRows - 1) / kTileRows;
    // stage step st (kSub tiles = 128 rows x 256 B = 2048 16-byte pieces, eight per thread) into buffer st & 1; rows beyond nB repeat
    // the last row with a norm that never wins
    const int n_steps = (n_tiles + kSub - 1) / kSub;
    static_assert(kSub == 2, "eight named staging registers");
    struct Stage { uint4 v0, v1, v2, v3, v4, v5, v6, v7; uint32_t nv; };  // (named members, returned by value: stays in registers)
    auto stage_load = [&](int st) -> Stage {
        Stage sg;
        auto piece = [&](int i) -> uint4 {
            const int pc = i * 256 + tid;
            return trB[(size_t)min(st * kSub * kTileRows + (pc >> 4), f.nB - 1) * 16 + (pc & 15)];
        };
        sg.v0 = piece(0); sg.v1 = piece(1); sg.v2 = piece(2); sg.v3 = piece(3);
        sg.v4 = piece(4); sg.v5 = piece(5); sg.v6 = piece(6); sg.v7 = piece(7);
        const int r = st * kSub * kTileRows + (tid & (kSub * kTileRows - 1));
        const Val big = kBig;
        sg.nv = (tid < kSub * kTileRows) ? (r < f.nB ? nrmB[r] : __builtin_bit_cast(uint32_t, big)) : 0u;
        return sg;
    };
    auto stage_store = [&](int st, const Stage& sg) {
        unsigned char* base = lds + (uint32_t)(st & 1) * kStepBytes;
        auto put = [&](int i, const uint4& v) {
            const int pc = i * 256 + tid;
            *reinterpret_cast<uint4*>(base + (uint32_t)(pc >> 4) * kRowPitch + (uint32_t)(pc & 15) * 16u) = v;
        };
        put(0, sg.v0); put(1, sg.v1); put(2, sg.v2); put(3, sg.v3);
        put(4, sg.v4); put(5, sg.v5); put(6, sg.v6); put(7, sg.v7);
        if (tid < kSub * kTileRows) reinterpret_cast<uint32_t*>(lds + kNormOff)[(st & 1) * kSub * kTileRows + tid] = sg.nv;
    };
    Stage sv = stage_load(0);
    stage_store(0, sv);
    __syncthreads();
    for (int st = 0; st < n_steps; ++st) {
        const bool more = st + 1 < n_steps;
        if (more) sv = stage_load(st + 1);  // in flight during this step's MFMAs
#pragma unroll
        for (int u = 0; u < kSub; ++u) {
        const int t = st * kSub + u;
        if (t >= n_tiles) break;            // workgroup-uniform
        const unsigned char* tb = lds + (uint32_t)(st & 1) * kStepBytes + (uint32_t)u * kTileBytes;
        const uint32_t* nb = reinterpret_cast<const uint32_t*>(lds + kNormOff) + ((st & 1) * kSub + u) * kTileRows;
        Acc acc[2][2];
        // C-in = the norm of the accumulator's row: rows (reg & 3) + 8 (reg >> 2) + 4 half of the 32-row block
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint4 n4 = *reinterpret_cast<const uint4*>(nb + rb * 32 + 8 * g + 4 * half);
                acc[rb][0][4 * g + 0] = __builtin_bit_cast(Val, n4.x); acc[rb][0][4 * g + 1] = __builtin_bit_cast(Val, n4.y);
                acc[rb][0][4 * g + 2] = __builtin_bit_cast(Val, n4.z); acc[rb][0][4 * g + 3] = __builtin_bit_cast(Val, n4.w);
            }
            acc[rb][1] = acc[rb][0];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            // A operand: train row (32 rb + col), bytes 32 s + 16 half .. + 16 (the same bytes of the row as the B operand's: the
            // dot product pairs equal positions whatever order the instruction visits them in)
            const uint4 a0 = *reinterpret_cast<const uint4*>(tb + (uint32_t)col * kRowPitch + (uint32_t)(32 * s + 16 * half));
            const uint4 a1 = *reinterpret_cast<const uint4*>(tb + (uint32_t)(32 + col) * kRowPitch + (uint32_t)(32 * s + 16 * half));
            if constexpr (HAM) {
                acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a0), __builtin_bit_cast(i32x4, bq[0][s]), acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a0), __builtin_bit_cast(i32x4, bq[1][s]), acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a1), __builtin_bit_cast(i32x4, bq[0][s]), acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a1), __builtin_bit_cast(i32x4, bq[1][s]), acc[1][1], 0, 0, 0);
            } else {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bq[0][s]), acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bq[1][s]), acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, bq[0][s]), acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, bq[1][s]), acc[1][1], 0, 0, 0);
            }
        }
        // the tile's minimum per query column (this lane's 32 of the 64 rows), and the first tile that reached it
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            Val mn = acc[0][c][0];
#pragma unroll
            for (int reg = 1; reg < 16; ++reg) mn = min(mn, acc[0][c][reg]);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) mn = min(mn, acc[1][c][reg]);
            const bool lt = mn < bestv[c];
            bestv[c] = lt ? mn : bestv[c];
            bestt[c] = lt ? t : bestt[c];
        }
        }
        if (more) stage_store(st + 1, sv);  // the other buffer: last read one barrier ago
        __syncthreads();
    }
    // ---- the two halves of the wave hold disjoint row sets of the same query: lower value, then earlier tile
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const Val ov = __builtin_bit_cast(Val, __shfl_xor(__builtin_bit_cast(int, bestv[c]), 32));
        const int ot = __shfl_xor(bestt[c], 32);
        const bool take = ov < bestv[c] || (ov == bestv[c] && ot < bestt[c]);
        bestt[c] = take ? ot : bestt[c];
    }
    // ---- every query's winning tile once more, one lane per train row, in the norm's own arithmetic: the first minimal row
    gms_dmatch* __restrict__ out = matches + pr.match_off;
    for (int i = 0; i < 64; ++i) {
        const int q = q0 + wave * 64 + i;
        if (q >= f.m) break;                                                          // wave-uniform
        const int tstar = __shfl(i < 32 ? bestt[0] : bestt[1], i & 31);
        uint32_t key;
        if constexpr (HAM) {
            const int r = tstar * kTileRows + lane, rr = min(r, f.nB - 1);
            const uint4* qa = reinterpret_cast<const uint4*>(reinterpret_cast<const uint32_t*>(raw) + (size_t)(f.offA + q) * 8);
            const uint4* tb4 = reinterpret_cast<const uint4*>(reinterpret_cast<const uint32_t*>(raw) + (size_t)(f.offB + rr) * 8);
            const uint4 x0 = qa[0], x1 = qa[1], y0 = tb4[0], y1 = tb4[1];
            const uint32_t d = (uint32_t)(__builtin_popcount(x0.x ^ y0.x) + __builtin_popcount(x0.y ^ y0.y) + __builtin_popcount(x0.z ^ y0.z) +
                                          __builtin_popcount(x0.w ^ y0.w) + __builtin_popcount(x1.x ^ y1.x) + __builtin_popcount(x1.y ^ y1.y) +
                                          __builtin_popcount(x1.z ^ y1.z) + __builtin_popcount(x1.w ^ y1.w));
            key = r < f.nB ? (d << 6) | (uint32_t)lane : 0xFFFFFFFFu;
        } else {
            // The tile's 64 rows are 16 KB of consecutive memory: read them coalesced -- a wave instruction covers four whole rows,
            // lane = (row of the four, 16-byte piece of the row) -- and add a row's sixteen partial dot products up across its lanes.
            // bf16 rows hold the integers exactly, so |a|^2 + |b|^2 - 2 a.b (two elements per v_dot2c_f32_bf16) is made of exact
            // integers below 2^24 in any order.
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            const int piece = lane & 15, rsub = lane >> 4;
            const uint4 x = rows[(size_t)(f.offA + q) * 16 + piece];
            const float na = __uint_as_float(norms[f.offA + q]);
            key = 0xFFFFFFFFu;
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const int row = tstar * kTileRows + 4 * j + rsub;
                const int rowc = min(row, f.nB - 1);
                const uint4 y = rows[(size_t)(f.offB + rowc) * 16 + piece];
                float dot = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x.x), __builtin_bit_cast(bf16x2, y.x), 0.0f, false);
                dot = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x.y), __builtin_bit_cast(bf16x2, y.y), dot, false);
                dot = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x.z), __builtin_bit_cast(bf16x2, y.z), dot, false);
                dot = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x.w), __builtin_bit_cast(bf16x2, y.w), dot, false);
#pragma unroll
                for (int d = 8; d >= 1; d >>= 1) dot += __shfl_xor(dot, d, 16);
                const float d2 = (na + __uint_as_float(nrmB[rowc])) - 2.0f * dot;
                const uint32_t kj = row < f.nB ? ((uint32_t)d2 << 6) | (uint32_t)(4 * j + rsub) : 0xFFFFFFFFu;  // d^2 <= 128 * 255^2 < 2^23
                key = min(key, kj);
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, d));
        if (lane == 0) {
            gms_dmatch m;
            m.queryIdx = q;
            m.trainIdx = f.nB > 0 ? tstar * kTileRows + (int)(key & 63u) : -1;
            m.imgIdx = 0;
            m.distance = f.nB > 0 ? (HAM ? (float)(key >> 6) : sqrtf((float)(key >> 6))) : 3.402823466e+38f;
            *reinterpret_cast<uint4*>(&out[q]) = *reinterpret_cast<const uint4*>(&m);
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
// prepared block:  L2       [total][128] bf16 | [total] float norms | [n_frames] u32 "not SIFT-like" flags
//                  Hamming  [total][256] int8 | [total] int32 popcounts
size_t bf_prepared_bytes(int kind, int64_t total, int n_frames)
{
    if (total < 0 || n_frames < 0) return 0;
    if (kind == GMS_DESC_L2_F32X128) return ((size_t)total * 260 + (size_t)n_frames * 4 + 15) & ~(size_t)15;
    if (kind == GMS_DESC_HAMMING256) return ((size_t)total * 260 + 15) & ~(size_t)15;
    return 0;
}

hipError_t launch_bf_prepare(int kind, const void* d_desc, const int64_t* d_frame_off, int n_frames, int64_t total, void* d_prep,
                             hipStream_t stream)
{
    if (total <= 0) return hipSuccess;
    char* base = reinterpret_cast<char*>(d_prep);
    if (kind == GMS_DESC_HAMMING256) {
        const int64_t blocks = (total * 16 + 255) / 256;
        hipLaunchKernelGGL(bf_ham_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const uint16_t*>(d_desc), total,
                           reinterpret_cast<uint4*>(base), reinterpret_cast<int32_t*>(base + (size_t)total * 256));
        return hipGetLastError();
    }
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
    const char* base = reinterpret_cast<const char*>(d_prep);
    const uint32_t* norms = reinterpret_cast<const uint32_t*>(base + (size_t)total * 256);
    const int tiles = (max_query + kQueriesPerBlock - 1) / kQueriesPerBlock;
    const uint32_t n = (uint32_t)tiles * (uint32_t)n_pairs;
    if (kind == GMS_DESC_HAMMING256 && d_prep != nullptr) {
        hipLaunchKernelGGL(bf_mfma_kernel<true>, dim3(n), dim3(256), 0, stream, reinterpret_cast<const uint4*>(base), norms,
                           (const uint32_t*)nullptr, d_desc, d_frame_off, n_frames, d_pairs, tiles, n, d_matches);
        return hipGetLastError();
    }
    if (kind == GMS_DESC_HAMMING256) {
        // no prepared block: the vector-ALU kernel on the raw rows; four query rows per lane when the launch fills the chip anyway
        const int tiles4 = (max_query + 1023) / 1024, tiles1 = (max_query + 255) / 256;
        if ((int64_t)tiles4 * n_pairs >= 1024) {
            const uint32_t n4 = (uint32_t)tiles4 * (uint32_t)n_pairs;
            hipLaunchKernelGGL(bf_hamming_kernel<4>, dim3(n4), dim3(256), 0, stream, reinterpret_cast<const uint32_t*>(d_desc), d_frame_off,
                               n_frames, d_pairs, tiles4, n4, d_matches);
        } else {
            const uint32_t n1 = (uint32_t)tiles1 * (uint32_t)n_pairs;
            hipLaunchKernelGGL(bf_hamming_kernel<1>, dim3(n1), dim3(256), 0, stream, reinterpret_cast<const uint32_t*>(d_desc), d_frame_off,
                               n_frames, d_pairs, tiles1, n1, d_matches);
        }
        return hipGetLastError();
    }
    if (kind == GMS_DESC_L2_F32X128) {
        const uint32_t* bad = reinterpret_cast<const uint32_t*>(base + (size_t)total * 260);
        hipLaunchKernelGGL(bf_mfma_kernel<false>, dim3(n), dim3(256), 0, stream, reinterpret_cast<const uint4*>(base), norms, bad, d_desc,
                           d_frame_off, n_frames, d_pairs, tiles, n, d_matches);
        // pairs with a frame that is not SIFT-like (every block of the other pairs returns at once)
        hipLaunchKernelGGL(bf_l2_loop_kernel<kL2Dim>, dim3(n), dim3(256), 0, stream, reinterpret_cast<const float*>(d_desc), bad, d_frame_off,
                           n_frames, d_pairs, tiles, d_matches);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

}  // namespace gms
