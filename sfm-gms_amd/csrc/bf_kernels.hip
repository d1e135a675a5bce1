// bf_kernels.hip -- brute-force descriptor matcher for gfx950: the producer of the match array the GMS filter consumes.
//
// What the reference runs in front of matchGMS (FeatureMatchUtil.cpp:66-68; DisparityUtil.cpp:104-109,143):
//     BFMatcher::create()->match(descriptors1, descriptors2, matches)        NORM_L2, no cross-check
// -- one DMatch per query row i: {queryIdx = i, trainIdx = arg min_j dist(i, j) (first minimum), imgIdx = 0, distance}.
// Here it works on the same resident frame table as the filter: descriptor i of a frame belongs to keypoint i, pair
// (frame_a, frame_b) gets its n(frame_a) matches written at d_matches[match_off ...] -- straight into the batch's match
// array, so the matches never cross PCIe.
//
//   bf_mfma_kernel<true>               NORM_HAMMING, 256-bit rows (ORB), with a prepared block: the bits as FP4 elements on
//       v_mfma_scale_f32_32x32x64_f8f6f4, the arg-min riding in the accumulator's fraction (details at the kernel).
//   bf_mfma_kernel<false>              NORM_L2, 128 floats (SIFT): SIFT descriptors are integers 0..255 stored as floats; as int8
//       (a - 128 on the train side, 127 - b on the query side) the cross term of d^2 runs on v_mfma_i32_32x32x32_i8 and a short
//       exact search of the winning 32-row block settles the arg-min; d^2 is the exact integer the reference's fp32 loop produces
//       (every partial sum below 2^24), so sqrtf(d^2) is bit-identical. gms_bf_prepare_device checks the integer property per
//       frame while it builds the int8 table; a pair with a frame that fails it takes
//   bf_l2_loop_kernel                  the reference's own arithmetic, sum_k (a_k - b_k)^2 in fp32 in index order (no FMA
//       contraction), one lane per query, train rows broadcast from LDS. Slow and exact; also any dimension other than 128.
//   bf_hamming_kernel                  NORM_HAMMING without a prepared block: integer VALU. A lane keeps Q query rows in registers;
//       the train rows are wave-uniform, so they arrive through the SCALAR cache (s_load_dwordx8) and feed v_xor / v_bcnt directly
//       as SGPR operands: no LDS, no vector loads in the loop. 18 VALU instructions per (query, train): 8 xor, 8 popcount-
//       accumulate, one shift-or that packs (distance, trainIdx), one unsigned min (lowest trainIdx wins ties for free).
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

// ---- the per-frame tables of the matrix-core paths ---------------------------------------------------------------------------------
// NORM_L2: one 32-lane group per 128-float row (16 bytes per lane). A SIFT-like row (integers 0..255) becomes 128 int8 a' = a - 128
// with w = sum (a' + 1)^2 beside it; a row with any other value flags its frame (whose pairs take the
// loop kernel).
__global__ void __launch_bounds__(256)
bf_l2_prepare_kernel(const float4* __restrict__ desc, const int64_t* __restrict__ frame_off, int n_frames, int64_t total,
                     uint32_t* __restrict__ rows_i8, int32_t* __restrict__ w_norms, uint32_t* __restrict__ frame_bad)
{
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
    const int part = (int)(threadIdx.x & 31);
    if (row >= total) return;
    const float4 v = desc[row * 32 + part];
    auto is_u8 = [](float x) { return x >= 0.0f && x <= 255.0f && x == floorf(x); };
    const bool bad = !(is_u8(v.x) && is_u8(v.y) && is_u8(v.z) && is_u8(v.w));
    auto shifted = [](float x) { return (int)fminf(fmaxf(x, 0.0f), 255.0f) - 128; };  // (the clamp only matters for a flagged row)
    const int e0 = shifted(v.x), e1 = shifted(v.y), e2 = shifted(v.z), e3 = shifted(v.w);
    rows_i8[row * 32 + part] = (uint32_t)(e0 & 255) | ((uint32_t)(e1 & 255) << 8) | ((uint32_t)(e2 & 255) << 16) | ((uint32_t)(e3 & 255) << 24);
    int w = (e0 + 1) * (e0 + 1) + (e1 + 1) * (e1 + 1) + (e2 + 1) * (e2 + 1) + (e3 + 1) * (e3 + 1);
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) w += __shfl_xor(w, d, 32);
    if (part == 0) w_norms[row] = w;
    if (__ballot(bad) != 0ull && bad) {
        int lo = 0, hi = n_frames - 1;  // last frame f with frame_off[f] <= row
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (frame_off[mid] <= row) lo = mid; else hi = mid - 1;
        }
        frame_bad[lo] = 1u;  // benign race: every writer stores 1
    }
}

// NORM_HAMMING: each of the 256 bits becomes one FP4 (E2M1) element -- 1.0 (0b0010) or 0 -- so a row is 128 bytes; its popcount
// is kept beside it as a float. One thread per (row, 16 bits).
__global__ void __launch_bounds__(256)
bf_ham_prepare_kernel(const uint16_t* __restrict__ desc, int64_t total, uint2* __restrict__ rows_fp4, float* __restrict__ norms)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // 16 pieces of 16 bits per row
    if (i >= total * 16) return;
    const uint32_t bits = desc[i];
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j >> 3] |= ((bits >> j) & 1u) << (4 * (j & 7) + 1);
    rows_fp4[i] = make_uint2(w[0], w[1]);
    uint32_t pc = (uint32_t)__builtin_popcount(bits);
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) pc += (uint32_t)__shfl_xor((int)pc, d, 16);
    if ((threadIdx.x & 15) == 0) norms[i >> 4] = (float)pc;
}

// ---- both norms on the matrix cores ------------------------------------------------------------------------------------------------
// One 256-thread workgroup = 256 query rows of one pair (four waves x 64 query columns, the B operands, resident in registers)
// against all train rows of the pair's other frame, streamed through LDS in tiles of 64 rows (the A operands), four tiles per
// staging step. Tracking the arg-min per accumulator element would cost more vector instructions than the MFMAs take; a lane only
// takes the MINIMUM of its 16 values per query column and 32-row block (one v_min3 per two values) and remembers the block.
//   Hamming  rows = 256 x FP4 (128 B), v_mfma_scale_f32_32x32x64_f8f6f4 with both formats FP4 and unit scales (four per block and
//            accumulator, at twice the int8 rate). hamming = |a| + |b| - 2 a.b over 0/1 elements: the B operand is the query row
//            times -2 (FP4 holds -2 exactly: every nibble 0x2 -> 0xC); the accumulator starts from the train row's popcount PLUS
//            (train row mod 32768) / 32768, so it ends as |a| - 2 a.b + row / 32768 -- |value| < 512 with 15 fraction bits: exact
//            in fp32 -- and the plain minimum over ALL the accumulators a lane ever sees already is "smallest distance, then
//            lowest row": integer part = distance - |b|, fraction = the row. One v_min3 per two values is the whole bookkeeping;
//            nothing is compared per block. Frames beyond 32768 rows fold the running minimum once per 32768 rows (kHamChunk).
//   L2       rows = 128 x int8 (128 B), v_mfma_i32_32x32x32_i8 (four per block and accumulator; the bf16 form needs eight at half the
//            rate). With a' = a - 128 and the query side as ~b' = 127 - b (a bytewise NOT: both fit int8 for every value 0..255),
//            a - b = (a' + 1) + ~b', so d^2 = w(a) + 2 a'.~b' + [sum ~b'^2 + 2 sum ~b'] with w(a) = sum (a' + 1)^2; the bracket is
//            the query's alone. The accumulator starts from floor(w / 2) and ends as P = floor((w + 2 a'.~b') / 2): the row of
//            the smallest d^2 has the smallest P, but two rows whose d^2 differ by one can share it. The winning BLOCK of a query
//            is therefore searched once more at the end, in exact integers, for the first minimal row. A query whose minimal P
//            shows up in a second block as well (d^2 within one of the minimum, in another block: a few per cent of the queries
//            that have no true match) gets both searched; in more than two, every block from the first such one on.
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) int i32x16;

constexpr int kL2Dim = 128;
constexpr int kTileRows = 64;                    // train rows per LDS tile
constexpr uint32_t kRowBytes = 128u;             // a prepared row, either kind
constexpr uint32_t kRowPitch = kRowBytes + 16u;  // + 16 B: the rows a ds_read_b128 touches spread over the banks
constexpr uint32_t kTileBytes = kTileRows * kRowPitch;               // 9216
constexpr int kSub = 2;                          // tiles per staging step (one workgroup barrier per step)
constexpr int kStepRows = kSub * kTileRows;      // 128
constexpr uint32_t kStepBytes = kSub * kTileBytes;                   // 18 432
constexpr uint32_t kNormOff = 2u * kStepBytes;                       // two steps' tiles, then two steps' norms
constexpr uint32_t kMfmaLdsBytes = kNormOff + 2u * kStepRows * 4u;   // 37 888
constexpr int kNQ = 4;                           // 32-column query sets (= accumulators) per wave: 128 query columns
constexpr int kQueriesPerBlock = 4 * 32 * kNQ;   // 512: 4 waves x 128 query columns
constexpr int kKSteps = 4;                       // MFMAs per (tile, accumulator): 4 x 32 bytes of a row
constexpr int kFp4UnitScale = 0x7F7F7F7F;        // E8M0 127 = 2^0 in every byte
constexpr int kHamChunk = 32768;                 // Hamming: train rows whose index rides in one fp32 fraction (a multiple of kStepRows)

// L2, the exact search of a 32-row block, eight queries per wave: the eight lanes of a group (lane >> 3) hold the 16-byte pieces
// (lane & 7) of the group's query and walk the 32 rows of the group's OWN block, so a group's load is one whole 128-byte row.
// With the query side as nb = ~b' (the B operand of the matrix pass): d^2 = w(a) + 2 a'.nb + [nb.nb + 2 sum nb]; four elements per
// v_dot4_i32_i8, the eight pieces added up with three DPP adds (the full sum lands in lanes 4..7 of the group, which all keep the
// same running minimum). Returns (d^2 << 5 | row inside the block) of the block's first nearest row -- valid in lanes 4..7.
__device__ __forceinline__ int dot16(const uint4& a, const uint4& b)
{
    int d = __builtin_amdgcn_sdot4((int)a.x, (int)b.x, 0, false);
    d = __builtin_amdgcn_sdot4((int)a.y, (int)b.y, d, false);
    d = __builtin_amdgcn_sdot4((int)a.z, (int)b.z, d, false);
    return __builtin_amdgcn_sdot4((int)a.w, (int)b.w, d, false);
}

// ROWS = 32: the whole block. ROWS = 16: only the rows whose accumulators sat in lane half `hsel` of the matrix pass -- rows
// (reg & 3) + 8 (reg >> 2) + 4 hsel -- when the minimum is known to have come from that half.
template <int ROWS>
__device__ __forceinline__ uint32_t l2_block_key(const uint4* __restrict__ rows, const uint32_t* __restrict__ w_norms, const uint4& nb,
                                                 int query_part, int64_t offB, int nB, int blk, int piece, int hsel)
{
    uint32_t key = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {   // (all loads in flight together)
        const int r = ROWS == 32 ? i : 8 * (i >> 2) + 4 * hsel + (i & 3);
        const int row = blk * 32 + r;
        const int64_t rr = offB + min(row, nB - 1);
        const uint4 y = rows[(size_t)rr * 8 + piece];
        const int w = (int)w_norms[rr];
        int d2 = 2 * dot16(y, nb) + query_part;
        d2 += __builtin_amdgcn_update_dpp(0, d2, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
        d2 += __builtin_amdgcn_update_dpp(0, d2, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]: every lane of a quad has the quad's sum
        d2 += __builtin_amdgcn_update_dpp(0, d2, 0x114, 0xF, 0xF, true);   // row_shr:4: lanes 4..7 add the quad before them
        d2 += w;
        key = min(key, row < nB ? ((uint32_t)d2 << 5) | (uint32_t)r : 0xFFFFFFFFu);   // d^2 <= 128 * 255^2 < 2^23
    }
    return key;
}

template <bool HAM>
__global__ void __launch_bounds__(256, 2)
bf_mfma_kernel(const uint4* __restrict__ rows, const uint32_t* __restrict__ norms, const uint32_t* __restrict__ frame_bad, const int64_t* __restrict__ frame_off, int n_frames,
               const gms_pair* __restrict__ pairs, int tiles_per_pair, uint32_t n_tasks, gms_dmatch* __restrict__ matches)
{
    using Acc = typename std::conditional<HAM, f32x16, i32x16>::type;
    using Val = typename std::conditional<HAM, float, int32_t>::type;   // an accumulator element / a norm
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
    const Val kBig = HAM ? (Val)3.0e38f : (Val)0x3FFFFFFF;
    if (f.nB <= 0) {  // nothing to match against: the reference's matcher returns no match for the row
        for (int q = q0 + tid; q < min(f.m, q0 + kQueriesPerBlock); q += 256) {
            gms_dmatch m;
            m.queryIdx = q;
            m.trainIdx = -1;
            m.imgIdx = 0;
            m.distance = 3.402823466e+38f;
            *reinterpret_cast<uint4*>(&matches[pr.match_off + q]) = *reinterpret_cast<const uint4*>(&m);
        }
        return;
    }

    // ---- this wave's 128 query columns as B operands, resident for the whole kernel:
    //      bq[c][s] = transformed Q[32 c + col][bytes 32 s + 16 half .. + 16)
    uint4 bq[kNQ][kKSteps];
    int qrow[kNQ];
#pragma unroll
    for (int c = 0; c < kNQ; ++c) {
        qrow[c] = q0 + wave * (32 * kNQ) + c * 32 + col;
        const uint4* src = rows + (size_t)(f.offA + min(qrow[c], f.m - 1)) * (kRowBytes / 16);
#pragma unroll
        for (int s = 0; s < kKSteps; ++s) {
            const uint4 r4 = src[2 * s + half];
            if (HAM) bq[c][s] = make_uint4(r4.x * 6u, r4.y * 6u, r4.z * 6u, r4.w * 6u);  // nibbles 0x2 -> 0xC (-2.0): no carries
            else bq[c][s] = make_uint4(~r4.x, ~r4.y, ~r4.z, ~r4.w);
        }
    }
    Val bestv[kNQ], bestd[kNQ];   // L2: the smallest block minimum so far / Hamming: the smallest distance part (distance - |b|) of the chunks folded so far
    int bestt[kNQ];               // L2: the block (of 32 train rows) bestv came from, the first one that reached it; Hamming: bestd's train row
    int ties[kNQ], bestt2[kNQ];   // L2: in how many later blocks the minimal P was seen again, and the last of them
    int whalf[kNQ];               // L2: which lane half's rows of the winning block hold the minimum (2 = either)
    Val mn[kNQ];                  // the running minimum (L2: of the block being folded; Hamming: of everything since the last chunk fold)
#pragma unroll
    for (int c = 0; c < kNQ; ++c) {
        bestv[c] = bestd[c] = mn[c] = kBig;
        bestt[c] = ties[c] = bestt2[c] = 0;
        whalf[c] = half;
    }

    const uint4* __restrict__ trB = rows + (size_t)f.offB * (kRowBytes / 16);
    const uint32_t* __restrict__ nrmB = norms + f.offB;
    const int n_tiles = (f.nB + kTileRows - 1) / kTileRows;
    // stage step st (kSub tiles = 128 rows x 128 B = 1024 16-byte pieces, four per thread) into buffer st & 1; rows beyond nB repeat
    // the last row with a norm that never wins
    const int n_steps = (n_tiles + kSub - 1) / kSub;
    static_assert(kStepRows * (kRowBytes / 16) == 4 * 256, "four named staging registers");
    struct Stage { uint4 v0, v1, v2, v3; uint32_t nv; };  // (named members, returned by value: stays in registers)
    auto stage_load = [&](int st) -> Stage {
        Stage sg;
        auto piece = [&](int i) -> uint4 {
            const int pc = i * 256 + tid;
            return trB[(size_t)min(st * kStepRows + (pc >> 3), f.nB - 1) * 8 + (pc & 7)];
        };
        sg.v0 = piece(0); sg.v1 = piece(1); sg.v2 = piece(2); sg.v3 = piece(3);
        // (the norm stays raw here: turning it into the accumulator's start value would wait for it -- and with it for the row loads
        //  just issued -- a whole memory latency before the step's products; stage_store does that a step later)
        sg.nv = nrmB[min(st * kStepRows + (tid & (kStepRows - 1)), f.nB - 1)];
        return sg;
    };
    auto stage_store = [&](int st, const Stage& sg) {
        unsigned char* base = lds + (uint32_t)(st & 1) * kStepBytes;
        auto put = [&](int i, const uint4& v) {
            const int pc = i * 256 + tid;
            *reinterpret_cast<uint4*>(base + (uint32_t)(pc >> 3) * kRowPitch + (uint32_t)(pc & 7) * 16u) = v;
        };
        put(0, sg.v0); put(1, sg.v1); put(2, sg.v2); put(3, sg.v3);
        if (tid < kStepRows) {
            const int r = st * kStepRows + tid;
            Val nv = kBig;
            if (r < f.nB) {
                if constexpr (HAM) nv = __uint_as_float(sg.nv) + (float)(r & (kHamChunk - 1)) * (1.0f / (float)kHamChunk);   // popcount + row / 32768
                else nv = (int32_t)sg.nv >> 1;                                                                            // floor(w / 2)
            }
            reinterpret_cast<Val*>(lds + kNormOff)[(st & 1) * kStepRows + tid] = nv;
        }
    };
    // ---- the main loop, scheduled by hand in the source (scheduling barriers keep the compiler from re-ordering it).
    // A BLOCK = 32 train rows x the wave's 128 query columns = four accumulators in two PAIRS (A: columns 0..63, B: 64..127), sixteen
    // MFMAs: the four A pieces of the block feed pair A's four k-steps, then pair B's -- every staged row and every LDS read serves
    // twice the products of the two-accumulator kernel of round 2 (the kernel runs at the chip's power limit: fewer bytes moved per
    // product is what buys throughput). C-in = the norm of the accumulator's row (rows (reg & 3) + 8 (reg >> 2)
    // + 4 half of the block), so after the K loop an accumulator holds |a| - 2 a.b + row / 32768 (Hamming) / h + a'.~b' (L2). The
    // matrix pipe takes an MFMA every 32 cycles and leaves the vector ALU free for most of them: while one pair's MFMAs run, the
    // OTHER pair's finished accumulators are folded into the minima (v_min3), so all four accumulators are live and none is copied.
    // The LDS reads of the next block's A pieces go out under pair A's k-steps, its norms under pair B's (after B's first k-step
    // has consumed this block's). Every step runs both tiles: rows past the end carry norms that never win.
    const uint32_t a_lane = (uint32_t)col * kRowPitch + 16u * (uint32_t)half;   // this lane's A row inside a block + its 16 bytes of a k-step
    auto lds_a = [&](int st, int blk, int s) -> uint4 {   // blk = 2 * tile + row block
        return *reinterpret_cast<const uint4*>(lds + (uint32_t)(st & 1) * kStepBytes + (uint32_t)blk * (32u * kRowPitch) + a_lane + 32u * (uint32_t)s);
    };
    auto lds_n = [&](int st, int blk, int g) -> uint4 {
        return *reinterpret_cast<const uint4*>(lds + kNormOff + (uint32_t)(((st & 1) * kSub * 2 + blk) * 32 + 8 * g + 4 * half) * 4u);
    };
    auto as_acc = [](const uint4& n0, const uint4& n1, const uint4& n2, const uint4& n3) -> Acc {
        const uint32_t w[16] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, n3.x, n3.y, n3.z, n3.w};
        Acc r;
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = __builtin_bit_cast(Val, w[i]);
        return r;
    };
    auto mfma = [&](const uint4& a, const uint4& b, const Acc& c) -> Acc {
        if constexpr (HAM) {
            auto wide = [](const uint4& v) { return i32x8{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0}; };
            // formats: cbsz = blgp = 4 (FP4 E2M1); both scales 2^0
            return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wide(a), wide(b), c, 4, 4, 0, kFp4UnitScale, 0, kFp4UnitScale);
        } else {
            return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
        }
    };
    Acc acc[kNQ];   // pair A = acc[0], acc[1]; pair B = acc[2], acc[3]
    auto min_regs = [&](int pair, int from, int to) {   // fold registers [from, to) of the pair's two accumulators
#pragma unroll
        for (int reg = from; reg < to; ++reg) {
            mn[2 * pair] = min(mn[2 * pair], acc[2 * pair][reg]);
            mn[2 * pair + 1] = min(mn[2 * pair + 1], acc[2 * pair + 1][reg]);
        }
    };
    auto block_compare = [&](int pair, int blk) {   // L2: the first block that reached the minimum stays (Hamming: mn simply keeps running)
        if constexpr (!HAM) {
#pragma unroll
            for (int c = 2 * pair; c < 2 * pair + 2; ++c) {
                const bool lt = mn[c] < bestv[c], eq = mn[c] == bestv[c];
                ties[c] = lt ? 0 : ties[c] + (eq ? 1 : 0);
                bestt2[c] = eq ? blk : bestt2[c];
                bestv[c] = lt ? mn[c] : bestv[c];
                bestt[c] = lt ? blk : bestt[c];
                mn[c] = kBig;
            }
        }
    };
    auto ham_fold = [&](int chunk) {   // Hamming: the running minimum of a chunk of kHamChunk rows into (distance part, row)
#pragma unroll
        for (int c = 0; c < kNQ; ++c) {
            const float d = floorf((float)mn[c]);   // equal distances: the earlier chunk holds the lower rows
            const bool lt = d < (float)bestd[c];
            bestt[c] = lt ? chunk * kHamChunk + (int)(((float)mn[c] - d) * (float)kHamChunk) : bestt[c];
            bestd[c] = lt ? (Val)d : bestd[c];
            mn[c] = kBig;
        }
    };
    auto set_big = [&](int pair) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) acc[2 * pair][reg] = acc[2 * pair + 1][reg] = kBig;
    };
    // staging runs a step ahead in memory and a step ahead in LDS: step st + 2 is requested while step st + 1, requested a step ago,
    // goes into the other LDS buffer (last read before the barrier that ended step st - 1) -- all of it BEFORE the step's products
    Stage sv = stage_load(0);
    stage_store(0, sv);
    if (n_steps > 1) sv = stage_load(1);
    __syncthreads();
    set_big(0);   // at first dummies that beat nothing
    set_big(1);
    for (int st = 0; st < n_steps; ++st) {
        if constexpr (HAM) {
            if (st != 0 && (st & (kHamChunk / kStepRows - 1)) == 0) {   // a new chunk of rows (rare): settle the one before, the lagging pair included
                min_regs(1, 0, 16);
                ham_fold(st / (kHamChunk / kStepRows) - 1);
                set_big(1);
            }
        }
#if !defined(BF_DIAG) || BF_DIAG != 2
        if (st + 1 < n_steps) stage_store(st + 1, sv);
        if (st + 2 < n_steps) sv = stage_load(st + 2);
#endif
        uint4 a_cur[kKSteps], a_nxt[kKSteps];
#pragma unroll
        for (int s = 0; s < kKSteps; ++s) a_cur[s] = lds_a(st, 0, s);
        uint4 n0 = lds_n(st, 0, 0), n1 = lds_n(st, 0, 1), n2 = lds_n(st, 0, 2), n3 = lds_n(st, 0, 3);
#pragma unroll
        for (int blk = 0; blk < 2 * kSub; ++blk) {
            const int gblk = st * 2 * kSub + blk;   // the block's number in the frame
            const bool more = blk + 1 < 2 * kSub;
            const Acc nrm = as_acc(n0, n1, n2, n3);
            // ---- pair A; pair B of the block before is folded meanwhile (its last MFMAs are still in the pipe during k-step 0)
            acc[1] = mfma(a_cur[0], bq[1][0], nrm);
            acc[0] = mfma(a_cur[0], bq[0][0], nrm);
            if (more) a_nxt[0] = lds_a(st, blk + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 1; s < kKSteps; ++s) {
                acc[0] = mfma(a_cur[s], bq[0][s], acc[0]);
                acc[1] = mfma(a_cur[s], bq[1][s], acc[1]);
                if (more) a_nxt[s] = lds_a(st, blk + 1, s);
                min_regs(1, s == 1 ? 0 : (s == 2 ? 6 : 11), s == 1 ? 6 : (s == 2 ? 11 : 16));
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- pair B; pair A of this block is folded meanwhile
            acc[3] = mfma(a_cur[0], bq[3][0], nrm);
            acc[2] = mfma(a_cur[0], bq[2][0], nrm);
            block_compare(1, gblk - 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 1; s < kKSteps; ++s) {
                acc[2] = mfma(a_cur[s], bq[2][s], acc[2]);
                acc[3] = mfma(a_cur[s], bq[3][s], acc[3]);
                if (more) {
                    if (s == 1) { n0 = lds_n(st, blk + 1, 0); n1 = lds_n(st, blk + 1, 1); }
                    if (s == 2) { n2 = lds_n(st, blk + 1, 2); n3 = lds_n(st, blk + 1, 3); }
                }
                min_regs(0, s == 1 ? 0 : (s == 2 ? 6 : 11), s == 1 ? 6 : (s == 2 ? 11 : 16));
                __builtin_amdgcn_sched_barrier(0);
            }
            block_compare(0, gblk);
#pragma unroll
            for (int s = 0; s < kKSteps; ++s) a_cur[s] = a_nxt[s];
        }
#if !defined(BF_DIAG) || BF_DIAG != 1   // (BF_DIAG: timing-only diagnostic builds with wrong results, never the product -- DESIGN.md section 4.5)
        __syncthreads();
#endif
    }
    min_regs(1, 0, 16);   // the last block's pair B
    block_compare(1, n_steps * 2 * kSub - 1);
    if constexpr (HAM) ham_fold((n_steps - 1) / (kHamChunk / kStepRows));
    // ---- the two halves of the wave hold disjoint row sets of the same query
    gms_dmatch* __restrict__ out = matches + pr.match_off;
    if constexpr (HAM) {
        int rowi[kNQ];
#pragma unroll
        for (int c = 0; c < kNQ; ++c) {   // lower distance, then lower row
            rowi[c] = bestt[c];
            const float od = __shfl_xor(bestd[c], 32);
            const int orow = __shfl_xor(rowi[c], 32);
            const bool take = od < bestd[c] || (od == bestd[c] && orow < rowi[c]);
            bestd[c] = take ? od : bestd[c];
            rowi[c] = take ? orow : rowi[c];
        }
#pragma unroll
        for (int pp = 0; pp < kNQ / 2; ++pp) {   // half 0 writes column set 2 pp, half 1 set 2 pp + 1: two records per lane
            const int q = half ? qrow[2 * pp + 1] : qrow[2 * pp];
            if (q < f.m) {
                gms_dmatch m;
                m.queryIdx = q;
                m.trainIdx = half ? rowi[2 * pp + 1] : rowi[2 * pp];
                m.imgIdx = 0;
                m.distance = (half ? bestd[2 * pp + 1] : bestd[2 * pp]) + __uint_as_float(norms[f.offA + q]);
                *reinterpret_cast<uint4*>(&out[q]) = *reinterpret_cast<const uint4*>(&m);
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < kNQ; ++c) {   // lower value, then earlier block
            const Val ov = __builtin_bit_cast(Val, __shfl_xor(__builtin_bit_cast(int, bestv[c]), 32));
            const int ot = __shfl_xor(bestt[c], 32);
            const int oties = __shfl_xor(ties[c], 32), ot2 = __shfl_xor(bestt2[c], 32);
            if (ov < bestv[c]) {
                bestt[c] = ot;
                bestt2[c] = ot2;
                ties[c] = oties;
                whalf[c] = half ^ 1;
            } else if (ov == bestv[c]) {   // both halves reached the minimal P: in the same block, in two blocks, or (rare) in more
                const int lo = min(bestt[c], ot), hi = max(bestt[c], ot);
                ties[c] = (ties[c] | oties) != 0 ? 2 : (lo != hi ? 1 : 0);
                whalf[c] = lo == hi ? 2 : (bestt[c] == lo ? half : half ^ 1);
                bestt[c] = lo;
                bestt2[c] = hi;
            }
        }
        // every query's winning block once more in exact integers: the first minimal row; eight queries at a time
        const int n_blocks = (f.nB + 31) / 32;
        const int piece = lane & 7, grp = lane >> 3;
        const uint4 ones = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
        auto pick = [&](const int (&v)[kNQ], int c) { return c == 0 ? v[0] : c == 1 ? v[1] : c == 2 ? v[2] : v[3]; };   // (c is wave-uniform)
        for (int i0 = 0; i0 < 32 * kNQ; i0 += 8) {
            if (q0 + wave * (32 * kNQ) + i0 >= f.m) break;                                // wave-uniform
            const int i = i0 + grp, q = q0 + wave * (32 * kNQ) + i;                       // this group's query: column i of the wave
            const int cs = i0 >> 5;                                                       // its column set
            int blk = __shfl(pick(bestt, cs), i & 31);
            const int nt = __shfl(pick(ties, cs), i & 31);
            const int blk2 = __shfl(pick(bestt2, cs), i & 31);
            const int hsel = __shfl(pick(whalf, cs), i & 31);
            const uint4 x = rows[(size_t)(f.offA + min(q, f.m - 1)) * 8 + piece];
            const uint4 nb = make_uint4(~x.x, ~x.y, ~x.z, ~x.w);
            const int query_part = dot16(nb, nb) + 2 * dot16(nb, ones);                   // this lane's 16 elements of the bracket
            uint32_t key;
            if (__ballot(hsel == 2) == 0ull) key = l2_block_key<16>(rows, norms, nb, query_part, f.offB, f.nB, blk, piece, hsel);
            else key = l2_block_key<32>(rows, norms, nb, query_part, f.offB, f.nB, blk, piece, 0);
            if (__ballot(nt == 1) != 0ull) {   // some group's minimal P was seen in a second block (a group without one repeats its own)
                const int b = nt == 1 ? blk2 : blk;
                const uint32_t k2 = l2_block_key<32>(rows, norms, nb, query_part, f.offB, f.nB, b, piece, 0);
                if ((k2 >> 5) < (key >> 5)) {   // a later block only wins with a strictly smaller distance
                    key = k2;
                    blk = b;
                }
            }
            if (__ballot(nt >= 2) != 0ull) {   // ... in more than two (rows repeated all over the frame): every block from the first on
                const int first = blk;   // (untouched so far: the step above leaves groups with nt >= 2 alone)
                for (int bb = 0; bb < n_blocks; ++bb) {
                    if (__ballot(nt >= 2 && bb > first) == 0ull) continue;
                    const int b = (nt >= 2 && bb > first) ? bb : blk;
                    const uint32_t k2 = l2_block_key<32>(rows, norms, nb, query_part, f.offB, f.nB, b, piece, 0);
                    if ((k2 >> 5) < (key >> 5)) {
                        key = k2;
                        blk = b;
                    }
                }
            }
            if (piece == 7 && q < f.m) {
                gms_dmatch m;
                m.queryIdx = q;
                m.trainIdx = blk * 32 + (int)(key & 31u);
                m.imgIdx = 0;
                m.distance = sqrtf((float)(key >> 5));
                *reinterpret_cast<uint4*>(&out[q]) = *reinterpret_cast<const uint4*>(&m);
            }
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
// prepared block:  L2       [total][128] int8 (a - 128) | [total] int32 w = sum (a' + 1)^2 | [n_frames] u32 "not SIFT-like" flags
//                  Hamming  [total][256] FP4 (128 B)    | [total] float popcounts
size_t bf_prepared_bytes(int kind, int64_t total, int n_frames)
{
    if (total < 0 || n_frames < 0) return 0;
    if (kind == GMS_DESC_L2_F32X128) return ((size_t)total * 132 + (size_t)n_frames * 4 + 15) & ~(size_t)15;
    if (kind == GMS_DESC_HAMMING256) return ((size_t)total * 132 + 15) & ~(size_t)15;
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
                           reinterpret_cast<uint2*>(base), reinterpret_cast<float*>(base + (size_t)total * 128));
        return hipGetLastError();
    }
    uint32_t* bad = reinterpret_cast<uint32_t*>(base + (size_t)total * 132);
    hipError_t e = hipMemsetAsync(bad, 0, (size_t)n_frames * 4, stream);
    if (e != hipSuccess) return e;
    const int64_t blocks = (total * 32 + 255) / 256;
    hipLaunchKernelGGL(bf_l2_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const float4*>(d_desc),
                       d_frame_off, n_frames, total, reinterpret_cast<uint32_t*>(base), reinterpret_cast<int32_t*>(base + (size_t)total * 128), bad);
    return hipGetLastError();
}

hipError_t launch_bf_match(int kind, const void* d_desc, const void* d_prep, int64_t total, const int64_t* d_frame_off, int n_frames,
                           const gms_pair* d_pairs, int n_pairs, int max_query, gms_dmatch* d_matches, hipStream_t stream)
{
    if (n_pairs <= 0 || max_query <= 0) return hipSuccess;
    const char* base = reinterpret_cast<const char*>(d_prep);
    const uint32_t* norms = reinterpret_cast<const uint32_t*>(base + (size_t)total * 128);
    const int tiles = (max_query + kQueriesPerBlock - 1) / kQueriesPerBlock;
    const uint32_t n = (uint32_t)tiles * (uint32_t)n_pairs;
    if (kind == GMS_DESC_HAMMING256 && d_prep != nullptr) {
        hipLaunchKernelGGL(bf_mfma_kernel<true>, dim3(n), dim3(256), 0, stream, reinterpret_cast<const uint4*>(base), norms,
                           (const uint32_t*)nullptr, d_frame_off, n_frames, d_pairs, tiles, n, d_matches);
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
        const uint32_t* bad = reinterpret_cast<const uint32_t*>(base + (size_t)total * 132);
        hipLaunchKernelGGL(bf_mfma_kernel<false>, dim3(n), dim3(256), 0, stream, reinterpret_cast<const uint4*>(base), norms, bad,
                           d_frame_off, n_frames, d_pairs, tiles, n, d_matches);
        // pairs with a frame that is not SIFT-like (every block of the other pairs returns at once); one query per lane, 256 per workgroup
        const int tiles_loop = (max_query + 255) / 256;
        hipLaunchKernelGGL(bf_l2_loop_kernel<kL2Dim>, dim3((uint32_t)tiles_loop * (uint32_t)n_pairs), dim3(256), 0, stream, reinterpret_cast<const float*>(d_desc),
                           bad, d_frame_off, n_frames, d_pairs, tiles_loop, d_matches);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

}  // namespace gms
