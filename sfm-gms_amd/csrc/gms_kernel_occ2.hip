// gms_kernel_occ2.hip -- the GMS filter built for TWO workgroups per CU (32 waves instead of 16).
//
// filter_kernel in gms_kernels.hip is bound by latency, not by any unit: with its 147 KB of LDS per pair only one
// 16-wave workgroup fits a CU, and the SQ counters show each wave waiting three quarters of its life. This variant
// trades table slack and registers for residency: <= 80 KB of LDS and <= 64 VGPRs per thread, so that two pairs are
// in flight per CU and one pair's waits are the other's issue slots. Same algorithm and bit-exact results; the
// differences from filter_kernel are:
//   * table regions of ~1.25 slots per match and no region header: the arg-max of each left cell's row is a scan
//     of its region after the insert (one more barrier per grid type);
//   * the per-grid-type cell tables (nLeft, region descriptor, half-cell view) are rebuilt from the half-cell
//     histogram for every grid type instead of being kept for all four;
//   * the code word carries the half-cell column and row separately, and mark derives the match's left cell from
//     them arithmetically, so the verified cell results sit in a small 21 x 21 table;
//   * two to five matches in flight per thread (register budget), frame B staged in LDS only when it fits.
// It serves m <= 10 240; the host picks it over filter_kernel when GMS_OCC2=1 (see gms_capi.cpp).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>

#include "gms_kernels.h"

#ifndef GMS_OCC2_CHUNK
#define GMS_OCC2_CHUNK 5  // matches in flight per thread through the LDS stages (must divide 5 and 10)
#endif

namespace gms {
namespace {

__constant__ int8_t c_rot2[8][9] = {  // mRotationPatterns - 1 (DLL .rdata 0x18012f520)
    {0, 1, 2, 3, 4, 5, 6, 7, 8}, {3, 0, 1, 6, 4, 2, 7, 8, 5}, {6, 3, 0, 7, 4, 1, 8, 5, 2},
    {7, 6, 3, 8, 4, 0, 5, 2, 1}, {8, 7, 6, 5, 4, 3, 2, 1, 0}, {5, 8, 7, 2, 4, 6, 1, 0, 3},
    {2, 5, 8, 1, 4, 7, 0, 3, 6}, {1, 2, 5, 0, 4, 8, 3, 6, 7}};

constexpr int NT = 1024;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;
// code word: [r:11 | hx:6 | hy:6 | 1 spare | acc:8]; hx = 63 marks a match that is never binned
constexpr uint32_t kRMask = 0x7FFu;
constexpr int kHxShift = 11, kHyShift = 17, kAccShift = 24;
constexpr uint32_t kHInvalid = 63u;
constexpr int kFineStride = 1664;
constexpr uint32_t kFineInvalid = kFineN;          // fdesc[1600..1663] = 0
constexpr int kResW = 21;                        // cellres indexed x + 21 * y, x, y in 0..20 (20 = outside)
constexpr uint32_t kNoMatch = 0xFFFFFF00u;
constexpr int kSlotRShift = 21;
constexpr uint32_t kSlotCountMask = (1u << kSlotRShift) - 1u;

// data buckets (4 slots each) of a left cell with n matches: ~1.25 slots per match, >= n + 1 slots
__device__ __forceinline__ uint32_t region_buckets(uint32_t n) { return n ? min((n + (n >> 2) + 3u) >> 2, 2048u) : 0u; }
__device__ __forceinline__ uint32_t bucket_of(uint32_t r, uint32_t nb) { return __umul24(__umul24(r, 2531u) & 0xFFFu, nb) >> 12; }
__device__ __forceinline__ int bucket_find(const uint4& v, uint32_t kr)
{
    int o = -1;
    o = ((v.w ^ kr) <= kSlotCountMask) ? 12 : o;
    o = ((v.z ^ kr) <= kSlotCountMask) ? 8 : o;
    o = ((v.y ^ kr) <= kSlotCountMask) ? 4 : o;
    o = ((v.x ^ kr) <= kSlotCountMask) ? 0 : o;
    return o;
}
__device__ __forceinline__ int bucket_first_empty(const uint4& v)
{
    int o = -1;
    o = (v.w == kEmpty) ? 12 : o;
    o = (v.z == kEmpty) ? 8 : o;
    o = (v.y == kEmpty) ? 4 : o;
    o = (v.x == kEmpty) ? 0 : o;
    return o;
}
__device__ __forceinline__ uint32_t bucket_count(const uint4& v, uint32_t kr)
{
    uint32_t c = 0;
    c = ((v.w ^ kr) <= kSlotCountMask) ? v.w : c;
    c = ((v.z ^ kr) <= kSlotCountMask) ? v.z : c;
    c = ((v.y ^ kr) <= kSlotCountMask) ? v.y : c;
    c = ((v.x ^ kr) <= kSlotCountMask) ? v.x : c;
    return c & kSlotCountMask;
}
__device__ __forceinline__ uint32_t* lds_at(uint32_t* base, uint32_t byte_off)
{
    return reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(base) + byte_off);
}

// motion[l][r]++, general walk over the region d = (first bucket << 16) | buckets
__device__ __forceinline__ void insert_general(uint32_t* tab, uint32_t d, uint32_t r)
{
    const uint32_t nb = d & 0xFFFFu, first = d >> 16;
    if (nb == 0) return;
    const uint32_t kr = r << kSlotRShift;
    uint32_t b = bucket_of(r, nb);
    for (uint32_t guard = 0; guard < 8u * nb + 8u; ++guard) {
        const uint32_t boff = (first + b) << 4;
        const uint4 v = *reinterpret_cast<const uint4*>(lds_at(tab, boff));
        const int f = bucket_find(v, kr);
        if (f >= 0) {
            atomicAdd(lds_at(tab, boff + (uint32_t)f), 1u);
            return;
        }
        const int e = bucket_first_empty(v);
        if (e >= 0) {
            const uint32_t prev = atomicCAS(lds_at(tab, boff + (uint32_t)e), kEmpty, kr | 1u);
            if (prev == kEmpty) return;
            continue;  // the slot went to somebody else (maybe to this very right cell): look again
        }
        if (++b == nb) b = 0;
    }
}

// motion[l][r], general walk
__device__ __forceinline__ uint32_t lookup_general(const uint32_t* tab, uint32_t d, uint32_t r)
{
    const uint32_t nb = d & 0xFFFFu, first = d >> 16;
    if (nb == 0) return 0;
    const uint32_t kr = r << kSlotRShift;
    uint32_t b = bucket_of(r, nb);
    for (uint32_t guard = 0; guard < nb; ++guard) {
        const uint4 v = *reinterpret_cast<const uint4*>(tab + ((first + b) << 2));
        if (bucket_find(v, kr) >= 0) return bucket_count(v, kr);
        if (bucket_first_empty(v) >= 0) return 0;
        if (++b == nb) b = 0;
    }
    return 0;
}

__device__ __forceinline__ bool threshold_rejects(uint32_t T, uint32_t n, uint32_t score, double factor, bool fast_ok)
{
    const double dT = (double)T, dN = (double)n, dS = (double)score;
    if (fast_ok) {  // see gms_kernels.hip: the squared form decides unless it is a near-tie
        const double a = dT * factor * factor, b = dS * dS * dN;
        if (fabs(a - b) > fmax(a, b) * 0x1p-40) return a > b;
    }
    return sqrt(dT / dN) * factor > dS;  // divsd, sqrtsd, mulsd, comisd (DLL@0x180049171)
}

__device__ __forceinline__ uint32_t dpp_xor1(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);
}

}  // namespace

template <int KPT, bool ROT, int CH>
__global__ void __launch_bounds__(NT, 8)
filter_kernel_occ2(FilterParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int kMcap = KPT * NT;
    constexpr int kNRot = ROT ? 8 : 1;
    static_assert(KPT % CH == 0, "KPT must be a multiple of the chunk");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    const gms_pair pr = p.pairs[blockIdx.x];
    const int m = pr.m;
    const gms_dmatch* __restrict__ matches = p.matches + pr.match_off;

    const uint32_t T = p.table_slots;              // multiple of 4
    uint32_t* tab = smem;                          // per-left-cell regions of [r | count] slots (no headers)
    uint32_t* nfine = tab + T;                     // [1664] 40 x 40 half-cell histogram of the left points
    uint32_t* fdesc = nfine + kFineStride;         // [1664] region of the cell a half-cell falls in, this grid type
    uint32_t* nleft = fdesc + kFineStride;         // [400]  mNumberPointsInPerCellLeft, this grid type
    uint32_t* desc = nleft + kLeftN;               // [400]  (first bucket << 16) | buckets, this grid type
    uint32_t* best = desc + kLeftN;                // [400]  (max count << 11) | (2047 - lowest right cell)
    uint32_t* cellres = best + kLeftN;             // [441]  (j* << 8) | rotation bits that pass; x + 21 * y
    uint32_t* bestmask = cellres + 448;            // kMcap / 32
    uint32_t* chunk_base = bestmask + (kMcap >> 5);// kMcap / 64 + 1
    uint32_t* misc = chunk_base + (kMcap >> 6) + 1;// [0..7] rotation counts, [8] error, [9] carry, [12] bucket allocator,
                                                   // [16..31] scan scratch
    uint32_t* trash = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(misc + 48) + 15) & ~uintptr_t(15));
                                                   // [0..63] atomic sink per lane, [64..67] an always-empty bucket

    if (p.stagger_cycles > 0 && blockIdx.x < (unsigned)p.stagger_blocks) {  // see filter_kernel
        const unsigned slot = (blockIdx.x * 37u) & 63u;
        const long long until = (long long)__builtin_readcyclecounter() + (long long)slot * (p.stagger_cycles >> 6);
        while ((long long)__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(32);
    }
    if (tid < 48) misc[tid] = 0;
    if (tid < 64) trash[tid] = 0;
    if (tid >= 64 && tid < 68) trash[tid] = kEmpty;
    for (int i = tid; i < (kMcap >> 5); i += NT) bestmask[i] = 0;
    for (int i = tid; i < kFineStride; i += NT) nfine[i] = 0;
    if (tid < 448) cellres[tid] = kNoMatch;

    const bool bad_pair = m < 0 || m > kMcap || pr.frame_a < 0 || pr.frame_a >= p.n_frames ||
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
    const bool thr_fast = p.threshold_factor > 1e-100 && p.threshold_factor < 1e100;
    uint32_t best_count = 0;
    int best_scale = -1, best_rot = -1;
    if (mm == 0 || nA <= 0 || nB <= 0) {  // workgroup-uniform
        if (tid == 0) {
            gms_pair_result r;
            r.n_inliers = 0;
            r.best_scale = -1;
            r.best_rot = -1;
            r.status = (bad_pair || m > 0) ? GMS_ERR_DOMAIN : GMS_OK;
            p.results[blockIdx.x] = r;
        }
        return;
    }
    // frame B into the (idle) table area when it fits: the train-side gather then comes from LDS
    const bool stage_b = (uint32_t)nB * 2u <= T && nB <= 4 * mm;
    if (stage_b) {
        float2* lds_b = reinterpret_cast<float2*>(tab);
        for (int j = tid; j < nB; j += NT) lds_b[j] = ptsB[j];
    }
    __syncthreads();

    // ---- both sides of every match, scale 0 ------------------------------------------------------------------
    uint32_t code[KPT];
    {
        const int wr = p.right_w[0];
        const uint32_t nr = (uint32_t)(wr * p.right_h[0]);
        const float fwr = (float)wr, fhr = (float)p.right_h[0];
        const float2* lds_b = reinterpret_cast<const float2*>(tab);
        bool any_bad = false;
#pragma unroll
        for (int k0 = 0; k0 < KPT; k0 += CH) {
            int2 qt[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) qt[c] = *reinterpret_cast<const int2*>(&matches[min((k0 + c) * NT + tid, mm - 1)]);
            float2 a[CH], b[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = ptsA[min((uint32_t)qt[c].x, (uint32_t)(nA - 1))];
            if (stage_b) {
#pragma unroll
                for (int c = 0; c < CH; ++c) b[c] = lds_b[min((uint32_t)qt[c].y, (uint32_t)(nB - 1))];
            } else {
#pragma unroll
                for (int c = 0; c < CH; ++c) b[c] = ptsB[min((uint32_t)qt[c].y, (uint32_t)(nB - 1))];
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const bool live = (k0 + c) * NT + tid < mm;
                // parity domain: finite, non-negative, < 2^20 (one unsigned compare on the bit patterns)
                const uint32_t worst = max(max(__float_as_uint(a[c].x), __float_as_uint(a[c].y)),
                                           max(__float_as_uint(b[c].x), __float_as_uint(b[c].y)));
                const float fx = 20.0f * a[c].x, fy = 20.0f * a[c].y;   // mulss, rounded to fp32
                const uint32_t hx = (uint32_t)(int)(fx + fx), hy = (uint32_t)(int)(fy + fy);  // floor(2f), exact
                const uint32_t r = (uint32_t)((int)(fwr * b[c].x) + (int)(fhr * b[c].y) * wr);  // no bounds test in the reference
                const bool ok = (uint32_t)qt[c].x < (uint32_t)nA && (uint32_t)qt[c].y < (uint32_t)nB &&
                                worst < 0x49800000u && r < nr;
                const bool binned = live && ok && hx < 40u && hy < 40u;  // else x >= 20 or y >= 20 under every grid type
                if (binned) atomicAdd(&nfine[hy * kFineW + hx], 1u);
                any_bad |= live && !ok;
                code[k0 + c] = binned ? (r | (hx << kHxShift) | (hy << kHyShift)) : (kHInvalid << kHxShift);
            }
        }
        if (any_bad) misc[8] = 1;  // benign race
    }
    __syncthreads();  // nfine complete, frame B no longer needed

    for (int s = 0; s < n_scales; ++s) {
        const int wr = p.right_w[s], hr = p.right_h[s];
        if (s > 0) {
            const uint32_t nr = (uint32_t)(wr * hr);
            const float fwr = (float)wr, fhr = (float)hr;
            bool any_bad = false;
#pragma unroll
            for (int k0 = 0; k0 < KPT; k0 += CH) {
                int t[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) t[c] = matches[min((k0 + c) * NT + tid, mm - 1)].trainIdx;
                float2 b[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) b[c] = ptsB[min((uint32_t)t[c], (uint32_t)(nB - 1))];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const uint32_t hpart = code[k0 + c] & ((63u << kHxShift) | (63u << kHyShift));
                    const bool had = ((hpart >> kHxShift) & 63u) != kHInvalid;
                    const uint32_t r = (uint32_t)((int)(fwr * b[c].x) + (int)(fhr * b[c].y) * wr);
                    const bool ok = r < nr;
                    any_bad |= had && !ok;
                    code[k0 + c] = (had && ok) ? (hpart | r) : (kHInvalid << kHxShift);
                }
            }
            if (any_bad) misc[8] = 1;
        }

        for (int g = 0; g < 4; ++g) {
            const int gx = g & 1, gy = g >> 1;
            // ---- motion.setTo(0); nLeft, regions and their half-cell view for this grid type ---------------------
            {
                const uint4 e4 = make_uint4(kEmpty, kEmpty, kEmpty, kEmpty);
                uint4* tab4 = reinterpret_cast<uint4*>(tab);
                for (uint32_t i = tid; i < (T >> 2); i += NT) tab4[i] = e4;
            }
            if (tid < kLeftN) {
                const int x = tid % kLeftW, y = tid / kLeftW;
                const int hx0 = 2 * x - gx, hy0 = 2 * y - gy;
                uint32_t n = 0;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int hx = hx0 + dx, hy = hy0 + dy;
                        if (hx >= 0 && hy >= 0) n += nfine[hy * kFineW + hx];
                    }
                const uint32_t nb = region_buckets(n);
                uint32_t d = 0;
                if (nb) d = (atomicAdd(&misc[12], nb) << 16) | nb;
                nleft[tid] = n;
                desc[tid] = d;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int hx = hx0 + dx, hy = hy0 + dy;
                        if (hx >= 0 && hy >= 0) fdesc[hy * kFineW + hx] = d;
                    }
            } else if (tid < kLeftN + 40) {
                if (gx) fdesc[(tid - kLeftN) * kFineW + 39] = 0;       // half-cell column 39: x = 20 when shifted
            } else if (tid < kLeftN + 80) {
                if (gy) fdesc[39 * kFineW + (tid - kLeftN - 40)] = 0;  // half-cell row 39: y = 20 when shifted
            } else if (tid < kLeftN + 80 + 64) {
                fdesc[kFineN + (tid - kLeftN - 80)] = 0;               // the "never binned" entries
            }
            __syncthreads();

            // ---- assignMatchPairs: motion[l][r]++ ------------------------------------------------------------------
            {
                uint32_t pending = 0;
                const uint32_t trash_add = (uint32_t)((trash - tab) + lane) << 2;
                const uint32_t trash_bkt = (uint32_t)((trash - tab) + 64) << 2;
#pragma unroll
                for (int k0 = 0; k0 < KPT; k0 += CH) {
                    uint32_t d[CH];
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const uint32_t cw = code[k0 + c];
                        const uint32_t hx = (cw >> kHxShift) & 63u, hy = (cw >> kHyShift) & 63u;
                        d[c] = fdesc[hx == kHInvalid ? kFineInvalid : hy * kFineW + hx];
                    }
                    uint32_t slot[CH];
                    uint4 v[CH];
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const uint32_t nb = d[c] & 0xFFFFu;
                        const uint32_t bo = ((d[c] >> 16) + bucket_of(code[k0 + c] & kRMask, nb)) << 4;
                        slot[c] = nb ? bo : trash_bkt;
                        v[c] = *reinterpret_cast<const uint4*>(lds_at(tab, slot[c]));
                    }
                    // "+1" where the bucket already holds the right cell, CAS into its first empty slot where not
                    uint32_t o_cas[CH];
                    uint32_t canput = 0;
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const bool valid = (d[c] & 0xFFFFu) != 0;
                        const uint32_t kr = (code[k0 + c] & kRMask) << kSlotRShift;
                        const int f = bucket_find(v[c], kr);
                        const int e = bucket_first_empty(v[c]);
                        const bool fnd = valid && f >= 0;
                        const bool put = valid && f < 0 && e >= 0;
                        if (valid && f < 0 && e < 0) pending |= 1u << (k0 + c);  // full bucket
                        slot[c] += (uint32_t)(f >= 0 ? f : (e & 12));
                        canput |= (put ? 1u : 0u) << c;
                        atomicAdd(lds_at(tab, fnd ? slot[c] : trash_add), 1u);
                        o_cas[c] = atomicCAS(lds_at(tab, put ? slot[c] : trash_add), kEmpty, kr | 1u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // a lost CAS whose winner was the same right cell becomes "+1"; any other winner: leftovers
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const uint32_t kr = (code[k0 + c] & kRMask) << kSlotRShift;
                        const bool put = (canput >> c) & 1u;
                        const bool w = put && o_cas[c] == kEmpty;
                        const bool sm = put && !w && (o_cas[c] ^ kr) <= kSlotCountMask;
                        if (put && !w && !sm) pending |= 1u << (k0 + c);
                        atomicAdd(lds_at(tab, sm ? slot[c] : trash_add), 1u);
                    }
                }
                while (pending) {
                    const int k1 = __ffs(pending) - 1;
                    pending &= pending - 1u;
                    uint32_t cw = 0;
#pragma unroll
                    for (int k = 0; k < KPT; ++k) cw = (k == k1) ? code[k] : cw;
                    const uint32_t hx = (cw >> kHxShift) & 63u, hy = (cw >> kHyShift) & 63u;
                    insert_general(tab, fdesc[hy * kFineW + hx], cw & kRMask);
                }
            }
            __syncthreads();

            // ---- arg-max of every left cell's row: max count, lowest right cell on ties. Two lanes per cell ------------
            {
                const int i = tid >> 1, half = tid & 1;
                uint32_t bp = 0;
                if (i < kLeftN) {
                    const uint32_t dd = desc[i];
                    const uint32_t nb = dd & 0xFFFFu;
                    const uint4* reg4 = reinterpret_cast<const uint4*>(tab) + (dd >> 16);
                    for (uint32_t b = half; b < nb; b += 2) {
                        const uint4 q = reg4[b];
                        // an empty slot gets key 0: it never wins (real counts are >= 1)
                        const uint32_t k0 = (q.x == kEmpty) ? 0u : ((q.x & kSlotCountMask) << 11) | (2047u - (q.x >> kSlotRShift));
                        const uint32_t k1 = (q.y == kEmpty) ? 0u : ((q.y & kSlotCountMask) << 11) | (2047u - (q.y >> kSlotRShift));
                        const uint32_t k2 = (q.z == kEmpty) ? 0u : ((q.z & kSlotCountMask) << 11) | (2047u - (q.z >> kSlotRShift));
                        const uint32_t k3 = (q.w == kEmpty) ? 0u : ((q.w & kSlotCountMask) << 11) | (2047u - (q.w >> kSlotRShift));
                        bp = max(max(bp, k0), max(max(k1, k2), k3));
                    }
                }
                bp = max(bp, dpp_xor1(bp));
                if (i < kLeftN && half == 0) best[i] = bp;
                if (tid == 0) misc[12] = 0;  // bucket allocator of the next grid type
            }
            __syncthreads();

            // ---- verifyCellPairs ------------------------------------------------------------------------------------
            {
                constexpr int kItems = ROT ? kLeftN * 8 : kLeftN * 2;
                for (int item = tid; item < ((kItems + 63) & ~63); item += NT) {
                    const bool live = item < kItems;
                    const int i = live ? (ROT ? (item >> 3) : (item >> 1)) : 0;
                    const int rot = ROT ? (item & 7) : 0;
                    const int half = item & 1;  // !ROT only
                    const uint32_t ni = live ? nleft[i] : 0u;
                    const uint32_t bi = best[i];
                    const int j = 2047 - (int)(bi & kRMask);
                    const int jx = j % wr, jy = j / wr;
                    const int ix = i % kLeftW, iy = i / kLeftW;
                    uint32_t score = 0, tn = 0;  // tn = (sum of nLeft << 4) | numpair
#pragma unroll
                    for (int h = 0; h < (ROT ? 8 : 4); h += 4) {
                        uint32_t dn[4], rq[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            int k;
                            if (ROT) {
                                const int k8 = h + c;
                                k = k8 < 4 ? k8 : k8 + 1;
                            } else {
                                k = half ? c + 5 : c;
                            }
                            const int q = ROT ? c_rot2[rot][k] : k;
                            int ldx, ldy, rdx, rdy;
                            if (ROT) {
                                ldx = (k % 3) - 1; ldy = (k / 3) - 1;
                                rdx = (q % 3) - 1; rdy = (q / 3) - 1;
                            } else {
                                ldx = half ? ((c + 5) % 3) - 1 : (c % 3) - 1;
                                ldy = half ? ((c + 5) / 3) - 1 : (c / 3) - 1;
                                rdx = ldx; rdy = ldy;
                            }
                            const int lx = ix + ldx, ly = iy + ldy;
                            const int rx = jx + rdx, ry = jy + rdy;
                            const bool okl = ni != 0 && (uint32_t)lx < (uint32_t)kLeftW && (uint32_t)ly < (uint32_t)kLeftH;  // ll != -1
                            const int ll = okl ? lx + ly * kLeftW : 0;
                            const uint32_t nll = nleft[ll], dll = desc[ll];
                            const bool okp = okl && (uint32_t)rx < (uint32_t)wr && (uint32_t)ry < (uint32_t)hr;             // rr != -1
                            rq[c] = okp ? (uint32_t)(rx + ry * wr) : 0u;
                            tn += okp ? ((nll << 4) | 1u) : 0u;
                            dn[c] = okp ? dll : 0u;
                        }
                        uint4 v[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const uint32_t nb = dn[c] & 0xFFFFu;
                            v[c] = make_uint4(kEmpty, kEmpty, kEmpty, kEmpty);
                            if (nb) v[c] = *reinterpret_cast<const uint4*>(tab + (((dn[c] >> 16) + bucket_of(rq[c], nb)) << 2));
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const uint32_t kr = rq[c] << kSlotRShift;
                            if (bucket_find(v[c], kr) >= 0) {
                                score += bucket_count(v[c], kr);
                            } else if (bucket_first_empty(v[c]) < 0) {  // full bucket without the key: walk
                                score += lookup_general(tab, dn[c], rq[c]);
                            }
                        }
                    }
                    if (!ROT) {
                        score += dpp_xor1(score);
                        tn += dpp_xor1(tn);
                    }
                    score += bi >> 11;        // centre pair: ll = i, rr = j*, count = the arg-max count
                    tn += (ni << 4) | 1u;
                    uint32_t pass = 0;
                    if (ni != 0 && (ROT || half == 0))
                        pass = threshold_rejects(tn >> 4, tn & 15u, score, p.threshold_factor, thr_fast) ? 0u : 1u;
                    uint32_t bits = pass;
                    bool writer = live && half == 0;
                    if (ROT) {
                        const unsigned long long bal = __ballot(pass);
                        bits = (uint32_t)(bal >> (lane & 56)) & 0xFFu;
                        writer = live && (lane & 7) == 0;
                    }
                    if (writer) cellres[ix + kResW * iy] = ni ? (((uint32_t)j << 8) | bits) : kNoMatch;
                }
            }
            __syncthreads();

            // ---- mark inliers: cellPairs[l] == r, all rotations at once; the left cell comes from the half-cell indices
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const uint32_t cw = code[k];
                const uint32_t hx = (cw >> kHxShift) & 63u, hy = (cw >> kHyShift) & 63u;
                // x, y <= 20 for binned matches (20 = outside under this grid type: that entry never matches);
                // never-binned matches (hx = 63) are sent to entry (20, 20)
                const uint32_t x = hx == kHInvalid ? 20u : (hx + (uint32_t)gx) >> 1;
                const uint32_t y = hx == kHInvalid ? 20u : (hy + (uint32_t)gy) >> 1;
                const uint32_t cr = cellres[x + kResW * y];
                if ((cr >> 8) == (cw & kRMask)) code[k] = cw | (cr << kAccShift);
            }
            // (the next grid type rewrites fdesc / nleft / desc / best, none of which mark reads; cellres is next written
            //  three barriers from here)
        }

        // ---- run() return value for each rotation of this scale; keep on strict '>' -------------------------------
        {
            uint32_t cnt[kNRot];
#pragma unroll
            for (int r = 0; r < kNRot; ++r) cnt[r] = 0;
#pragma unroll
            for (int k = 0; k < KPT; ++k)
#pragma unroll
                for (int r = 0; r < kNRot; ++r)
                    cnt[r] += (uint32_t)__popcll(__ballot((code[k] >> (kAccShift + r)) & 1u));
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < kNRot; ++r)
                    if (cnt[r]) atomicAdd(&misc[r], cnt[r]);
            }
        }
        __syncthreads();
        int winner = -1;
#pragma unroll
        for (int r = 0; r < kNRot; ++r) {
            const uint32_t c = misc[r];
            if (c > best_count) {
                best_count = c;
                best_scale = s;
                best_rot = r + 1;
                winner = r;
            }
        }
        if (winner >= 0) {
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const unsigned long long b = __ballot((code[k] >> (kAccShift + winner)) & 1u);
                if (lane == 0) {
                    const int ch = k * (NT / 64) + wave;
                    bestmask[2 * ch] = (uint32_t)b;
                    bestmask[2 * ch + 1] = (uint32_t)(b >> 32);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KPT; ++k) code[k] &= (1u << kAccShift) - 1u;
        __syncthreads();
        if (tid < 8) misc[tid] = 0;
        // cellres of the last grid type must not leak into the next scale's first mark: it is rewritten by that
        // scale's first verify before any mark reads it (every entry of a non-empty cell; empty cells hold kNoMatch)
    }
    __syncthreads();

    // ---- copy-out: surviving DMatch verbatim, in input order (DLL@0x180048340) ------------------------------------
    const bool failed = misc[8] != 0;
    const int n_chunks = (mm + 63) >> 6;
    {
        uint32_t* wave_tot = misc + 16;
        for (int base = 0; base < n_chunks; base += NT) {
            const int c = base + tid;
            const uint32_t v = (c < n_chunks && !failed) ? __popc(bestmask[2 * c]) + __popc(bestmask[2 * c + 1]) : 0u;
            uint32_t incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d);
                if (lane >= d) incl += t;
            }
            if (lane == 63) wave_tot[wave] = incl;
            __syncthreads();
            uint32_t wave_off = misc[9];
            for (int w = 0; w < wave; ++w) wave_off += wave_tot[w];
            if (c < n_chunks) chunk_base[c] = wave_off + incl - v;
            __syncthreads();
            if (tid == NT - 1) misc[9] = wave_off + incl;
            __syncthreads();
        }
    }
    const uint32_t total = misc[9];
    gms_dmatch* __restrict__ out = p.out + pr.match_off;
    uint8_t* mask_out = p.mask ? p.mask + pr.match_off : nullptr;
#pragma unroll
    for (int k0 = 0; k0 < KPT; k0 += CH) {
        uint32_t pos[CH];
        uint4 v[CH];
        uint32_t inm = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = (k0 + c) * NT + tid;
            const int ch = i >> 6;
            pos[c] = 0;
            v[c] = make_uint4(0, 0, 0, 0);
            if (i < mm) {
                const unsigned long long bits =
                    failed ? 0ull : ((unsigned long long)bestmask[2 * ch] | ((unsigned long long)bestmask[2 * ch + 1] << 32));
                const bool in = (bits >> lane) & 1ull;
                if (mask_out) mask_out[i] = in ? 1 : 0;
                if (in) {
                    inm |= 1u << c;
                    pos[c] = chunk_base[ch] + (uint32_t)__popcll(bits & ((1ull << lane) - 1ull));
                    v[c] = *reinterpret_cast<const uint4*>(&matches[i]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if ((inm >> c) & 1u) *reinterpret_cast<uint4*>(&out[pos[c]]) = v[c];
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

// ------------------------------------------------------------------------------------------------------------
uint32_t occ2_table_slots(int kpt)
{
    // sum over cells of 4 * region_buckets(n) <= 1.25 * M + 3 * 400
    const uint32_t mcap = (uint32_t)kpt * NT;
    return (mcap + (mcap >> 2) + 3 * kLeftN + 3u) & ~3u;
}

size_t occ2_lds_bytes(int kpt)
{
    const size_t mcap = (size_t)kpt * NT;
    const size_t dwords = occ2_table_slots(kpt) + 2 * kFineStride + 3 * kLeftN + 448 + (mcap >> 5) + (mcap >> 6) + 1 + 48 + 4 + 68 + 12;
    return dwords * 4;
}

int occ2_pick_kpt(int max_m)
{
    static const int kKpt[] = {5, 10};
    for (int k : kKpt)
        if (max_m <= k * NT && occ2_lds_bytes(k) <= 80 * 1024) return k;
    return 0;
}

template <int KPT, bool ROT, int CH>
static hipError_t launch_occ2_t(const FilterParams& p, int n_pairs, size_t lds, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(filter_kernel_occ2<KPT, ROT, CH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
        if (std::getenv("GMS_OCC2_REPORT")) {
            int nblk = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, filter_kernel_occ2<KPT, ROT, CH>, NT, lds);
            std::fprintf(stderr, "occ2<%d,%d,%d>: lds %zu B, resident workgroups per CU (API) %d\n", KPT, (int)ROT, CH, lds, nblk);
        }
    }
    hipLaunchKernelGGL((filter_kernel_occ2<KPT, ROT, CH>), dim3((unsigned)n_pairs), dim3(NT), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_filter_occ2(const FilterParams& p, int kpt, int n_pairs, hipStream_t stream)
{
    if (n_pairs <= 0) return hipSuccess;
    const size_t lds = occ2_lds_bytes(kpt);
    const bool rot = p.with_rotation != 0;
    switch (kpt) {
    case 5: return rot ? launch_occ2_t<5, true, GMS_OCC2_CHUNK>(p, n_pairs, lds, stream) : launch_occ2_t<5, false, GMS_OCC2_CHUNK>(p, n_pairs, lds, stream);
    case 10: return rot ? launch_occ2_t<10, true, GMS_OCC2_CHUNK>(p, n_pairs, lds, stream) : launch_occ2_t<10, false, GMS_OCC2_CHUNK>(p, n_pairs, lds, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace gms
