// gms_kernel_band.hip -- large pairs (16 385 ... 262 144 matches) under the reference's default flags (no rotation,
// no scale hypotheses): BASELINE config 4 and the reference's dense one-keypoint-per-pixel disparity call
// (DisparityUtil.cpp:123-149, W*H matches).
//
// The byte-matrix idea of gms_kernels.hip (dense_pair) with 16-bit counters: the 400 x 400 motion matrix is
// 320 KB then, so the 20 left-grid rows are cut into three bands (7 + 7 + 6 rows). One workgroup owns one band of one
// pair and keeps the band's rows plus one halo row on either side in LDS -- [header dword | 400 x u16] per left cell,
// 145 KB for 9 rows -- so that every neighbour count verifyCellPairs needs for the band's own cells is local. A left
// cell (of whatever grid type) belongs to exactly one band, so the bands of a pair never exchange anything: each
// marks the matches its own cells accept in the pair's byte mask (all writers store 1), and the mask is the OR over
// the four grid types by construction. Three stream-ordered launches per batch:
//   band_codes_kernel    one thread per match: (queryIdx, trainIdx) + the two keypoint gathers -> a code word (right
//                        cell, half-cell coordinates of the left point) appended, with the match index, to the list of
//                        every band that keeps the match's left row (1.2 lists per match on average), and the pair's
//                        40 x 40 half-cell histogram (both LDS-privatised per block, then global atomics);
//   band_filter_kernel   grid (3 bands [x 4 grid types when the batch is small], pairs): per grid type clear, stream the band's list (coalesced, 8 B per
//                        entry) for assignMatchPairs, verify the band's cells, stream again to mark;
//   band_compact_kernel  grid (16k-match tiles, pairs): order-preserving compaction of the DMatch records by the mask.
// A pair with a left cell above 65 535 matches (a 16-bit entry could wrap) is flagged instead and left to the
// HBM-slab kernel of gms_kernel_big.hip, which runs afterwards on flagged pairs only. Bit-exactness rules are the ones
// of gms_kernels.hip (same float -> cell arithmetic, same threshold, same arg-max and tie rules).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gms_device_common.h"

namespace gms {
namespace {

constexpr int kRightW = 20, kRightN = 400;                 // right grid of scale 0
constexpr uint32_t kRowBytes = 4u + 2u * kRightN;          // header dword + 400 x u16
constexpr int kBandRowsMax = 9;                            // 7 own rows + 2 halo rows
constexpr uint32_t kMatrixBytes = kBandRowsMax * kLeftW * kRowBytes;  // 144 720
constexpr uint32_t kNleftOff = kMatrixBytes;               // [400] u32: nLeft of every cell under the current grid type
constexpr uint32_t kFineOff = kNleftOff + 4u * kLeftN;     // [1600] u32: the pair's half-cell histogram
constexpr uint32_t kMiscOff = kFineOff + 4u * kFineN;      // [16] u32
constexpr uint32_t kBandLdsBytes = kMiscOff + 64u;
static_assert(kBandLdsBytes <= kLdsBytes, "band layout exceeds the LDS");
static_assert(kMatrixBytes % 16 == 0, "the band matrix is cleared in uint4s");

// code word: E' = 400 - r (1..400; higher = lower right cell) : 9 | hx : 6 | hy : 6 | binned : 1
constexpr int kHxShift = 9, kHyShift = 15;
constexpr uint32_t kBinned = 1u << 21;

constexpr uint32_t kFlagDomain = 1u;    // an input outside the parity domain: the pair fails as a whole
constexpr uint32_t kFlagGeneral = 2u;   // a left cell above 65 535 matches: gms_kernel_big.hip takes the pair

__device__ __forceinline__ void band_rows(int band, int& lo, int& hi)  // own rows [lo, hi)
{
    lo = band * 7;
    hi = band == 2 ? kLeftH : lo + 7;
}

// Streams a band's list of (code word, match index) entries through the workgroup, 4 per thread per step, the next step's
// loads issued before the current step is worked on. body(code word, match index) for every entry.
template <typename F>
__device__ __forceinline__ void stream_list(const uint2* __restrict__ list, int len, int tid, F&& body)
{
    constexpr int kDepth = 4;
    uint2 cur[kDepth], nxt[kDepth];
#pragma unroll
    for (int k = 0; k < kDepth; ++k) {
        const int i = k * 1024 + tid;
        cur[k] = i < len ? list[i] : make_uint2(0u, 0u);
    }
    for (int i0 = 0; i0 < len; i0 += kDepth * 1024) {
#pragma unroll
        for (int k = 0; k < kDepth; ++k) {
            const int i = i0 + kDepth * 1024 + k * 1024 + tid;
            nxt[k] = i < len ? list[i] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < kDepth; ++k) body(cur[k].x, (int)cur[k].y);  // code word 0 (not binned) beyond the list
#pragma unroll
        for (int k = 0; k < kDepth; ++k) cur[k] = nxt[k];
    }
}

// rows a band keeps in LDS: its own [lo, hi) plus one halo row on either side
__device__ __forceinline__ bool band_holds(int band, uint32_t row)
{
    int lo, hi;
    band_rows(band, lo, hi);
    return (int)row >= lo - 1 && (int)row < hi + 1;
}

}  // namespace

// ---- code words + half-cell histogram -----------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
band_codes_kernel(FilterParams p, uint2* lists, uint32_t* list_len, uint32_t* nfine_g, uint32_t* flags, uint8_t* mask_ws, int mcap)
{
    __shared__ uint32_t hist[kFineN];
    __shared__ uint32_t cnt_l[3], base_g[3];
    const int pi = blockIdx.y, tid = threadIdx.x;
    const gms_pair pr = p.pairs[pi];
    const int m = pr.m;
    const bool bad_pair = m < 0 || m > mcap || pr.frame_a < 0 || pr.frame_a >= p.n_frames || pr.frame_b < 0 ||
                          pr.frame_b >= p.n_frames;
    if (bad_pair) {
        if (blockIdx.x == 0 && tid == 0) atomicOr(&flags[pi], kFlagDomain);
        return;
    }
    const int base = blockIdx.x * 4096;
    if (base >= m) return;  // workgroup-uniform
    const int64_t offA = p.frame_off[pr.frame_a], offB = p.frame_off[pr.frame_b];
    const int nA = (int)(p.frame_off[pr.frame_a + 1] - offA), nB = (int)(p.frame_off[pr.frame_b + 1] - offB);
    if (nA <= 0 || nB <= 0) {  // matches, but nothing valid to index
        if (blockIdx.x == 0 && tid == 0) atomicOr(&flags[pi], kFlagDomain);
        return;
    }
    const float2* __restrict__ ptsA = p.pts + offA;
    const float2* __restrict__ ptsB = p.pts + offB;
    const gms_dmatch* __restrict__ matches = p.matches + pr.match_off;
    uint8_t* mask = p.mask ? p.mask + pr.match_off : mask_ws + (size_t)pi * mcap;

    for (int j = tid; j < kFineN; j += 1024) hist[j] = 0;
    if (tid < 3) cnt_l[tid] = 0;
    __syncthreads();
    uint2 qt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) qt[k] = *reinterpret_cast<const uint2*>(&matches[min(base + k * 1024 + tid, m - 1)]);
    float2 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = ptsA[min(qt[k].x, (uint32_t)(nA - 1))];
        b[k] = ptsB[min(qt[k].y, (uint32_t)(nB - 1))];
    }
    bool any_bad = false;
    uint32_t cw[4], rank[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = base + k * 1024 + tid;
        const bool live = i < m;
        // parity domain: coordinates finite, non-negative, < 2^20 (one unsigned compare on the bit patterns)
        const uint32_t worst = max(max(__float_as_uint(a[k].x), __float_as_uint(a[k].y)),
                                   max(__float_as_uint(b[k].x), __float_as_uint(b[k].y)));
        const float fx = 20.0f * a[k].x, fy = 20.0f * a[k].y;   // mulss, rounded to fp32
        const uint32_t hx = (uint32_t)(int)(fx + fx), hy = (uint32_t)(int)(fy + fy);  // floor(2f), 2f exact
        // getGridIndexRight: x + y * 20 with no bounds test; clamped operands keep the 24-bit form exact below 400
        const uint32_t rx = (uint32_t)(int)(20.0f * b[k].x), ry = (uint32_t)(int)(20.0f * b[k].y);
        const uint32_t r = __umul24(min(ry, 4096u), (uint32_t)kRightW) + min(rx, 4096u);
        const bool ok = qt[k].x < (uint32_t)nA && qt[k].y < (uint32_t)nB && worst < 0x49800000u && r < (uint32_t)kRightN;
        const bool binned = live && ok && hx < 40u && hy < 40u;
        if (binned) atomicAdd(&hist[hy * kFineW + hx], 1u);
        any_bad |= live && !ok;
        if (live) mask[i] = 0;
        cw[k] = binned ? (((uint32_t)kRightN - r) | (hx << kHxShift) | (hy << kHyShift) | kBinned) : 0u;
        // the bands that keep this match's left row under the unshifted (hy >> 1) or the y-shifted ((hy + 1) >> 1) grid types
#pragma unroll
        for (int bnd = 0; bnd < 3; ++bnd) {
            const bool need = binned && (band_holds(bnd, hy >> 1) || (hy < 39u && band_holds(bnd, (hy + 1u) >> 1)));
            rank[k][bnd] = need ? atomicAdd(&cnt_l[bnd], 1u) : 0xFFFFFFFFu;
        }
    }
    if (any_bad) atomicOr(&flags[pi], kFlagDomain);
    __syncthreads();
    uint32_t* nf = nfine_g + (size_t)pi * kFineN;
    for (int j = tid; j < kFineN; j += 1024)
        if (hist[j]) atomicAdd(&nf[j], hist[j]);
    // this block's entries go to the end of the pair's three band lists (list order does not matter: every consumer is
    // either a commutative atomic or a store of 1 to the match's own mask byte)
    if (tid < 3) base_g[tid] = cnt_l[tid] ? atomicAdd(&list_len[pi * 3 + tid], cnt_l[tid]) : 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int bnd = 0; bnd < 3; ++bnd)
            if (rank[k][bnd] != 0xFFFFFFFFu)
                lists[((size_t)pi * 3 + bnd) * mcap + base_g[bnd] + rank[k][bnd]] = make_uint2(cw[k], (uint32_t)(base + k * 1024 + tid));
}

// ---- one band of one pair -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
band_filter_kernel(FilterParams p, const uint2* lists, const uint32_t* list_len, const uint32_t* nfine_g, uint32_t* flags,
                   uint8_t* mask_ws, int mcap)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int band = blockIdx.x % 3, pi = blockIdx.y, tid = threadIdx.x;
    const int g_only = (gridDim.x == 12) ? (int)(blockIdx.x / 3) : -1;  // one (band, grid type) per workgroup, or all four types
    const gms_pair pr = p.pairs[pi];
    const int m = pr.m;
    if (m <= 0 || m > mcap) return;
    if (flags[pi] & kFlagDomain) return;  // written by the previous launch
    const uint2* __restrict__ list = lists + ((size_t)pi * 3 + band) * mcap;
    const int len = (int)list_len[pi * 3 + band];
    const uint32_t* __restrict__ nf = nfine_g + (size_t)pi * kFineN;
    uint8_t* mask = p.mask ? p.mask + pr.match_off : mask_ws + (size_t)pi * mcap;

    const uint8_t* bytes = reinterpret_cast<const uint8_t*>(smem);
    uint32_t* nleft = smem + kNleftOff / 4;
    uint32_t* misc = smem + kMiscOff / 4;
    int lo, hi;
    band_rows(band, lo, hi);
    const int row0 = max(lo - 1, 0), row1 = min(hi + 1, kLeftH);  // rows held in LDS: [row0, row1)
    const uint32_t clear16 = (uint32_t)((row1 - row0) * kLeftW) * kRowBytes / 16u;

    uint32_t* nfine = smem + kFineOff / 4;
    for (int j = tid; j < kFineN; j += 1024) nfine[j] = nf[j];  // one coalesced read instead of four scattered ones per cell
    auto nleft_of = [&](int cell, int gx, int gy) -> uint32_t {
        const int hx0 = 2 * (cell % kLeftW) - gx, hy0 = 2 * (cell / kLeftW) - gy;  // hx0 + 1, hy0 + 1 <= 39
        uint32_t n = 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int hx = hx0 + dx, hy = hy0 + dy;
                if (hx >= 0 && hy >= 0) n += nfine[hy * kFineW + hx];
            }
        return n;
    };

    // a 16-bit entry never exceeds its left cell's population: every cell of every grid type has to stay below 2^16
    if (tid == 0) misc[0] = 0;
    __syncthreads();
    for (int item = tid; item < 4 * kLeftN; item += 1024) {
        const int g = item / kLeftN;
        if (nleft_of(item - g * kLeftN, g & 1, g >> 1) > 65535u) misc[0] = 1;
    }
    __syncthreads();
    if (misc[0]) {  // the same decision in all three bands of the pair
        if (blockIdx.x == 0 && tid == 0) atomicOr(&flags[pi], kFlagGeneral);
        return;
    }

    // The band's list is walked eight times (binning and marking under each grid type) and a batch's lists together do not
    // stay in L2: the first kResident entries per thread are read once and kept in registers, only what is beyond them
    // is streamed again every time.
    constexpr int kResident = 24;
    uint2 res[kResident];
#pragma unroll
    for (int k = 0; k < kResident; ++k) {
        const int i = k * 1024 + tid;
        res[k] = i < len ? list[i] : make_uint2(0u, 0u);
    }

    for (int g = (g_only < 0 ? 0 : g_only); g < (g_only < 0 ? 4 : g_only + 1); ++g) {
        const int gx = g & 1, gy = g >> 1;
        // opaque per grid type: otherwise every resident entry's field extraction is hoisted out of this loop and the
        // extracted fields (several registers per entry) spill
#pragma unroll
        for (int k = 0; k < kResident; ++k) asm volatile("" : "+v"(res[k].x));
        if (tid < kLeftN) nleft[tid] = nleft_of(tid, gx, gy);
        {
            const uint4 z4 = make_uint4(0, 0, 0, 0);
            uint4* d4 = reinterpret_cast<uint4*>(smem);
            for (uint32_t i = tid; i < clear16; i += 1024) d4[i] = z4;  // motion.setTo(0), headers included
        }
        __syncthreads();

        // ---- assignMatchPairs for the rows this band holds (own + halo): +1 on the 16-bit entry, the count it produced
        //      into the row's running arg-max ((count - 1) << 9 | E', atomicMax: highest count, then lowest right cell)
        auto bin_one = [&](uint32_t cw, int) {
            const uint32_t lx = (((cw >> kHxShift) & 63u) + (uint32_t)gx) >> 1;
            const uint32_t ly = (((cw >> kHyShift) & 63u) + (uint32_t)gy) >> 1;
            // x >= 20 || y >= 20 -> -1 (DLL@0x180047d3d); rows outside [row0, row1) belong to another band
            if ((cw & kBinned) && lx < (uint32_t)kLeftW && ly >= (uint32_t)row0 && ly < (uint32_t)row1) {
                const uint32_t e = cw & 0x1FFu;  // E' = 400 - r
                const uint32_t row = (__umul24(ly - (uint32_t)row0, (uint32_t)kLeftW) + lx) * kRowBytes;
                const uint32_t at = row + 4u + 2u * (e - 1u);
                const uint32_t sh = (at & 2u) << 3;
                const uint32_t old = atomicAdd(lds_at(smem, at & ~3u), 1u << sh);
                atomicMax(lds_at(smem, row), (((old >> sh) & 0xFFFFu) << 9) | e);
            }
        };
#pragma unroll
        for (int k = 0; k < kResident; ++k) bin_one(res[k].x, 0);  // code word 0 (not binned) beyond the list
        if (len > kResident * 1024) stream_list(list + kResident * 1024, len - kResident * 1024, tid, bin_one);
        __syncthreads();

        // ---- verifyCellPairs for the band's own cells: two lanes per cell, four neighbours each, joined by one DPP exchange
        {
            const int n_items = (hi - lo) * kLeftW * 2;
            for (int item = tid; item < ((n_items + 63) & ~63); item += 1024) {
                const bool live = item < n_items;
                const int c = live ? item >> 1 : 0, half = item & 1;
                const int ix = c % kLeftW, iy = lo + c / kLeftW;
                const int i = iy * kLeftW + ix;
                const uint32_t hdr_off = (uint32_t)((iy - row0) * kLeftW + ix) * kRowBytes;
                const uint32_t ni = live ? nleft[i] : 0u;
                const uint32_t best = smem[hdr_off >> 2];  // ((max count - 1) << 9) | E'(j*), lowest j* among maxima
                const uint32_t ej = ni ? (best & 0x1FFu) : (uint32_t)kRightN;
                const int j = kRightN - (int)ej;
                const int jx = j % kRightW, jy = j / kRightW;
                uint32_t score = 0, T = 0, np = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = half ? q + 5 : q;  // lane 0: neighbours 0..3, lane 1: neighbours 5..8 (4 is the centre)
                    const int dx = (k % 3) - 1, dy = (k / 3) - 1;
                    const int lx = ix + dx, ly = iy + dy, rx = jx + dx, ry = jy + dy;
                    const bool okl = ni != 0 && (uint32_t)lx < (uint32_t)kLeftW && (uint32_t)ly < (uint32_t)kLeftH;   // ll != -1
                    const bool okp = okl && (uint32_t)rx < (uint32_t)kRightW && (uint32_t)ry < (uint32_t)kRightW;    // rr != -1
                    if (okp) {  // ly is within one row of an own row, so the band holds it
                        // entries are stored by E' - 1 = 399 - r
                        const uint32_t at = (uint32_t)((ly - row0) * kLeftW + lx) * kRowBytes + 4u +
                                            2u * (uint32_t)(kRightN - 1 - (rx + ry * kRightW));
                        score += *reinterpret_cast<const uint16_t*>(bytes + at);
                        T += nleft[ly * kLeftW + lx];
                        np += 1;
                    }
                }
                score += dpp_xor1(score);
                T += dpp_xor1(T);
                np += dpp_xor1(np);
                score += (best >> 9) + 1u;  // centre pair (k = 4): ll = i, rr = j*, the arg-max count itself
                T += ni;
                np += 1;
                if (live && half == 0 && ni != 0) {
                    const uint32_t pass = threshold_rejects(T, np, score, p.threshold_factor, threshold_fast_ok(p.threshold_factor)) ? 0u : 1u;
                    smem[hdr_off >> 2] = (ej << 1) | pass;  // cellPairs[i] (both lanes of the cell have read the header above)
                }
            }
        }
        __syncthreads();

        // ---- mark: cellPairs[l] == r for the matches whose left cell is one of the band's own
        auto mark_one = [&](uint32_t cw, int i) {
            const uint32_t lx = (((cw >> kHxShift) & 63u) + (uint32_t)gx) >> 1;
            const uint32_t ly = (((cw >> kHyShift) & 63u) + (uint32_t)gy) >> 1;
            if ((cw & kBinned) && lx < (uint32_t)kLeftW && ly >= (uint32_t)lo && ly < (uint32_t)hi) {
                const uint32_t cp = smem[((__umul24(ly - (uint32_t)row0, (uint32_t)kLeftW) + lx) * kRowBytes) >> 2];
                if (cp == (((cw & 0x1FFu) << 1) | 1u)) mask[i] = 1;
            }
        };
#pragma unroll
        for (int k = 0; k < kResident; ++k) asm volatile("" : "+v"(res[k].x));  // (as above: no fields kept across verify)
#pragma unroll
        for (int k = 0; k < kResident; ++k) mark_one(res[k].x, (int)res[k].y);
        if (len > kResident * 1024) stream_list(list + kResident * 1024, len - kResident * 1024, tid, mark_one);
        __syncthreads();  // the next grid type clears the matrix
    }
}

// ---- order-preserving compaction: workgroup (tile, pair) writes the survivors among matches [tile * 16384, +16384) ----
// Its output offset is the number of survivors in front of the tile, which it counts from the mask itself (at most
// 256 KB of bytes to sum) instead of waiting for other workgroups; the pair's last tile also knows the total.
__device__ __forceinline__ uint32_t mask_bits16(const uint8_t* mask, int first, int m)
{
    uint32_t bits = 0;
    if (first + 16 <= m && ((reinterpret_cast<uintptr_t>(mask) + (uintptr_t)first) & 15u) == 0) {
        const uint4 v = *reinterpret_cast<const uint4*>(mask + first);  // 16 mask bytes, each 0 or 1
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        // bytes b0..b3 (bit 0 of each) -> bits 24..27: b_k * 2^(8k) * 2^(24 - 7k); the cross terms stay below bit 20
#pragma unroll
        for (int q = 0; q < 4; ++q) bits |= ((((w[q] & 0x01010101u) * 0x01020408u) >> 24) & 15u) << (4 * q);
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (first + k < m && mask[first + k]) bits |= 1u << k;
    }
    return bits;
}

__global__ void __launch_bounds__(1024)
band_compact_kernel(FilterParams p, const uint32_t* flags, uint8_t* mask_ws, const uint32_t* state, int mcap)
{
    __shared__ uint32_t wave_tile[16], wave_before[16];
    const int tile = blockIdx.x, pi = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const gms_pair pr = p.pairs[pi];
    const int m = pr.m;
    const uint32_t fl = flags[pi];
    if (fl & kFlagGeneral) return;  // gms_kernel_big.hip produces this pair
    const bool failed = (fl & kFlagDomain) != 0 || m < 0 || m > mcap;
    const int n_tiles = (failed || m <= 0) ? 1 : (m + 16383) >> 14;
    if (tile >= n_tiles) return;
    uint32_t total = 0;
    if (!failed && m > 0) {
        const gms_dmatch* __restrict__ matches = p.matches + pr.match_off;
        gms_dmatch* __restrict__ out = p.out + pr.match_off;
        const uint8_t* mask = p.mask ? p.mask + pr.match_off : mask_ws + (size_t)pi * mcap;
        // survivors in front of this tile (this thread's share of them)
        uint32_t before = 0;
        for (int f = tid * 16; f < tile * 16384; f += 16 * 1024) before += (uint32_t)__popc(mask_bits16(mask, f, m));
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d);
        // a wave owns 1024 consecutive matches of the tile, 64 at a time: loads and stores are coalesced
        const int wbase = tile * 16384 + wave * 1024;
        uint8_t mb[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = wbase + j * 64 + lane;
            mb[j] = mask[min(i, m - 1)];  // (unconditional loads: sixteen in flight, not sixteen round trips)
            if (i >= m) mb[j] = (uint8_t)0;
        }
        unsigned long long bal[16];
        uint32_t wave_count = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            bal[j] = __ballot(mb[j] != 0);
            wave_count += (uint32_t)__popcll(bal[j]);
        }
        if (lane == 0) {
            wave_tile[wave] = wave_count;
            wave_before[wave] = before;
        }
        __syncthreads();
        uint32_t pos = 0, tile_total = 0, before_total = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) pos += wave_tile[w];
            tile_total += wave_tile[w];
            before_total += wave_before[w];
        }
        pos += before_total;
        const unsigned long long lt = (1ull << lane) - 1ull;
        // the wave's records, eight loads in flight at a time (requested unconditionally and pinned before the stores: a load that
        // only a conditional store uses is sunk into the branch by the compiler and waited for there, one round trip per record)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 rec[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) rec[j] = *reinterpret_cast<const uint4*>(&matches[min(wbase + (h * 8 + j) * 64 + lane, m - 1)]);
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(rec[j].x), "+v"(rec[j].y), "+v"(rec[j].z), "+v"(rec[j].w));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (mb[h * 8 + j]) *reinterpret_cast<uint4*>(&out[pos + (uint32_t)__popcll(bal[h * 8 + j] & lt)]) = rec[j];
                pos += (uint32_t)__popcll(bal[h * 8 + j]);
            }
        }
        total = before_total + tile_total;
    } else if (failed && m > 0 && m <= mcap && p.mask) {
        uint8_t* mask = p.mask + pr.match_off;  // a failed pair keeps nothing
        for (int i = tid; i < m; i += 1024) mask[i] = 0;
    }
    if (tid == 0 && tile == n_tiles - 1) {
        gms_pair_result r;
        r.n_inliers = failed ? 0 : (int)total;
        // default flags: the one hypothesis (scale 0, rotation 1); otherwise what tile_apply_kernel recorded
        r.best_scale = (failed || total == 0) ? -1 : (state ? (int)state[pi * 4 + 1] : 0);
        r.best_rot = (failed || total == 0) ? -1 : (state ? (int)state[pi * 4 + 2] : 1);
        r.status = failed ? GMS_ERR_DOMAIN : GMS_OK;
        p.results[pi] = r;
    }
}

// ================================================================================================================================
// Large pairs with rotation and/or scale hypotheses (BASELINE config 4: 4K pairs, 50k features, FeatureMatchUtil.cpp:69 flags).
//
// The same 16-bit LDS matrix, generalised from bands of rows to TILES of left cells: a scale's right grid has nr = wr * wr
// cells, a left cell's row is [header | nr x u16], and a workgroup keeps an own block of R x C left cells plus a halo ring
// (whatever of the ring is inside the grid) -- sized so that the block fits 144 KB: 7 x 20 own cells at scale 0 (20 x 20 right
// cells), the whole grid at scale 1 (10 x 10), 16 x 20 at scale 2 (14 x 14), 7 x 7 at scale 3 (28 x 28), 4 x 5 at scale 4
// (40 x 40). One workgroup per (tile, grid type); tiles and types are independent (each ORs rotation bits into the pair's
// byte-per-match hypothesis mask). Per scale four launches -- tile_codes_kernel (code words with this scale's right cell,
// appended to the lists of the at most 2 x 2 tiles that keep the match's left cell), tile_filter_kernel (bin, verify under
// all rotations, mark), tile_count_kernel + tile_apply_kernel (inliers per rotation, getInlierMask's strict '>' against the
// best hypothesis so far, whose mask is replaced when beaten) -- and after the last scale band_compact_kernel copies the survivors out.
// Pairs with a cell above 65 535 matches are flagged for the HBM-slab kernel as in the default-flag path.
// ================================================================================================================================
struct TileGeom {
    int scale;          // index into FilterParams::right_w / right_h
    int wr, nr;         // right grid width, cells
    int own_r, own_c;   // own block of left cells
    int tiles_y, tiles_x;
    uint32_t row_bytes; // 4 + 2 * nr
};
constexpr int kMaxTiles = 32;
constexpr uint32_t kTileMatrixBytes = 144u * 1024u;
constexpr uint32_t kTileNleftOff = kTileMatrixBytes;                 // [400] u32
constexpr uint32_t kTileFineOff = kTileNleftOff + 4u * kLeftN;       // [1600] u32
constexpr uint32_t kTileMiscOff = kTileFineOff + 4u * kFineN;        // [16] u32
constexpr uint32_t kTileLdsBytes = kTileMiscOff + 64u;
static_assert(kTileLdsBytes <= kLdsBytes, "tile layout exceeds the LDS");
// code word: E' = nr - r (1..nr) : 11 | hx : 6 | hy : 6 | binned : 1
constexpr int kTHxShift = 11, kTHyShift = 17;
constexpr uint32_t kTBinned = 1u << 23;

namespace {
// tiles whose kept rows (own + halo) contain left row r: [lo, hi]
__device__ __forceinline__ void tiles_holding(int r, int own, int n_tiles, int& lo, int& hi)
{
    lo = r >= 1 ? (r - 1) / own : 0;
    hi = min((r + 1) / own, n_tiles - 1);
}
}  // namespace

__global__ void __launch_bounds__(1024)
tile_codes_kernel(FilterParams p, TileGeom gm, uint2* lists, uint32_t* list_len, uint32_t* nfine_g, uint32_t* flags,
                  uint8_t* rotmask, int mcap)
{
    __shared__ uint32_t hist[kFineN];
    __shared__ uint32_t cnt_l[kMaxTiles], base_g[kMaxTiles];
    const int pi = blockIdx.y, tid = threadIdx.x;
    const int n_tiles = gm.tiles_y * gm.tiles_x;
    const gms_pair pr = p.pairs[pi];
    const int m = pr.m;
    const bool bad_pair = m < 0 || m > mcap || pr.frame_a < 0 || pr.frame_a >= p.n_frames || pr.frame_b < 0 ||
                          pr.frame_b >= p.n_frames;
    if (bad_pair) {
        if (blockIdx.x == 0 && tid == 0) atomicOr(&flags[pi], kFlagDomain);
        return;
    }
    const int base = blockIdx.x * 4096;
    if (base >= m) return;  // workgroup-uniform
    const int64_t offA = p.frame_off[pr.frame_a], offB = p.frame_off[pr.frame_b];
    const int nA = (int)(p.frame_off[pr.frame_a + 1] - offA), nB = (int)(p.frame_off[pr.frame_b + 1] - offB);
    if (nA <= 0 || nB <= 0) {
        if (blockIdx.x == 0 && tid == 0) atomicOr(&flags[pi], kFlagDomain);
        return;
    }
    const float2* __restrict__ ptsA = p.pts + offA;
    const float2* __restrict__ ptsB = p.pts + offB;
    const gms_dmatch* __restrict__ matches = p.matches + pr.match_off;
    uint8_t* rm = rotmask + (size_t)pi * mcap;

    for (int j = tid; j < kFineN; j += 1024) hist[j] = 0;
    if (tid < kMaxTiles) cnt_l[tid] = 0;
    __syncthreads();
    uint2 qt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) qt[k] = *reinterpret_cast<const uint2*>(&matches[min(base + k * 1024 + tid, m - 1)]);
    float2 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = ptsA[min(qt[k].x, (uint32_t)(nA - 1))];
        b[k] = ptsB[min(qt[k].y, (uint32_t)(nB - 1))];
    }
    const float fwr = (float)gm.wr;
    bool any_bad = false;
    uint32_t cw[4], rank[4][4], tile_of[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = base + k * 1024 + tid;
        const bool live = i < m;
        const uint32_t worst = max(max(__float_as_uint(a[k].x), __float_as_uint(a[k].y)),
                                   max(__float_as_uint(b[k].x), __float_as_uint(b[k].y)));
        const float fx = 20.0f * a[k].x, fy = 20.0f * a[k].y;   // mulss, rounded to fp32
        const uint32_t hx = (uint32_t)(int)(fx + fx), hy = (uint32_t)(int)(fy + fy);  // floor(2f), 2f exact
        // getGridIndexRight of this scale: (int)(wr * x) + (int)(wr * y) * wr, no bounds test (clamped 24-bit form, exact below nr)
        const uint32_t rx = (uint32_t)(int)(fwr * b[k].x), ry = (uint32_t)(int)(fwr * b[k].y);
        const uint32_t r = __umul24(min(ry, 4096u), (uint32_t)gm.wr) + min(rx, 4096u);
        const bool ok = qt[k].x < (uint32_t)nA && qt[k].y < (uint32_t)nB && worst < 0x49800000u && r < (uint32_t)gm.nr;
        const bool binned = live && ok && hx < 40u && hy < 40u;
        if (binned) atomicAdd(&hist[hy * kFineW + hx], 1u);
        any_bad |= live && !ok;
        if (live) rm[i] = 0;
        cw[k] = binned ? (((uint32_t)gm.nr - r) | (hx << kTHxShift) | (hy << kTHyShift) | kTBinned) : 0u;
        // the tiles that keep this match's left cell under any grid type: rows hy >> 1 and (hy + 1) >> 1, columns likewise
        int ty_lo = 0, ty_hi = -1, tx_lo = 0, tx_hi = -1, dummy;
        if (binned) {
            tiles_holding((int)(hy >> 1), gm.own_r, gm.tiles_y, ty_lo, dummy);
            tiles_holding((int)min((hy + 1u) >> 1, 19u), gm.own_r, gm.tiles_y, dummy, ty_hi);
            tiles_holding((int)(hx >> 1), gm.own_c, gm.tiles_x, tx_lo, dummy);
            tiles_holding((int)min((hx + 1u) >> 1, 19u), gm.own_c, gm.tiles_x, dummy, tx_hi);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ty = ty_lo + (q >> 1), tx = tx_lo + (q & 1);
            const bool need = ty <= ty_hi && tx <= tx_hi;
            tile_of[k][q] = need ? (uint32_t)(ty * gm.tiles_x + tx) : 0u;
            rank[k][q] = need ? atomicAdd(&cnt_l[tile_of[k][q]], 1u) : 0xFFFFFFFFu;
        }
    }
    if (any_bad) atomicOr(&flags[pi], kFlagDomain);
    __syncthreads();
    uint32_t* nf = nfine_g + (size_t)pi * kFineN;
    for (int j = tid; j < kFineN; j += 1024)
        if (hist[j]) atomicAdd(&nf[j], hist[j]);
    if (tid < n_tiles) base_g[tid] = cnt_l[tid] ? atomicAdd(&list_len[pi * kMaxTiles + tid], cnt_l[tid]) : 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (rank[k][q] != 0xFFFFFFFFu)
                lists[((size_t)pi * n_tiles + tile_of[k][q]) * mcap + base_g[tile_of[k][q]] + rank[k][q]] =
                    make_uint2(cw[k], (uint32_t)(base + k * 1024 + tid));
}

template <bool ROT>
__global__ void __launch_bounds__(1024)
tile_filter_kernel(FilterParams p, TileGeom gm, const uint2* lists, const uint32_t* list_len, const uint32_t* nfine_g,
                   uint32_t* flags, uint32_t* rotmask32, int mcap)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int n_tiles = gm.tiles_y * gm.tiles_x;
    const int tile = blockIdx.x % n_tiles, g = blockIdx.x / n_tiles, pi = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const gms_pair pr = p.pairs[pi];
    const int m = pr.m;
    if (m <= 0 || m > mcap) return;
    if (flags[pi] & (kFlagDomain | kFlagGeneral)) return;  // written by earlier launches
    const uint2* __restrict__ list = lists + ((size_t)pi * n_tiles + tile) * mcap;
    const int len = (int)list_len[pi * kMaxTiles + tile];
    const uint32_t* __restrict__ nf = nfine_g + (size_t)pi * kFineN;
    uint32_t* rm32 = rotmask32 + ((size_t)pi * mcap >> 2);

    const uint8_t* bytes = reinterpret_cast<const uint8_t*>(smem);
    uint32_t* nleft = smem + kTileNleftOff / 4;
    uint32_t* nfine = smem + kTileFineOff / 4;
    uint32_t* misc = smem + kTileMiscOff / 4;
    const int ty = tile / gm.tiles_x, tx = tile % gm.tiles_x;
    const int y0 = ty * gm.own_r, y1 = min(y0 + gm.own_r, kLeftH), x0 = tx * gm.own_c, x1 = min(x0 + gm.own_c, kLeftW);  // own cells
    const int ky0 = max(y0 - 1, 0), ky1 = min(y1 + 1, kLeftH), kx0 = max(x0 - 1, 0), kx1 = min(x1 + 1, kLeftW);         // cells kept
    const uint32_t kw = (uint32_t)(kx1 - kx0), kh = (uint32_t)(ky1 - ky0), ow = (uint32_t)(x1 - x0), oh = (uint32_t)(y1 - y0);
    const uint32_t row_bytes = gm.row_bytes, wr = (uint32_t)gm.wr, nr = (uint32_t)gm.nr;
    const uint32_t clear16 = (kw * kh * row_bytes + 15u) >> 4;
    const int gx = g & 1, gy = g >> 1;

    for (int j = tid; j < kFineN; j += 1024) nfine[j] = nf[j];
    if (tid == 0) misc[0] = 0;
    __syncthreads();
    auto nleft_of = [&](int cell, int sx, int sy) -> uint32_t {
        const int hx0 = 2 * (cell % kLeftW) - sx, hy0 = 2 * (cell / kLeftW) - sy;
        uint32_t n = 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int hx = hx0 + dx, hy = hy0 + dy;
                if (hx >= 0 && hy >= 0) n += nfine[hy * kFineW + hx];
            }
        return n;
    };
    for (int item = tid; item < 4 * kLeftN; item += 1024) {
        const int gg = item / kLeftN;
        if (nleft_of(item - gg * kLeftN, gg & 1, gg >> 1) > 65535u) misc[0] = 1;
    }
    if (tid < kLeftN) nleft[tid] = nleft_of(tid, gx, gy);
    {
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        uint4* d4 = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < clear16; i += 1024) d4[i] = z4;
    }
    __syncthreads();
    if (misc[0]) {  // the same decision in every workgroup of the pair
        if (blockIdx.x == 0 && tid == 0) atomicOr(&flags[pi], kFlagGeneral);
        return;
    }

    // ---- assignMatchPairs for the cells this tile keeps
    stream_list(list, len, tid, [&](uint32_t cw, int) {
        const uint32_t lx = (((cw >> kTHxShift) & 63u) + (uint32_t)gx) >> 1;
        const uint32_t ly = (((cw >> kTHyShift) & 63u) + (uint32_t)gy) >> 1;
        // x >= 20 || y >= 20 -> -1 (DLL@0x180047d3d); cells outside the kept block belong to other tiles
        if ((cw & kTBinned) && lx - (uint32_t)kx0 < kw && ly - (uint32_t)ky0 < kh && lx < (uint32_t)kLeftW && ly < (uint32_t)kLeftH) {
            const uint32_t e = cw & 0x7FFu;  // E' = nr - r
            const uint32_t row = (__umul24(ly - (uint32_t)ky0, kw) + (lx - (uint32_t)kx0)) * row_bytes;
            const uint32_t at = row + 4u + 2u * (e - 1u);
            const uint32_t sh = (at & 2u) << 3;
            const uint32_t old = atomicAdd(lds_at(smem, at & ~3u), 1u << sh);
            atomicMax(lds_at(smem, row), (((old >> sh) & 0xFFFFu) << 11) | e);
        }
    });
    __syncthreads();

    // ---- verifyCellPairs for the own cells: two lanes per cell without rotation, one lane per (cell, rotation) with
    uint32_t rot_pack = 0;
    if (ROT) {
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {  // k8-th outer neighbour = position k8 (k8 < 4) or k8 + 1
            const int u = (int)((0x45637210u >> (4 * k8)) & 15u);  // position -> ring index: {0,1,2,7,.,3,6,5,4}
            // ring 0,1,2,5,8,7,6,3 clockwise; pattern rot sends ring index u to (u - rot) mod 8 (see gms_kernels.hip)
            const int q = (int)((0x36785210u >> ((((u - (tid & 7)) & 7)) << 2)) & 15u);
            const int dx = (int)((0x24924u >> (q << 1)) & 3u) - 1, dy = (int)((0x2a540u >> (q << 1)) & 3u) - 1;
            rot_pack |= (uint32_t)((dx + 1) | ((dy + 1) << 2)) << (4 * k8);
        }
    }
    {
        const int n_items = (int)(ow * oh) * (ROT ? 8 : 2);
        for (int item = tid; item < ((n_items + 63) & ~63); item += 1024) {
            const bool live = item < n_items;
            const uint32_t c = live ? (uint32_t)(ROT ? item >> 3 : item >> 1) : 0u;
            const int half = item & 1;  // !ROT only
            const int ix = x0 + (int)(c % ow), iy = y0 + (int)(c / ow);
            const int i = iy * kLeftW + ix;
            const uint32_t hdr_off = (__umul24((uint32_t)(iy - ky0), kw) + (uint32_t)(ix - kx0)) * row_bytes;
            const uint32_t ni = live ? nleft[i] : 0u;
            if (__ballot(ni != 0) == 0ull) continue;
            const uint32_t best = smem[hdr_off >> 2];  // ((max count - 1) << 11) | E'(j*), lowest j* among maxima
            const uint32_t ej = ni ? (best & 0x7FFu) : nr;
            const uint32_t j = nr - ej;
            const int jy = (int)(j / wr), jx = (int)(j - (uint32_t)jy * wr);
            uint32_t score = 0, T = 0, np = 0;
#pragma unroll
            for (int q = 0; q < (ROT ? 8 : 4); ++q) {
                int ldx, ldy, rdx, rdy;
                if (ROT) {
                    const int k = q < 4 ? q : q + 1;
                    ldx = (k % 3) - 1; ldy = (k / 3) - 1;
                    rdx = (int)((rot_pack >> (4 * q)) & 3u) - 1;
                    rdy = (int)((rot_pack >> (4 * q + 2)) & 3u) - 1;
                } else {
                    const int k = half ? q + 5 : q;  // lane 0: neighbours 0..3, lane 1: neighbours 5..8 (4 is the centre)
                    ldx = (k % 3) - 1; ldy = (k / 3) - 1;
                    rdx = ldx; rdy = ldy;
                }
                const int lx = ix + ldx, ly = iy + ldy, rx = jx + rdx, ry = jy + rdy;
                const bool okl = ni != 0 && (uint32_t)lx < (uint32_t)kLeftW && (uint32_t)ly < (uint32_t)kLeftH;  // ll != -1
                const bool okp = okl && (uint32_t)rx < wr && (uint32_t)ry < wr;                                    // rr != -1
                if (okp) {  // within one cell of an own cell: kept
                    const uint32_t at = (__umul24((uint32_t)(ly - ky0), kw) + (uint32_t)(lx - kx0)) * row_bytes + 4u +
                                        2u * (nr - 1u - (uint32_t)(rx + ry * (int)wr));  // entries are stored by E' - 1 = nr - 1 - r
                    score += *reinterpret_cast<const uint16_t*>(bytes + at);
                    T += nleft[ly * kLeftW + lx];
                    np += 1;
                }
            }
            if (!ROT) {
                score += dpp_xor1(score);
                T += dpp_xor1(T);
                np += dpp_xor1(np);
            }
            score += (best >> 11) + 1u;  // centre pair (k = 4): ll = i, rr = j*, the arg-max count itself
            T += ni;
            np += 1;
            uint32_t pass = 0;
            if (live && ni != 0 && (ROT || half == 0)) pass = threshold_rejects(T, np, score, p.threshold_factor, threshold_fast_ok(p.threshold_factor)) ? 0u : 1u;
            uint32_t bits = pass;
            bool writer = live && ni != 0 && half == 0;
            if (ROT) {
                const unsigned long long bal = __ballot(pass);
                bits = (uint32_t)(bal >> (lane & 56)) & 0xFFu;
                writer = live && ni != 0 && (lane & 7) == 0;
            }
            if (writer) smem[hdr_off >> 2] = (ej << 8) | bits;  // cellPairs[i] (every lane of the cell has read the header above)
        }
    }
    __syncthreads();

    // ---- mark: cellPairs[l] == r for the matches whose left cell is one of the own cells; rotation bits into the byte mask
    stream_list(list, len, tid, [&](uint32_t cw, int i) {
        const uint32_t lx = (((cw >> kTHxShift) & 63u) + (uint32_t)gx) >> 1;
        const uint32_t ly = (((cw >> kTHyShift) & 63u) + (uint32_t)gy) >> 1;
        if ((cw & kTBinned) && lx - (uint32_t)x0 < ow && ly - (uint32_t)y0 < oh) {
            const uint32_t cp = smem[((__umul24(ly - (uint32_t)ky0, kw) + (lx - (uint32_t)kx0)) * row_bytes) >> 2];
            if ((cp >> 8) == (cw & 0x7FFu) && (cp & 0xFFu) != 0) atomicOr(&rm32[i >> 2], (cp & 0xFFu) << ((i & 3) << 3));
        }
    });
}

// Inliers per rotation of the scale just filtered (tile_count_kernel: grid 16k-match tiles x pairs, into cnt_g[pair][8]), then
// getInlierMask's strict '>' (scale outer, rotation inner) against the best hypothesis so far (tile_apply_kernel: every
// workgroup of a pair reads the same counts and the same incoming state, so all take the same decision; the winner's inlier
// bytes replace the best mask tile by tile, workgroup 0 writes the outgoing state). state = {best count, best scale, best
// rotation, -} per pair, ping-ponged between two arrays from scale to scale.
template <bool ROT>
__global__ void __launch_bounds__(1024)
tile_count_kernel(FilterParams p, const uint8_t* rotmask, uint32_t* cnt_g, const uint32_t* flags, int mcap)
{
    __shared__ uint32_t cnt[8];
    const int tile = blockIdx.x, pi = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int m = p.pairs[pi].m;
    if (m <= 0 || m > mcap || tile * 16384 >= m || (flags[pi] & (kFlagDomain | kFlagGeneral))) return;
    constexpr int kNRot = ROT ? 8 : 1;
    if (tid < 8) cnt[tid] = 0;
    __syncthreads();
    const int first = tile * 16384 + tid * 16;  // 16 consecutive mask bytes per thread (the slab is 64-byte aligned)
    uint4 v = make_uint4(0, 0, 0, 0);
    if (first < m) v = *reinterpret_cast<const uint4*>(rotmask + (size_t)pi * mcap + first);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
    if (first + 16 > m) {  // the slab beyond this pair's m matches holds whatever an earlier, larger pair left there
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int left = m - (first + 4 * q);  // bytes of this dword that belong to the pair
            w[q] = left >= 4 ? w[q] : left <= 0 ? 0u : (w[q] & ((1u << (8 * left)) - 1u));
        }
    }
    uint32_t c[kNRot];
#pragma unroll
    for (int r = 0; r < kNRot; ++r) {
        uint32_t acc = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += (((w[q] >> r) & 0x01010101u) * 0x01010101u) >> 24;  // bytes with bit r set
        c[r] = acc;
    }
#pragma unroll
    for (int r = 0; r < kNRot; ++r) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c[r] += __shfl_xor(c[r], d);
        if (lane == 0 && c[r]) atomicAdd(&cnt[r], c[r]);
    }
    __syncthreads();
    if (tid < kNRot && cnt[tid]) atomicAdd(&cnt_g[pi * 8 + tid], cnt[tid]);
}

template <bool ROT>
__global__ void __launch_bounds__(1024)
tile_apply_kernel(FilterParams p, int scale, const uint8_t* rotmask, uint8_t* bestmask_ws, const uint32_t* cnt_g,
                  const uint32_t* state_in, uint32_t* state_out, const uint32_t* flags, int mcap)
{
    const int tile = blockIdx.x, pi = blockIdx.y, tid = threadIdx.x;
    const gms_pair pr = p.pairs[pi];
    const int m = pr.m;
    if (m <= 0 || m > mcap || (flags[pi] & (kFlagDomain | kFlagGeneral))) return;
    constexpr int kNRot = ROT ? 8 : 1;
    uint32_t best = state_in[pi * 4 + 0];
    int w = -1;
#pragma unroll
    for (int r = 0; r < kNRot; ++r) {
        const uint32_t c = cnt_g[pi * 8 + r];
        if (c > best) {
            best = c;
            w = r;
        }
    }
    if (tile == 0 && tid == 0) {
        state_out[pi * 4 + 0] = best;
        state_out[pi * 4 + 1] = w >= 0 ? (uint32_t)scale : state_in[pi * 4 + 1];
        state_out[pi * 4 + 2] = w >= 0 ? (uint32_t)(w + 1) : state_in[pi * 4 + 2];
    }
    if (w < 0 && scale != 0) return;  // (the first scale also initialises the mask: nothing may ever win)
    const uint8_t* rm = rotmask + (size_t)pi * mcap;
    uint8_t* bm = p.mask ? p.mask + pr.match_off : bestmask_ws + (size_t)pi * mcap;
    for (int i = tile * 16384 + tid; i < min(m, (tile + 1) * 16384); i += 1024) bm[i] = w >= 0 ? (rm[i] >> w) & 1u : 0u;
}

// ---- launch helpers ----------------------------------------------------------------------------------------------------
hipError_t init_band_kernels()  // once per context: see init_filter_kernels
{
    const void* fns[] = {reinterpret_cast<const void*>(band_filter_kernel), reinterpret_cast<const void*>(tile_filter_kernel<true>),
                         reinterpret_cast<const void*>(tile_filter_kernel<false>)};
    for (const void* fn : fns) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

size_t band_ws_bytes_per_pair(int mcap, bool need_mask)
{
    return (size_t)mcap * 24 + (size_t)kFineN * 4 + 16 + (need_mask ? (size_t)mcap : 0);
}

// ws layout for n pairs: lists [n][3][mcap] uint2 | nfine [n][1600] u32 | list_len [n][3] u32 | flags [n] u32 |
//                        mask [n][mcap] u8 (if p.mask is null)
hipError_t launch_filter_band(const FilterParams& p, int mcap, void* ws, const uint32_t** flags_out, hipStream_t stream)
{
    const int n = p.n_pairs;
    if (n <= 0) return hipSuccess;
    uint2* lists = reinterpret_cast<uint2*>(ws);
    uint32_t* nfine = reinterpret_cast<uint32_t*>(lists + (size_t)n * 3 * mcap);
    uint32_t* list_len = nfine + (size_t)n * kFineN;
    uint32_t* flags = list_len + (size_t)n * 3;
    uint8_t* mask_ws = reinterpret_cast<uint8_t*>(flags + n);
    hipError_t e = hipMemsetAsync(nfine, 0, ((size_t)n * kFineN + (size_t)n * 4) * 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(band_codes_kernel, dim3((unsigned)((mcap + 4095) / 4096), (unsigned)n), dim3(1024), 0, stream, p, lists,
                       list_len, nfine, flags, mask_ws, mcap);
    // few pairs: one workgroup per (band, grid type) -- 12 per pair -- so that a single large pair spreads over more CUs (the
    // grid types of a band are independent: each only ORs into the mask); many pairs: one per band, which streams less
    const unsigned per_pair = n < 64 ? 12u : 3u;
    hipLaunchKernelGGL(band_filter_kernel, dim3(per_pair, (unsigned)n), dim3(1024), kBandLdsBytes, stream, p, lists, list_len,
                       nfine, flags, mask_ws, mcap);
    hipLaunchKernelGGL(band_compact_kernel, dim3((unsigned)((mcap + 16383) / 16384), (unsigned)n), dim3(1024), 0, stream, p, flags,
                       mask_ws, (const uint32_t*)nullptr, mcap);
    *flags_out = flags;
    return hipGetLastError();
}


// ---- rotation / scale hypotheses for large pairs ------------------------------------------------------------------------------
static TileGeom tile_geom(const FilterParams& p, int scale)
{
    TileGeom g;
    g.scale = scale;
    g.wr = p.right_w[scale];
    g.nr = g.wr * p.right_h[scale];
    g.row_bytes = 4u + 2u * (uint32_t)g.nr;
    // the own block (R x C left cells) whose kept block (own + halo ring, clipped to the grid) fits the matrix area with the
    // fewest tiles, then the fewest kept cells (every kept cell beyond the own ones is a match streamed twice)
    const int cap = (int)(kTileMatrixBytes / g.row_bytes);  // cells that fit
    auto kept = [](int own, int n) { return own >= n ? n : (own + 2 < n ? own + 2 : n); };
    int best_tiles = 1 << 30, best_kept = 1 << 30;
    g.own_r = g.own_c = 0;
    for (int r = 3; r <= kLeftH; ++r)
        for (int c = 3; c <= kLeftW; ++c) {
            if (kept(r, kLeftH) * kept(c, kLeftW) > cap) continue;
            const int ty = (kLeftH + r - 1) / r, tx = (kLeftW + c - 1) / c;
            const int total_kept = ty * tx * kept(r, kLeftH) * kept(c, kLeftW);
            if (ty * tx < best_tiles || (ty * tx == best_tiles && total_kept < best_kept)) {
                best_tiles = ty * tx;
                best_kept = total_kept;
                g.own_r = r;
                g.own_c = c;
            }
        }
    if (g.own_r == 0) {  // (cannot happen for the reference's five right grids: 3 x 3 own cells need 25 kept cells)
        g.own_r = g.own_c = 1;
        g.tiles_y = g.tiles_x = 1 << 10;
        return g;
    }
    g.tiles_y = (kLeftH + g.own_r - 1) / g.own_r;
    g.tiles_x = (kLeftW + g.own_c - 1) / g.own_c;
    return g;
}

size_t tile_ws_bytes_per_pair(const FilterParams& p, int mcap, bool need_mask)
{
    int max_tiles = 1;
    const int n_scales = p.with_scale ? 5 : 1;
    for (int s = 0; s < n_scales; ++s) {
        const TileGeom g = tile_geom(p, s);
        max_tiles = g.tiles_y * g.tiles_x > max_tiles ? g.tiles_y * g.tiles_x : max_tiles;
    }
    return (size_t)max_tiles * mcap * 8 + (size_t)kFineN * 4 + kMaxTiles * 4 + 32 + 4 + 32 + (size_t)mcap + (need_mask ? (size_t)mcap : 0) + 64;
}

// ws layout for n pairs: lists [n][tiles][mcap] uint2 | nfine [n][1600] | list_len [n][32] | cnt [n][8] | flags [n] |
//                        state [2][n][4] | rotmask [n][mcap] u8 | bestmask [n][mcap] u8 (if p.mask is null)
hipError_t launch_filter_tiles(const FilterParams& p, int mcap, void* ws, const uint32_t** flags_out, hipStream_t stream)
{
    const int n = p.n_pairs;
    if (n <= 0) return hipSuccess;
    const int n_scales = p.with_scale ? 5 : 1;
    int max_tiles = 1;
    for (int s = 0; s < n_scales; ++s) {
        const TileGeom g = tile_geom(p, s);
        if (g.tiles_y * g.tiles_x > kMaxTiles || g.own_r < 1 || g.own_c < 1) return hipErrorInvalidValue;
        max_tiles = g.tiles_y * g.tiles_x > max_tiles ? g.tiles_y * g.tiles_x : max_tiles;
    }
    uint2* lists = reinterpret_cast<uint2*>(ws);
    uint32_t* nfine = reinterpret_cast<uint32_t*>(lists + (size_t)n * max_tiles * mcap);
    uint32_t* list_len = nfine + (size_t)n * kFineN;
    uint32_t* cnt = list_len + (size_t)n * kMaxTiles;
    uint32_t* flags = cnt + (size_t)n * 8;
    uint32_t* state = flags + n;  // two arrays of n x 4, used alternately
    uint8_t* rotmask = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(state + (size_t)n * 8) + 15) & ~(uintptr_t)15);  // read 16 bytes at a time
    uint8_t* bestmask = rotmask + (size_t)n * mcap;
    hipError_t e = hipMemsetAsync(flags, 0, (size_t)n * 9 * 4, stream);  // flags + both states
    if (e != hipSuccess) return e;
    if (p.mask == nullptr) {
        e = hipMemsetAsync(bestmask, 0, (size_t)n * mcap, stream);
        if (e != hipSuccess) return e;
    }
    const bool rot = p.with_rotation != 0;
    for (int s = 0; s < n_scales; ++s) {
        const TileGeom g = tile_geom(p, s);
        const int n_tiles = g.tiles_y * g.tiles_x;
        e = hipMemsetAsync(nfine, 0, ((size_t)n * kFineN + (size_t)n * kMaxTiles + (size_t)n * 8) * 4, stream);  // histogram, list lengths, counts
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(tile_codes_kernel, dim3((unsigned)((mcap + 4095) / 4096), (unsigned)n), dim3(1024), 0, stream, p, g, lists,
                           list_len, nfine, flags, rotmask, mcap);
        if (rot)
            hipLaunchKernelGGL(tile_filter_kernel<true>, dim3((unsigned)(n_tiles * 4), (unsigned)n), dim3(1024), kTileLdsBytes, stream, p, g,
                               lists, list_len, nfine, flags, reinterpret_cast<uint32_t*>(rotmask), mcap);
        else
            hipLaunchKernelGGL(tile_filter_kernel<false>, dim3((unsigned)(n_tiles * 4), (unsigned)n), dim3(1024), kTileLdsBytes, stream, p, g,
                               lists, list_len, nfine, flags, reinterpret_cast<uint32_t*>(rotmask), mcap);
        const dim3 tg((unsigned)((mcap + 16383) / 16384), (unsigned)n);
        const uint32_t* st_in = state + (size_t)(s & 1) * n * 4;
        uint32_t* st_out = state + (size_t)((s + 1) & 1) * n * 4;
        if (rot) {
            hipLaunchKernelGGL(tile_count_kernel<true>, tg, dim3(1024), 0, stream, p, rotmask, cnt, flags, mcap);
            hipLaunchKernelGGL(tile_apply_kernel<true>, tg, dim3(1024), 0, stream, p, s, rotmask, bestmask, cnt, st_in, st_out, flags, mcap);
        } else {
            hipLaunchKernelGGL(tile_count_kernel<false>, tg, dim3(1024), 0, stream, p, rotmask, cnt, flags, mcap);
            hipLaunchKernelGGL(tile_apply_kernel<false>, tg, dim3(1024), 0, stream, p, s, rotmask, bestmask, cnt, st_in, st_out, flags, mcap);
        }
    }
    hipLaunchKernelGGL(band_compact_kernel, dim3((unsigned)((mcap + 16383) / 16384), (unsigned)n), dim3(1024), 0, stream, p, flags,
                       bestmask, (const uint32_t*)(state + (size_t)(n_scales & 1) * n * 4), mcap);
    *flags_out = flags;
    return hipGetLastError();
}

}  // namespace gms
