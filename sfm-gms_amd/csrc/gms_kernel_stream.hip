// gms_kernel_stream.hip -- pairs of 16 385 ... 65 536 matches (BASELINE config 4: 4K pairs, 50k features) under every flag
// combination, on the BYTE matrix of gms_kernels.hip (dense_pair / dense_scales_pair): one 1024-thread workgroup per (pair, scale
// hypothesis) keeps the scale's 400 x N_right motion matrix -- or a band of its left rows -- in the CU's LDS, one byte per entry.
//
// What a workgroup of the small-pair kernels keeps in registers (a code word per match) does not fit here, so the matches live in a
// per-pair workspace as 8-byte ENTRIES, written once per launch by two small kernels and streamed by the filter:
//   stream_index_kernel<0>   per match: its left point's code and its right point's scale code from the frame table (no float ->
//                            cell arithmetic per pair: normalize_kernel did it per keypoint), the pair's half-cell histogram and the
//                            number of matches per left-grid ROW;
//   stream_index_kernel<1>   the codes again (as pass 0 stored them: no second gather), now placing every entry in the slot of its
//                            half row: the entry array is sorted by the left HALF row (2 x row under grid type 1 + the y parity), so
//                            that a band of left rows under any grid type is a contiguous range of it (plus nLeft of every cell
//                            under the four grid types, 16 bits each);
//   stream_filter_kernel     one workgroup per (pair, scale, grid type, band of left rows): clear the band's rows, stream the band's
//                            range of entries to bin them (one returning LDS atomic on the entry's byte, one atomicMax on the row
//                            header: the running arg-max), verify the band's own cells under all rotations, and leave per own cell
//                            [E(cellPairs[cell]) | the rotations that accept the pair] in the pair's TABLE of this (scale, grid type):
//                            400 words -- everything the marking loop of run() needs to know;
//   stream_mark_kernel       run()'s marking loop for all scales and grid types in ONE pass over the matches in their original order
//                            (their codes as stream_index_kernel<0> left them; the pair's twenty tables in LDS): per match and scale
//                            the rotations under which some grid type accepts it, counted per (scale, rotation) = run()'s return
//                            values -- per pair, and per tile of 8192 matches;
//   stream_compact_kernel    getInlierMask's strict '>' over the (scale, rotation) counts in the reference's order, the winner's bit of
//                            every match of a tile (four table look-ups), the survivors in front of the tile from the tile counts, and
//                            the order-preserving copy-out.
// Right grids: 20 x 20 fits the LDS whole; 28 x 28 takes three bands of eight rows (+ a halo row either side), 40 x 40 seven bands of
// three; a band streams only the rows it holds -- not the pair. 10 x 10 and 14 x 14 are never binned: their matrices are the 20 x 20 and
// 28 x 28 ones' rows summed two by two, in place, by the item that holds them (pool_rows).
// A byte entry that would exceed its byte (more than 255 matches in ONE (left cell, right cell) pair), a left cell above 65 535 matches or
// a frame table without code arrays flag the pair for the HBM-slab kernel (gms_kernel_big.hip), which runs behind on flagged pairs
// only. Bit-exactness rules are those of gms_kernels.hip (same codes, same arg-max and tie rules, same threshold arithmetic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "gms_device_common.h"

// Diagnostic build only (-DGMS_PHASE_TIMING, libgms_hip_diag.so; tools/stream_phase_timing.py): thread 0 of every stream_filter_kernel
// workgroup leaves the shader-clock cycles of its phases in p.diag[workgroup][phase]. No stamp exists in the product build.
#ifdef GMS_PHASE_TIMING
#define GMS_SSTAMP_DECL unsigned long long sph_[8] = {0}; unsigned long long st_prev_ = __builtin_readcyclecounter();
#define GMS_SSTAMP(k) do { unsigned long long t_ = __builtin_readcyclecounter(); sph_[k] += t_ - st_prev_; st_prev_ = t_; } while (0)
#define GMS_SSTAMP_FLUSH do { if (threadIdx.x == 0 && p.diag) { for (int k_ = 0; k_ < 8; ++k_) p.diag[(size_t)blockIdx.x * 8 + k_] = sph_[k_]; } } while (0)
#else
#define GMS_SSTAMP_DECL
#define GMS_SSTAMP(k)
#define GMS_SSTAMP_FLUSH
#endif

namespace gms {
static size_t align16s(size_t x) { return (x + 15) & ~(size_t)15; }
namespace {

constexpr int kSMaxMatches = 1 << 16;                      // (an entry holds 26 bits of original index; beyond 65 536 matches the 16-bit band / tile kernels are the better fit: entries above 255 get likely)
constexpr uint32_t kSMatrixBytes = kLeftN * (4u + 400u);   // 161 600: scale 0's matrix; every other scale's block is smaller
constexpr uint32_t kSNleftOff = kSMatrixBytes;             // [400] u16: nLeft of every cell under the current grid type
constexpr uint32_t kSRowOff = kSNleftOff + 2u * kLeftN;    // [24] u32: first entry of every left row (21 used: [20] = entries binned)
constexpr uint32_t kSMiscOff = kSRowOff + 96u;             // [16] u32: [0..7] inlier counts per rotation, [8] overflow
constexpr uint32_t kSLdsBytes = kSMiscOff + 64u;           // 162 560
static_assert(kSLdsBytes <= kLdsBytes, "stream layout exceeds the LDS");
static_assert(10u * 20u * (4u + 784u) <= kSMatrixBytes && 5u * 20u * (4u + 1600u) <= kSMatrixBytes, "the banded scales fit");

// entry.x = [q : 5 | never, edgeX, edgeY : 3 | left cell under grid type 1 : 9 | original index, low 15 bits]
// entry.y = [right cell on the 20 x 20 grid : 9 | on the 28 x 28 grid : 10 | low bit of the 40 x 40 cell's x, y : 2 | original index, bits 15..25]
// (the low 8 bits of x are the dense code word's, the low 21 of y the frame table's scale code as it stands)
constexpr int kSCellShift = 8, kSOrigShift = 17, kSOrigHiShift = 21;
constexpr uint32_t kSFlagDomain = 1u, kSFlagGeneral = 2u;  // (the values gms_kernel_band.hip / gms_kernel_big.hip use)
constexpr int kSItemsScales = 4 * (7 + 3 + 1);             // (scale, grid type, band) work items of a pair with scale hypotheses (the 10 x 10 and
                                                           // 14 x 14 grids ride on the 20 x 20 and 28 x 28 items)
constexpr int kSRowBuckets = 41;                           // 40 half rows of the left grid (2 x row of grid type 1 + the y parity) + "binned under no grid type":
                                                           // the rows a band holds under ANY grid type are a range of half rows
constexpr int kSRowWords = 192;                            // per pair: [h] matches per bucket; [64 + h] fill cursors; [128 + h] first entry of bucket h
constexpr int kSMarkTile = 8192;                           // matches per workgroup of the marking / compacting kernels
constexpr int kSTilesMax = kSMaxMatches / kSMarkTile;      // 8

struct StreamWs {
    uint2* entries;      // [n][mcap], sorted by left row
    uint2* codes;        // [n][mcap], the same words in the matches' original order
    uint32_t* nfine;     // [n][1600] half-cell histogram
    uint32_t* row_cnt;   // [n][kSRowWords]: [h] matches per half-row bucket; [64 + h] fill cursors; [128 + h] first entry of bucket h ([128 + 40]: entries binned)
    uint16_t* nleft;     // [n][4][400]
    uint32_t* counts;    // [n][5][8]
    uint32_t* tile_cnt;  // [n][kSTilesMax][5][8]: the same per tile of kSMarkTile matches
    uint32_t* flags;     // [n]
    uint32_t* tables;    // [n][scales][4 grid types][400]: E(cellPairs[cell]) << 8 | the rotations that accept the cell pair (0: none)
};


// E(r) - 3 = nr - r of the entry's right cell under scale hypothesis s (see dense_scales_pair: the coarse grids' cells are halves of
// the fine ones', the 40 x 40 cell is twice the 20 x 20 one plus a stored bit)
template <int S>
__device__ __forceinline__ uint32_t right_cell(uint32_t aux)
{
    if constexpr (S == 0) return aux & 0x1FFu;
    if constexpr (S == 3) return (aux >> 9) & 0x3FFu;
    if constexpr (S == 4) {
        const uint32_t c20 = aux & 0x1FFu, cy = (c20 * 3277u) >> 16, cx = c20 - cy * 20u;
        return (2u * cy + ((aux >> 20) & 1u)) * 40u + 2u * cx + ((aux >> 19) & 1u);
    }
    if constexpr (S == 1 || S == 2) {
        const uint32_t fine = S == 1 ? aux & 0x1FFu : (aux >> 9) & 0x3FFu, wf = S == 1 ? 20u : 28u;
        const uint32_t fy = (fine * (S == 1 ? 3277u : 2341u)) >> 16, fx = fine - fy * wf;
        return (fy >> 1) * (wf >> 1) + (fx >> 1);
    }
    return 0u;
}

}  // namespace

// ---- per match: codes, histograms (PASS 0) / its slot in the row-sorted entry array (PASS 1) ---------------------------------------------
template <int PASS>
__global__ void __launch_bounds__(1024)
stream_index_kernel(FilterParams p, StreamWs w, int mcap)
{
    __shared__ uint32_t hist[kFineN];
    __shared__ uint32_t cnt_l[64], base_g[64], row_start[64];
    const int pi = blockIdx.y, tid = threadIdx.x;
    const gms_pair pr = p.pairs[pi];
    const int m = pr.m;
    const bool bad_pair = m < 0 || m > mcap || pr.frame_a < 0 || pr.frame_a >= p.n_frames || pr.frame_b < 0 || pr.frame_b >= p.n_frames;
    if (bad_pair) {
        if (PASS == 0 && blockIdx.x == 0 && tid == 0) atomicOr(&w.flags[pi], kSFlagDomain);
        return;
    }
    uint32_t* rc = w.row_cnt + (size_t)pi * kSRowWords;
    uint2* codes = w.codes + (size_t)pi * mcap;
    const int base = blockIdx.x * 4096;
    bool any_bad = false;
    uint32_t rank[4], bucket[4];
    uint2 ent[4];
    if constexpr (PASS == 0) {
        const int64_t offA = p.frame_off[pr.frame_a], offB = p.frame_off[pr.frame_b];
        const int nA = (int)(p.frame_off[pr.frame_a + 1] - offA), nB = (int)(p.frame_off[pr.frame_b + 1] - offB);
        const int64_t total_kp = table_total_kp(p);
        if (m > 0 && (nA <= 0 || nB <= 0)) {  // matches, but nothing valid to index
            if (blockIdx.x == 0 && tid == 0) atomicOr(&w.flags[pi], kSFlagDomain);
            return;
        }
        if (total_kp < 0 || offA + nA > total_kp || offB + nB > total_kp) {  // no code arrays to work from: the general kernel's pair
            if (blockIdx.x == 0 && tid == 0) atomicOr(&w.flags[pi], kSFlagGeneral);
            return;
        }
        if (base >= m) return;  // workgroup-uniform
        const gms_dmatch* __restrict__ matches = p.matches + pr.match_off;
        const uint16_t* __restrict__ lcode = reinterpret_cast<const uint16_t*>(p.pts + total_kp) + offA;
        const uint32_t* __restrict__ scode = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(p.pts + total_kp) + 2 * total_kp) + offB;
        for (int j = tid; j < kFineN; j += 1024) hist[j] = 0;
        if (tid < 64) cnt_l[tid] = 0;
        __syncthreads();
        uint2 qt[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) qt[k] = *reinterpret_cast<const uint2*>(&matches[min(base + k * 1024 + tid, m - 1)]);
        uint32_t ca[4], cb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ca[k] = lcode[min(qt[k].x, (uint32_t)(nA - 1))];
            cb[k] = scode[min(qt[k].y, (uint32_t)(nB - 1))];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = base + k * 1024 + tid;
            const bool live = i < m;
            const uint32_t cell = ca[k] >> 7;  // under grid type 1; 510 = binned under no grid type, 511 = outside the parity domain
            const bool ok = qt[k].x < (uint32_t)nA && qt[k].y < (uint32_t)nB && cell != 511u && (cb[k] >> 31) == 0u;
            const bool binned = live && ok && cell < 510u;
            any_bad |= live && !ok;
            bucket[k] = binned ? 2u * (cell / (uint32_t)kLeftW) + ((ca[k] >> 2) & 1u) : 40u;
            if (live) atomicAdd(&cnt_l[bucket[k]], 1u);
            if (binned) {
                const uint32_t hx = 2u * (cell % (uint32_t)kLeftW) + (ca[k] & 1u), hy = 2u * (cell / (uint32_t)kLeftW) + ((ca[k] >> 2) & 1u);
                atomicAdd(&hist[hy * kFineW + hx], 1u);
            }
            // the dense code word's low byte (q as it stands, the two edge bits one place up, "never" for what is not binned), the cell, the index
            const uint32_t low = binned ? ((ca[k] & 31u) | ((ca[k] & 0x60u) << 1)) : (1u << 5);
            ent[k].x = low | ((binned ? cell : 0u) << kSCellShift) | (((uint32_t)i & 0x7FFFu) << kSOrigShift);
            ent[k].y = (cb[k] & 0x1FFFFFu) | (((uint32_t)i >> 15) << kSOrigHiShift);
            if (live) codes[i] = ent[k];  // (for pass 1, the marking pass and the copy-out: nobody gathers a second time)
        }
        // an index out of range, a point outside the parity domain or outside one of the right grids: the general kernel decides what
        // the reference would make of the pair (a domain error, or -- a right coordinate of exactly 1.0 -- a wrapped cell it counts)
        if (any_bad) atomicOr(&w.flags[pi], kSFlagGeneral);
        __syncthreads();
        uint32_t* nf = w.nfine + (size_t)pi * kFineN;
        for (int j = tid; j < kFineN; j += 1024)
            if (hist[j]) atomicAdd(&nf[j], hist[j]);
        if (tid < kSRowBuckets && cnt_l[tid]) atomicAdd(&rc[tid], cnt_l[tid]);
    } else {
        if (w.flags[pi] & (kSFlagDomain | kSFlagGeneral)) return;  // (written by pass 0)
        if (blockIdx.x == 0) {
            // nLeft of every cell under the four grid types (16 bits: a cell above 65 535 matches flags the pair)
            const uint32_t* __restrict__ nf = w.nfine + (size_t)pi * kFineN;
            bool big = false;
            for (int item = tid; item < 4 * kLeftN; item += 1024) {
                const int g = item / kLeftN, cell = item - g * kLeftN;
                const int hx0 = 2 * (cell % kLeftW) - (g & 1), hy0 = 2 * (cell / kLeftW) - (g >> 1);
                uint32_t part[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) part[d] = nf[max(hy0 + (d >> 1), 0) * kFineW + max(hx0 + (d & 1), 0)];  // (unconditional loads)
                uint32_t n = 0;
#pragma unroll
                for (int d = 0; d < 4; ++d) n += (hx0 + (d & 1) >= 0 && hy0 + (d >> 1) >= 0) ? part[d] : 0u;
                big |= n > 65535u;
                w.nleft[(size_t)pi * 4 * kLeftN + item] = (uint16_t)n;
            }
            if (big) atomicOr(&w.flags[pi], kSFlagGeneral);
        }
        if (base >= m) return;  // workgroup-uniform
#pragma unroll
        for (int k = 0; k < 4; ++k) ent[k] = codes[min(base + k * 1024 + tid, m - 1)];
        if (tid < 64) cnt_l[tid] = 0;
        if (tid < 64) {  // the first entry of every row bucket: the running sum of the counts, over the lanes of wave 0
            const uint32_t mine = tid < kSRowBuckets ? rc[tid] : 0u;
            uint32_t incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d);
                if (tid >= d) incl += up;
            }
            if (tid < kSRowBuckets) {
                row_start[tid] = incl - mine;
                if (blockIdx.x == 0) rc[128 + tid] = incl - mine;  // (for the filter's workgroups: a band of left rows is a range of entries)
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool live = base + k * 1024 + tid < m, binned = live && (ent[k].x & (1u << 5)) == 0u;
            bucket[k] = binned ? 2u * (((ent[k].x >> kSCellShift) & 0x1FFu) / (uint32_t)kLeftW) + ((ent[k].x >> 2) & 1u) : 40u;
            rank[k] = live ? atomicAdd(&cnt_l[bucket[k]], 1u) : 0u;
        }
        __syncthreads();
        // this block's entries of a row go behind whatever other blocks have placed there (order inside a row does not matter: every
        // consumer is a commutative atomic)
        if (tid < kSRowBuckets) base_g[tid] = cnt_l[tid] ? atomicAdd(&rc[64 + tid], cnt_l[tid]) : 0u;
        __syncthreads();
        uint2* ents = w.entries + (size_t)pi * mcap;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (base + k * 1024 + tid < m) ents[row_start[bucket[k]] + base_g[bucket[k]] + rank[k]] = ent[k];
    }
}

// ---- verifyCellPairs for the own cells of a band (dense_scales_pair's: two lanes per cell without rotation, four lanes per cell and
//      two rotations per lane with). The matrix rows are STRIDE bytes apart; a row's header word holds the arg-max key
//      ((count - 1) << 11 | E, E = WR * WR + 3 - right cell) and receives [E << 8 | the rotations that accept the cell pair].
//      POOLED: the row holds a finer grid's counts summed two by two (pool_rows below) -- 16-bit entries, the one of right cell
//      (rx, ry) at byte 4 WR * WR + 2 - 4 WR ry - 2 rx of the row; otherwise bytes, at WR * WR + 3 - (ry WR + rx).
//      WIDE (with ROT): eight lanes per cell, one rotation each, instead of four with two -- for the bands of few own cells, whose
//      verification is a chain of latencies on a handful of waves.
template <bool ROT, uint32_t WR, uint32_t STRIDE, bool POOLED, bool WIDE = false>
__device__ __forceinline__ void verify_cells(const FilterParams& p, uint32_t* smem, const uint16_t* nleft, const uint32_t own0, const uint32_t n_own,
                                             const uint32_t cell0, const bool thr_fast)
{
    constexpr uint32_t wr = WR, nr = WR * WR, stride = STRIDE;
    constexpr uint32_t wr_magic = 65535u / wr + 1u;  // j / wr == (j * magic) >> 16 for j * wr < 65536
    const int tid = threadIdx.x;
    const uint8_t* bytes = reinterpret_cast<const uint8_t*>(smem);
    auto count_at = [&](uint32_t rowb, bool ok, int rx, int ry) -> uint32_t {
        if (POOLED) return reinterpret_cast<const uint16_t*>(bytes)[(rowb + (ok ? 4u * nr + 2u - 4u * wr * (uint32_t)ry - 2u * (uint32_t)rx : 4u)) >> 1];
        return bytes[rowb + (ok ? nr + 3u - (uint32_t)(rx + ry * (int)wr) : 4u)];
    };
    {
        static_assert(ROT || !WIDE, "eight lanes per cell: one per rotation");
        constexpr int kNR = (ROT && !WIDE) ? 2 : 1;
        constexpr int kLanesPerCell = ROT ? (WIDE ? 8 : 4) : 2, kCellShift = ROT ? (WIDE ? 3 : 2) : 1;
        const int n_items = (int)n_own * kLanesPerCell;
        for (int item = tid; item < ((n_items + 63) & ~63); item += 1024) {
            const bool live = item < n_items;
            const int i = (int)own0 + (live ? (item >> kCellShift) : 0);
            const int sub = item & (kLanesPerCell - 1);
            const int half = item & 1;  // !ROT only
            const int ix = i % kLeftW, iy = i / kLeftW;
            const uint32_t ni = live ? (uint32_t)nleft[i] : 0u;
            if (__ballot(ni != 0) == 0ull) continue;  // none of this wave's cells has a match under this grid type
            const uint32_t hdr = ((uint32_t)i - cell0) * (stride >> 2);
            const uint32_t best = smem[hdr];  // ((max count - 1) << 11) | E(j*), lowest j* among maxima
            const uint32_t ej = ni ? (best & 0x7FFu) : nr + 3u;
            const uint32_t j = nr + 3u - ej;
            const int jy = (int)((j * wr_magic) >> 16), jx = (int)j - jy * (int)wr;
            uint32_t score[kNR], tn[kNR], rpack[kNR];  // tn = (sum of nLeft << 4) | numpair
#pragma unroll
            for (int jr = 0; jr < kNR; ++jr) {
                score[jr] = tn[jr] = 0;
                if (WIDE)
                    rpack[jr] = sub == 0 ? rotation_pack(0) : sub == 1 ? rotation_pack(1) : sub == 2 ? rotation_pack(2) : sub == 3 ? rotation_pack(3)
                              : sub == 4 ? rotation_pack(4) : sub == 5 ? rotation_pack(5) : sub == 6 ? rotation_pack(6) : rotation_pack(7);
                else
                    rpack[jr] = sub == 0 ? rotation_pack(jr) : sub == 1 ? rotation_pack(2 + jr) : sub == 2 ? rotation_pack(4 + jr) : rotation_pack(6 + jr);
            }
#pragma unroll
            for (int c = 0; c < (ROT ? 8 : 4); ++c) {
                int ldx, ldy;
                if (ROT) {
                    const int k = c < 4 ? c : c + 1;
                    ldx = (k % 3) - 1; ldy = (k / 3) - 1;
                } else {
                    ldx = half ? ((c + 5) % 3) - 1 : (c % 3) - 1;
                    ldy = half ? ((c + 5) / 3) - 1 : (c / 3) - 1;
                }
                const int lx = ix + ldx, ly = iy + ldy;
                const bool okl = ni != 0 && (uint32_t)lx < (uint32_t)kLeftW && (uint32_t)ly < (uint32_t)kLeftH;
                const uint32_t ll = okl ? (uint32_t)(lx + ly * kLeftW) : (uint32_t)i;  // within one row of an own row: held
                const uint32_t nll = (uint32_t)nleft[ll];
                const uint32_t rowb = (ll - cell0) * stride;
#pragma unroll
                for (int jr = 0; jr < kNR; ++jr) {
                    int rdx = ldx, rdy = ldy;
                    if (ROT) {
                        rdx = (int)((rpack[jr] >> (4 * c)) & 3u) - 1;
                        rdy = (int)((rpack[jr] >> (4 * c + 2)) & 3u) - 1;
                    }
                    const int rx = jx + rdx, ry = jy + rdy;
                    const bool okp = okl && (uint32_t)rx < wr && (uint32_t)ry < wr;
                    const uint32_t cnt = count_at(rowb, okp, rx, ry);
                    score[jr] += okp ? cnt : 0u;
                    tn[jr] += okp ? ((nll << 4) | 1u) : 0u;
                }
            }
            uint32_t vbits = 0;
            if (!ROT) {
                score[0] += dpp_xor1(score[0]);
                tn[0] += dpp_xor1(tn[0]);
            }
#pragma unroll
            for (int jr = 0; jr < kNR; ++jr) {
                const uint32_t sc = score[jr] + (best >> 11) + 1u, t = tn[jr] + ((ni << 4) | 1u);
                uint32_t pass = 0;
                if (ni != 0 && (ROT || half == 0)) pass = threshold_rejects(t >> 4, t & 15u, sc, p.threshold_factor, thr_fast) ? 0u : 1u;
                vbits |= pass << jr;
            }
            if (ROT && WIDE) {  // the cell's eight lanes hold one rotation each: their byte of the wave's ballot
                const unsigned long long bal = __ballot(vbits != 0u);
                vbits = (uint32_t)(bal >> (threadIdx.x & 56u)) & 0xFFu;
            } else if (ROT) {  // the cell's four lanes hold rotations (0,1) (2,3) (4,5) (6,7): gather the quad's bit pairs
                const uint32_t b0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x00, 0xF, 0xF, false);
                const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x55, 0xF, 0xF, false);
                const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0xAA, 0xF, 0xF, false);
                const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0xFF, 0xF, 0xF, false);
                vbits = b0 | (b1 << 2) | (b2 << 4) | (b3 << 6);
            }
            if (ni != 0 && sub == 0) smem[hdr] = (ej << 8) | vbits;  // cellPairs[i] and the rotations that accept it
        }
    }
}

// ---- a finer grid's rows summed two by two: the motion matrix of the 10 x 10 (14 x 14) right grid from the 20 x 20 (28 x 28) one's ----
// A right cell of the coarse grid is four cells of the fine one (right_cell<1>, <2>), the left grid is the same: the coarse matrix is
// the fine one pooled along its rows -- no second binning pass. In place: a row's bytes sit at WF * WF + 3 - (y WF + x), so an
// aligned word holds x = 4 t + 3 ... 4 t of one fine row; the word of row 2 Ry and the one of row 2 Ry + 1 (WF bytes below) add up, byte
// pairs as 16-bit fields, to the counts of coarse cells (Ry, 2 t) [high half] and (Ry, 2 t + 1) [low half], written back over the first
// word: nobody else reads or writes it. Sixteen lanes share a row; its arg-max key (count - 1) << 11 | E goes into the row's header.
template <uint32_t WF>
__device__ __forceinline__ void pool_rows(uint32_t* smem, const uint32_t n_rows)
{
    constexpr uint32_t nrf = WF * WF, stride = 4u + nrf, wc = WF / 2u, nrc = wc * wc, wpr = WF / 4u;  // words per fine row
    static_assert(WF % 4u == 0u && stride % 4u == 0u, "rows are whole words");
    // of the sixteen lanes of a row, 16 / wpr groups of wpr take coarse rows ph, ph + 16 / wpr, ...: lane (ph, t) walks one column of words
    constexpr uint32_t kPhases = 16u / wpr;
    const uint32_t sub = threadIdx.x & 15u, grp = threadIdx.x >> 4, ph = sub / wpr, t = sub - ph * wpr;
    for (uint32_t row = grp; row < ((n_rows + 3u) & ~3u); row += 64u) {  // (a wave's four groups stay together for the DPP steps)
        const uint32_t rowb = row * stride;
        int best = 0;
        if (row < n_rows && ph < kPhases) {
            uint32_t at = (rowb + nrf - 2u * ph * WF - 4u * t) >> 2;
            int e0 = (int)(nrc + 3u - (ph * wc + 2u * t)) - 2048;  // key = (count << 11) + E - 2048 = (count - 1) << 11 | E; negative for a count of zero
#pragma unroll
            for (uint32_t ry = 0; ry < (wc + kPhases - 1u) / kPhases; ++ry) {
                if (wc % kPhases == 0u || ph + ry * kPhases < wc) {
                    const uint32_t a = smem[at], b = smem[at - wpr];
                    const uint32_t sum = (a & 0x00FF00FFu) + ((a >> 8) & 0x00FF00FFu) + (b & 0x00FF00FFu) + ((b >> 8) & 0x00FF00FFu);
                    smem[at] = sum;
                    best = max(best, max((int)((sum >> 16) << 11) + e0, (int)((sum & 0xFFFFu) << 11) + e0 - 1));
                }
                at -= kPhases * (WF / 2u);
                e0 -= (int)(kPhases * wc);
            }
        }
        best = max(best, __builtin_amdgcn_update_dpp(0, best, 0x111, 0xF, 0xF, true));  // row_shr 1, 2, 4, 8: lane 15 of the sixteen ends
        best = max(best, __builtin_amdgcn_update_dpp(0, best, 0x112, 0xF, 0xF, true));  // with their maximum
        best = max(best, __builtin_amdgcn_update_dpp(0, best, 0x114, 0xF, 0xF, true));
        best = max(best, __builtin_amdgcn_update_dpp(0, best, 0x118, 0xF, 0xF, true));
        if (row < n_rows && sub == 15u) smem[rowb >> 2] = (uint32_t)best;
    }
}

// ---- one scale hypothesis of one pair --------------------------------------------------------------------------------------------------
// POOL (scales 0 and 3 of a launch with scale hypotheses): the item goes on to the coarser grid whose cells are this one's two by two
// -- scale 1 after 0, scale 2 after 3 -- on the pooled rows, and leaves that scale's table as well.
template <bool ROT, int S, bool POOL>
__device__ __forceinline__ void stream_scale(const FilterParams& p, const StreamWs& w, uint32_t* smem, const int mcap, const int n_scales, const int pi,
                                             const int g, const int band)
{
    constexpr int kChunk = 8;  // entries per thread and round; two rounds in flight
    GMS_SSTAMP_DECL
    const int tid = threadIdx.x;
    constexpr uint32_t wr = S == 0 ? 20u : S == 1 ? 10u : S == 2 ? 14u : S == 3 ? 28u : 40u, nr = wr * wr, stride = 4u + nr;  // (checked by the launcher)
    constexpr int band_rows = S == 4 ? 3 : S == 3 ? 8 : kLeftH;
    const int gy_top = g >> 1, lo_top = band * band_rows, hi_top = min(lo_top + band_rows, kLeftH);
    const int hlo_top = max(lo_top - 1, 0), hhi_top = min(hi_top + 1, kLeftH);
    uint16_t* nleft = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(smem) + kSNleftOff);
    uint32_t* misc = smem + kSMiscOff / 4;
    // Everything the item needs before its entries, requested TOGETHER (one round trip, not three): the pair's match count, the two
    // ends of the band's range of entries (written by stream_index_kernel<1>), nLeft of this grid type, and the pair's flag word --
    // the index kernels write it before the launch, but the pair's OTHER workgroups of this launch may OR kSFlagGeneral into it at any
    // time (below); waves reading it one by one could disagree and a part of the workgroup would run the barriers alone, so one thread
    // reads it into scratch word [15] and everybody takes that value behind the first barrier.
    const uint32_t* __restrict__ first_of_half = w.row_cnt + (size_t)pi * kSRowWords + 128;
    uint32_t m_word = reinterpret_cast<const uint32_t*>(p.pairs + pi)[2];
    uint32_t lo_word = first_of_half[max(2 * hlo_top - gy_top, 0)], hi_word = first_of_half[2 * hhi_top - gy_top];
    if (tid == 0) misc[15] = __hip_atomic_load(&w.flags[pi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 15) misc[tid] = 0;
    if (tid >= 64 && tid < 64 + kLeftN / 2)
        reinterpret_cast<uint32_t*>(nleft)[tid - 64] = reinterpret_cast<const uint32_t*>(w.nleft + ((size_t)pi * 4 + g) * kLeftN)[tid - 64];
    asm volatile("" : "+v"(m_word), "+v"(lo_word), "+v"(hi_word));
    const int m = uniform((int)m_word);
    if (m <= 0 || m > mcap) return;
    const uint2* __restrict__ ents = w.entries + (size_t)pi * mcap;
    // the table of this (scale, grid type): a cell is written by the one band that owns its left row
    uint32_t* tab = w.tables + (((size_t)pi * n_scales + (n_scales == 5 ? S : 0)) * 4 + (size_t)g) * kLeftN;

    const bool thr_fast = threshold_fast_ok(p.threshold_factor);
    static_assert((kLeftH + band_rows - 1) / band_rows == (S == 4 ? 7 : S == 3 ? 3 : 1), "bands per scale as the item table assumes");

    // The entries at positions [p_lo, p_hi): a thread takes p_lo rounded down to a multiple of 1024, plus tid, plus 1024 k, kChunk
    // entries per round. The loads are UNCONDITIONAL (a position outside the range reads the pair's last entry and is turned into
    // "binned under no grid type" afterwards): a load under a condition is a branch with its own wait, and sixteen of those are sixteen
    // round trips one after the other.
    auto load_chunk = [&](uint2 (&e)[kChunk], uint32_t pos0) {
#pragma unroll
        for (int j = 0; j < kChunk; ++j) e[j] = ents[min(pos0 + (uint32_t)j * 1024u, (uint32_t)m - 1u)];
    };
    auto clip_chunk = [&](uint2 (&e)[kChunk], uint32_t pos0, uint32_t p_lo, uint32_t p_hi) {
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            const uint32_t pos = pos0 + (uint32_t)j * 1024u;
            if (!(pos >= p_lo && pos < p_hi)) e[j].x = 1u << 5;
        }
    };

    {
        const int gx = g & 1, gy = g >> 1;
        const uint32_t q_mask = (uint32_t)(gx + 20 * gy);                               // l = l1 + (q & q_mask)
        const uint32_t out_mask = (1u << 5) | (gx ? 1u << 6 : 0u) | (gy ? 1u << 7 : 0u);  // never | x >= 20 | y >= 20 under this grid type
        {
            const int lo = band * band_rows, hi = min(lo + band_rows, kLeftH);          // own rows
            const int hlo = max(lo - 1, 0), hhi = min(hi + 1, kLeftH);                  // rows held (own + halo)
            const uint32_t cell0 = (uint32_t)(hlo * kLeftW), n_held = (uint32_t)((hhi - hlo) * kLeftW);
            const uint32_t own0 = (uint32_t)(lo * kLeftW), n_own = (uint32_t)((hi - lo) * kLeftW);
            // the entries of the rows held: half rows 2 hlo .. 2 hhi - 1, one half row down under the y-shifted grid types (whose row R
            // is made of the odd half row 2 R - 1 and the even one 2 R); the first round of them is on its way while the rows are cleared
            const uint32_t p_lo = (uint32_t)uniform((int)lo_word), p_hi = (uint32_t)uniform((int)hi_word);  // first_of_half[max(2 hlo - gy, 0)], [2 hhi - gy]
            uint32_t pos0 = (p_lo & ~1023u) + (uint32_t)tid;
            uint2 ea[kChunk], eb[kChunk];
            load_chunk(ea, pos0);
            {   // motion.setTo(0) for the rows held, headers included
                const uint4 z4 = make_uint4(0, 0, 0, 0);
                uint4* d4 = reinterpret_cast<uint4*>(smem);
                GMS_SSTAMP(0);   // start: pair, flags, row counts
                for (uint32_t i = tid; i < (n_held * stride + 15u) / 16u; i += 1024) d4[i] = z4;
            }
            __syncthreads();
            if (misc[15] & (kSFlagDomain | kSFlagGeneral)) return;  // (workgroup-uniform: one word, read behind the barrier)
            GMS_SSTAMP(1);   // clear + barrier
            // ---- assignMatchPairs for the rows held: +1 on the entry's byte; the count it produced goes into the row's running
            //      arg-max. The next round's entries are requested before this round's atomics.
            {
                auto bin_chunk = [&](const uint2 (&e)[kChunk]) {
                    uint32_t old[kChunk], at[kChunk], hdr[kChunk], key[kChunk];
#pragma unroll
                    for (int j = 0; j < kChunk; ++j) {
                        const uint32_t cw = e[j].x;
                        const uint32_t l = ((cw >> kSCellShift) & 0x1FFu) + (cw & q_mask) - cell0;
                        const bool in = (cw & out_mask) == 0u && l < n_held;
                        const uint32_t er = nr + 3u - right_cell<S>(e[j].y);  // the byte's offset in its row
                        const uint32_t row = __umul24(l, stride);
                        at[j] = row + er;
                        hdr[j] = in ? row : 0xFFFFFFFFu;
                        key[j] = er;
                        old[j] = 0u;
                        if (in) old[j] = atomicAdd(lds_at(smem, at[j] & ~3u), 1u << ((at[j] << 3) & 31u));
                    }
                    __builtin_amdgcn_sched_barrier(0);  // all of the round's atomics are issued before any result is read
#pragma unroll
                    for (int c = 0; c < kChunk; ++c) {
                        const uint32_t before = (old[c] >> ((at[c] << 3) & 31u)) & 255u;
                        if (hdr[c] != 0xFFFFFFFFu) {
                            if (before == 255u) misc[8] = 1u;  // the entry's byte has just wrapped
                            atomicMax(lds_at(smem, hdr[c]), (before << 11) | key[c]);  // highest count, then lowest right cell
                        }
                    }
                };
                while (pos0 < p_hi) {  // (per thread: no barrier inside)
                    const uint32_t pos1 = pos0 + kChunk * 1024u, pos2 = pos1 + kChunk * 1024u;
                    if (pos1 < p_hi) load_chunk(eb, pos1);
                    clip_chunk(ea, pos0, p_lo, p_hi);
                    bin_chunk(ea);
                    if (pos1 >= p_hi) break;
                    if (pos2 < p_hi) load_chunk(ea, pos2);
                    clip_chunk(eb, pos1, p_lo, p_hi);
                    bin_chunk(eb);
                    pos0 = pos2;
                }
            }
            __syncthreads();
            GMS_SSTAMP(2);   // bin + barrier
            if (misc[8] != 0u) {  // a (left cell, right cell) pair above 255 matches (workgroup-uniform): the general kernel's pair
                if (tid == 0) {
                    if (!(atomicOr(&w.flags[pi], kSFlagGeneral) & kSFlagGeneral) && p.overflow_events) atomicAdd(p.overflow_events, 1u);
                }
                return;
            }
            verify_cells<ROT, wr, stride, false, ROT && S == 4>(p, smem, nleft, own0, n_own, cell0, thr_fast);
            __syncthreads();
            GMS_SSTAMP(3);   // verify + barrier
            // ---- what the marking loop needs of this band: per own cell E(cellPairs[cell]) and the rotations that accept the pair
            //      (a cell without matches under this grid type still holds the zero of the clear: accepts nothing)
            for (uint32_t c = (uint32_t)tid; c < n_own; c += 1024u) tab[own0 + c] = smem[(own0 + c - cell0) * (stride >> 2)];
            GMS_SSTAMP(4);   // table
            if constexpr (POOL) {
                static_assert(S == 0 || S == 3, "the grids with a coarser one of half the width");
                constexpr int kCoarse = S == 0 ? 1 : 2;
                __syncthreads();  // (the headers have been copied out)
                pool_rows<wr>(smem, n_held);
                __syncthreads();
                GMS_SSTAMP(5);   // pooling + barriers
                verify_cells<ROT, wr / 2u, stride, true>(p, smem, nleft, own0, n_own, cell0, thr_fast);
                __syncthreads();
                uint32_t* tab2 = w.tables + (((size_t)pi * n_scales + kCoarse) * 4 + (size_t)g) * kLeftN;
                for (uint32_t c = (uint32_t)tid; c < n_own; c += 1024u) tab2[own0 + c] = smem[(own0 + c - cell0) * (stride >> 2)];
                GMS_SSTAMP(6);   // the coarser scale's verify + table
            }
            GMS_SSTAMP_FLUSH;
        }
    }
}

template <bool ROT>
__global__ void __launch_bounds__(1024)
stream_filter_kernel(FilterParams p, StreamWs w, int mcap, int n_scales)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // One workgroup per (pair, scale, grid type, band of left rows): all independent (each clears its rows and writes its own cells'
    // verdicts into its (scale, grid type)'s table). Items of a pair, longest first: 20 x 20 (four grid types; each goes on to the
    // 10 x 10 grid on the pooled rows), 28 x 28 (three bands x four; each goes on to 14 x 14), 40 x 40 (seven x four): 44 with scale
    // hypotheses, 4 without.
    // Workgroups are handed to the eight XCDs round-robin; the items of a pair all stream the pair's entry array, so they are dealt to
    // ONE XCD (its L2 then serves every re-read): workgroup L = 8 slot + xcd works on pair 8 (slot / items) + xcd, item slot % items.
    const int items = n_scales == 5 ? kSItemsScales : 4;
    const int L = (int)blockIdx.x, n8 = p.n_pairs & ~7;
    int pi, item;
    if (L < n8 * items) {
        const int slot = L >> 3;
        pi = (slot / items) * 8 + (L & 7);
        item = slot % items;
    } else {  // the last n_pairs % 8 pairs
        pi = n8 + (L - n8 * items) / items;
        item = (L - n8 * items) % items;
    }
    if (n_scales != 5) {
        stream_scale<ROT, 0, false>(p, w, smem, mcap, n_scales, pi, item & 3, 0);
    } else if (item < 4) {
        stream_scale<ROT, 0, true>(p, w, smem, mcap, n_scales, pi, item, 0);
    } else if (item < 16) {
        stream_scale<ROT, 3, true>(p, w, smem, mcap, n_scales, pi, (item - 4) & 3, (item - 4) >> 2);
    } else {
        stream_scale<ROT, 4, false>(p, w, smem, mcap, n_scales, pi, (item - 16) & 3, (item - 16) >> 2);
    }
}

// ---- run()'s marking loop for every scale and grid type in one pass over the entries -----------------------------------------------
namespace {
// the left cell of an entry under the four grid types (kLeftN = "under this grid type the match is outside the left grid": its
// table slot holds zero)
__device__ __forceinline__ void left_cells(uint32_t cw, uint32_t (&lg)[4])
{
    const uint32_t l1 = (cw >> kSCellShift) & 0x1FFu;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t q_mask = (uint32_t)((g & 1) + 20 * (g >> 1));
        const uint32_t out_mask = (1u << 5) | ((g & 1) ? 1u << 6 : 0u) | ((g >> 1) ? 1u << 7 : 0u);
        lg[g] = (cw & out_mask) == 0u ? l1 + (cw & q_mask) : (uint32_t)kLeftN;
    }
}

// the rotations under which a match is an inlier of scale hypothesis S: cellPairs[l] == r under some grid type, and that cell pair
// verified (tab = the scale's four tables, kSTabStride words apart)
constexpr uint32_t kSTabStride = kLeftN + 1;  // (+ the always-zero slot)
template <int S>
__device__ __forceinline__ uint32_t inlier_rotations(const uint32_t* tab, uint32_t aux, const uint32_t (&lg)[4])
{
    constexpr uint32_t wr = S == 0 ? 20u : S == 1 ? 10u : S == 2 ? 14u : S == 3 ? 28u : 40u, nr = wr * wr;
    const uint32_t e8 = (nr + 3u - right_cell<S>(aux)) << 8;
    uint32_t rot = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t x = tab[g * kSTabStride + lg[g]] ^ e8;  // < 256: the same right cell, x = the rotations that accept
        rot |= min(x, 256u) & 255u;  // (x itself below 256, else nothing: two instructions, no compare + select)
    }
    return rot;
}

__device__ __forceinline__ void load_tables(uint32_t* tab, const uint32_t* src, int n_tables, int tid)
{
    for (int i = tid; i < n_tables * (int)kSTabStride; i += 1024) {
        const int t = i / (int)kSTabStride, c = i - t * (int)kSTabStride;
        tab[i] = c < kLeftN ? src[t * kLeftN + c] : 0u;
    }
}
}  // namespace

template <bool ROT>
__global__ void __launch_bounds__(1024)
stream_mark_kernel(FilterParams p, StreamWs w, int mcap, int n_scales)
{
    __shared__ uint32_t tab[5 * 4 * kSTabStride];
    __shared__ uint32_t cnt[5 * 4];  // per scale: rotations (0, 2) (1, 3) (4, 6) (5, 7) as two 16-bit fields each
    const int tile = blockIdx.x, pi = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int m = p.pairs[pi].m;
    if (m <= 0 || m > mcap || tile * kSMarkTile >= m || (w.flags[pi] & (kSFlagDomain | kSFlagGeneral))) return;
    const uint2* __restrict__ codes = w.codes + (size_t)pi * mcap;
    constexpr int kPer = kSMarkTile / 1024;
    uint2 e[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int i = tile * kSMarkTile + k * 1024 + tid;
        e[k] = codes[min(i, m - 1)];  // (unconditional loads: all of them in flight together, under the tables' loads)
        if (i >= m) e[k].x = 1u << 5;  // an entry binned under no grid type
    }
    load_tables(tab, w.tables + (size_t)pi * n_scales * 4 * kLeftN, n_scales * 4, tid);
    if (tid < 20) cnt[tid] = 0;
    __syncthreads();
    uint32_t acc_lo[5] = {0u, 0u, 0u, 0u, 0u}, acc_hi[5] = {0u, 0u, 0u, 0u, 0u};  // four byte fields each: a thread adds at most eight
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        uint32_t lg[4];
        left_cells(e[k].x, lg);
        auto one = [&](auto sc, int slot) {
            constexpr int S = decltype(sc)::value;
            const uint32_t rot = inlier_rotations<S>(tab + slot * 4 * kSTabStride, e[k].y, lg);
            acc_lo[S] += __umul24(rot & 15u, 0x204081u) & 0x01010101u;  // bit i of the nibble to bit 8 i
            if (ROT) acc_hi[S] += __umul24(rot >> 4, 0x204081u) & 0x01010101u;
        };
        one(std::integral_constant<int, 0>{}, 0);
        if (n_scales == 5) {
            one(std::integral_constant<int, 1>{}, 1);
            one(std::integral_constant<int, 2>{}, 2);
            one(std::integral_constant<int, 3>{}, 3);
            one(std::integral_constant<int, 4>{}, 4);
        }
    }
    for (int s = 0; s < n_scales; ++s) {
        const uint32_t lo = s == 0 ? acc_lo[0] : s == 1 ? acc_lo[1] : s == 2 ? acc_lo[2] : s == 3 ? acc_lo[3] : acc_lo[4];
        const uint32_t hi = s == 0 ? acc_hi[0] : s == 1 ? acc_hi[1] : s == 2 ? acc_hi[2] : s == 3 ? acc_hi[3] : acc_hi[4];
        const uint32_t f0 = wave_sum(lo & 0x00FF00FFu), f1 = wave_sum((lo >> 8) & 0x00FF00FFu);
        if (lane == 0 && f0) atomicAdd(&cnt[s * 4 + 0], f0);
        if (lane == 0 && f1) atomicAdd(&cnt[s * 4 + 1], f1);
        if (ROT) {
            const uint32_t f2 = wave_sum(hi & 0x00FF00FFu), f3 = wave_sum((hi >> 8) & 0x00FF00FFu);
            if (lane == 0 && f2) atomicAdd(&cnt[s * 4 + 2], f2);
            if (lane == 0 && f3) atomicAdd(&cnt[s * 4 + 3], f3);
        }
    }
    __syncthreads();
    if (tid < n_scales * 8) {
        const int s = tid >> 3, r = tid & 7;
        const uint32_t c = (cnt[s * 4 + (r >> 2) * 2 + (r & 1)] >> ((r & 2) << 3)) & 0xFFFFu;
        w.tile_cnt[(((size_t)pi * kSTilesMax + tile) * 5 + s) * 8 + r] = c;  // (every tile in front of m writes its forty words: no clearing)
        if (c) atomicAdd(&w.counts[((size_t)pi * 5 + s) * 8 + r], c);
    }
}

// ---- getInlierMask over the scales' counts, the winner's bit of every match of a tile, the survivors copied out in order -------------
template <bool ROT>
__global__ void __launch_bounds__(1024)
stream_compact_kernel(FilterParams p, StreamWs w, int mcap, int n_scales)
{
    __shared__ uint32_t tab[4 * kSTabStride];
    __shared__ uint32_t wave_tile[16];
    const int tile = blockIdx.x, pi = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const gms_pair pr = p.pairs[pi];
    const int m = pr.m;
    const uint32_t fl = w.flags[pi];
    if (fl & kSFlagGeneral) return;  // gms_kernel_big.hip produces this pair
    const bool failed = (fl & kSFlagDomain) != 0 || m < 0 || m > mcap;
    const int n_tiles = (failed || m <= 0) ? 1 : (m + kSMarkTile - 1) / kSMarkTile;
    if (tile >= n_tiles) return;
    if (failed || m <= 0) {  // nothing kept (a pair outside the reference's domain fails as a whole)
        if (failed && m > 0 && m <= mcap && p.mask) {
            uint8_t* mask = p.mask + pr.match_off;
            for (int i = tid; i < m; i += 1024) mask[i] = 0;
        }
        if (tid == 0) {
            gms_pair_result r;
            r.n_inliers = 0;
            r.best_scale = r.best_rot = -1;
            r.status = failed ? GMS_ERR_DOMAIN : GMS_OK;
            p.results[pi] = r;
        }
        return;
    }
    constexpr int kPer = kSMarkTile / 1024;
    const uint2* __restrict__ codes = w.codes + (size_t)pi * mcap;
    const gms_dmatch* __restrict__ matches = p.matches + pr.match_off;
    gms_dmatch* __restrict__ out = p.out + pr.match_off;
    // a wave owns kPer * 64 consecutive matches of the tile, 64 at a time: loads and stores are coalesced
    const int wbase = tile * kSMarkTile + wave * (kPer * 64);
    uint2 e[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) e[k] = codes[min(wbase + k * 64 + lane, m - 1)];
    // scale outer, rotation inner, strict '>' from 0 (DLL@0x180047dc0): the first hypothesis with the largest count. Lane h of every
    // wave holds hypothesis h = 8 scale + rotation; the largest (count, -h) wins.
    constexpr int kNRot = ROT ? 8 : 1;
    uint32_t key = 0;
    if (lane < 40 && (lane >> 3) < n_scales && (lane & 7) < kNRot) {
        const uint32_t c = w.counts[(size_t)pi * 40 + lane];
        key = c ? (c << 6) | (uint32_t)(63 - lane) : 0u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) key = max(key, (uint32_t)__shfl_xor((int)key, d));
    const uint32_t best = key >> 6;
    const int bh = key ? 63 - (int)(key & 63u) : -1, bs = bh < 0 ? -1 : bh >> 3, br = bh < 0 ? -1 : bh & 7;
    // survivors in front of this tile: the winner's counts of the tiles before it
    uint32_t before = 0;
    if (bh >= 0) {
        const uint32_t mine = lane < tile ? w.tile_cnt[((size_t)pi * kSTilesMax + lane) * 40 + bh] : 0u;
        before = wave_sum(mine);
        load_tables(tab, w.tables + ((size_t)pi * n_scales + bs) * 4 * kLeftN, 4, tid);
    }
    __syncthreads();
    unsigned long long bal[kPer];
    uint32_t wave_count = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int i = wbase + k * 64 + lane;
        uint32_t lg[4];
        left_cells(e[k].x, lg);
        uint32_t rot = 0;
        switch (bs) {  // (workgroup-uniform)
            case 0: rot = inlier_rotations<0>(tab, e[k].y, lg); break;
            case 1: rot = inlier_rotations<1>(tab, e[k].y, lg); break;
            case 2: rot = inlier_rotations<2>(tab, e[k].y, lg); break;
            case 3: rot = inlier_rotations<3>(tab, e[k].y, lg); break;
            case 4: rot = inlier_rotations<4>(tab, e[k].y, lg); break;
            default: break;
        }
        const bool keep = i < m && bs >= 0 && ((rot >> br) & 1u) != 0u;
        if (p.mask && i < m) p.mask[pr.match_off + i] = keep ? 1 : 0;
        bal[k] = __ballot(keep);
        wave_count += (uint32_t)__popcll(bal[k]);
    }
    if (lane == 0) wave_tile[wave] = wave_count;
    __syncthreads();
    uint32_t pos = before;
    for (int wv = 0; wv < wave; ++wv) pos += wave_tile[wv];
    const unsigned long long lt = (1ull << lane) - 1ull;
    // the wave's records, all requested before the first store (a load that only a conditional store uses is sunk into the branch by
    // the compiler and waited for there, one round trip per record)
    uint4 rec[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) rec[k] = *reinterpret_cast<const uint4*>(&matches[min(wbase + k * 64 + lane, m - 1)]);
#pragma unroll
    for (int k = 0; k < kPer; ++k) asm volatile("" : "+v"(rec[k].x), "+v"(rec[k].y), "+v"(rec[k].z), "+v"(rec[k].w));
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        if ((bal[k] >> lane) & 1ull) *reinterpret_cast<uint4*>(&out[pos + (uint32_t)__popcll(bal[k] & lt)]) = rec[k];
        pos += (uint32_t)__popcll(bal[k]);
    }
    if (tid == 0 && tile == n_tiles - 1) {
        gms_pair_result r;
        r.n_inliers = (int)best;
        r.best_scale = best == 0 ? -1 : bs;
        r.best_rot = best == 0 ? -1 : br + 1;
        r.status = GMS_OK;
        p.results[pi] = r;
    }
}

// ================================================================================================================================
// No scale hypotheses (the default flags of DisparityUtil.cpp:149,299, or rotation alone): ONE workgroup per pair, dense_pair() of
// gms_kernels.hip with the per-match code words in an L2-resident scratch array instead of registers -- the 400 x 400 byte matrix
// fills the LDS, zeroed once per pair (every match takes its own increment back after each grid type), the row headers carry the
// running arg-max under a grid-type tag. What is different at this size:
//   * nLeft is 16 bits per cell, from a 32-bit half-cell histogram that lives in the matrix area before the matrix does;
//   * an entry that would pass 255 (the increment that wraps its byte notices) or a cell above 65 535 matches sends the pair to the
//     HBM-slab kernel before anything is written out, and is reported (p.overflow_events): a context that keeps meeting such pairs
//     goes back to the 16-bit band kernels for a while (gms_capi.cpp);
//   * a match's rotation bits ride in its code word (read-modify-write by the one thread that owns index i = tid + 1024 k);
//   * copy-out: three sweeps over the code words (counts per rotation, survivors per 64-match chunk + scan, records).
// code word: [qx | qy | never | edgeX | edgeY | E(r) : 9 | left cell under grid type 1 : 9 | inlier-under-rotation bits : 8]
// ================================================================================================================================
namespace {
constexpr int kDEShiftS = 5, kDCellShiftS = 14, kDAccShiftS = 23;
constexpr uint32_t kDRow = 4u + 400u;                       // header dword + one byte per right cell (offset E(r) = 403 - r)
constexpr uint32_t kDSNleftOff = kLeftN * kDRow;            // 161 600: [400] u16
constexpr uint32_t kDSMiscOff = kDSNleftOff + 2u * kLeftN;  // [32] u32: [0..7] rotation counts, [8] bad input, [9] entry / cell too big, [16..31] wave totals
constexpr uint32_t kDSLdsBytes = kDSMiscOff + 128u;         // 162 528
static_assert(kDSLdsBytes <= kLdsBytes, "stream-dense layout exceeds the LDS");
constexpr int kKeyCountShift = 11, kKeyTagShift = 27;       // row header while binning: grid type << 27 | (count - 1) << 11 | E
}  // namespace

template <bool ROT>
__global__ void __launch_bounds__(1024)
stream_dense_kernel(FilterParams p, uint32_t* __restrict__ codes_ws, uint16_t* __restrict__ nleft_ws, uint32_t* __restrict__ flags, int mcap)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int kNRot = ROT ? 8 : 1;
    constexpr int kC = 8;  // code words a thread has in flight (16: no faster)
    const int pi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const gms_pair pr = p.pairs[pi];
    const int m = pr.m;
    uint8_t* bytes = reinterpret_cast<uint8_t*>(smem);
    uint16_t* nleft = reinterpret_cast<uint16_t*>(bytes + kDSNleftOff);
    uint32_t* misc = smem + kDSMiscOff / 4;
    const int64_t total_kp = table_total_kp(p);
    bool general = m <= 0 || m > mcap || pr.frame_a < 0 || pr.frame_a >= p.n_frames || pr.frame_b < 0 || pr.frame_b >= p.n_frames || total_kp < 0;
    int64_t offA = 0, offB = 0;
    int nA = 0, nB = 0;
    if (!general) {
        offA = p.frame_off[pr.frame_a];
        offB = p.frame_off[pr.frame_b];
        nA = (int)(p.frame_off[pr.frame_a + 1] - offA);
        nB = (int)(p.frame_off[pr.frame_b + 1] - offB);
        general = nA <= 0 || nB <= 0 || offA + nA > total_kp || offB + nB > total_kp;
    }
    if (general) {  // (workgroup-uniform) nothing this kernel can take: the general kernel decides what the pair is
        if (tid == 0) atomicOr(&flags[pi], kSFlagGeneral);
        return;
    }
    const gms_dmatch* __restrict__ matches = p.matches + pr.match_off;
    const uint16_t* __restrict__ lcode = reinterpret_cast<const uint16_t*>(p.pts + total_kp) + offA;
    const uint16_t* __restrict__ rcode = reinterpret_cast<const uint16_t*>(p.pts + total_kp) + total_kp + offB;
    uint32_t* __restrict__ codes = codes_ws + (size_t)pi * mcap;
    uint16_t* __restrict__ nl_g = nleft_ws + (size_t)pi * 4 * kLeftN;
    const int kpt = (m + 1023) >> 10;

    // ---- the code words and the half-cell histogram (u32, one dword per half cell: [cell][qx + 2 qy], in the still unused matrix area)
    uint32_t* hist = smem;
    for (int j = tid; j < kFineN; j += 1024) hist[j] = 0;
    if (tid < 32) misc[tid] = 0;
    __syncthreads();
    {
        bool any_bad = false;
        for (int k0 = 0; k0 < kpt; k0 += kC) {
            uint2 qt[kC];
#pragma unroll
            for (int j = 0; j < kC; ++j) qt[j] = *reinterpret_cast<const uint2*>(&matches[min((k0 + j) * 1024 + tid, m - 1)]);
            uint32_t ca[kC], cb[kC];
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                ca[j] = lcode[min(qt[j].x, (uint32_t)(nA - 1))];
                cb[j] = rcode[min(qt[j].y, (uint32_t)(nB - 1))];
            }
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                const int i = (k0 + j) * 1024 + tid;
                const bool live = i < m;
                const uint32_t cell = ca[j] >> 7, e0 = cb[j] & 0x1FFu;  // cell: 510 = binned under no grid type, 511 = outside the parity domain
                const bool ok = qt[j].x < (uint32_t)nA && qt[j].y < (uint32_t)nB && cell != 511u && (cb[j] >> 15) == 0u && e0 != 0u;
                const bool binned = live && ok && cell < 510u;
                any_bad |= live && !ok;
                const uint32_t qx = ca[j] & 1u, qy = (ca[j] >> 2) & 1u;
                if (binned) atomicAdd(&hist[cell * 4u + qx + 2u * qy], 1u);
                const uint32_t cw = binned ? (qx | (qy << 1) | (((ca[j] >> 5) & 3u) << 3) | (e0 << kDEShiftS) | (cell << kDCellShiftS)) : (1u << 2);
                if (live) codes[i] = cw;
            }
        }
        if (any_bad) misc[8] = 1;
    }
    __syncthreads();
    if (misc[8] != 0) {  // an index out of range, a point outside the parity domain or outside the right grid: the general kernel's pair
        if (tid == 0) atomicOr(&flags[pi], kSFlagGeneral);
        return;
    }
    for (int item = tid; item < 4 * kLeftN; item += 1024) {
        const int g = item / kLeftN, cell = item - g * kLeftN;
        const int hx0 = 2 * (cell % kLeftW) - (g & 1), hy0 = 2 * (cell / kLeftW) - (g >> 1);
        uint32_t n = 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int hx = hx0 + dx, hy = hy0 + dy;
                if (hx >= 0 && hy >= 0) n += hist[(((hy >> 1) * kLeftW + (hx >> 1)) << 2) + (hx & 1) + ((hy & 1) << 1)];
            }
        if (n > 65535u) misc[9] = 1;
        nl_g[item] = (uint16_t)n;
    }
    __syncthreads();
    if (misc[9] != 0) {  // a cell above 65 535 matches
        if (tid == 0) {
            atomicOr(&flags[pi], kSFlagGeneral);
            if (p.overflow_events) atomicAdd(p.overflow_events, 1u);
        }
        return;
    }
    {   // motion.setTo(0), once: every grid type leaves the matrix as it found it
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        uint4* d4 = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < kDSNleftOff / 16u; i += 1024) d4[i] = z4;
    }
    const bool thr_fast = threshold_fast_ok(p.threshold_factor);

    for (int g = 0; g < 4; ++g) {
        const uint32_t gx = (uint32_t)(g & 1), gy = (uint32_t)(g >> 1);
        const uint32_t out_mask = (1u << 2) | (gx << 3) | (gy << 4);  // never | x >= 20 | y >= 20 under this grid type (DLL@0x180047d3d)
        const uint32_t tag = (uint32_t)g << kKeyTagShift;
        if (tid < kLeftN / 2) reinterpret_cast<uint32_t*>(nleft)[tid] = reinterpret_cast<const uint32_t*>(nl_g + g * kLeftN)[tid];
        __syncthreads();  // (the matrix clear / the previous grid type's undo and side-table reset are complete)
        // ---- assignMatchPairs
        for (int k0 = 0; k0 < kpt; k0 += kC) {
            uint32_t cw[kC];
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                const int i = (k0 + j) * 1024 + tid;
                cw[j] = i < m ? codes[i] : (1u << 2);
            }
            uint32_t old[kC], at[kC];
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                const uint32_t l = ((cw[j] >> kDCellShiftS) & 0x1FFu) + (cw[j] & gx) + 20u * ((cw[j] >> 1) & gy);
                at[j] = __umul24(l, kDRow) + ((cw[j] >> kDEShiftS) & 0x1FFu);
                old[j] = 0;
                if ((cw[j] & out_mask) == 0u) old[j] = atomicAdd(lds_at(smem, at[j] & ~3u), 1u << ((at[j] << 3) & 31u));
            }
            __builtin_amdgcn_sched_barrier(0);  // all of the chunk's atomics are issued before any result is read
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                if ((cw[j] & out_mask) != 0u) continue;
                const uint32_t e = (cw[j] >> kDEShiftS) & 0x1FFu, row = at[j] - e;
                const uint32_t before = (old[j] >> ((at[j] << 3) & 31u)) & 255u;
                if (before == 255u) misc[9] = 1;  // the entry's byte has just wrapped: more than 255 matches in one (left cell, right cell) pair
                atomicMax(lds_at(smem, row), tag | (before << kKeyCountShift) | e);  // highest count, then lowest right cell
            }
        }
        __syncthreads();
        if (misc[9] != 0) {  // (workgroup-uniform) the general kernel's pair; nothing has been written out
            if (tid == 0) {
                atomicOr(&flags[pi], kSFlagGeneral);
                if (p.overflow_events) atomicAdd(p.overflow_events, 1u);
            }
            return;
        }
        // ---- verifyCellPairs (dense_pair's: two lanes per cell without rotation, one lane per (cell, rotation) with)
        {
            constexpr int kItems = ROT ? kLeftN * 8 : kLeftN * 2;
            for (int item = tid; item < ((kItems + 63) & ~63); item += 1024) {
                const bool live = item < kItems;
                const int i = live ? (ROT ? (item >> 3) : (item >> 1)) : 0;
                const int rot = ROT ? (item & 7) : 0;
                const int half = item & 1;  // !ROT only
                const int ix = i % kLeftW, iy = i / kLeftW;
                const uint32_t ni = live ? (uint32_t)nleft[i] : 0u;
                if (__ballot(ni != 0) == 0ull) continue;  // none of this wave's cells has a match under this grid type
                const uint32_t best = smem[i * (kDRow / 4)] & ((1u << kKeyTagShift) - 1u);  // ((max count - 1) << 11) | E(j*), lowest j* among maxima
                const uint32_t ej = ni ? (best & 0x7FFu) : 403u;
                const int j = 403 - (int)ej;
                const int jx = j % 20, jy = j / 20;
                uint32_t score = 0, tsum = 0, np = 0;
#pragma unroll
                for (int c = 0; c < (ROT ? 8 : 4); ++c) {
                    int ldx, ldy, rdx, rdy;
                    if (ROT) {
                        const int k = c < 4 ? c : c + 1;
                        constexpr int kRingIndex[9] = {0, 1, 2, 7, -1, 3, 6, 5, 4};  // position -> ring index
                        const int q = rotated_position(rot, kRingIndex[k]);
                        ldx = (k % 3) - 1; ldy = (k / 3) - 1;
                        rdx = position_dx(q); rdy = position_dy(q);
                    } else {
                        ldx = half ? ((c + 5) % 3) - 1 : (c % 3) - 1;  // lane 0: neighbours 0..3, lane 1: 5..8
                        ldy = half ? ((c + 5) / 3) - 1 : (c / 3) - 1;
                        rdx = ldx; rdy = ldy;
                    }
                    const int lx = ix + ldx, ly = iy + ldy, rx = jx + rdx, ry = jy + rdy;
                    const bool okl = ni != 0 && (uint32_t)lx < (uint32_t)kLeftW && (uint32_t)ly < (uint32_t)kLeftH;  // ll != -1
                    const bool okp = okl && (uint32_t)rx < 20u && (uint32_t)ry < 20u;                                 // rr != -1
                    const uint32_t ll = okl ? (uint32_t)(lx + ly * kLeftW) : 0u;
                    const uint32_t at = ll * kDRow + (okp ? (uint32_t)(403 - (rx + ry * 20)) : 4u);
                    const uint32_t cnt = bytes[at];
                    score += okp ? cnt : 0u;
                    tsum += okp ? (uint32_t)nleft[ll] : 0u;
                    np += okp ? 1u : 0u;
                }
                if (!ROT) {
                    score += dpp_xor1(score);
                    tsum += dpp_xor1(tsum);
                    np += dpp_xor1(np);
                }
                score += (best >> kKeyCountShift) + 1u;  // centre pair (k = 4): ll = i, rr = j*, the arg-max count itself
                tsum += ni;
                np += 1u;
                uint32_t pass = 0;
                if (ni != 0 && (ROT || half == 0)) pass = threshold_rejects(tsum, np, score, p.threshold_factor, thr_fast) ? 0u : 1u;
                uint32_t vbits = pass;
                bool writer = ni != 0 && half == 0;
                if (ROT) {
                    const unsigned long long bal = __ballot(pass);
                    vbits = (uint32_t)(bal >> (lane & 56)) & 0xFFu;
                    writer = ni != 0 && (lane & 7) == 0;
                }
                // every lane of the cell has read the header above (same wave, program order): it now holds cellPairs[i]
                if (writer) smem[i * (kDRow / 4)] = (ej << 8) | vbits;
            }
        }
        __syncthreads();
        // ---- mark inliers (cellPairs[l] == r, all rotations at once) and take this grid type's increments back
        for (int k0 = 0; k0 < kpt; k0 += kC) {
            uint32_t cw[kC];
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                const int i = (k0 + j) * 1024 + tid;
                cw[j] = i < m ? codes[i] : (1u << 2);
            }
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                if ((cw[j] & out_mask) != 0u) continue;
                const uint32_t l = ((cw[j] >> kDCellShiftS) & 0x1FFu) + (cw[j] & gx) + 20u * ((cw[j] >> 1) & gy);
                const uint32_t e = (cw[j] >> kDEShiftS) & 0x1FFu, row = __umul24(l, kDRow);
                const uint32_t x = smem[row >> 2] ^ (e << 8);  // < 256: the same right cell, x = the rotations that accept the cell
                if (g < 3) bytes[row + e] = 0;                 // (every reader of the entry is past the barrier: see dense_pair)
                if (x < 256u && ((cw[j] >> kDAccShiftS) | x) != (cw[j] >> kDAccShiftS)) codes[(k0 + j) * 1024 + tid] = cw[j] | (x << kDAccShiftS);
            }
        }
    }
    __syncthreads();

    // ---- run()'s return value per rotation, getInlierMask's strict '>' over the rotations (one scale)
    int winner = 0;
    if (ROT) {
        uint32_t cnt[kNRot];
#pragma unroll
        for (int r = 0; r < kNRot; ++r) cnt[r] = 0;
        for (int k = 0; k < kpt; ++k) {
            const int i = k * 1024 + tid;
            const uint32_t acc = i < m ? codes[i] >> kDAccShiftS : 0u;
#pragma unroll
            for (int r = 0; r < kNRot; ++r) cnt[r] += (uint32_t)__popcll(__ballot((acc >> r) & 1u));
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < kNRot; ++r)
                if (cnt[r]) atomicAdd(&misc[r], cnt[r]);
        }
        __syncthreads();
        uint32_t best_count = 0;
        winner = -1;
#pragma unroll
        for (int r = 0; r < kNRot; ++r) {
            const uint32_t c = misc[r];
            if (c > best_count) {
                best_count = c;
                winner = r;
            }
        }
    }
    // ---- copy-out: survivors per chunk of 64 consecutive matches (chunk k * 16 + wave), scanned; then the records, in input order
    uint32_t* cnt_tab = smem;  // in the matrix area (every reader of the matrix is past the barrier above)
    uint32_t* wave_tot = misc + 16;
    for (int k = 0; k < kpt; ++k) {
        const int i = k * 1024 + tid;
        const bool keep = winner >= 0 && i < m && ((codes[i] >> (kDAccShiftS + max(winner, 0))) & 1u) != 0u;
        const unsigned long long b = __ballot(keep);
        if (lane == 0) cnt_tab[k * 16 + wave] = (uint32_t)__popcll(b);
    }
    __syncthreads();
    uint32_t total = 0;
    {
        const int n_chunks = kpt * 16;  // <= 1024: one scan entry per thread
        const uint32_t c = tid < n_chunks ? cnt_tab[tid] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t off = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t tw = wave_tot[w];
            off += w < wave ? tw : 0u;
            total += tw;
        }
        if (tid < n_chunks) cnt_tab[tid] = off + incl - c;
    }
    __syncthreads();
    gms_dmatch* __restrict__ out = p.out + pr.match_off;
    uint8_t* mask_out = p.mask ? p.mask + pr.match_off : nullptr;
    for (int k = 0; k < kpt; ++k) {
        const int i = k * 1024 + tid;
        const bool keep = winner >= 0 && i < m && ((codes[i] >> (kDAccShiftS + max(winner, 0))) & 1u) != 0u;
        const unsigned long long b = __ballot(keep);
        if (i < m && mask_out) mask_out[i] = keep ? 1 : 0;
        if (keep) {
            const uint32_t pos = cnt_tab[k * 16 + wave] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            *reinterpret_cast<uint4*>(&out[pos]) = *reinterpret_cast<const uint4*>(&matches[i]);
        }
    }
    if (tid == 0) {
        gms_pair_result r;
        r.n_inliers = (int)total;
        r.best_scale = total ? 0 : -1;
        r.best_rot = total ? winner + 1 : -1;
        r.status = GMS_OK;
        p.results[pi] = r;
    }
}

// ================================================================================================================================
// The same size class under the DEFAULT flags (no rotation either: DisparityUtil.cpp:149,299 -- BASELINE config 4 as the reference's
// disparity demo calls it): stream_plain_kernel = stream_dense_kernel<false> rebuilt the way dense_pair_plain (gms_kernels.hip) rebuilt
// the register kernel -- entry-offset code words ([404 * cell + E : 18 | E : 9 | q and edge bits : 5]) in the scratch array, a per-lane
// sink word instead of predication, LDS by absolute offset, the two-lane verification on base + s * 403 * d -- and with what the
// streaming allows on top: the inlier flag of a match is one bit of two registers of its thread (a thread owns matches tid + 1024 k,
// k < 64), so the code words are written once and only read afterwards (the old kernel read-modify-wrote them in every marking pass
// and swept them three more times for the copy-out), and the records travel non-temporally.
// ================================================================================================================================
namespace {
constexpr uint32_t kSPTrashOff = kDSMiscOff + 128u;         // [16] dwords: the sinks
constexpr uint32_t kSPLdsBytes = kSPTrashOff + 64u;         // 162 592
static_assert(kSPLdsBytes <= kLdsBytes, "stream-plain layout exceeds the LDS");
constexpr uint32_t kSPEdgeX = 1u << 1, kSPEdgeY = 1u << 3;  // plain code word (as in gms_kernels.hip)
constexpr int kSPEShift = 5, kSPAtShift = 14, kSPTagShift = 20;
}  // namespace

__global__ void __launch_bounds__(1024)
stream_plain_kernel(FilterParams p, uint32_t* __restrict__ codes_ws, uint16_t* __restrict__ nleft_ws, uint32_t* __restrict__ flags, int mcap)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int kC = 8;  // code words a thread has in flight
    const int pi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const gms_pair pr = p.pairs[pi];
    const int m = pr.m;
    uint32_t* misc = smem + kDSMiscOff / 4;
    const int64_t total_kp = table_total_kp(p);
    bool general = m <= 0 || m > mcap || m > 64 * 1024 || pr.frame_a < 0 || pr.frame_a >= p.n_frames || pr.frame_b < 0 || pr.frame_b >= p.n_frames ||
                   total_kp < 0 || (uint32_t)(uintptr_t)((lds_u32_t*)smem) != 0u;
    int64_t offA = 0, offB = 0;
    int nA = 0, nB = 0;
    if (!general) {
        offA = p.frame_off[pr.frame_a];
        offB = p.frame_off[pr.frame_b];
        nA = (int)(p.frame_off[pr.frame_a + 1] - offA);
        nB = (int)(p.frame_off[pr.frame_b + 1] - offB);
        general = nA <= 0 || nB <= 0 || offA + nA > total_kp || offB + nB > total_kp;
    }
    if (general) {  // (workgroup-uniform) nothing this kernel can take: the general kernel decides what the pair is
        if (tid == 0) atomicOr(&flags[pi], kSFlagGeneral);
        return;
    }
    const gms_dmatch* __restrict__ matches = p.matches + pr.match_off;
    const uint16_t* __restrict__ lcode = reinterpret_cast<const uint16_t*>(p.pts + total_kp) + offA;
    const uint16_t* __restrict__ rcode = reinterpret_cast<const uint16_t*>(p.pts + total_kp) + total_kp + offB;
    uint32_t* __restrict__ codes = codes_ws + (size_t)pi * mcap;
    uint16_t* __restrict__ nl_g = nleft_ws + (size_t)pi * 4 * kLeftN;
    const int kpt = (m + 1023) >> 10;
    const uint32_t cw_sink = (kSPTrashOff + 4u * (uint32_t)(lane & 15)) << kSPAtShift;  // E = 0, q = 0, no edge bit

    // ---- the code words and the half-cell histogram (u32, one dword per half cell: [cell][qx + 2 qy], in the still unused matrix area)
    uint32_t* hist = smem;
    for (int j = tid; j < kFineN; j += 1024) hist[j] = 0;
    if (tid < 32) misc[tid] = 0;
    // frame A's left codes staged behind the histogram when they fit (77 600 keypoints): a gather from L2 costs the CU a cycle per
    // lane, one from LDS a few per wave -- and there are two gathers per match. (Frame B's right codes stay where they are: both
    // frames of a 50k-keypoint pair do not fit.)
    constexpr uint32_t kStageOff = 4u * kFineN;                          // bytes: behind the 1600 histogram dwords
    const uint32_t phA = (uint32_t)(reinterpret_cast<uintptr_t>(lcode) >> 1) & 7u;   // the copy keeps the source's 16-byte phase
    const uint32_t qA = (phA + (uint32_t)nA + 7u) >> 3;                  // uint4s
    const bool stagedA = kStageOff + 16u * qA <= kDSNleftOff;
    if (stagedA) {
        const uint4* __restrict__ srcA = reinterpret_cast<const uint4*>(lcode - phA);
        uint4* d4 = reinterpret_cast<uint4*>(smem) + kStageOff / 16u;
        for (uint32_t j = tid; j < qA; j += 1024) d4[j] = srcA[j];
    }
    const uint32_t ldsA = kStageOff + 2u * phA;                          // left code of keypoint q at byte ldsA + 2 q
    __syncthreads();
    // The first kR chunks of a thread's code words (32 matches) stay in registers for the whole kernel; only the rest is written to
    // the scratch array and streamed back in every binning and marking pass (a 20k-match pair streams nothing, a 50k-match pair a
    // third of what it did: at 256 pairs per launch the scratch arrays do not stay in L2 and that traffic is what the launch waits for).
    constexpr int kR = 4;
    uint32_t creg[kR * kC];
    {
        bool any_bad = false;
        auto build_chunk = [&](const int k0, uint32_t* cw_out) {
            uint2 qt[kC];
#pragma unroll
            for (int j = 0; j < kC; ++j) qt[j] = *reinterpret_cast<const uint2*>(&matches[(uint32_t)min((k0 + j) * 1024 + tid, m - 1)]);
            uint32_t ca[kC], cb[kC];
#pragma unroll
            for (int j = 0; j < kC; ++j) cb[j] = rcode[min(qt[j].y, (uint32_t)(nB - 1))];
            if (stagedA) {
#pragma unroll
                for (int j = 0; j < kC; ++j) ca[j] = ldsa_ld16(ldsA + 2u * min(qt[j].x, (uint32_t)(nA - 1)));
            } else {
#pragma unroll
                for (int j = 0; j < kC; ++j) ca[j] = lcode[min(qt[j].x, (uint32_t)(nA - 1))];
            }
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                const int i = (k0 + j) * 1024 + tid;
                const bool live = i < m;
                const uint32_t cell = ca[j] >> 7, e0 = cb[j] & 0x1FFu;  // cell: 510 = binned under no grid type, 511 = outside the parity domain
                const bool ok = qt[j].x < (uint32_t)nA && qt[j].y < (uint32_t)nB && cell != 511u && (cb[j] >> 15) == 0u && e0 != 0u;
                const bool binned = live && ok && cell < 510u;
                any_bad |= live && !ok;
                const uint32_t qx = ca[j] & 1u, qy = (ca[j] >> 2) & 1u;
                if (binned) atomicAdd(&hist[cell * 4u + qx + 2u * qy], 1u);
                const uint32_t qe = (ca[j] & 21u) | ((ca[j] >> 4) & kSPEdgeX) | ((ca[j] >> 3) & kSPEdgeY);
                cw_out[j] = binned ? (((__umul24(cell, kDRow) + e0) << kSPAtShift) | (e0 << kSPEShift) | qe) : cw_sink;
            }
        };
#pragma unroll
        for (int c = 0; c < kR; ++c) {
            if (c * kC < kpt) build_chunk(c * kC, &creg[c * kC]);
            else {
#pragma unroll
                for (int j = 0; j < kC; ++j) creg[c * kC + j] = cw_sink;
            }
        }
#pragma unroll 1
        for (int k0 = kR * kC; k0 < kpt; k0 += kC) {
            uint32_t cw[kC];
            build_chunk(k0, cw);
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                const uint32_t i = (uint32_t)((k0 + j) * 1024 + tid);
                if (i < (uint32_t)m) codes[i] = cw[j];
            }
        }
        if (any_bad) misc[8] = 1;
    }
    __syncthreads();
    if (misc[8] != 0) {  // an index out of range, a point outside the parity domain or outside the right grid: the general kernel's pair
        if (tid == 0) atomicOr(&flags[pi], kSFlagGeneral);
        return;
    }
    for (int item = tid; item < 4 * kLeftN; item += 1024) {
        const int g = item / kLeftN, cell = item - g * kLeftN;
        const int hx0 = 2 * (cell % kLeftW) - (g & 1), hy0 = 2 * (cell / kLeftW) - (g >> 1);
        uint32_t n = 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int hx = hx0 + dx, hy = hy0 + dy;
                if (hx >= 0 && hy >= 0) n += hist[(((hy >> 1) * kLeftW + (hx >> 1)) << 2) + (hx & 1) + ((hy & 1) << 1)];
            }
        if (n > 65535u) misc[9] = 1;
        nl_g[item] = (uint16_t)n;
    }
    __syncthreads();
    if (misc[9] != 0) {  // a cell above 65 535 matches
        if (tid == 0) {
            atomicOr(&flags[pi], kSFlagGeneral);
            if (p.overflow_events) atomicAdd(p.overflow_events, 1u);
        }
        return;
    }
    {   // motion.setTo(0), once: every grid type leaves the matrix as it found it; the sinks get bit 31 (nothing below ever clears it:
        // a sink is never equal to an E, so a sink word is nobody's inlier -- see dense_pair_plain)
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        uint4* d4 = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < kDSNleftOff / 16u; i += 1024) d4[i] = z4;
        if (tid < 16) smem[kSPTrashOff / 4 + tid] = 0x80000000u;
    }
    const bool thr_fast = threshold_fast_ok(p.threshold_factor);
    uint32_t acc[2] = {0u, 0u};  // match tid + 1024 k: bit (k & 24) + 7 - (k & 7) of acc[k >> 5]

    for (int g = 0; g < 4; ++g) {
        const uint32_t gx = (uint32_t)(g & 1), gy = (uint32_t)(g >> 1);
        const uint32_t q_mask = gx + 20u * gy;                                      // entry = entry1 + 404 * (q & q_mask)
        const uint32_t x_mask = (gx ? kSPEdgeX : 0u) | (gy ? kSPEdgeY : 0u);        // x >= 20 || y >= 20 -> -1 (DLL@0x180047d3d)
        const uint32_t tag = (uint32_t)g << kSPTagShift;  // (bits 20, 21: byte 3 of a row header stays zero)
        if (tid < kLeftN / 2) ldsa_st32(kDSNleftOff + 4u * (uint32_t)tid, reinterpret_cast<const uint32_t*>(nl_g + g * kLeftN)[tid]);
        __syncthreads();  // (the matrix clear / the previous grid type's undo are complete)
        // ---- assignMatchPairs
        auto bin_chunk = [&](const uint32_t* cwc) {
            uint32_t cg[kC], old[kC], at[kC];
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                cg[j] = (cwc[j] & x_mask) ? cw_sink : cwc[j];
                at[j] = mad24_vsv(cg[j] & q_mask, kDRow, cg[j] >> kSPAtShift);
                old[j] = ldsa_add_rtn(at[j] & ~3u, 1u << ((at[j] << 3) & 31u));
            }
            __builtin_amdgcn_sched_barrier(0);  // all of the chunk's atomics are issued before any result is read
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                const uint32_t e = (cg[j] >> kSPEShift) & 0x1FFu;
                const uint32_t before = __builtin_amdgcn_ubfe(old[j], at[j] << 3, 8);
                if (e != 0u && before == 255u) misc[9] = 1;  // the entry's byte has just wrapped: more than 255 matches in one (left cell, right cell) pair
                ldsa_max(at[j] - e, tag | (before << kKeyCountShift) | e);  // highest count, then lowest right cell
            }
        };
        uint32_t cwn[kC];  // the streamed part: the next chunk's code words are requested a chunk ahead (an L2 round trip per chunk otherwise)
        if (kR * kC < kpt) {
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                const uint32_t i = (uint32_t)((kR * kC + j) * 1024 + tid);
                cwn[j] = i < (uint32_t)m ? codes[i] : cw_sink;
            }
        }
#pragma unroll
        for (int c = 0; c < kR; ++c)
            if (c * kC < kpt) bin_chunk(&creg[c * kC]);
#pragma unroll 1
        for (int k0 = kR * kC; k0 < kpt; k0 += kC) {
            uint32_t cw[kC];
#pragma unroll
            for (int j = 0; j < kC; ++j) cw[j] = cwn[j];
            if (k0 + kC < kpt) {
#pragma unroll
                for (int j = 0; j < kC; ++j) {
                    const uint32_t i = (uint32_t)((k0 + kC + j) * 1024 + tid);
                    cwn[j] = i < (uint32_t)m ? codes[i] : cw_sink;
                }
            }
            bin_chunk(cw);
        }
        __syncthreads();
        if (misc[9] != 0) {  // (workgroup-uniform) the general kernel's pair; nothing has been written out
            if (tid == 0) {
                atomicOr(&flags[pi], kSFlagGeneral);
                if (p.overflow_events) atomicAdd(p.overflow_events, 1u);
            }
            return;
        }
        // ---- verifyCellPairs: two lanes per cell (see dense_pair_plain)
        if (tid < 2 * kLeftN) {
            const uint32_t vi = (uint32_t)tid >> 1;
            const uint32_t viy = (vi * 3277u) >> 16, vix = vi - 20u * viy;
            const bool vodd = (tid & 1) != 0;
            const int s1 = vodd ? -1 : 1;
            const uint32_t nlb = kDSNleftOff + 2u * vi, hdr = vi * kDRow;
            const uint32_t ni = ldsa_ld16(nlb);
            const uint32_t hdr_word = ldsa_ld32(hdr);
            uint32_t nl4[4];
            {
                constexpr int kD[4] = {-21, -20, -19, -1};
#pragma unroll
                for (int c = 0; c < 4; ++c) nl4[c] = ldsa_ld16(nlb + (uint32_t)(2 * s1 * kD[c]));
            }
            if (__ballot(ni != 0) != 0ull) {
                const uint32_t best = hdr_word & ((1u << kSPTagShift) - 1u);  // ((max count - 1) << 11) | E(j*), lowest j* among maxima
                const uint32_t ej = ni ? (best & 0x7FFu) : 403u;
                const uint32_t j = 403u - ej;
                const uint32_t jy = (j * 3277u) >> 16, jx = j - 20u * jy;
                const uint32_t lo = vodd ? 19u : 0u, hi = 19u - lo;
                const bool okA = (vix != lo) & (jx != lo), okB = (vix != hi) & (jx != hi), okC = (viy != lo) & (jy != lo);
                const int s403 = vodd ? -403 : 403;
                const uint32_t base = hdr + ej;
                uint32_t score = 0, tsum = 0, np = 0;
                auto side = [&](int c, int d, bool valid) {
                    score += ldsa_ld8(valid ? base + (uint32_t)(s403 * d) : 3u);  // (byte 3 of a row header is zero at all times)
                    tsum += valid ? nl4[c] : 0u;
                    np += valid ? 1u : 0u;
                };
                side(0, -21, okA & okC);
                side(1, -20, okC);
                side(2, -19, okB & okC);
                side(3, -1, okA);
                score += dpp_xor1(score);
                tsum += dpp_xor1(tsum);
                np += dpp_xor1(np);
                score += (best >> kKeyCountShift) + 1u;  // centre pair: ll = i, rr = j*, the arg-max count itself
                tsum += ni;
                np += 1u;
                if (ni != 0 && !vodd) ldsa_st32(hdr, threshold_rejects(tsum, np, score, p.threshold_factor, thr_fast) ? 0u : ej);  // cellPairs[i] as E(j*), 0 = none
            }
        }
        __syncthreads();
        // ---- mark inliers (cellPairs[l] == r) and take this grid type's increments back
        auto mark_chunk = [&](const int k0, const uint32_t* cwc) {
            uint32_t cur = 0;
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                const uint32_t cg = (cwc[j] & x_mask) ? cw_sink : cwc[j];
                const uint32_t at = mad24_vsv(cg & q_mask, kDRow, cg >> kSPAtShift), e = (cg >> kSPEShift) & 0x1FFu;
                const uint32_t cr = ldsa_ld32(at - e);  // (a sink's own dword: never equal to its E = 0)
                ldsa_st8(at, 0u);
                cur = shift_in_equal(cur, cr, e);       // match j of the chunk ends up in bit 7 - j
            }
            if (k0 < 32) acc[0] |= cur << (k0 & 31);
            else acc[1] |= cur << (k0 & 31);
        };
        if (kR * kC < kpt) {
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                const uint32_t i = (uint32_t)((kR * kC + j) * 1024 + tid);
                cwn[j] = i < (uint32_t)m ? codes[i] : cw_sink;
            }
        }
#pragma unroll
        for (int c = 0; c < kR; ++c)
            if (c * kC < kpt) mark_chunk(c * kC, &creg[c * kC]);
#pragma unroll 1
        for (int k0 = kR * kC; k0 < kpt; k0 += kC) {
            uint32_t cw[kC];
#pragma unroll
            for (int j = 0; j < kC; ++j) cw[j] = cwn[j];
            if (k0 + kC < kpt) {
#pragma unroll
                for (int j = 0; j < kC; ++j) {
                    const uint32_t i = (uint32_t)((k0 + kC + j) * 1024 + tid);
                    cwn[j] = i < (uint32_t)m ? codes[i] : cw_sink;
                }
            }
            mark_chunk(k0, cw);
        }
    }
    __syncthreads();

    // ---- copy-out: survivors per chunk of 64 consecutive matches (chunk k * 16 + wave), scanned; then the records, in input order
    const unsigned long long accq = (unsigned long long)acc[0] | ((unsigned long long)acc[1] << 32);
    auto kept = [&](int k) -> bool { return ((accq >> ((k & 56) + 7 - (k & 7))) & 1ull) != 0ull; };
    uint32_t* cnt_tab = smem;  // in the matrix area (every reader of the matrix is past the barrier above)
    uint32_t* wave_tot = misc + 16;
#pragma unroll 1
    for (int k = 0; k < kpt; ++k) {
        const unsigned long long b = __ballot(k * 1024 + tid < m && kept(k));
        if (lane == 0) cnt_tab[k * 16 + wave] = (uint32_t)__popcll(b);
    }
    __syncthreads();
    uint32_t total = 0;
    {
        const int n_chunks = kpt * 16;  // <= 1024: one scan entry per thread
        const uint32_t c = tid < n_chunks ? cnt_tab[tid] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t off = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t tw = wave_tot[w];
            off += w < wave ? tw : 0u;
            total += tw;
        }
        if (tid < n_chunks) cnt_tab[tid] = off + incl - c;
    }
    __syncthreads();
    gms_dmatch* __restrict__ out = p.out + pr.match_off;
    uint8_t* mask_out = p.mask ? p.mask + pr.match_off : nullptr;
#pragma unroll 1
    for (int k = 0; k < kpt; ++k) {
        const int i = k * 1024 + tid;
        const bool keep = i < m && kept(k);
        const unsigned long long b = __ballot(keep);
        if (i < m && mask_out) mask_out[i] = keep ? 1 : 0;
        if (keep) {
            const uint32_t pos = cnt_tab[k * 16 + wave] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(&matches[i])), reinterpret_cast<u32x4_t*>(&out[pos]));
        }
    }
    if (tid == 0) {
        gms_pair_result r;
        r.n_inliers = (int)total;
        r.best_scale = total ? 0 : -1;
        r.best_rot = total ? 1 : -1;
        r.status = GMS_OK;
        p.results[pi] = r;
    }
}

// GMS_STREAM_PLAIN=0 (diagnostic, read once per process): the default-flags pairs of this size class on stream_dense_kernel<false> instead
static bool stream_plain_on()
{
    static const bool on = [] { const char* e = getenv("GMS_STREAM_PLAIN"); return !(e && atoi(e) == 0); }();
    return on;
}

size_t stream_dense_ws_bytes_per_pair(int mcap) { return (size_t)mcap * 4 + 4 * (size_t)kLeftN * 2 + 4 + 64; }

// ws layout for n pairs: codes [n][mcap] u32 | nleft [n][4][400] u16 | flags [n]; *flags_out marks the pairs left to launch_filter_big (bit 1)
hipError_t launch_filter_stream_dense(const FilterParams& p, int mcap, void* ws, const uint32_t** flags_out, hipStream_t stream)
{
    const int n = p.n_pairs;
    if (n <= 0) return hipSuccess;
    if (p.right_w[0] != 20 || p.right_h[0] != 20 || p.with_scale) return hipErrorInvalidValue;
    char* q = reinterpret_cast<char*>(ws);
    uint32_t* codes = reinterpret_cast<uint32_t*>(q);
    q += align16s((size_t)n * mcap * 4);
    uint16_t* nleft = reinterpret_cast<uint16_t*>(q);
    q += align16s((size_t)n * 4 * kLeftN * 2);
    uint32_t* flags = reinterpret_cast<uint32_t*>(q);
    hipError_t e = hipMemsetAsync(flags, 0, (size_t)n * 4, stream);
    if (e != hipSuccess) return e;
    if (p.with_rotation) hipLaunchKernelGGL(stream_dense_kernel<true>, dim3((unsigned)n), dim3(1024), kDSLdsBytes, stream, p, codes, nleft, flags, mcap);
    else if (stream_plain_on()) hipLaunchKernelGGL(stream_plain_kernel, dim3((unsigned)n), dim3(1024), kSPLdsBytes, stream, p, codes, nleft, flags, mcap);
    else hipLaunchKernelGGL(stream_dense_kernel<false>, dim3((unsigned)n), dim3(1024), kDSLdsBytes, stream, p, codes, nleft, flags, mcap);
    *flags_out = flags;
    return hipGetLastError();
}

// ---- launch helpers ----------------------------------------------------------------------------------------------------------------------
hipError_t init_stream_kernels()  // once per context: see init_filter_kernels
{
    const void* fns[] = {reinterpret_cast<const void*>(stream_filter_kernel<true>), reinterpret_cast<const void*>(stream_filter_kernel<false>),
                         reinterpret_cast<const void*>(stream_dense_kernel<true>), reinterpret_cast<const void*>(stream_dense_kernel<false>),
                         reinterpret_cast<const void*>(stream_plain_kernel)};
    for (const void* fn : fns) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

int stream_max_matches() { return kSMaxMatches; }


size_t stream_ws_bytes_per_pair(const FilterParams& p, int mcap, bool)
{
    const size_t n_scales = p.with_scale ? 5 : 1;
    return (size_t)mcap * 16 + (size_t)kFineN * 4 + (size_t)kSRowWords * 4 + 4 * (size_t)kLeftN * 2 + 5 * 8 * 4 + (size_t)kSTilesMax * 5 * 8 * 4 + 4 +
           4 * n_scales * (size_t)kLeftN * 4 + 128;  // (+ the alignment of the arrays of a slice)
}

// ws layout for n pairs: entries | codes | nfine | row_cnt | counts | flags | tile_cnt | tables | nleft; *flags_out marks the pairs left
// to launch_filter_big (bit 1)
hipError_t launch_filter_stream(const FilterParams& p, int mcap, void* ws, const uint32_t** flags_out, hipStream_t stream)
{
    const int n = p.n_pairs;
    if (n <= 0) return hipSuccess;
    const int n_scales = p.with_scale ? 5 : 1;
    static const int kGrid[5] = {20, 10, 14, 28, 40};  // the right grids the kernels are written for (setScale, DLL@0x180048c10)
    for (int s = 0; s < n_scales; ++s)
        if (p.right_w[s] != kGrid[s] || p.right_h[s] != kGrid[s]) return hipErrorInvalidValue;
    StreamWs w;
    char* q = reinterpret_cast<char*>(ws);
    w.entries = reinterpret_cast<uint2*>(q);
    q += align16s((size_t)n * mcap * 8);
    w.codes = reinterpret_cast<uint2*>(q);
    q += align16s((size_t)n * mcap * 8);
    char* zero_from = q;  // the histograms, counters and flags start at zero (tables and tile counts do not need to: every word has one writer)
    w.nfine = reinterpret_cast<uint32_t*>(q);
    q += (size_t)n * kFineN * 4;
    w.row_cnt = reinterpret_cast<uint32_t*>(q);
    q += (size_t)n * kSRowWords * 4;
    w.counts = reinterpret_cast<uint32_t*>(q);
    q += (size_t)n * 5 * 8 * 4;
    w.flags = reinterpret_cast<uint32_t*>(q);
    q += align16s((size_t)n * 4);
    char* zero_to = q;
    w.tile_cnt = reinterpret_cast<uint32_t*>(q);
    q += (size_t)n * kSTilesMax * 5 * 8 * 4;
    w.tables = reinterpret_cast<uint32_t*>(q);
    q += align16s((size_t)n * n_scales * 4 * kLeftN * 4);
    w.nleft = reinterpret_cast<uint16_t*>(q);
    hipError_t e = hipMemsetAsync(zero_from, 0, (size_t)(zero_to - zero_from), stream);
    if (e != hipSuccess) return e;
    const dim3 ig((unsigned)((mcap + 4095) / 4096), (unsigned)n);
    hipLaunchKernelGGL(stream_index_kernel<0>, ig, dim3(1024), 0, stream, p, w, mcap);
    hipLaunchKernelGGL(stream_index_kernel<1>, ig, dim3(1024), 0, stream, p, w, mcap);
    const bool rot = p.with_rotation != 0;
    const dim3 fg((unsigned)(n_scales == 5 ? kSItemsScales : 4) * (unsigned)n), sg((unsigned)((mcap + kSMarkTile - 1) / kSMarkTile), (unsigned)n);
    if (rot) {
        hipLaunchKernelGGL(stream_filter_kernel<true>, fg, dim3(1024), kSLdsBytes, stream, p, w, mcap, n_scales);
        hipLaunchKernelGGL(stream_mark_kernel<true>, sg, dim3(1024), 0, stream, p, w, mcap, n_scales);
        hipLaunchKernelGGL(stream_compact_kernel<true>, sg, dim3(1024), 0, stream, p, w, mcap, n_scales);
    } else {
        hipLaunchKernelGGL(stream_filter_kernel<false>, fg, dim3(1024), kSLdsBytes, stream, p, w, mcap, n_scales);
        hipLaunchKernelGGL(stream_mark_kernel<false>, sg, dim3(1024), 0, stream, p, w, mcap, n_scales);
        hipLaunchKernelGGL(stream_compact_kernel<false>, sg, dim3(1024), 0, stream, p, w, mcap, n_scales);
    }
    *flags_out = w.flags;
    return hipGetLastError();
}

}  // namespace gms
