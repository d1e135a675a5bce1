// gms_kernel_big.hip -- the GMS filter for pairs whose matches do not fit one workgroup's registers + LDS
// (m > 16 384: BASELINE config 4 at 50k matches, the reference's dense one-keypoint-per-pixel disparity call,
// DisparityUtil.cpp:123-149, at W*H matches).
//
// Same algorithm, same phases and the same region/bucket/header table layout as filter_kernel in
// gms_kernels.hip (read that file's header first); what changes is where the two big arrays live:
//   code[m]   one dword per match      -> a per-workgroup slab in HBM (L2-resident), each thread owns i = tid + k*1024
//   table     the (left cell, right cell) -> count non-zeros -> the same slab; device-scope atomics, and every
//             table read is an agent-scope atomic load (served by L2: the atomics never update the CU's L1)
// The small per-cell and per-half-cell tables, the winner's bit mask and the copy-out scan stay in LDS.
// At most one persistent workgroup per CU walks the pairs, so the workspace does not grow with the batch.
// This path is about coverage, not speed: it is latency-bound on L2 atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gms_device_common.h"

namespace gms {
namespace {

__constant__ int8_t c_rot_big[8][9] = {  // mRotationPatterns - 1 (DLL .rdata 0x18012f520)
    {0, 1, 2, 3, 4, 5, 6, 7, 8}, {3, 0, 1, 6, 4, 2, 7, 8, 5}, {6, 3, 0, 7, 4, 1, 8, 5, 2},
    {7, 6, 3, 8, 4, 0, 5, 2, 1}, {8, 7, 6, 5, 4, 3, 2, 1, 0}, {5, 8, 7, 2, 4, 6, 1, 0, 3},
    {2, 5, 8, 1, 4, 7, 0, 3, 6}, {1, 2, 5, 0, 4, 8, 3, 6, 7}};

constexpr int kDescShift = 12;                 // desc = (header bucket << 12) | data buckets (<= 2048)
constexpr uint32_t kDescNbMask = (1u << kDescShift) - 1u;

// table reads: agent scope, i.e. from L2, where the atomics land
__device__ __forceinline__ uint32_t tload(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// code words: written and re-read by the same thread across phases and pairs; kept out of the L1 as well
__device__ __forceinline__ void cstore(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// motion[l][r]++ and the running arg-max of the row, general walk (see region_insert_general in gms_kernels.hip)
__device__ void big_insert(uint32_t* tab, uint32_t d, uint32_t r)
{
    const uint32_t nb = d & kDescNbMask, hb = d >> kDescShift;
    if (nb == 0) return;
    const uint32_t kr = r << kSlotRShift;
    uint32_t b = bucket_of(r, nb);
    for (uint32_t guard = 0; guard < 8u * nb + 8u; ++guard) {
        uint32_t* bk = tab + ((size_t)(hb + 1u + b) << 2);
        bool again = false, moved = false;
        for (int e = 0; e < 4 && !moved; ++e) {
            const uint32_t v = tload(bk + e);
            uint32_t count = 0;
            if ((v ^ kr) <= kSlotCountMask) {
                count = (atomicAdd(bk + e, 1u) & kSlotCountMask) + 1u;
            } else if (v == kEmpty) {
                const uint32_t prev = atomicCAS(bk + e, kEmpty, kr | 1u);
                if (prev == kEmpty) count = 1u;
                else if ((prev ^ kr) <= kSlotCountMask) count = (atomicAdd(bk + e, 1u) & kSlotCountMask) + 1u;
                else { again = true; moved = true; }  // another right cell took the slot: look at this bucket again
            } else {
                continue;
            }
            if (count) {
                atomicMin(tab + ((size_t)hb << 2), ~((count << 11) | (2047u - r)));
                return;
            }
        }
        if (!again && ++b == nb) b = 0;
    }
}

// motion[l][r]
__device__ uint32_t big_lookup(const uint32_t* tab, uint32_t d, uint32_t r)
{
    const uint32_t nb = d & kDescNbMask, hb = d >> kDescShift;
    if (nb == 0) return 0;
    const uint32_t kr = r << kSlotRShift;
    uint32_t b = bucket_of(r, nb);
    for (uint32_t guard = 0; guard < nb; ++guard) {
        const uint32_t* bk = tab + ((size_t)(hb + 1u + b) << 2);
        for (int e = 0; e < 4; ++e) {
            const uint32_t v = tload(bk + e);
            if ((v ^ kr) <= kSlotCountMask) return v & kSlotCountMask;
            if (v == kEmpty) return 0;
        }
        if (++b == nb) b = 0;
    }
    return 0;
}

}  // namespace

template <bool ROT>
__global__ void __launch_bounds__(kThreads)
filter_kernel_big(FilterParams p, uint32_t* ws, size_t ws_stride, int mcap, uint32_t T)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int kNRot = ROT ? 8 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    uint32_t* code = ws + (size_t)blockIdx.x * ws_stride;  // [mcap]
    uint32_t* tab = code + mcap;                           // [T], 16-byte aligned (mcap is a multiple of 64)

    uint32_t* nfine = smem;                        // [1664], later fres
    uint32_t* fres = nfine;
    uint32_t* nleft4 = nfine + kFineStride;        // [4][400]
    uint32_t* desc4 = nleft4 + 4 * kLeftN;         // [4][400]
    uint32_t* fdesc4 = desc4 + 4 * kLeftN;         // [4][1664]
    uint32_t* misc = fdesc4 + 4 * kFineStride;     // [0..7] counts, [8] error, [9] carry, [10] capacity, [12..15] allocators, [16..31] scan
    // the winner's bit mask and the copy-out scan: in LDS up to kBigLdsMaskMatches matches, behind the table in the slab beyond
    // (only this workgroup touches them, always across a workgroup barrier)
    uint32_t* bestmask = mcap <= kBigLdsMaskMatches ? misc + 64 : tab + T;  // mcap / 32
    uint32_t* chunk_base = bestmask + (mcap >> 5);                          // mcap / 64 + 1

    for (int pi = blockIdx.x; pi < p.n_pairs; pi += gridDim.x) {
        if (p.pair_flags && !(p.pair_flags[pi] & 2u)) continue;  // the band kernels (gms_kernel_band.hip) did this pair
        const gms_pair pr = p.pairs[pi];
        const int m = pr.m;
        const gms_dmatch* __restrict__ matches = p.matches + pr.match_off;
        __syncthreads();  // the previous pair's readers of LDS are done
        if (tid < 48) misc[tid] = 0;
        for (int i = tid; i < (mcap >> 5); i += kThreads) bestmask[i] = 0;
        for (int i = tid; i < kFineStride; i += kThreads) nfine[i] = 0;
        for (int i = tid; i < 4 * kFineStride; i += kThreads) fdesc4[i] = 0;

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
        if (mm == 0 || nA <= 0 || nB <= 0) {  // workgroup-uniform
            if (tid == 0) {
                gms_pair_result r;
                r.n_inliers = 0;
                r.best_scale = -1;
                r.best_rot = -1;
                r.status = (bad_pair || m > 0) ? GMS_ERR_DOMAIN : GMS_OK;
                p.results[pi] = r;
            }
            continue;
        }
        __syncthreads();

        // ---- bin: half-cell index of the left point, right cell at scale 0 --------------------------------
        {
            const int wr = p.right_w[0];
            const uint32_t nr = (uint32_t)(wr * p.right_h[0]);
            const float fwr = (float)wr, fhr = (float)p.right_h[0];
            bool any_bad = false;
            for (int i = tid; i < mm; i += kThreads) {
                const int2 qt = *reinterpret_cast<const int2*>(&matches[i]);
                const float2 a = ptsA[min((uint32_t)qt.x, (uint32_t)(nA - 1))];
                const float2 b = ptsB[min((uint32_t)qt.y, (uint32_t)(nB - 1))];
                const uint32_t worst = max(max(__float_as_uint(a.x), __float_as_uint(a.y)),
                                           max(__float_as_uint(b.x), __float_as_uint(b.y)));
                const float fx = 20.0f * a.x, fy = 20.0f * a.y;
                const uint32_t hx = (uint32_t)(int)(fx + fx), hy = (uint32_t)(int)(fy + fy);
                const uint32_t r = (uint32_t)((int)(fwr * b.x) + (int)(fhr * b.y) * wr);
                const bool ok = (uint32_t)qt.x < (uint32_t)nA && (uint32_t)qt.y < (uint32_t)nB &&
                                worst < 0x49800000u && r < nr;
                const uint32_t f = (ok && hx < 40u && hy < 40u) ? hy * kFineW + hx : kFineInvalid;
                if (f != kFineInvalid) atomicAdd(&nfine[f], 1u);
                any_bad |= !ok;
                cstore(&code[i], (ok ? r : 0u) | (f << kFShift));
            }
            if (any_bad) misc[8] = 1;
        }
        __syncthreads();

        // ---- region tables -------------------------------------------------------------------------------------
        for (int item = tid; item < 4 * kLeftN; item += kThreads) {
            const int g = item / kLeftN, cell = item - g * kLeftN;
            const int x = cell % kLeftW, y = cell / kLeftW;
            const int hx0 = 2 * x - (g & 1), hy0 = 2 * y - (g >> 1);
            uint32_t n = 0;
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    const int hx = hx0 + dx, hy = hy0 + dy;
                    if (hx >= 0 && hy >= 0) n += nfine[hy * kFineW + hx];
                }
            const uint32_t nb = region_buckets(n, 0);
            uint32_t d = 0;
            if (n > kSlotCountMask) misc[10] = 1;  // a table slot counts in 21 bits: more than 2 097 151 matches in ONE left cell
            if (nb) d = (atomicAdd(&misc[12 + g], nb + 1u) << kDescShift) | nb;
            nleft4[item] = n;
            desc4[item] = d;
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    const int hx = hx0 + dx, hy = hy0 + dy;
                    if (hx >= 0 && hy >= 0) fdesc4[g * kFineStride + hy * kFineW + hx] = d;
                }
        }
        __syncthreads();

        const int n_scales = p.with_scale ? 5 : 1;
        uint32_t best_count = 0;
        int best_scale = -1, best_rot = -1;

        for (int s = 0; s < n_scales; ++s) {
            const int wr = p.right_w[s], hr = p.right_h[s];
            if (s > 0) {
                const uint32_t nr = (uint32_t)(wr * hr);
                const float fwr = (float)wr, fhr = (float)hr;
                bool any_bad = false;
                for (int i = tid; i < mm; i += kThreads) {
                    const uint32_t fpart = tload(&code[i]) & (kFMask << kFShift);
                    const bool had = fpart != (kFineInvalid << kFShift);
                    const int t = matches[i].trainIdx;
                    const float2 b = ptsB[min((uint32_t)t, (uint32_t)(nB - 1))];
                    const uint32_t r = (uint32_t)((int)(fwr * b.x) + (int)(fhr * b.y) * wr);
                    const bool ok = r < nr;
                    any_bad |= had && !ok;
                    cstore(&code[i], (had && ok) ? (fpart | r) : (kFineInvalid << kFShift));
                }
                if (any_bad) misc[8] = 1;
            }
            for (int i = tid; i < mm; i += kThreads) cstore(&code[i], tload(&code[i]) & ((1u << kAccShift) - 1u));

            for (int g = 0; g < 4; ++g) {
                const uint32_t* nleft = nleft4 + g * kLeftN;
                const uint32_t* desc = desc4 + g * kLeftN;
                const uint32_t* fdesc = fdesc4 + g * kFineStride;
                {
                    const uint4 e4 = make_uint4(kEmpty, kEmpty, kEmpty, kEmpty);
                    uint4* tab4 = reinterpret_cast<uint4*>(tab);
                    for (uint32_t i = tid; i < (T >> 2); i += kThreads) tab4[i] = e4;
                }
                __threadfence();  // the empty table is in L2 before any other wave's atomic can reach it
                __syncthreads();
                for (int i = tid; i < kFineStride; i += kThreads) fres[i] = kNoMatch;

                for (int i = tid; i < mm; i += kThreads) {
                    const uint32_t c = tload(&code[i]);
                    big_insert(tab, fdesc[(c >> kFShift) & kFMask], c & kRMask);
                }
                __syncthreads();

                for (int item = tid; item < ((kLeftN * kNRot + 63) & ~63); item += kThreads) {
                    const bool live = item < kLeftN * kNRot;
                    const int i = live ? (ROT ? (item >> 3) : item) : 0;
                    const int rot = ROT ? (item & 7) : 0;
                    const uint32_t ni = live ? nleft[i] : 0u;
                    uint32_t pass = 0, j = 0;
                    if (ni != 0) {
                        const uint32_t bi = ~tload(tab + ((size_t)(desc[i] >> kDescShift) << 2));
                        j = 2047u - (bi & kRMask);
                        const int jx = (int)j % wr, jy = (int)j / wr;
                        const int ix = i % kLeftW, iy = i / kLeftW;
                        uint32_t score = 0, tsum = 0, numpair = 0;
                        for (int k = 0; k < 9; ++k) {
                            const int q = ROT ? c_rot_big[rot][k] : k;
                            const int lx = ix + (k % 3) - 1, ly = iy + (k / 3) - 1;
                            const int rx = jx + (q % 3) - 1, ry = jy + (q / 3) - 1;
                            if ((uint32_t)lx >= (uint32_t)kLeftW || (uint32_t)ly >= (uint32_t)kLeftH) continue;  // ll == -1
                            if ((uint32_t)rx >= (uint32_t)wr || (uint32_t)ry >= (uint32_t)hr) continue;          // rr == -1
                            const int ll = lx + ly * kLeftW;
                            score += big_lookup(tab, desc[ll], (uint32_t)(rx + ry * wr));
                            tsum += nleft[ll];
                            numpair++;
                        }
                        const double thresh = sqrt((double)tsum / (double)numpair) * p.threshold_factor;
                        pass = (thresh > (double)score) ? 0u : 1u;
                    }
                    uint32_t bits = pass;
                    bool writer = ni != 0;
                    if (ROT) {
                        const unsigned long long bal = __ballot(pass);
                        bits = (uint32_t)(bal >> (lane & 56)) & 0xFFu;
                        writer = writer && (lane & 7) == 0;
                    }
                    if (writer) {
                        const uint32_t cr = (j << 8) | bits;
                        const int hx0 = 2 * (i % kLeftW) - (g & 1), hy0 = 2 * (i / kLeftW) - (g >> 1);
                        for (int dy = 0; dy < 2; ++dy)
                            for (int dx = 0; dx < 2; ++dx) {
                                const int hx = hx0 + dx, hy = hy0 + dy;
                                if (hx >= 0 && hy >= 0) fres[hy * kFineW + hx] = cr;
                            }
                    }
                }
                __syncthreads();

                for (int i = tid; i < mm; i += kThreads) {
                    const uint32_t c = tload(&code[i]);
                    const uint32_t cr = fres[(c >> kFShift) & kFMask];
                    if ((cr >> 8) == (c & kRMask)) cstore(&code[i], c | (cr << kAccShift));
                }
                __syncthreads();  // mark reads fres, the next grid type resets it
            }

            // run() return value per rotation; keep on strict '>' (scale outer, rotation inner)
            for (int i0 = tid - lane; i0 < mm; i0 += kThreads) {
                const int i = i0 + lane;
                const uint32_t bits = (i < mm) ? (tload(&code[i]) >> kAccShift) : 0u;
                for (int r = 0; r < kNRot; ++r) {
                    const unsigned long long b = __ballot((bits >> r) & 1u);
                    if (lane == 0 && b) atomicAdd(&misc[r], (uint32_t)__popcll(b));
                }
            }
            __syncthreads();
            int winner = -1;
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
                for (int i0 = tid - lane; i0 < mm; i0 += kThreads) {
                    const int i = i0 + lane;
                    const uint32_t bit = (i < mm) ? ((tload(&code[i]) >> (kAccShift + winner)) & 1u) : 0u;
                    const unsigned long long b = __ballot(bit);
                    if (lane == 0) {
                        bestmask[i0 >> 5] = (uint32_t)b;
                        bestmask[(i0 >> 5) + 1] = (uint32_t)(b >> 32);
                    }
                }
            }
            __syncthreads();
            if (tid < 8) misc[tid] = 0;
            __syncthreads();
        }

        // ---- copy-out ------------------------------------------------------------------------------------------
        const bool too_many = misc[10] != 0;
        const bool failed = misc[8] != 0 || too_many;
        const int n_chunks = (mm + 63) >> 6;
        {
            uint32_t* wave_tot = misc + 16;
            for (int base = 0; base < n_chunks; base += kThreads) {
                const int c = base + tid;
                const uint32_t v = (c < n_chunks && !failed) ? __popc(bestmask[2 * c]) + __popc(bestmask[2 * c + 1]) : 0u;
                uint32_t incl = v;
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
                if (tid == kThreads - 1) misc[9] = wave_off + incl;
                __syncthreads();
            }
        }
        const uint32_t total = misc[9];
        gms_dmatch* __restrict__ out = p.out + pr.match_off;
        uint8_t* mask_out = p.mask ? p.mask + pr.match_off : nullptr;
        for (int i0 = tid - lane; i0 < mm; i0 += kThreads) {
            const int i = i0 + lane, c = i0 >> 6;
            const unsigned long long bits =
                failed ? 0ull : ((unsigned long long)bestmask[2 * c] | ((unsigned long long)bestmask[2 * c + 1] << 32));
            const bool in = (bits >> lane) & 1ull;
            if (i < mm) {
                if (mask_out) mask_out[i] = in ? 1 : 0;
                if (in) {
                    const uint32_t pos = chunk_base[c] + (uint32_t)__popcll(bits & ((1ull << lane) - 1ull));
                    *reinterpret_cast<uint4*>(&out[pos]) = *reinterpret_cast<const uint4*>(&matches[i]);
                }
            }
        }
        if (tid == 0) {
            gms_pair_result r;
            r.n_inliers = failed ? 0 : (int)total;
            r.best_scale = failed ? -1 : best_scale;
            r.best_rot = failed ? -1 : best_rot;
            r.status = too_many ? GMS_ERR_CAPACITY : failed ? GMS_ERR_DOMAIN : GMS_OK;
            p.results[pi] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
int big_mcap(int max_m) { return (max_m + 63) & ~63; }

// a left cell's region never has more than 2048 data buckets + its header (1600 right cells at most), so the table of one grid
// type never needs more than 400 * 2049 buckets however many matches the pair has -- and the descriptor's 20-bit bucket index holds
uint32_t big_table_slots(int mcap)
{
    const uint64_t want = ((uint64_t)mcap * 2u + 7u * kLeftN + 3u) & ~(uint64_t)3;
    const uint64_t most = (uint64_t)kLeftN * 2049u * 4u;
    return (uint32_t)(want < most ? want : most);
}

static size_t big_mask_dwords(int mcap) { return (size_t)(mcap >> 5) + (size_t)(mcap >> 6) + 1; }

size_t big_ws_stride_dwords(int mcap)
{
    const size_t d = (size_t)mcap + big_table_slots(mcap) + (mcap > kBigLdsMaskMatches ? big_mask_dwords(mcap) : 0);
    return (d + 3) & ~(size_t)3;
}

size_t big_lds_bytes(int mcap)
{
    return ((size_t)kFineStride + 8 * kLeftN + 4 * kFineStride + 64 + (mcap <= kBigLdsMaskMatches ? big_mask_dwords(mcap) : 0)) * 4;
}

hipError_t init_big_kernels()  // once per context: see init_filter_kernels
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(filter_kernel_big<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(filter_kernel_big<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    return e;
}

hipError_t launch_filter_big(const FilterParams& p, int mcap, int n_workgroups, uint32_t* ws, hipStream_t stream)
{
    if (p.n_pairs <= 0) return hipSuccess;
    const size_t lds = big_lds_bytes(mcap);
    const uint32_t T = big_table_slots(mcap);
    const size_t stride = big_ws_stride_dwords(mcap);
    const bool rot = p.with_rotation != 0;
    if (rot)
        hipLaunchKernelGGL(filter_kernel_big<true>, dim3((unsigned)n_workgroups), dim3(kThreads), lds, stream, p, ws, stride, mcap, T);
    else
        hipLaunchKernelGGL(filter_kernel_big<false>, dim3((unsigned)n_workgroups), dim3(kThreads), lds, stream, p, ws, stride, mcap, T);
    return hipGetLastError();
}

}  // namespace gms
