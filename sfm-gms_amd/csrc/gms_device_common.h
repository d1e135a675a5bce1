// gms_device_common.h -- device helpers shared by every kernel file of the GMS filter (one definition each:
// the hashed table's slot/bucket arithmetic, verifyCellPairs' threshold test, the rotation patterns, small
// LDS/DPP idioms). Internal; included by the .hip files only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gms_kernels.h"

namespace gms {

constexpr uint32_t kEmpty = 0xFFFFFFFFu;

// code word of the hashed kernels (one dword per match)
constexpr uint32_t kRMask = 0x7FFu;        // bits 0..10   right cell of the current scale (< 1600)
constexpr int kFShift = 11;                // bits 11..21  half-cell ("fine") index hy * 40 + hx of the left point,
constexpr uint32_t kFMask = 0x7FFu;        //              or kFineInvalid when the point is never binned
constexpr int kAccShift = 24;              // bits 24..31  inlier-under-rotation bits (OR over the 4 grid types)
constexpr uint32_t kFineInvalid = kFineN;  // entries [1600, 1664) of the fine tables are "nothing here"
constexpr int kFineStride = 1664;
constexpr uint32_t kNoMatch = 0xFFFFFF00u; // fres value that equals no right cell

// table slot: [right cell : 11 | count : 21]; a left cell's region holds only its own right cells and is
// organised in 4-slot buckets so that one 16-byte read sees a whole bucket
constexpr int kSlotRShift = 21;
constexpr uint32_t kSlotCountMask = (1u << kSlotRShift) - 1u;

// Data buckets (of 4 slots) of a left cell holding n matches: slots >= distinct right cells + 1, so an
// empty slot always exists and ends every probe chain. sh = 0, 1, 2 gives about 2, 1.5, 1.25 slots per match;
// 2048 buckets already exceed the 1600 right cells any region can hold.
__device__ __forceinline__ uint32_t region_buckets(uint32_t n, int sh)
{
    return n ? min((n + (n >> sh) + 3u) >> 2, 2048u) : 0u;
}

// bucket of right cell r in a region of nb <= 2048 buckets: Fibonacci hash on 12 bits, all 24-bit multiplies
__device__ __forceinline__ uint32_t bucket_of(uint32_t r, uint32_t nb)
{
    return __umul24(__umul24(r, 2531u) & 0xFFFu, nb) >> 12;
}

__device__ __forceinline__ uint32_t* lds_at(uint32_t* base, uint32_t byte_off)
{
    return reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(base) + byte_off);
}

// ---- idioms of the byte-matrix kernels that count their instructions (dense_pair_plain in gms_kernels.hip, stream_plain_kernel in
//      gms_kernel_stream.hip): LDS by absolute byte offset (the dynamic segment of these kernels starts at 0, so no "+ base" per access),
//      non-temporal record traffic, two instructions the compiler does not pick by itself
// The match records are read once and the survivors written once: non-temporal, so that what the L2 keeps is the lines a
// workgroup touches ahead for its successor (below) -- with plain loads and stores a good part of those is evicted before use.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
using lds_u32_t = __attribute__((address_space(3))) uint32_t;
using lds_u16_t = __attribute__((address_space(3))) uint16_t;
using lds_u8_t = __attribute__((address_space(3))) uint8_t;
__device__ __forceinline__ uint32_t ldsa_add_rtn(uint32_t a, uint32_t v)
{
    return __hip_atomic_fetch_add(reinterpret_cast<lds_u32_t*>((uintptr_t)a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void ldsa_add(uint32_t a, uint32_t v)
{
    (void)__hip_atomic_fetch_add(reinterpret_cast<lds_u32_t*>((uintptr_t)a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void ldsa_max(uint32_t a, uint32_t v)
{
    (void)__hip_atomic_fetch_max(reinterpret_cast<lds_u32_t*>((uintptr_t)a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t ldsa_ld32(uint32_t a) { return *reinterpret_cast<lds_u32_t*>((uintptr_t)a); }
__device__ __forceinline__ uint32_t ldsa_ld16(uint32_t a) { return *reinterpret_cast<lds_u16_t*>((uintptr_t)a); }
__device__ __forceinline__ uint32_t ldsa_ld8(uint32_t a) { return *reinterpret_cast<lds_u8_t*>((uintptr_t)a); }
__device__ __forceinline__ void ldsa_st32(uint32_t a, uint32_t v) { *reinterpret_cast<lds_u32_t*>((uintptr_t)a) = v; }
__device__ __forceinline__ void ldsa_st8(uint32_t a, uint32_t v) { *reinterpret_cast<lds_u8_t*>((uintptr_t)a) = (uint8_t)v; }

// a * b + c on the 24-bit multiplier, b in a scalar register (the compiler turns the builtin multiply + add into the quarter-rate
// v_mad_u64_u32 when it cannot see that the factors are short)
__device__ __forceinline__ uint32_t mad24_vsv(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
}
// bits = 2 * bits + (a == b): a compare and an add-with-carry
__device__ __forceinline__ uint32_t shift_in_equal(uint32_t bits, uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %3, %3, vcc" : "=v"(d) : "v"(a), "v"(b), "v"(bits) : "vcc");
    return d;
}

// verifyCellPairs' test "thresh = sqrt(T / n) * factor; reject iff thresh > score" (divsd, sqrtsd, mulsd, comisd at
// DLL@0x180049171). In exact arithmetic (factor > 0) it is T * factor^2 > score^2 * n. b = score^2 * n is exact in
// fp64 (< 2^53) and a = fl(fl(T * factor) * factor) is within 2^-51 of exact, while the reference's three roundings
// move thresh by less than 2^-50 relative: when a and b differ by more than 2^-40 relative, the reference's answer is
// the sign of a - b. Only near-ties (exact ties, in practice) run the divide and the square root.
__device__ __forceinline__ bool threshold_rejects(uint32_t T, uint32_t n, uint32_t score, double factor, bool fast_ok)
{
    const double dT = (double)T, dN = (double)n, dS = (double)score;
    if (fast_ok) {
        const double a = dT * factor * factor, b = dS * dS * dN;
        if (fabs(a - b) > fmax(a, b) * 0x1p-40) return a > b;
    }
    return sqrt(dT / dN) * factor > dS;
}
// factor ranges where factor^2 neither overflows nor loses precision to underflow
__device__ __forceinline__ bool threshold_fast_ok(double factor) { return factor > 1e-100 && factor < 1e100; }

// mRotationPatterns without a table: the eight outer positions of the 3 x 3 block form a ring
// (0,1,2,5,8,7,6,3 clockwise); pattern rot sends the position with ring index u to the one with ring index
// (u - rot) mod 8, the centre stays (checked against the DLL's table in tests/test_oracle_pins.py). A per-lane
// rotation would otherwise index constant memory per lane, which the compiler serialises over the distinct values.
__device__ __forceinline__ int rotated_position(int rot, int u) { return (int)((0x36785210u >> (((u - rot) & 7) << 2)) & 15u); }
__device__ __forceinline__ int position_dx(int q) { return (int)((0x24924u >> (q << 1)) & 3u) - 1; }  // q % 3 - 1
__device__ __forceinline__ int position_dy(int q) { return (int)((0x2a540u >> (q << 1)) & 3u) - 1; }  // q / 3 - 1

// All of a rotation pattern in one word: for the eight outer neighbours k8 = 0..7 (positions 0,1,2,3,5,6,7,8 of the 3 x 3 block),
// where pattern rot + 1 sends it, as (dx + 1) | (dy + 1) << 2 in bits 4 k8 .. 4 k8 + 3. A compile-time constant per rotation.
constexpr uint32_t rotation_pack(int rot)
{
    constexpr int kRingIndex[9] = {0, 1, 2, 7, -1, 3, 6, 5, 4};  // position -> ring index
    uint32_t w = 0;
    for (int k8 = 0; k8 < 8; ++k8) {
        const int k = k8 < 4 ? k8 : k8 + 1;
        const int q = (int)((0x36785210u >> (((kRingIndex[k] - rot) & 7) << 2)) & 15u);
        const int dx = (int)((0x24924u >> (q << 1)) & 3u), dy = (int)((0x2a540u >> (q << 1)) & 3u);  // already + 1
        w |= (uint32_t)(dx | (dy << 2)) << (4 * k8);
    }
    return w;
}

// Number of keypoints the frame table behind p.pts was built for (normalize_kernel's header), or -1 when the block does not
// start with the header: the byte-matrix kernels then leave the pair to the path that works from the points alone.
__device__ __forceinline__ int64_t table_total_kp(const FilterParams& p)
{
    const uint32_t* __restrict__ h = reinterpret_cast<const uint32_t*>(p.pts) - kTableHeaderBytes / 4;
    const uint2 magic = *reinterpret_cast<const uint2*>(h);
    const int64_t total = *reinterpret_cast<const int64_t*>(h + 2);
    return (magic.x == kTableMagic0 && magic.y == kTableMagic1) ? total : (int64_t)-1;
}

// A value every lane of the wave holds alike (read through a pointer the compiler cannot prove read-only, so it arrives by vector
// load): moved to scalar registers, where it costs no vector register for the rest of the kernel.
// (inline assembly with a scalar-register result: the builtin is folded away once the compiler has proven the value uniform, and the
//  value then stays in the vector registers it was loaded into)
__device__ __forceinline__ int uniform(int v)
{
    int s;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(v));
    return s;
}
__device__ __forceinline__ int64_t uniform(int64_t v)
{
    const uint32_t lo = (uint32_t)uniform((int)(uint32_t)v), hi = (uint32_t)uniform((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
// A pair's record in scalar registers. Its three 8-byte words are requested TOGETHER and only then moved: a v_readfirstlane per
// field straight behind each field's load waits for every load in turn -- five round trips at the start of every pair.
__device__ __forceinline__ gms_pair load_pair(const gms_pair* pairs, int idx)
{
    const uint2* __restrict__ q = reinterpret_cast<const uint2*>(pairs + idx);
    uint2 a = q[0], b = q[1], c = q[2];
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(c.x), "+v"(c.y));  // (all of them have arrived before the first is used)
    gms_pair u;
    u.frame_a = uniform((int)a.x);
    u.frame_b = uniform((int)a.y);
    u.m = uniform((int)b.x);
    u.reserved = 0;
    u.match_off = (int64_t)(((uint64_t)(uint32_t)uniform((int)c.y) << 32) | (uint32_t)uniform((int)c.x));
    return u;
}

// The same with the frame table's header word fetched alongside (table_total_kp): one round trip for both.
__device__ __forceinline__ gms_pair load_pair(const gms_pair* pairs, int idx, const FilterParams& p, int64_t& total_kp)
{
    const uint2* __restrict__ q = reinterpret_cast<const uint2*>(pairs + idx);
    const uint32_t* __restrict__ h = reinterpret_cast<const uint32_t*>(p.pts) - kTableHeaderBytes / 4;
    uint2 a = q[0], b = q[1], c = q[2];
    uint2 magic = *reinterpret_cast<const uint2*>(h), total = *reinterpret_cast<const uint2*>(h + 2);
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(c.x), "+v"(c.y), "+v"(magic.x), "+v"(magic.y), "+v"(total.x), "+v"(total.y));
    gms_pair u;
    u.frame_a = uniform((int)a.x);
    u.frame_b = uniform((int)a.y);
    u.m = uniform((int)b.x);
    u.reserved = 0;
    u.match_off = (int64_t)(((uint64_t)(uint32_t)uniform((int)c.y) << 32) | (uint32_t)uniform((int)c.x));
    const bool ok = (uint32_t)uniform((int)magic.x) == kTableMagic0 && (uint32_t)uniform((int)magic.y) == kTableMagic1;
    const int64_t t = (int64_t)(((uint64_t)(uint32_t)uniform((int)total.y) << 32) | (uint32_t)uniform((int)total.x));
    total_kp = ok ? t : (int64_t)-1;
    return u;
}

// load_frame_ranges in two halves, for a caller with loads of its own to request in between (they then travel beside these four
// instead of behind them; the counter of outstanding loads is in order, so theirs must come AFTER these)
struct FrameRangeWords { uint2 a0, a1, b0, b1; };
__device__ __forceinline__ FrameRangeWords request_frame_ranges(const int64_t* frame_off, int fa, int fb)
{
    const uint2* __restrict__ qa = reinterpret_cast<const uint2*>(frame_off + fa);
    const uint2* __restrict__ qb = reinterpret_cast<const uint2*>(frame_off + fb);
    FrameRangeWords w;
    w.a0 = qa[0]; w.a1 = qa[1]; w.b0 = qb[0]; w.b1 = qb[1];
    return w;
}
__device__ __forceinline__ void take_frame_ranges(FrameRangeWords w, int64_t& offA, int& nA, int64_t& offB, int& nB)
{
    asm volatile("" : "+v"(w.a0.x), "+v"(w.a0.y), "+v"(w.a1.x), "+v"(w.a1.y), "+v"(w.b0.x), "+v"(w.b0.y), "+v"(w.b1.x), "+v"(w.b1.y));
    auto s64 = [](const uint2& v) { return (int64_t)(((uint64_t)(uint32_t)uniform((int)v.y) << 32) | (uint32_t)uniform((int)v.x)); };
    offA = s64(w.a0);
    offB = s64(w.b0);
    nA = (int)(s64(w.a1) - offA);
    nB = (int)(s64(w.b1) - offB);
}

// first keypoint and number of keypoints of two frames of the table, in scalar registers: four loads requested together (see load_pair)
__device__ __forceinline__ void load_frame_ranges(const int64_t* frame_off, int fa, int fb, int64_t& offA, int& nA, int64_t& offB, int& nB)
{
    const uint2* __restrict__ qa = reinterpret_cast<const uint2*>(frame_off + fa);
    const uint2* __restrict__ qb = reinterpret_cast<const uint2*>(frame_off + fb);
    uint2 a0 = qa[0], a1 = qa[1], b0 = qb[0], b1 = qb[1];
    asm volatile("" : "+v"(a0.x), "+v"(a0.y), "+v"(a1.x), "+v"(a1.y), "+v"(b0.x), "+v"(b0.y), "+v"(b1.x), "+v"(b1.y));
    auto s64 = [](const uint2& v) { return (int64_t)(((uint64_t)(uint32_t)uniform((int)v.y) << 32) | (uint32_t)uniform((int)v.x)); };
    offA = s64(a0);
    offB = s64(b0);
    nA = (int)(s64(a1) - offA);
    nB = (int)(s64(b1) - offB);
}

// lane ^ 1 exchange on the VALU (DPP quad_perm [1,0,3,2]), no LDS round trip
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);
}

// sum of v over the wave's 64 lanes, in a scalar register: inclusive row scans on the DPP path (row_shr 1, 2, 4, 8 with zeros
// shifted in: lane 15 of every row of sixteen ends with the row's sum), then the four row sums by v_readlane
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 15) + (uint32_t)__builtin_amdgcn_readlane((int)v, 31) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 47) + (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

}  // namespace gms
