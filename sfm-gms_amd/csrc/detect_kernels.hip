// detect_kernels.hip -- the keypoint source in front of the matcher: single-scale FAST-9 corners + steered-BRIEF rows, for gfx950.
//
// What the reference does here is OpenCV's (FeatureMatchUtil.cpp:9-12 SIFT::create(10000)->detectAndCompute;
// DisparityUtil.cpp:108,123-138 ORB::create(), detectAndCompute or compute() at every pixel): a binary dependency. This is the
// build's own minimal detector in its place, defined in integer arithmetic so that every result is
// exact: same keypoints, same order, same bits on every run and against the CPU statement of the definition.
//
//   image      8-bit grey, row-major, pitch = width; n_images of one size back to back. Keypoints sit at x in [16, W-16), y in [16, H-16).
//   FAST-9     d_i = I(circle_i) - I(p) on the 16-pixel circle of radius 3; score = max over the 16 arcs of nine contiguous pixels of
//              min d (bright arc) or min -d (dark arc), floored at 0 = the largest t for which the pixel still is a FAST-9 corner.
//              Candidate: score > threshold and score > all 8 neighbours' scores.
//   selection  the max_keypoints highest scores, equal scores in raster order; output in raster order (no sort: a 256-bin histogram
//              gives the cut-off score, per-row counts and a scan give every survivor its slot).
//   smoothing  S = 5 x 5 box sum (u16).
//   direction  integer moments over the disc of radius 15, the best of 32 directions by integer dot product; angle = 11.25 * bin.
//   rows       256 comparisons S(p + a_k) < S(p + b_k) of a fixed pseudo-random pattern (disc of radius 12) turned by the bin.
// Output: cv::KeyPoint records {x, y, 31, angle, score, 0, -1} and 32-byte rows -- what gms_frame_table / gms_bf_prepare_device take.
//
// Kernels (one launch each, every image of the batch in grid.y / grid.z):
//   det_maps_kernel      64 x 16 pixel tile (+3 halo) through LDS: FAST score (u8) and box sum (u16) per pixel. HBM: 1 B in, 3 B out.
//   det_nms_kernel       3 x 3 non-maximum suppression on the score image -> candidate image, per-image score histogram.
//   det_cut_kernel       the cut-off score and the quota of candidates AT the cut-off, from the histogram.
//   det_rows_kernel      per image row: candidates above / at the cut-off.
//   det_scan_kernel      exclusive scan of those over the rows.
//   det_emit_kernel      per image row: survivors to their slots (x, y, score), raster order.
//   det_describe_kernel  one wave per keypoint: moments by wave reduction, direction, 256 comparisons by four ballots.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gms_kernels.h"

namespace gms {
namespace {

constexpr int kBorder = 16;
constexpr int kBins = 32;
constexpr int kTests = 256;
constexpr int kTileW = 64, kTileH = 16, kHalo = 3;

struct DirTable { int c[kBins], s[kBins]; };
__constant__ const DirTable kDir = {
    {4096, 4017, 3784, 3406, 2896, 2276, 1567, 799, 0, -799, -1567, -2276, -2896, -3406, -3784, -4017,
     -4096, -4017, -3784, -3406, -2896, -2276, -1567, -799, 0, 799, 1567, 2276, 2896, 3406, 3784, 4017},
    {0, 799, 1567, 2276, 2896, 3406, 3784, 4017, 4096, 4017, 3784, 3406, 2896, 2276, 1567, 799,
     0, -799, -1567, -2276, -2896, -3406, -3784, -4017, -4096, -4017, -3784, -3406, -2896, -2276, -1567, -799}};

// The comparison pattern, generated at compile time: a 32-bit linear congruential generator (1664525, 1013904223, seed 0x2545F491),
// coordinates (state >> 16) % 25 - 12, points outside the disc of radius 12 drawn again, b drawn again while it equals a.
struct Pattern { int8_t p[kTests][4]; };
constexpr Pattern make_pattern()
{
    Pattern t{};
    uint32_t state = 0x2545F491u;
    auto coord = [&state]() {
        state = state * 1664525u + 1013904223u;
        return (int)((state >> 16) % 25u) - 12;
    };
    for (int k = 0; k < kTests; ++k) {
        int ax = 0, ay = 0, bx = 0, by = 0;
        do { ax = coord(); ay = coord(); } while (ax * ax + ay * ay > 144);
        do {
            do { bx = coord(); by = coord(); } while (bx * bx + by * by > 144);
        } while (bx == ax && by == ay);
        t.p[k][0] = (int8_t)ax; t.p[k][1] = (int8_t)ay; t.p[k][2] = (int8_t)bx; t.p[k][3] = (int8_t)by;
    }
    return t;
}
__constant__ const Pattern kPattern = make_pattern();

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(a, min(b, c)); }

// ---- FAST score + box sum ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
det_maps_kernel(const uint8_t* __restrict__ images, int w, int h, uint8_t* __restrict__ score, uint16_t* __restrict__ box)
{
    constexpr int LW = kTileW + 2 * kHalo, LH = kTileH + 2 * kHalo;   // 70 x 22
    __shared__ uint8_t tile[LH][LW + 2];
    const size_t plane = (size_t)w * h;
    const uint8_t* __restrict__ img = images + (size_t)blockIdx.z * plane;
    const int x0 = (int)blockIdx.x * kTileW, y0 = (int)blockIdx.y * kTileH;
    for (int i = (int)threadIdx.x; i < LW * LH; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        const int x = min(max(x0 + lx - kHalo, 0), w - 1), y = min(max(y0 + ly - kHalo, 0), h - 1);   // (clamped reads: values there are never used)
        tile[ly][lx] = img[(size_t)y * w + x];
    }
    __syncthreads();
    const int lx = (int)threadIdx.x & 63, ry = (int)threadIdx.x >> 6;   // a thread: one column, rows ry, ry + 4, ry + 8, ry + 12
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ly = ry + 4 * j, x = x0 + lx, y = y0 + ly;
        if (x >= w || y >= h) continue;
        const int cx = lx + kHalo, cy = ly + kHalo;
        int sc = 0;
        if (x >= kBorder && x < w - kBorder && y >= kBorder && y < h - kBorder) {
            const int p = tile[cy][cx];
            int d[16];
            d[0] = tile[cy - 3][cx] - p;      d[1] = tile[cy - 3][cx + 1] - p;  d[2] = tile[cy - 2][cx + 2] - p;  d[3] = tile[cy - 1][cx + 3] - p;
            d[4] = tile[cy][cx + 3] - p;      d[5] = tile[cy + 1][cx + 3] - p;  d[6] = tile[cy + 2][cx + 2] - p;  d[7] = tile[cy + 3][cx + 1] - p;
            d[8] = tile[cy + 3][cx] - p;      d[9] = tile[cy + 3][cx - 1] - p;  d[10] = tile[cy + 2][cx - 2] - p; d[11] = tile[cy + 1][cx - 3] - p;
            d[12] = tile[cy][cx - 3] - p;     d[13] = tile[cy - 1][cx - 3] - p; d[14] = tile[cy - 2][cx - 2] - p; d[15] = tile[cy - 3][cx - 1] - p;
            // window minima / maxima by doubling: 2, 4, 8, then the ninth value
            int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { lo2[i] = min(d[i], d[(i + 1) & 15]); hi2[i] = max(d[i], d[(i + 1) & 15]); }
#pragma unroll
            for (int i = 0; i < 16; ++i) { lo4[i] = min(lo2[i], lo2[(i + 2) & 15]); hi4[i] = max(hi2[i], hi2[(i + 2) & 15]); }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int lo9 = min3i(lo4[i], lo4[(i + 4) & 15], d[(i + 8) & 15]);
                const int hi9 = max(max(hi4[i], hi4[(i + 4) & 15]), d[(i + 8) & 15]);
                sc = max(sc, max(lo9, -hi9));
            }
        }
        score[(size_t)blockIdx.z * plane + (size_t)y * w + x] = (uint8_t)sc;
        int bs = 0;
        if (x >= 2 && x < w - 2 && y >= 2 && y < h - 2) {
#pragma unroll
            for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
                for (int dx = -2; dx <= 2; ++dx) bs += tile[cy + dy][cx + dx];
        }
        box[(size_t)blockIdx.z * plane + (size_t)y * w + x] = (uint16_t)bs;
    }
}

// ---- non-maximum suppression + histogram of the candidates' scores --------------------------------------------------------------------
__global__ void __launch_bounds__(256)
det_nms_kernel(const uint8_t* __restrict__ score, int w, int h, int threshold, uint8_t* __restrict__ cand, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const size_t plane = (size_t)w * h;
    const uint8_t* __restrict__ sc = score + (size_t)blockIdx.z * plane;
    const int x = (int)blockIdx.x * 64 + ((int)threadIdx.x & 63), y = (int)blockIdx.y * 4 + ((int)threadIdx.x >> 6);
    if (x < w && y < h) {
        int keep = 0;
        if (x >= kBorder && x < w - kBorder && y >= kBorder && y < h - kBorder) {   // (so every neighbour is inside the image)
            const int s = sc[(size_t)y * w + x];
            if (s > threshold) {
                const uint8_t* r0 = sc + (size_t)(y - 1) * w + x;
                const uint8_t* r1 = r0 + w;
                const uint8_t* r2 = r1 + w;
                const int m = max(max(max(r0[-1], r0[0]), max(r0[1], r1[-1])), max(max(r1[1], r2[-1]), max(r2[0], r2[1])));
                if (s > m) keep = s;
            }
        }
        cand[(size_t)blockIdx.z * plane + (size_t)y * w + x] = (uint8_t)keep;
        if (keep) atomicAdd(&lh[keep], 1u);
    }
    __syncthreads();
    if (lh[threadIdx.x] != 0) atomicAdd(&hist[(size_t)blockIdx.z * 256 + threadIdx.x], lh[threadIdx.x]);
}

// cut[img] = {cut-off score, quota at the cut-off, keypoints kept, candidates}
__global__ void __launch_bounds__(256)
det_cut_kernel(const uint32_t* __restrict__ hist, int max_keypoints, int32_t* __restrict__ cut, int32_t* __restrict__ counts)
{
    __shared__ uint32_t above[257];   // above[s] = candidates with score >= s
    const uint32_t* hh = hist + (size_t)blockIdx.x * 256;
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        above[256] = 0;
        for (int s = 255; s >= 0; --s) {
            run += hh[s];
            above[s] = run;
        }
    }
    __syncthreads();
    const uint32_t total = above[0];
    int32_t* out = cut + (size_t)blockIdx.x * 4;
    if (total <= (uint32_t)max_keypoints) {
        if (threadIdx.x == 0) {
            out[0] = 0; out[1] = 0; out[2] = (int32_t)total; out[3] = (int32_t)total;
            counts[blockIdx.x] = (int32_t)total;
        }
        return;
    }
    // the score s with above[s + 1] < max_keypoints <= above[s] (unique; s >= 1 because no candidate scores 0)
    const int s = (int)threadIdx.x;
    if (above[s + 1] < (uint32_t)max_keypoints && above[s] >= (uint32_t)max_keypoints) {
        out[0] = s; out[1] = max_keypoints - (int32_t)above[s + 1]; out[2] = max_keypoints; out[3] = (int32_t)total;
        counts[blockIdx.x] = max_keypoints;
    }
}

// per row: [0] candidates above the cut-off, [1] at it
__global__ void __launch_bounds__(64)
det_rows_kernel(const uint8_t* __restrict__ cand, int w, int h, const int32_t* __restrict__ cut, uint32_t* __restrict__ rows)
{
    const int y = (int)blockIdx.x, img = (int)blockIdx.y, lane = (int)threadIdx.x;
    const int c = cut[(size_t)img * 4];
    const uint8_t* __restrict__ row = cand + (size_t)img * w * h + (size_t)y * w;
    uint32_t gt = 0, eq = 0;
    for (int x = lane; x < w; x += 64) {
        const int s = row[x];
        gt += s > c ? 1u : 0u;
        eq += (s == c && s != 0) ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        gt += (uint32_t)__shfl_xor((int)gt, d);
        eq += (uint32_t)__shfl_xor((int)eq, d);
    }
    if (lane == 0) {
        rows[((size_t)img * h + y) * 2] = gt;
        rows[((size_t)img * h + y) * 2 + 1] = eq;
    }
}

// exclusive scan over the rows, in place: rows[y] = {candidates above the cut-off before row y, candidates at it before row y}
__global__ void __launch_bounds__(1024)
det_scan_kernel(uint32_t* __restrict__ rows, int h)
{
    __shared__ uint32_t part[2][1024];
    uint32_t* r = rows + (size_t)blockIdx.x * h * 2;
    const int tid = (int)threadIdx.x, per = (h + 1023) / 1024;
    uint32_t a = 0, b = 0;
    for (int i = 0; i < per; ++i) {
        const int y = tid * per + i;
        if (y < h) { a += r[2 * y]; b += r[2 * y + 1]; }
    }
    part[0][tid] = a; part[1][tid] = b;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {   // Hillis-Steele inclusive scan
        const uint32_t pa = tid >= d ? part[0][tid - d] : 0u, pb = tid >= d ? part[1][tid - d] : 0u;
        __syncthreads();
        part[0][tid] += pa; part[1][tid] += pb;
        __syncthreads();
    }
    uint32_t ra = part[0][tid] - a, rb = part[1][tid] - b;   // exclusive prefix of this thread's rows
    for (int i = 0; i < per; ++i) {
        const int y = tid * per + i;
        if (y < h) {
            const uint32_t ca = r[2 * y], cb = r[2 * y + 1];
            r[2 * y] = ra; r[2 * y + 1] = rb;
            ra += ca; rb += cb;
        }
    }
}

// survivors of a row to their slots: slot = (above before) + min(at-cut before, quota) + rank inside the row
__global__ void __launch_bounds__(64)
det_emit_kernel(const uint8_t* __restrict__ cand, int w, int h, const int32_t* __restrict__ cut, const uint32_t* __restrict__ rows,
                int max_keypoints, uint32_t* __restrict__ list)
{
    const int y = (int)blockIdx.x, img = (int)blockIdx.y, lane = (int)threadIdx.x;
    const int c = cut[(size_t)img * 4];
    const uint32_t quota = (uint32_t)cut[(size_t)img * 4 + 1];
    uint32_t gt_run = rows[((size_t)img * h + y) * 2], eq_run = rows[((size_t)img * h + y) * 2 + 1];
    const uint8_t* __restrict__ row = cand + (size_t)img * w * h + (size_t)y * w;
    uint32_t* __restrict__ out = list + (size_t)img * max_keypoints * 2;
    const uint64_t below = (1ull << lane) - 1ull;
    for (int x0 = 0; x0 < w; x0 += 64) {
        const int x = x0 + lane;
        const int s = x < w ? row[x] : 0;
        const bool gt = s > c, eq = s == c && s != 0;
        const uint64_t mg = __ballot(gt), me = __ballot(eq);
        const uint32_t eq_rank = eq_run + (uint32_t)__popcll(me & below);
        const bool take = gt || (eq && eq_rank < quota);
        if (take) {
            const uint32_t eq_before = min(eq_run + (uint32_t)__popcll(me & below), quota);   // at-cut survivors before this one
            const uint32_t slot = gt_run + (uint32_t)__popcll(mg & below) + eq_before;
            if (slot < (uint32_t)max_keypoints) {
                out[2 * slot] = (uint32_t)x | ((uint32_t)y << 16);
                out[2 * slot + 1] = (uint32_t)s;
            }
        }
        gt_run += (uint32_t)__popcll(mg);
        eq_run += (uint32_t)__popcll(me);
    }
}

// ---- direction + descriptor: one wave per keypoint ---------------------------------------------------------------------------------------
// FROM_LIST: keypoints come from det_emit_kernel's list (and the record is written here); otherwise from caller's records (compute()).
template <bool FROM_LIST>
__global__ void __launch_bounds__(256)
det_describe_kernel(const uint8_t* __restrict__ images, const uint16_t* __restrict__ box, int w, int h, const uint32_t* __restrict__ list,
                    const int32_t* __restrict__ counts, int max_keypoints, gms_keypoint* __restrict__ kp, uint8_t* __restrict__ desc,
                    int32_t* __restrict__ status)
{
    const int lane = (int)threadIdx.x & 63, img = (int)blockIdx.y;
    const int k = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    const int n = counts != nullptr ? min(counts[img], max_keypoints) : max_keypoints;
    if (k >= n) return;   // wave-uniform
    const size_t plane = (size_t)w * h;
    const uint8_t* __restrict__ im = images + (size_t)img * plane;
    const uint16_t* __restrict__ bx = box + (size_t)img * plane;
    gms_keypoint* rec = kp + (size_t)img * max_keypoints + k;
    int x, y, sc = 0;
    if constexpr (FROM_LIST) {
        const uint32_t xy = list[((size_t)img * max_keypoints + k) * 2];
        x = (int)(xy & 0xFFFFu); y = (int)(xy >> 16);
        sc = (int)list[((size_t)img * max_keypoints + k) * 2 + 1];
    } else {
        const float fx = rec->x, fy = rec->y;
        x = (int)fx; y = (int)fy;
        const bool ok = (float)x == fx && (float)y == fy && x >= kBorder && y >= kBorder && x < w - kBorder && y < h - kBorder;
        if (!ok) {   // wave-uniform; the caller sees the flag, the row stays as it was
            if (lane == 0) atomicMax(status, 1);
            return;
        }
    }
    // moments over the disc of radius 15: 31 x 31 positions, 64 at a time
    int m10 = 0, m01 = 0;
    for (int i = lane; i < 31 * 31; i += 64) {
        const int dy = i / 31 - 15, dx = i % 31 - 15;
        if (dx * dx + dy * dy <= 225) {
            const int v = im[(size_t)(y + dy) * w + (x + dx)];
            m10 += dx * v;
            m01 += dy * v;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        m10 += __shfl_xor(m10, d);
        m01 += __shfl_xor(m01, d);
    }
    // the best of 32 directions: key = dot << 5 | (31 - k), so the maximum is the largest dot product, then the lowest k
    const int kk = lane & 31;
    long long key = (((long long)m10 * kDir.c[kk] + (long long)m01 * kDir.s[kk]) << 5) | (long long)(31 - kk);
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) {
        const long long o = __shfl_xor(key, d);
        key = o > key ? o : key;
    }
    const int bin = 31 - (int)(key & 31);
    const int c = kDir.c[bin], s = kDir.s[bin];
    uint64_t bits[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = 64 * j + lane;
        const int ax = kPattern.p[t][0], ay = kPattern.p[t][1], bxx = kPattern.p[t][2], by = kPattern.p[t][3];
        const int rax = (ax * c - ay * s + 2048) >> 12, ray = (ax * s + ay * c + 2048) >> 12;
        const int rbx = (bxx * c - by * s + 2048) >> 12, rby = (bxx * s + by * c + 2048) >> 12;
        const int va = bx[(size_t)(y + ray) * w + (x + rax)], vb = bx[(size_t)(y + rby) * w + (x + rbx)];
        bits[j] = __ballot(va < vb);
    }
    if (lane < 4) reinterpret_cast<uint64_t*>(desc + ((size_t)img * max_keypoints + k) * 32)[lane] = bits[lane & 3];
    if (lane == 0) {
        if constexpr (FROM_LIST) {
            gms_keypoint r;
            r.x = (float)x; r.y = (float)y; r.size = 31.0f; r.angle = 11.25f * (float)bin; r.response = (float)sc; r.octave = 0; r.class_id = -1;
            *rec = r;
        } else {
            rec->angle = 11.25f * (float)bin;
        }
    }
}

}  // namespace

// workspace of a batch: score | candidates | box sums | histogram | cut | row counts | list
size_t detect_workspace_bytes(int w, int h, int n_images, int max_keypoints)
{
    if (w <= 0 || h <= 0 || n_images <= 0 || max_keypoints < 0) return 0;
    const size_t plane = (size_t)w * h;
    size_t b = 0;
    b += ((plane * n_images + 255) & ~(size_t)255) * 2;        // score, candidates
    b += (plane * n_images * 2 + 255) & ~(size_t)255;           // box sums
    b += (size_t)n_images * 256 * 4 + (size_t)n_images * 16;    // histogram, cut
    b += ((size_t)n_images * h * 8 + 255) & ~(size_t)255;       // row counts
    b += ((size_t)n_images * max_keypoints * 8 + 255) & ~(size_t)255;   // list
    return b;
}

namespace {
struct DetWs { uint8_t* score; uint8_t* cand; uint16_t* box; uint32_t* hist; int32_t* cut; uint32_t* rows; uint32_t* list; };
DetWs carve(void* ws, int w, int h, int n_images, int max_keypoints)
{
    const size_t plane = (size_t)w * h, a = (plane * n_images + 255) & ~(size_t)255;
    char* p = reinterpret_cast<char*>(ws);
    DetWs d;
    d.score = reinterpret_cast<uint8_t*>(p); p += a;
    d.cand = reinterpret_cast<uint8_t*>(p); p += a;
    d.box = reinterpret_cast<uint16_t*>(p); p += (plane * n_images * 2 + 255) & ~(size_t)255;
    d.hist = reinterpret_cast<uint32_t*>(p); p += (size_t)n_images * 256 * 4;
    d.cut = reinterpret_cast<int32_t*>(p); p += (size_t)n_images * 16;
    d.rows = reinterpret_cast<uint32_t*>(p); p += ((size_t)n_images * h * 8 + 255) & ~(size_t)255;
    d.list = reinterpret_cast<uint32_t*>(p);
    (void)max_keypoints;
    return d;
}
}  // namespace

hipError_t launch_detect(const uint8_t* d_images, int n_images, int w, int h, int threshold, int max_keypoints, void* d_ws,
                         gms_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts, hipStream_t stream)
{
    if (n_images <= 0) return hipSuccess;
    const DetWs ws = carve(d_ws, w, h, n_images, max_keypoints);
    hipError_t e = hipMemsetAsync(ws.hist, 0, (size_t)n_images * 256 * 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(det_maps_kernel, dim3((w + kTileW - 1) / kTileW, (h + kTileH - 1) / kTileH, n_images), dim3(256), 0, stream, d_images, w, h,
                       ws.score, ws.box);
    hipLaunchKernelGGL(det_nms_kernel, dim3((w + 63) / 64, (h + 3) / 4, n_images), dim3(256), 0, stream, ws.score, w, h, threshold, ws.cand, ws.hist);
    hipLaunchKernelGGL(det_cut_kernel, dim3(n_images), dim3(256), 0, stream, ws.hist, max_keypoints, ws.cut, d_counts);
    hipLaunchKernelGGL(det_rows_kernel, dim3(h, n_images), dim3(64), 0, stream, ws.cand, w, h, ws.cut, ws.rows);
    hipLaunchKernelGGL(det_scan_kernel, dim3(n_images), dim3(1024), 0, stream, ws.rows, h);
    hipLaunchKernelGGL(det_emit_kernel, dim3(h, n_images), dim3(64), 0, stream, ws.cand, w, h, ws.cut, ws.rows, max_keypoints, ws.list);
    if (max_keypoints > 0)
        hipLaunchKernelGGL(det_describe_kernel<true>, dim3((max_keypoints + 3) / 4, n_images), dim3(256), 0, stream, d_images, ws.box, w, h, ws.list,
                           d_counts, max_keypoints, d_kp, d_desc, (int32_t*)nullptr);
    return hipGetLastError();
}

// compute(): directions and rows at the caller's keypoints of ONE image; *d_status = 1 when a keypoint is off the pixel grid or outside the
// keypoint region (its row is left alone)
hipError_t launch_describe(const uint8_t* d_image, int w, int h, gms_keypoint* d_kp, int n, void* d_ws, uint8_t* d_desc, int32_t* d_status,
                           hipStream_t stream)
{
    const DetWs ws = carve(d_ws, w, h, 1, 0);
    hipError_t e = hipMemsetAsync(d_status, 0, 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(det_maps_kernel, dim3((w + kTileW - 1) / kTileW, (h + kTileH - 1) / kTileH, 1), dim3(256), 0, stream, d_image, w, h, ws.score, ws.box);
    if (n > 0)
        hipLaunchKernelGGL(det_describe_kernel<false>, dim3((n + 3) / 4, 1), dim3(256), 0, stream, d_image, ws.box, w, h, (const uint32_t*)nullptr,
                           (const int32_t*)nullptr, n, d_kp, d_desc, d_status);
    return hipGetLastError();
}

}  // namespace gms
