/* detect_ref.c -- TEST INFRASTRUCTURE ONLY (the checker of the keypoint source; never linked into or called by the product).
 *
 * CPU restatement of the build's own minimal keypoint detector / descriptor. The reference obtains its
 * keypoints from OpenCV (FeatureMatchUtil.cpp:9-12: SIFT::create(10000)->detectAndCompute; DisparityUtil.cpp:108,123-138:
 * ORB::create() / SIFT::create(), detectAndCompute or compute at every pixel) -- a binary dependency whose detector code is not in
 * /root/reference. This detector is therefore NOT a restatement of cv::ORB: it is a single-scale FAST-9 + steered-BRIEF design in
 * integer arithmetic, defined here, and "parity" for it means: the HIP kernels reproduce this file bit for bit. PARITY UNPINNED
 * against OpenCV by construction (nothing to pin to); the output FORMAT is the reference's (cv::KeyPoint records, 32-byte rows for
 * NORM_HAMMING), which is what the matcher and matchGMS consume.
 *
 * Definition (all integer):
 *   image      8-bit grey, row-major, pitch = width. A keypoint may sit at x in [16, W-16), y in [16, H-16).
 *   FAST-9     the 16-pixel Bresenham circle of radius 3; d_i = I(circle_i) - I(p). score = the largest t for which nine contiguous
 *              circle pixels are all >= p + t or all <= p - t: the max over the 16 arcs of the min over the arc's 9 values of d
 *              (bright) or of -d (dark), floored at 0. A candidate needs score > threshold and score > every one of its 8
 *              neighbours' scores (equal neighbours suppress each other).
 *   selection  at most max_keypoints: the highest scores, equal scores in raster order; OUTPUT in raster order (y, then x).
 *   smoothing  S = 5 x 5 box sum of I (0 where the box leaves the image).
 *   direction  m10 = sum dx I, m01 = sum dy I over the disc dx^2 + dy^2 <= 225; bin = arg max_k (m10 C_k + m01 S_k) over the 32
 *              directions (C_k, S_k) = round(4096 (cos, sin)(2 pi k / 32)), lowest k on ties; KeyPoint.angle = 11.25 * bin.
 *   descriptor 256 point pairs in the disc of radius 12 from the generator below, turned by the bin with the same table
 *              ((x C - y S + 2048) >> 12, (x S + y C + 2048) >> 12); bit k = S(p + a_k) < S(p + b_k); test k is bit k % 8 of byte k / 8.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DET_BORDER 16
#define DET_BINS 32
#define DET_TESTS 256

static const int kCos[DET_BINS] = {4096, 4017, 3784, 3406, 2896, 2276, 1567, 799, 0, -799, -1567, -2276, -2896, -3406, -3784, -4017,
                                   -4096, -4017, -3784, -3406, -2896, -2276, -1567, -799, 0, 799, 1567, 2276, 2896, 3406, 3784, 4017};
static const int kSin[DET_BINS] = {0, 799, 1567, 2276, 2896, 3406, 3784, 4017, 4096, 4017, 3784, 3406, 2896, 2276, 1567, 799,
                                   0, -799, -1567, -2276, -2896, -3406, -3784, -4017, -4096, -4017, -3784, -3406, -2896, -2276, -1567, -799};
static const int kCircle[16][2] = {{0, -3}, {1, -3}, {2, -2}, {3, -1}, {3, 0}, {3, 1}, {2, 2}, {1, 3},
                                   {0, 3}, {-1, 3}, {-2, 2}, {-3, 1}, {-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}};

typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } det_keypoint;   /* = cv::KeyPoint, 28 bytes */

static uint32_t g_state;
static int lcg_coord(void)
{
    g_state = g_state * 1664525u + 1013904223u;
    return (int)((g_state >> 16) % 25u) - 12;
}
static void lcg_point(int* x, int* y)
{
    do { *x = lcg_coord(); *y = lcg_coord(); } while (*x * *x + *y * *y > 144);
}
/* pattern[k] = {ax, ay, bx, by} of test k, unrotated */
void det_ref_pattern(int8_t* pattern /* [256][4] */)
{
    g_state = 0x2545F491u;
    for (int k = 0; k < DET_TESTS; ++k) {
        int ax, ay, bx, by;
        lcg_point(&ax, &ay);
        do { lcg_point(&bx, &by); } while (bx == ax && by == ay);
        pattern[4 * k + 0] = (int8_t)ax; pattern[4 * k + 1] = (int8_t)ay;
        pattern[4 * k + 2] = (int8_t)bx; pattern[4 * k + 3] = (int8_t)by;
    }
}
static int rot(int a, int b) { return (a + b + 2048) >> 12; }   /* (arithmetic shift: floor) */

int det_ref_fast_score(const uint8_t* img, int w, int x, int y)
{
    int d[16];
    const int p = img[(size_t)y * w + x];
    for (int i = 0; i < 16; ++i) d[i] = (int)img[(size_t)(y + kCircle[i][1]) * w + (x + kCircle[i][0])] - p;
    int best = 0;
    for (int s = 0; s < 16; ++s) {
        int lo = 255, hi = -255;
        for (int k = 0; k < 9; ++k) {
            const int v = d[(s + k) & 15];
            if (v < lo) lo = v;
            if (v > hi) hi = v;
        }
        if (lo > best) best = lo;      /* nine brighter pixels: the weakest of them */
        if (-hi > best) best = -hi;    /* nine darker pixels */
    }
    return best;
}

/* score image (0 outside the keypoint region) and the box sums */
void det_ref_maps(const uint8_t* img, int w, int h, uint8_t* score, uint16_t* box)
{
    memset(score, 0, (size_t)w * h);
    memset(box, 0, (size_t)w * h * 2);
    for (int y = DET_BORDER; y < h - DET_BORDER; ++y)
        for (int x = DET_BORDER; x < w - DET_BORDER; ++x) score[(size_t)y * w + x] = (uint8_t)det_ref_fast_score(img, w, x, y);
    for (int y = 2; y < h - 2; ++y)
        for (int x = 2; x < w - 2; ++x) {
            int s = 0;
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx) s += img[(size_t)(y + dy) * w + (x + dx)];
            box[(size_t)y * w + x] = (uint16_t)s;
        }
}

int det_ref_direction(const uint8_t* img, int w, int x, int y)
{
    long m10 = 0, m01 = 0;
    for (int dy = -15; dy <= 15; ++dy)
        for (int dx = -15; dx <= 15; ++dx)
            if (dx * dx + dy * dy <= 225) {
                const int v = img[(size_t)(y + dy) * w + (x + dx)];
                m10 += dx * v;
                m01 += dy * v;
            }
    int best = 0;
    long long best_dot = 0;
    for (int k = 0; k < DET_BINS; ++k) {
        const long long dot = (long long)m10 * kCos[k] + (long long)m01 * kSin[k];
        if (k == 0 || dot > best_dot) { best_dot = dot; best = k; }
    }
    return best;
}

void det_ref_describe_one(const uint16_t* box, int w, int x, int y, int bin, const int8_t* pattern, uint8_t* desc /* 32 */)
{
    memset(desc, 0, 32);
    const int c = kCos[bin], s = kSin[bin];
    for (int k = 0; k < DET_TESTS; ++k) {
        const int ax = pattern[4 * k], ay = pattern[4 * k + 1], bx = pattern[4 * k + 2], by = pattern[4 * k + 3];
        const int rax = rot(ax * c, -ay * s), ray = rot(ax * s, ay * c);
        const int rbx = rot(bx * c, -by * s), rby = rot(bx * s, by * c);
        const int va = box[(size_t)(y + ray) * w + (x + rax)], vb = box[(size_t)(y + rby) * w + (x + rbx)];
        if (va < vb) desc[k >> 3] |= (uint8_t)(1u << (k & 7));
    }
}

/* detect + describe. Returns the number of keypoints written (<= max_keypoints). */
int det_ref_detect(const uint8_t* img, int w, int h, int threshold, int max_keypoints, det_keypoint* kp, uint8_t* desc)
{
    if (w <= 2 * DET_BORDER || h <= 2 * DET_BORDER || max_keypoints <= 0) return 0;
    uint8_t* score = (uint8_t*)malloc((size_t)w * h);
    uint8_t* cand = (uint8_t*)calloc((size_t)w * h, 1);
    uint16_t* box = (uint16_t*)malloc((size_t)w * h * 2);
    int8_t pattern[DET_TESTS * 4];
    det_ref_pattern(pattern);
    det_ref_maps(img, w, h, score, box);
    long hist[256] = {0};
    long total = 0;
    for (int y = DET_BORDER; y < h - DET_BORDER; ++y)
        for (int x = DET_BORDER; x < w - DET_BORDER; ++x) {
            const int s = score[(size_t)y * w + x];
            if (s <= threshold) continue;
            int keep = 1;
            for (int dy = -1; dy <= 1 && keep; ++dy)
                for (int dx = -1; dx <= 1; ++dx)
                    if ((dx || dy) && score[(size_t)(y + dy) * w + (x + dx)] >= s) { keep = 0; break; }
            if (keep) { cand[(size_t)y * w + x] = (uint8_t)s; ++hist[s]; ++total; }
        }
    int cut = 0;          /* keep every score > cut and the first `quota` (raster order) with score == cut */
    long quota = 0;
    if (total > max_keypoints) {
        long above = 0;
        for (cut = 255; cut > 0; --cut) {
            if (above + hist[cut] >= max_keypoints) break;
            above += hist[cut];
        }
        quota = max_keypoints - above;
    }
    int n = 0;
    long eq_seen = 0;
    for (int y = DET_BORDER; y < h - DET_BORDER; ++y)
        for (int x = DET_BORDER; x < w - DET_BORDER; ++x) {
            const int s = cand[(size_t)y * w + x];
            if (s == 0) continue;
            int take = s > cut;
            if (!take && s == cut && eq_seen < quota) { take = 1; ++eq_seen; }
            if (!take) continue;
            const int bin = det_ref_direction(img, w, x, y);
            kp[n].x = (float)x; kp[n].y = (float)y; kp[n].size = 31.0f; kp[n].angle = 11.25f * (float)bin;
            kp[n].response = (float)s; kp[n].octave = 0; kp[n].class_id = -1;
            det_ref_describe_one(box, w, x, y, bin, pattern, desc + (size_t)n * 32);
            ++n;
        }
    free(score); free(cand); free(box);
    return n;
}

/* descriptors (and directions) at given keypoints: cv::Feature2D::compute as DisparityUtil.cpp:123-133 uses it. Every keypoint must
 * sit on an integer pixel inside the keypoint region; returns -1 otherwise (nothing written). */
int det_ref_describe(const uint8_t* img, int w, int h, det_keypoint* kp, int n, uint8_t* desc)
{
    for (int i = 0; i < n; ++i) {
        const int x = (int)kp[i].x, y = (int)kp[i].y;
        if ((float)x != kp[i].x || (float)y != kp[i].y || x < DET_BORDER || y < DET_BORDER || x >= w - DET_BORDER || y >= h - DET_BORDER) return -1;
    }
    uint8_t* score = (uint8_t*)malloc((size_t)w * h);
    uint16_t* box = (uint16_t*)malloc((size_t)w * h * 2);
    int8_t pattern[DET_TESTS * 4];
    det_ref_pattern(pattern);
    det_ref_maps(img, w, h, score, box);
    for (int i = 0; i < n; ++i) {
        const int x = (int)kp[i].x, y = (int)kp[i].y;
        const int bin = det_ref_direction(img, w, x, y);
        kp[i].angle = 11.25f * (float)bin;
        det_ref_describe_one(box, w, x, y, bin, pattern, desc + (size_t)i * 32);
    }
    free(score); free(box);
    return n;
}
