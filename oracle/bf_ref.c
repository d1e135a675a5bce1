/*
 * bf_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT. CPU restatement of the brute-force descriptor matcher the
 * reference runs in front of matchGMS:
 *
 *   FeatureMatchUtil.cpp:66-68   Ptr<DescriptorMatcher> matcher = BFMatcher::create();   // NORM_L2, crossCheck = false
 *                                matcher->match(descriptors1, descriptors2, matches);    // one DMatch per query row
 *   (BASELINE config 2 words it as "10k ORB features + BFMatcher": the same call with NORM_HAMMING on 32-byte rows.)
 *
 * The algorithm lives in opencv_world452 (features2d BFMatcher -> core batchDistance), which the reference vendors as
 * a Windows import library only: parity unpinned (no fixture, no runnable binary). Restated from the published
 * algorithm: for every query row i, scan the train rows j = 0..N2-1 in order, keep the first strict minimum
 * (`d < best`), emit DMatch{queryIdx = i, trainIdx = j*, imgIdx = 0, distance}. NORM_HAMMING: popcount of the xor as
 * float. NORM_L2: sqrt of the sum of squared differences, all in fp32; the sum runs over k in index order here
 * (OpenCV's SIMD kernel sums in another order -- identical whenever every partial sum is exact, which holds for
 * SIFT's integer-valued 0..255 descriptors: 128 * 255^2 < 2^24).
 * Built with -ffp-contract=off -fno-fast-math.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { int32_t queryIdx, trainIdx, imgIdx; float distance; } bf_dmatch;

/* queries: n1 rows of 32 bytes, trains: n2 rows; out: n1 records (trainIdx -1, distance FLT_MAX when n2 == 0) */
int bf_ref_hamming256(const uint8_t* q, int n1, const uint8_t* t, int n2, bf_dmatch* out)
{
    if (n1 < 0 || n2 < 0) return -1;
    for (int i = 0; i < n1; ++i) {
        int best = 1 << 30, bj = -1;
        uint32_t a[8];
        memcpy(a, q + (size_t)i * 32, 32);
        for (int j = 0; j < n2; ++j) {
            uint32_t b[8];
            memcpy(b, t + (size_t)j * 32, 32);
            int d = 0;
            for (int w = 0; w < 8; ++w) d += __builtin_popcount(a[w] ^ b[w]);
            if (d < best) { best = d; bj = j; }
        }
        out[i].queryIdx = i;
        out[i].trainIdx = bj;
        out[i].imgIdx = 0;
        out[i].distance = bj >= 0 ? (float)best : 3.402823466e+38f;
    }
    return 0;
}

/* queries: n1 rows of `dim` floats, trains: n2 rows */
int bf_ref_l2(const float* q, int n1, const float* t, int n2, int dim, bf_dmatch* out)
{
    if (n1 < 0 || n2 < 0 || dim <= 0) return -1;
    for (int i = 0; i < n1; ++i) {
        float best = 3.402823466e+38f;
        int bj = -1;
        const float* a = q + (size_t)i * dim;
        for (int j = 0; j < n2; ++j) {
            const float* b = t + (size_t)j * dim;
            float s = 0.0f;
            for (int k = 0; k < dim; ++k) {
                const float d = a[k] - b[k];
                s = s + d * d;
            }
            if (s < best) { best = s; bj = j; }
        }
        out[i].queryIdx = i;
        out[i].trainIdx = bj;
        out[i].imgIdx = 0;
        out[i].distance = bj >= 0 ? sqrtf(best) : 3.402823466e+38f;
    }
    return 0;
}
