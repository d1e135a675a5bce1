/*
 * consumer_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT. CPU restatement of the two consumers of matchGMS's output in the
 * reference, first-party code this time (the source is in the tree):
 *   disp_ref_map_and_rms   SfM-GMS/SfM-GMS/DisparityUtil.cpp:179-201 (matchBasedDispCalculate): the disparity map and the
 *                          RMS statistics against the ground truth. `long float` is double under MSVC.
 *   sfm_ref_gather         SfM-GMS/SfM-GMS/SfMUtil.cpp:25-35 (structureFromMotion): the matched-point arrays.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } ref_keypoint;
typedef struct { int32_t queryIdx, trainIdx, imgIdx; float distance; } ref_dmatch;

/* returns 0, or -2 when a match indexes outside the keypoints or lands outside the image (undefined in the reference) */
int disp_ref_map_and_rms(const ref_keypoint* kp1, int n1, const ref_keypoint* kp2, int n2, const ref_dmatch* matches, int m,
                         int w, int h, const uint8_t* gt, int disp_ratio, uint8_t* disparity, int64_t* count, int64_t* sum_sq,
                         int32_t* max_abs, double* rms)
{
    memset(disparity, 255, (size_t)w * h);                                  /* Mat disparity(h, w, CV_8U, Scalar(255))   :180 */
    for (int i = 0; i < m; ++i) {                                           /* :181 */
        if (matches[i].queryIdx < 0 || matches[i].queryIdx >= n1 || matches[i].trainIdx < 0 || matches[i].trainIdx >= n2) return -2;
        const int x = (int)kp1[matches[i].queryIdx].x;                      /* :183 */
        const int y = (int)kp1[matches[i].queryIdx].y;                      /* :184 */
        const int x1 = (int)kp2[matches[i].trainIdx].x;                     /* :185 */
        if (x < 0 || x >= w || y < 0 || y >= h) return -2;
        disparity[(size_t)y * w + x] = (uint8_t)abs(x - x1);                /* :186 */
    }
    double r = 0, c = 0, mx = 0;                                            /* long float rms = 0, count = 0, max_disp = 0  :188 */
    if (gt) {
        for (int i = 0; i < w; ++i)                                         /* :189 */
            for (int j = 0; j < h; ++j)
                if (disparity[(size_t)j * w + i] != 255) {                  /* :191 */
                    const int a = abs(disparity[(size_t)j * w + i] - gt[(size_t)j * w + i] / disp_ratio);   /* :193 */
                    if (a > mx) mx = a;
                    r = r + a * a;                                          /* :196 */
                    c = c + 1;
                }
    }
    *count = (int64_t)c;
    *sum_sq = (int64_t)r;
    *max_abs = (int32_t)mx;
    *rms = sqrt(r / c);                                                     /* :201 (NaN when nothing matched, as in the reference) */
    return 0;
}

int sfm_ref_gather(const ref_keypoint* kp1, int n1, const ref_keypoint* kp2, int n2, const ref_dmatch* matches, int m,
                   float* coords1, float* coords2)
{
    for (int i = 0; i < m; ++i) {                                           /* SfMUtil.cpp:26 */
        if (matches[i].queryIdx < 0 || matches[i].queryIdx >= n1 || matches[i].trainIdx < 0 || matches[i].trainIdx >= n2) return -2;
        coords1[2 * i] = kp1[matches[i].queryIdx].x;                        /* :28-30 */
        coords1[2 * i + 1] = kp1[matches[i].queryIdx].y;
        coords2[2 * i] = kp2[matches[i].trainIdx].x;                        /* :32-34 */
        coords2[2 * i + 1] = kp2[matches[i].trainIdx].y;
    }
    return 0;
}
