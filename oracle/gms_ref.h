/*
 * oracle/gms_ref.h -- TEST INFRASTRUCTURE, NOT PRODUCT (see gms_ref.c header: parity pinned only in part).
 * CPU restatement of the reference's cv::xfeatures2d::matchGMS (opencv_xfeatures2d452.dll).
 */
#ifndef GMS_REF_H
#define GMS_REF_H
#include "../include/gms.h" /* POD types and error codes only */

#ifdef __cplusplus
extern "C" {
#endif

extern const int gms_ref_rotation_patterns[8][9];
double gms_ref_scale_ratio(int s);

/* Whole call: matchGMS (DLL@0x180048280). mask_out (m bytes) and result may be NULL. */
int gms_ref_match(const gms_keypoint* kp1, int n1, int w1, int h1,
                  const gms_keypoint* kp2, int n2, int w2, int h2,
                  const gms_dmatch* matches, int m,
                  int with_rotation, int with_scale, double threshold_factor,
                  gms_dmatch* out, int* n_out, unsigned char* mask_out, gms_pair_result* result);

/* The same call on storage kept between calls (one scratch per host thread): identical arithmetic and loop order,
 * no allocation once the buffers have grown to the largest shape seen. */
typedef struct gms_ref_scratch gms_ref_scratch;
gms_ref_scratch* gms_ref_scratch_create(void);
void gms_ref_scratch_destroy(gms_ref_scratch* s);
int gms_ref_match_ws(gms_ref_scratch* s, const gms_keypoint* kp1, int n1, int w1, int h1,
                     const gms_keypoint* kp2, int n2, int w2, int h2,
                     const gms_dmatch* matches, int m,
                     int with_rotation, int with_scale, double threshold_factor,
                     gms_dmatch* out, int* n_out, unsigned char* mask_out, gms_pair_result* result);

/* Pieces, for pinning tests. */
int   gms_ref_grid_index_left(float nx, float ny, int type);          /* DLL@0x180047bc0 */
int   gms_ref_grid_index_right(float nx, float ny, int wr, int hr);   /* DLL@0x180047d60 */
void  gms_ref_right_grid(int scale, int* wr, int* hr);                /* DLL@0x180048c10 */
void  gms_ref_right_grid_from(int left_w, int left_h, int scale, int* wr, int* hr); /* DLL@0x180048c37 */
float gms_ref_normalize(float v, int extent);                         /* DLL@0x180048420 */
int   gms_ref_threshold_rejects(int T, int n, int score, double factor); /* DLL@0x180049171 */

void  gms_ref_neighbors(int gw, int gh, int* out);                      /* DLL@0x180048180 / 0x180048030 */

int   gms_ref_assign_pairs(const float* p1, const float* p2, const int* matches, int m, int wr, int hr,
                           int* pairs, int* nleft, int* motion);                 /* DLL@0x180047880 */
int   gms_ref_verify_cells(const int* motion, const int* nleft, int wr, int hr, int rotation_type, double factor,
                           int* cell_pairs_out);                                  /* DLL@0x180048d10 */
/* run()'s marking loop + count for one grid type (DLL@0x180048ae0-0x180048bd4); getInlierMask's loop nest on scripted run() results (DLL@0x180047dc0) */
int   gms_ref_selftest_mark(const int* pairs, const int* cell_pairs, int m, unsigned char* mask);
int   gms_ref_selftest_select(int with_rotation, int with_scale, int m, const int* counts, const unsigned char* masks,
                              unsigned char* best_mask, int* best, int* log, int* n_log);

/* gms_ref_mt.c: the same call over a batch of pairs, one pair per thread at a time (the algorithm
 * itself stays serial, as in the reference). Frames are (kp pointer, n, w, h) tables. Returns the
 * number of pairs that failed. Used only by bench.py's cpu_baseline leg and tests. */
int gms_ref_batch(const gms_keypoint* kp_all, const int64_t* frame_off, const int32_t* wh, int n_frames,
                  const gms_pair* pairs, int n_pairs, const gms_dmatch* matches,
                  int with_rotation, int with_scale, double threshold_factor,
                  gms_dmatch* out, gms_pair_result* results, unsigned char* mask, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
