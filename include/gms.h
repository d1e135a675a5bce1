/*
 * gms.h -- C ABI of the MI355X-native GMS (Grid-based Motion Statistics) match filter.
 *
 * Drop-in boundary for ONE reference call:
 *
 *   cv::xfeatures2d::matchGMS(size1, size2, keypoints1, keypoints2, matches1to2, matchesGMS,
 *                             withRotation=false, withScale=false, thresholdFactor=6.0)
 *
 * as called by the reference at
 *   SfM-GMS/SfM-GMS/FeatureMatchUtil.cpp:69      (withRotation=true, withScale=true, 6.0)
 *   SfM-GMS/SfM-GMS/DisparityUtil.cpp:149, :299  (defaults: false, false, 6.0)
 * and implemented (binary only) in SfM-GMS/bin/opencv_xfeatures2d452.dll, export ordinal 884,
 * RVA 0x48280 (opencv_contrib xfeatures2d 4.5.2, class GMSMatcher).
 *
 * Everything here is plain C: pointers, sizes, PODs. No torch, no OpenCV, no C++ types.
 * The work behind every entry point is done by hand-written HIP kernels for gfx950; there is no
 * CPU fallback in this library (a missing/failed GPU is an error code, never a silent detour).
 */
#ifndef MI355_GMS_H
#define MI355_GMS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (the reference signals nothing: void return, UB on bad input) ------------- */
#define GMS_OK             0
#define GMS_ERR_BAD_ARG   (-1) /* null pointer / negative size / zero image dimension            */
#define GMS_ERR_DOMAIN    (-2) /* input outside the domain on which the reference is defined     */
#define GMS_ERR_HIP       (-3) /* a HIP runtime call failed (gms_last_hip_error() has the code)  */
#define GMS_ERR_NO_DEVICE (-4) /* no usable gfx950 device                                        */
#define GMS_ERR_CAPACITY  (-5) /* m exceeds what this build supports (see gms_max_matches())     */
#define GMS_ERR_IO        (-7) /* gms_dataset_read / _write: cannot open, short file, or not a GMSFRM01 file             */
#define GMS_ERR_NO_MODEL  (-8) /* two-view stage: fewer than five correspondences, or no essential matrix / pose could be estimated */
#define GMS_ERR_NOT_RESERVED (-6) /* a workspace would have to grow while the stream is being captured: call
                                     gms_ctx_reserve() for this shape first                          */

/* ---- PODs, bit-compatible with the reference's element types ------------------------------- */

/* cv::KeyPoint: 28 bytes, stride 0x1c at DLL@0x1800485d4; only pt.x (+0) and pt.y (+4) are read. */
typedef struct gms_keypoint {
    float x, y;      /* pt                                                      */
    float size;      /* ignored by GMS                                          */
    float angle;     /* ignored                                                 */
    float response;  /* ignored                                                 */
    int32_t octave;  /* ignored                                                 */
    int32_t class_id;/* ignored                                                 */
} gms_keypoint;

/* cv::DMatch: 16 bytes, stride 0x10; queryIdx (+0) and trainIdx (+4) are read (DLL@0x180046aa3),
 * the whole struct is copied verbatim to the output (DLL@0x18004836a). */
typedef struct gms_dmatch {
    int32_t queryIdx;
    int32_t trainIdx;
    int32_t imgIdx;   /* opaque payload, carried through */
    float   distance; /* opaque payload, carried through */
} gms_dmatch;

/* One image pair of a batch: which two resident frames, and where its putative matches live. */
typedef struct gms_pair {
    int32_t frame_a;    /* index of the query ("left") frame  -> keypoints1 / size1 */
    int32_t frame_b;    /* index of the train ("right") frame -> keypoints2 / size2 */
    int32_t m;          /* number of putative matches of this pair                  */
    int32_t reserved;   /* must be 0                                                */
    int64_t match_off;  /* element offset of the pair's first gms_dmatch in the match (and out) array */
} gms_pair;

/* Per-pair result record. */
typedef struct gms_pair_result {
    int32_t n_inliers;  /* number of gms_dmatch written for this pair                           */
    int32_t best_scale; /* 0..4 index into {1, 1/2, 1/sqrt2, sqrt2, 2}; -1 if no hypothesis won  */
    int32_t best_rot;   /* 1..8 rotation pattern; -1 if no hypothesis won                        */
    int32_t status;     /* GMS_OK, GMS_ERR_DOMAIN (input outside the reference's domain) or GMS_ERR_BAD_ARG (the pair's match range overlaps another pair's) */
} gms_pair_result;

typedef struct gms_ctx gms_ctx;

/* ---- one-shot drop-in ------------------------------------------------------------------------
 * Same contract as the reference call: inputs borrowed, `out` must have room
 * for m entries, receives the surviving matches verbatim and in input order, *n_out their number.
 * Host pointers. Uses a lazily created process-wide context on device 0.
 * Replaces: cv::xfeatures2d::matchGMS (FeatureMatchUtil.cpp:69; DisparityUtil.cpp:149,299). */
int gms_match(const gms_keypoint* kp1, int n1, int w1, int h1,
              const gms_keypoint* kp2, int n2, int w2, int h2,
              const gms_dmatch* matches, int m,
              int with_rotation, int with_scale, double threshold_factor,
              gms_dmatch* out, int* n_out);

/* Same, on an explicit context; additionally reports the winning hypothesis (may be NULL). */
int gms_match_ctx(gms_ctx* ctx,
                  const gms_keypoint* kp1, int n1, int w1, int h1,
                  const gms_keypoint* kp2, int n2, int w2, int h2,
                  const gms_dmatch* matches, int m,
                  int with_rotation, int with_scale, double threshold_factor,
                  gms_dmatch* out, int* n_out, gms_pair_result* result);

/* ---- context --------------------------------------------------------------------------------
 * One context per (process, device). Every entry point that takes a context serialises on it (a context may be
 * shared by host threads); work is ordered on the context's current stream. If the stream is changed while earlier
 * launches are still running, later launches that reuse the context's internal workspaces wait for them (an event). */
int  gms_ctx_create(int device, gms_ctx** out_ctx);
int  gms_ctx_destroy(gms_ctx* ctx);
/* Launch on a caller-owned hipStream_t (pass it as void*); NULL selects the context's own stream. */
int  gms_ctx_set_stream(gms_ctx* ctx, void* hip_stream);
int  gms_ctx_synchronize(gms_ctx* ctx);
/* Sizes the internal workspaces for gms_filter_device calls of up to n_pairs pairs of up to max_m matches under the
 * given flags. After it, such calls neither allocate nor synchronise -- they are pure stream-ordered launches and can
 * be captured into a hipGraph. May allocate and synchronise itself. */
int  gms_ctx_reserve(gms_ctx* ctx, int n_pairs, int max_m, int with_rotation, int with_scale);
/* What the context's most recent gms_filter_device launch ran with, for reports (bench.py prints them beside its numbers). The
 * library picks between bit-identical kernel variants from what earlier launches of the context saw: a probe kernel behind
 * every sixteenth launch writes a verdict, and the first later launch that finds that kernel complete adopts it. */
#define GMS_QUERY_LAST_DEALT        1 /* 1: the byte-matrix kernel dealt the matches to its lanes (inputs in spatial order) */
#define GMS_QUERY_LAST_SCALE_PROBE  2 /* bit s set: scale hypothesis s was bounded by a probe before being evaluated; bit 8 + s: with four-bit entries first */
#define GMS_QUERY_LAST_KPT          3 /* matches per thread of the workgroup kernel (0: the large-pair kernels ran)        */
#define GMS_QUERY_LAUNCHES          4 /* filter launches of the context so far                                             */
#define GMS_QUERY_CUS               5 /* compute units of the context's device                                             */
#define GMS_QUERY_PREFETCH_TYPE     6 /* byte-matrix kernel: grid type (0..3) before which a workgroup touches the match records of
                                         the pair its CU's next workgroup will filter, so that they wait in L2 (-1: never)        */
#define GMS_QUERY_PREFETCH_AHEAD    7 /* ... how many pairs ahead that pair is (the number of CUs unless GMS_PREFETCH says otherwise) */
#define GMS_QUERY_STAGGER_TICKS     8 /* first-round start spread of the most recent launch, in 10 ns ticks (0: none)             */
int  gms_ctx_query(gms_ctx* ctx, int what, int64_t* value);
/* Forces one of those choices for the context's later launches (value 0 / 1), or hands it back to the library (-1, the default).
 * Speed only: every variant produces the same bytes. The environment switches GMS_DEAL / GMS_SCALE_PROBE do the same process-wide. */
#define GMS_OPTION_DEAL         1
#define GMS_OPTION_SCALE_PROBE  2
int  gms_ctx_set_option(gms_ctx* ctx, int option, int value);

/* ---- device-resident batch path (throughput API) ---------------------------------------------
 * All d_* pointers are device pointers on the context's device; calls are stream-ordered on the
 * context's stream. gms_normalize_device never synchronises. gms_filter_device does not synchronise or allocate
 * once the shape has been reserved (gms_ctx_reserve); without a reservation it grows its workspaces on first
 * use of a larger shape (one hipStreamSynchronize + hipMalloc then), and returns GMS_ERR_NOT_RESERVED instead
 * if that would have to happen inside a stream capture. Batches of pairs up to 16 384 matches under the
 * default flags need no workspace at all.
 *
 * gms_normalize_device: GMSMatcher::normalizePoints (DLL@0x180048420) for every keypoint of every
 * frame: d_pts[2*i] = kp[i].x / (float)w[frame], d_pts[2*i+1] = kp[i].y / (float)h[frame]
 * (IEEE fp32 divide). d_frame_off has n_frames+1 entries (keypoint offsets), d_wh 2*n_frames ints.
 * d_pts is the frame table the filter works from and needs gms_frame_table_bytes(total_kp) bytes (16 per keypoint + 32):
 * a 16-byte header (GMS_FRAME_TABLE_HEADER_BYTES: a magic word and total_kp, so that the table describes itself), the
 * normalised points (8 bytes each, point i at float index 4 + 2 i), then 8 bytes of cell codes per keypoint -- everything
 * about a keypoint that does not depend on the pair it is matched in (its cells on the left grid's four half-cell shifted
 * types and on the right grids of setScale) is worked out once per frame here, not once per pair. Opaque beyond the points;
 * always pass the block back whole (16-byte aligned). gms_filter_device may be given any n_frames / d_frame_off whose frames lie
 * inside the table (a prefix of the frames, say): where the code arrays are is read from the header, not derived from the call. */
#define GMS_FRAME_TABLE_HEADER_BYTES 16
int64_t gms_frame_table_bytes(int64_t total_kp);
int gms_normalize_device(gms_ctx* ctx, const gms_keypoint* d_kp, const int64_t* d_frame_off,
                         const int32_t* d_wh, int n_frames, int64_t total_kp, float* d_pts);

/* gms_filter_device: the GMS filter proper (GMSMatcher ctor..getInlierMask..copy-out, DLL@0x180046900,
 * 0x180047dc0, 0x180048630, 0x180048d10, 0x180048340) for n_pairs independent pairs.
 *   d_pts/d_frame_off  normalised keypoint table from gms_normalize_device
 *   d_pairs            n_pairs descriptors; max_m >= every d_pairs[i].m (host-known upper bound)
 *   d_matches          putative matches, pair i at [match_off, match_off+m). The ranges of a batch's pairs must be DISJOINT
 *                      (empty pairs aside): pair i's survivors are written over the head of the same range of d_out, so
 *                      overlapping ranges would make pairs overwrite each other. Validated on the device BEHIND the first launch
 *                      of a context and every sixteenth (every launch with GMS_CHECK_PAIRS=1; never inside a stream capture or a
 *                      graph replay): every pair whose range overlaps another's gets status GMS_ERR_BAD_ARG in d_results. The check
 *                      reports, it does not prevent: d_out of the flagged pairs AND of whatever their ranges touch has already been
 *                      written by then and is invalid, and a launch that is not checked returns GMS_OK on such a table. A caller
 *                      that cannot vouch for its table validates it once itself (gms_filter_host_batch does: every call, on the
 *                      host, GMS_ERR_BAD_ARG before anything is launched).
 *   d_out              same offsets/capacity; pair i's survivors are written at d_out[match_off ...]
 *   d_results          n_pairs result records
 *   d_mask             optional (may be NULL): per-match inlier byte (0/1) at the match's offset */
int gms_filter_device(gms_ctx* ctx, const float* d_pts, const int64_t* d_frame_off, int n_frames,
                      const gms_pair* d_pairs, int n_pairs, int max_m,
                      const gms_dmatch* d_matches,
                      int with_rotation, int with_scale, double threshold_factor,
                      gms_dmatch* d_out, gms_pair_result* d_results, uint8_t* d_mask);

/* ---- host-pointer batch path -------------------------------------------------------------------
 * The throughput entry for callers that hold everything in host memory (a C++ caller of the reference looping over
 * image pairs: FeatureMatchUtil.cpp:66-69 once per pair): the frames' keypoints are uploaded and normalised once,
 * then the pair list is cut into chunks that travel through pinned staging buffers on two streams -- chunk k+1 is
 * uploaded while chunk k is filtered and chunk k-1 comes back. Synchronous: returns when out/results are complete.
 *   kp/frame_off/wh   keypoints of all frames back to back, n_frames+1 offsets, (w, h) per frame
 *   pairs/matches     as gms_filter_device, host memory; match_off indexes `matches` and `out` alike
 *   out               pair i's survivors verbatim at out[match_off .. match_off + results[i].n_inliers)
 * Returns GMS_OK, or the first error; pairs outside the parity domain are reported per pair in results[i].status. */
int gms_filter_host_batch(gms_ctx* ctx, const gms_keypoint* kp, const int64_t* frame_off, const int32_t* wh,
                          int n_frames, const gms_pair* pairs, int n_pairs, const gms_dmatch* matches,
                          int with_rotation, int with_scale, double threshold_factor,
                          gms_dmatch* out, gms_pair_result* results);

/* ---- brute-force descriptor matcher: the producer of the match array ------------------------------------
 * Replaces, for a batch of pairs on the resident frame table, what the reference runs in front of matchGMS
 * (FeatureMatchUtil.cpp:66-68; DisparityUtil.cpp:104-109,143):
 *     BFMatcher::create(normType)->match(descriptors1, descriptors2, matches)          (no cross-check)
 * Descriptor i of a frame belongs to keypoint i (same d_frame_off as the keypoint table). Pair p gets
 * one match per query row -- {queryIdx = i, trainIdx = first minimum over the train rows, imgIdx = 0, distance} for
 * i < min(d_pairs[p].m, n(frame_a)) -- written at d_matches[match_off + i]: the array gms_filter_device reads next.
 *   GMS_DESC_HAMMING256    rows of 32 bytes (ORB), NORM_HAMMING, distance = popcount as float. With a prepared block the
 *                          cross term runs on the matrix cores (the bits as FP4 elements, exact); with d_prepared = NULL on
 *                          the vector ALUs straight from the raw rows. Same results.
 *   GMS_DESC_L2_F32X128    rows of 128 floats (SIFT), NORM_L2, distance = sqrtf(sum of squared differences in fp32); needs
 *                          the prepared block. Frames whose values are all integers 0..255 (what SIFT emits) run on the
 *                          matrix cores (as int8, exact arithmetic); any other frame is matched by the reference's fp32 loop.
 * gms_bf_prepare_device builds the per-frame tables (gms_bf_prepared_bytes bytes, caller-allocated) once per frame table.
 * Stream-ordered on the context's stream, no allocation, no synchronisation. A frame may hold at most 2^22 rows. */
#define GMS_DESC_NONE      (-1)
#define GMS_DESC_HAMMING256  0
#define GMS_DESC_L2_F32X128  1
int64_t gms_bf_prepared_bytes(int desc_kind, int64_t total_desc, int n_frames);
int gms_bf_prepare_device(gms_ctx* ctx, int desc_kind, const void* d_desc, const int64_t* d_frame_off, int n_frames,
                          int64_t total_desc, void* d_prepared);
int gms_bfmatch_device(gms_ctx* ctx, int desc_kind, const void* d_desc, const void* d_prepared, int64_t total_desc,
                       const int64_t* d_frame_off, int n_frames, const gms_pair* d_pairs, int n_pairs, int max_query,
                       gms_dmatch* d_matches);

/* ---- consumers of the filtered matches --------------------------------------------------------------------
 * Both read the survivors of ONE pair where gms_filter_device left them (d_matches = d_out + match_off, *d_n_matches =
 * d_results[i].n_inliers, max_matches >= that count, e.g. the pair's m) and the two frames' ORIGINAL keypoints (pixel
 * coordinates, cv::KeyPoint records). Stream-ordered, no allocation, no synchronisation.
 *
 * gms_disparity_device: DisparityUtil.cpp:179-201. d_disparity (w*h bytes, row-major) receives the disparity map (255 = no
 * match; a later match overwrites an earlier one on the same pixel); with a ground-truth image d_gt (w*h bytes, may be NULL)
 * d_stats receives count / sum of squares / maximum of |map - gt / disp_ratio| over the matched pixels, from which
 * rms = sqrt(sum_sq / count) (DisparityUtil.cpp:201). d_work: w*h uint32 of scratch. status: GMS_ERR_DOMAIN when a match
 * indexes outside the keypoints or lands outside the image (undefined behaviour in the reference). */
typedef struct gms_disparity_stats {
    int64_t count;    /* matched pixels (map != 255) */
    int64_t sum_sq;   /* sum of a^2, a = |map - gt / disp_ratio| */
    int32_t max_abs;  /* max a */
    int32_t status;   /* GMS_OK or GMS_ERR_DOMAIN */
} gms_disparity_stats;
int gms_disparity_device(gms_ctx* ctx, const gms_keypoint* d_kp1, int n1, const gms_keypoint* d_kp2, int n2,
                         const gms_dmatch* d_matches, const int32_t* d_n_matches, int max_matches, int width, int height,
                         const uint8_t* d_gt, int disp_ratio, uint8_t* d_disparity, uint32_t* d_work,
                         gms_disparity_stats* d_stats);
/* gms_gather_points_device: SfMUtil.cpp:25-35. coords1[i] = keypoints1[queryIdx].pt, coords2[i] = keypoints2[trainIdx].pt
 * (two floats each) for i < *d_n_matches: the arrays findEssentialMat / recoverPose / undistortPoints take (SfMUtil.cpp:39,
 * 45,78-79). *d_status: GMS_OK or GMS_ERR_DOMAIN. */
int gms_gather_points_device(gms_ctx* ctx, const gms_keypoint* d_kp1, int n1, const gms_keypoint* d_kp2, int n2,
                             const gms_dmatch* d_matches, const int32_t* d_n_matches, int max_matches,
                             float* d_coords1, float* d_coords2, int32_t* d_status);

/* gms_triangulate_device: SfMUtil.cpp:76-82 and 128-143 for the gathered points -- cv::undistortPoints (camera = fx, fy, cx, cy;
 * dist = k1, k2, p1, p2, k3 or NULL), cv::triangulatePoints with the 3 x 4 row-major projection matrices P1, P2 (the reference
 * uses [I|0] and [R|t] from recoverPose), division by the fourth coordinate. d_points3d receives 3 doubles per match;
 * d_stats the sums of squared reprojection errors in both views (normalised image coordinates), the number of finite points
 * and how many of them lie behind a camera. fp64; agrees with OpenCV's SVD-based routine to rounding, not bit for bit.
 * camera / dist / P1 / P2 are HOST pointers (they travel as kernel arguments). */
typedef struct gms_triangulation_stats {
    double  sum_sq_err1, sum_sq_err2;
    int64_t count, behind;
} gms_triangulation_stats;
int gms_triangulate_device(gms_ctx* ctx, const double camera[4], const double dist[5], const double P1[12], const double P2[12],
                           const float* d_coords1, const float* d_coords2, const int32_t* d_n_matches, int max_matches,
                           double* d_points3d, gms_triangulation_stats* d_stats);

/* gms_recover_pose_device: cv::recoverPose(E, points1, points2, cameraMatrix, R, t, mask) as SfMUtil.cpp:45 calls it (OpenCV 4.5.2:
 * distance threshold 50) for the gathered points: the four (R, t) the essential matrix decomposes into, each tried on every
 * correspondence by triangulation in normalised coordinates (positive depth below the threshold in both cameras); the first of
 * (R1, t), (R2, t), (R1, -t), (R2, -t) with the most such points wins. E (3 x 3 row-major) and camera = (fx, fy, cx, cy) are HOST
 * pointers; d_in_mask (optional: findEssentialMat's inlier mask, non-zero = use) and everything else device pointers.
 * *d_pose receives R, t, the winner's point count and its index; d_out_mask (optional) per correspondence what the reference's
 * bitwise_and leaves: the input mask's byte where the point passes (255 without an input mask), 0 elsewhere.
 * Stream-ordered; the context keeps max_matches + 48 bytes of scratch, which grows (one stream synchronisation) on first use of a
 * larger max_matches -- GMS_ERR_NOT_RESERVED instead when the stream is being captured; gms_ctx_reserve(ctx, 1, max_matches, ...) sizes it. fp64, agrees with OpenCV to rounding (R1 / R2 and the
 * sign of t may be numbered differently than by another SVD: `which` is informational). */
typedef struct gms_pose {
    double  R[9], t[3];
    int32_t n_good, which;
} gms_pose;
int gms_recover_pose_device(gms_ctx* ctx, const double E[9], const double camera[4], const float* d_coords1, const float* d_coords2,
                            const int32_t* d_n_matches, int max_matches, const uint8_t* d_in_mask, gms_pose* d_pose,
                            uint8_t* d_out_mask);

/* ---- the same consumers for a whole batch ---------------------------------------------------------------------------
 * What structureFromMotion does with the survivors of ONE pair (SfMUtil.cpp:25-82) and matchBasedDispCalculate with its map
 * (DisparityUtil.cpp:170-201), for every pair of a batch per launch: same pair table, same offsets as gms_filter_device. Every
 * per-match array (coords, mask, 3-D points) holds pair i's entries at its match_off (coords: 2 floats per match, points: 3 doubles),
 * d_tv holds one record per pair. Stream-ordered on the context's stream, no allocation, no synchronisation, capturable.
 *
 *   gms_gather_points_batch_device   SfMUtil.cpp:25-35. Zeroes d_tv, then n_points = the pair's n_inliers and the coordinates.
 *   gms_find_essential_batch_device  SfMUtil.cpp:39: cv::findEssentialMat(coords1, coords2, cameraMatrix, RANSAC, prob, threshold,
 *       mask) of OpenCV 4.5.2 -- points normalised with the camera matrix, threshold / ((fx + fy) / 2), RANSAC over five-point
 *       minimal solves (Nister) with cv::RNG((uint64)-1) drawing the samples, at most max_iters (OpenCV: 1000) iterations, the
 *       bound shrinking by RANSACUpdateNumIters(prob, ...) whenever a model with strictly more inliers appears; error =
 *       (x2^T E x1)^2 / (|E x1|_xy^2 + |E^T x2|_xy^2) as fp32 against (float)threshold^2. E: row-major, unit Frobenius norm, largest
 *       entry positive (an SVD leaves the sign open); the models of one sample are tried in ascending order of E[0], E[1], ...
 *       (OpenCV's order is that of its polynomial root finder: it only matters between models of equal inlier count).
 *       d_mask: 1 / 0 per correspondence. Pairs with fewer than five correspondences or no model: status GMS_ERR_NO_MODEL.
 *       fp64; agrees with an SVD / eigenvalue based implementation to rounding, not bit for bit.
 *   gms_recover_pose_batch_device    SfMUtil.cpp:45: cv::recoverPose(E, coords1, coords2, cameraMatrix, R, t, mask), distance
 *       threshold 50; d_mask is in/out as in the reference (use_in_mask = 0: output only, 255 / 0).
 *   gms_triangulate_batch_device     SfMUtil.cpp:65-82,128-143: the correspondences with a non-zero mask byte (d_mask NULL: all),
 *       compacted in order, cv::undistortPoints, cv::triangulatePoints with [I|0] and [R|t], division by the fourth coordinate:
 *       pair i's n_triangulated points at d_points3d[3 * match_off ...]; reprojection error sums in normalised coordinates.
 *   gms_two_view_batch_device        all four, in stream order.
 *   gms_disparity_batch_device       DisparityUtil.cpp:170-201 per pair: pair i's map (width x height of frame_a, row-major) at
 *       d_disparity + i * map_stride, its ground truth (optional) at d_gt + i * gt_stride (gt_stride 0: one image for all),
 *       d_work: n_pairs * map_stride uint32 of scratch, d_stats one record per pair. */
typedef struct gms_camera {   /* cameraMatrix and distCoeffs as structureFromMotion receives them (SfMUtil.cpp:4; main.cpp:59-67) */
    double fx, fy, cx, cy;
    double k1, k2, p1, p2, k3;
} gms_camera;
typedef struct gms_two_view {
    double  E[9];                      /* findEssentialMat                                                   */
    double  R[9], t[3];                /* recoverPose                                                        */
    double  sum_sq_err1, sum_sq_err2;  /* sums of squared reprojection errors of the triangulated points     */
    int64_t n_finite, n_behind;        /* triangulated points that are finite / of those, behind a camera    */
    int32_t n_points;                  /* correspondences of the pair (the filter's n_inliers)               */
    int32_t n_ransac;                  /* inliers of E (non-zero bytes of findEssentialMat's mask)           */
    int32_t ransac_iters;              /* RANSAC iterations run                                              */
    int32_t n_pose;                    /* recoverPose's return value                                         */
    int32_t pose_which;                /* which of (R1,t) (R2,t) (R1,-t) (R2,-t) won: informational          */
    int32_t n_triangulated;            /* points written to d_points3d                                       */
    int32_t status;                    /* GMS_OK, GMS_ERR_DOMAIN, GMS_ERR_NO_MODEL, or the filter's status   */
    int32_t reserved;
} gms_two_view;
int gms_gather_points_batch_device(gms_ctx* ctx, const gms_keypoint* d_kp, const int64_t* d_frame_off, int n_frames,
                                   const gms_pair* d_pairs, int n_pairs, int max_m, const gms_dmatch* d_filtered,
                                   const gms_pair_result* d_results, float* d_coords1, float* d_coords2, gms_two_view* d_tv);
int gms_find_essential_batch_device(gms_ctx* ctx, const gms_camera* camera, double prob, double threshold, int max_iters,
                                    const gms_pair* d_pairs, int n_pairs, const float* d_coords1, const float* d_coords2,
                                    uint8_t* d_mask, gms_two_view* d_tv);
int gms_recover_pose_batch_device(gms_ctx* ctx, const gms_camera* camera, int use_in_mask, const gms_pair* d_pairs, int n_pairs,
                                  const float* d_coords1, const float* d_coords2, uint8_t* d_mask, gms_two_view* d_tv);
int gms_triangulate_batch_device(gms_ctx* ctx, const gms_camera* camera, const gms_pair* d_pairs, int n_pairs,
                                 const float* d_coords1, const float* d_coords2, const uint8_t* d_mask, double* d_points3d,
                                 gms_two_view* d_tv);
int gms_two_view_batch_device(gms_ctx* ctx, const gms_camera* camera, double prob, double threshold, int max_iters,
                              const gms_keypoint* d_kp, const int64_t* d_frame_off, int n_frames, const gms_pair* d_pairs, int n_pairs,
                              int max_m, const gms_dmatch* d_filtered, const gms_pair_result* d_results, float* d_coords1,
                              float* d_coords2, uint8_t* d_mask, double* d_points3d, gms_two_view* d_tv);
int gms_disparity_batch_device(gms_ctx* ctx, const gms_keypoint* d_kp, const int64_t* d_frame_off, const int32_t* d_wh, int n_frames,
                               const gms_pair* d_pairs, int n_pairs, int max_m, const gms_dmatch* d_filtered,
                               const gms_pair_result* d_results, const uint8_t* d_gt, int64_t gt_stride, int disp_ratio,
                               uint8_t* d_disparity, int64_t map_stride, uint32_t* d_work, gms_disparity_stats* d_stats);

/* ---- ingest format -----------------------------------------------------------------------------------------
 * The reference keeps detector and matcher output in process (std::vector<cv::KeyPoint>, cv::Mat descriptors,
 * std::vector<cv::DMatch>: FeatureMatchUtil.cpp:9-12,58-68; DisparityUtil.cpp:108,137-143) and has no on-disk form. One
 * little-endian file ("GMSFRM01", layout in gms_io.cpp) carries a sequence in exactly the arrays the batch API takes --
 * cv::KeyPoint / cv::DMatch records verbatim -- so that a caller can dump its vectors and any process can filter them.
 * gms_dataset_read allocates one block (owner) that gms_dataset_free releases; on write, owner is ignored. Host only. */
typedef struct gms_dataset {
    int32_t n_frames, desc_kind;        /* desc_kind: GMS_DESC_NONE / _HAMMING256 / _L2_F32X128                    */
    int64_t n_pairs, total_matches;
    int32_t* wh;                        /* 2 * n_frames: (width, height)                                           */
    int64_t* frame_off;                 /* n_frames + 1 keypoint offsets; frame_off[n_frames] = number of keypoints */
    gms_keypoint* keypoints;
    void* descriptors;                  /* one row per keypoint (32 B or 128 floats), or NULL                      */
    gms_pair* pairs;                    /* match_off indexes `matches`                                             */
    gms_dmatch* matches;
    void* owner;
} gms_dataset;
int  gms_dataset_write(const char* path, const gms_dataset* d);
int  gms_dataset_read(const char* path, gms_dataset* d);
void gms_dataset_free(gms_dataset* d);

/* ---- introspection --------------------------------------------------------------------------- */
int         gms_max_matches(void);        /* largest m per pair this build accepts                 */
int         gms_last_hip_error(void);     /* last hipError_t seen by this thread's calls           */
/* ---- keypoint source ---------------------------------------------------------------------
 * Stands where the reference calls OpenCV's detectors (FeatureMatchUtil.cpp:9-12 SIFT::create(10000)->detectAndCompute;
 * DisparityUtil.cpp:108,123-138 ORB::create(), detectAndCompute / compute at every pixel). NOT cv::ORB: a single-scale FAST-9 +
 * steered-BRIEF detector of this library's own, in integer arithmetic (definition: DESIGN.md section 7b; CPU statement
 * oracle/detect_ref.c). Same records out: cv::KeyPoint {pt, size 31, angle = 11.25 * direction bin, response = FAST score, octave 0,
 * class_id -1} and one 32-byte row per keypoint for NORM_HAMMING -- what gms_normalize_device / gms_bf_prepare_device take.
 * Images: 8-bit grey, row-major, pitch = width, n_images of one size back to back in device memory. Keypoints sit at least
 * GMS_DETECT_BORDER pixels from every edge. */
#define GMS_DETECT_BORDER 16
size_t gms_detect_workspace_bytes(int width, int height, int n_images, int max_keypoints);

/* detectAndCompute for a batch: per image the max_keypoints strongest FAST-9 corners with score > threshold (equal scores in raster
 * order), written in raster order: d_keypoints[i * max_keypoints ..], d_descriptors[(i * max_keypoints ..) * 32], d_counts[i].
 * Errors: GMS_ERR_BAD_ARG (NULL, width/height outside (32, 65535], threshold outside [0, 254], workspace too small). */
int gms_detect_batch_device(gms_ctx* ctx, const uint8_t* d_images, int n_images, int width, int height, int threshold, int max_keypoints,
                            void* d_workspace, size_t workspace_bytes, gms_keypoint* d_keypoints, uint8_t* d_descriptors, int32_t* d_counts);

/* Feature2D::compute on ONE image (DisparityUtil.cpp:123-133, a keypoint per pixel): direction (written to angle) and row at each
 * of the caller's n keypoints. A keypoint off the integer pixel grid or inside the border sets *d_status = 1 and keeps its row
 * (OpenCV would have dropped the keypoint and renumbered the rest; here the caller keeps control of the indices).
 * Workspace: gms_detect_workspace_bytes(width, height, 1, 0). */
int gms_describe_device(gms_ctx* ctx, const uint8_t* d_image, int width, int height, gms_keypoint* d_keypoints, int n,
                        void* d_workspace, size_t workspace_bytes, uint8_t* d_descriptors, int32_t* d_status);

const char* gms_error_string(int code);
const char* gms_version(void);

/* Test hook (no production use): evaluates thresh = sqrt(T / n) * factor > score in device fp64 for
 * `count` (T, n, score) triples, so the tests can pin the device's div/sqrt/mul against IEEE.
 * Host pointers; out[i] = 1 iff the cell would be rejected. */
int gms_selftest_threshold(gms_ctx* ctx, const int32_t* T, const int32_t* n, const int32_t* score,
                           double factor, int count, uint8_t* out);

/* Test hook (no production use): the five-point minimal solver of gms_find_essential_batch_device alone, exactly as the RANSAC
 * kernel runs it, on n_samples caller-given samples of NORMALISED points. Host pointers: pts[20 s ..] = x1[5], y1[5], x2[5], y2[5]
 * of sample s; models[90 s ..] receives up to ten 3 x 3 matrices (zeros beyond counts[s]). */
int gms_selftest_five_point(gms_ctx* ctx, const double* pts, int n_samples, double* models, int32_t* counts);

#ifdef __cplusplus
}
#endif
#endif /* MI355_GMS_H */
