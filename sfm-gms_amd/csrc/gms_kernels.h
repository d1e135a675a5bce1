// gms_kernels.h -- internal interface between the C-ABI host code and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "gms.h"

namespace gms {

constexpr int kLeftW = 20, kLeftH = 20, kLeftN = 400;  // DLL@0x180046ac6: fixed 20 x 20 left grid
constexpr int kFineW = 40, kFineN = 1600;                  // half-cell grid: carries all four grid types
constexpr int kThreads = 1024;                         // one 16-wave workgroup per image pair
constexpr size_t kLdsBytes = 160 * 1024;               // gfx950 LDS per CU (and per workgroup)
// record handed from filter_kernel_dense_scales to filter_kernel: state, best count, best scale, best rotation, then one
// inlier bit per match of the best hypothesis so far (16 x 1024 matches at most)
constexpr uint32_t kPartialHeaderDw = 4;
constexpr uint32_t kPartialStrideDw = kPartialHeaderDw + 16 * 1024 / 32;

// The frame table (gms_frame_table_bytes) starts with a 16-byte header -- magic, then the number of keypoints it was built
// for -- so that the kernels find the code arrays behind the points by themselves, whatever n_frames / frame_off a filter call
// passes (a prefix or a subset of the table's frames is fine). A block without the header is filtered from its points alone.
constexpr uint32_t kTableMagic0 = 0x46534D47u, kTableMagic1 = 0x31424154u;  // "GMSF" "TAB1"
constexpr int kTableHeaderBytes = 16;

struct FilterParams {
    const float2* pts;          // normalised keypoints of all frames (the table's points: kTableHeaderBytes behind its start)
    const int64_t* frame_off;   // n_frames + 1
    int n_frames;
    const gms_pair* pairs;
    int n_pairs;
    int stagger_ticks;          // first-round workgroups start spread over this many 100 MHz wall-clock ticks (0 = off)
    int stagger_blocks;         // how many leading workgroups count as the first round
    const gms_dmatch* matches;
    gms_dmatch* out;
    gms_pair_result* results;
    uint8_t* mask;              // optional
    uint32_t table_slots;       // multiple of 4: data + header buckets of all 400 regions
    int region_shift;           // region slots per match = 1 + 2^-shift
    int with_rotation, with_scale;
    uint32_t* partial;          // scale hypotheses: per-pair records of the byte-matrix kernel (scales 0..3), or null
    const uint32_t* pair_flags; // large-pair kernel only: when set, it filters just the pairs whose flag word has bit 1 set
    int dealt;                  // byte-matrix kernel: deal the matches to the lanes (inputs in spatial order; see dense_pair)
    int probe_scales;           // scale hypotheses: bit s set = bound scale s's inlier count first and skip the scale when it cannot win
    int probe_nibble;           // scale hypotheses: bit 3 / bit 4 = bound the 28 x 28 / 40 x 40 grid with four-bit entries first (half the bands of the byte probe)
    uint32_t* probe_stats;      // optional device counters: [0] scales probed, [1] scales the probe let skip
    uint32_t* overflow_events;  // streamed byte-matrix kernels: a word (pinned host memory) that counts the pairs they had to hand on because an entry left its byte
    int prefetch_type, prefetch_ahead;  // byte-matrix kernel: before grid type prefetch_type a workgroup touches the records of pair + prefetch_ahead (0 = off)
    int dense;                  // try the byte-matrix path first (no scale hypotheses only); the general path is the fallback
    double threshold_factor;
    int right_w[5], right_h[5]; // setScale (DLL@0x180048c10): cvRound(20 * ratio[s])
#ifdef GMS_PHASE_TIMING
    unsigned long long* diag;   // diagnostic build only: [n_pairs][12] cycle sums
#endif
};

hipError_t init_filter_kernels();        // once per context, before the first launch: dynamic-LDS limits of every kernel
hipError_t init_band_kernels();
hipError_t init_big_kernels();
int        filter_pick_kpt(int max_m);   // matches per thread (template variant) for max_m, 0 = too large
uint32_t   filter_table_slots(int kpt);
int        filter_region_shift(int kpt);
size_t     filter_lds_bytes(int kpt, uint32_t table_slots);
// kp_stride_bytes: 28 = cv::KeyPoint records, 8 = packed (pt.x, pt.y) pairs
hipError_t launch_normalize(const void* d_kp, int kp_stride_bytes, const int64_t* d_frame_off, const int32_t* d_wh,
                            int n_frames, int64_t total_kp, float* d_pts, hipStream_t stream);
hipError_t launch_filter(const FilterParams& p, int kpt, int n_pairs, hipStream_t stream);
hipError_t launch_order_probe(const FilterParams& p, uint32_t* flag, hipStream_t stream);  // *flag: pinned host word
hipError_t launch_probe_verdict(uint32_t* stats, uint32_t* flag, hipStream_t stream);      // FilterParams::probe_stats -> pinned host word
// pair-table validation: ranges [match_off, match_off + m) must be disjoint; offenders get GMS_ERR_BAD_ARG in d_results (d_flag: a device word)
hipError_t launch_check_pairs(const gms_pair* d_pairs, int n_pairs, gms_pair_result* d_results, uint32_t* d_flag, hipStream_t stream);
hipError_t launch_filter_scales(const FilterParams& p, int kpt, int n_pairs, hipStream_t stream);
// gms_filter_host_batch: pair i's n_inliers survivors (at d_out + match_off) packed back to back into d_packed, the total into *d_total
hipError_t launch_compact_survivors(const gms_pair* d_pairs, const gms_pair_result* d_results, int n_pairs, const gms_dmatch* d_out,
                                    gms_dmatch* d_packed, int64_t* d_total, hipStream_t stream);
// large pairs (gms_kernel_big.hip): code words and table in a per-workgroup HBM slab
constexpr int kBigMaxMatches = 1 << 22;       // per pair: 4 194 304 (a 2594 x 1131 one-keypoint-per-pixel frame of DisparityUtil.cpp:299 has 2.93 M)
constexpr int kBigLdsMaskMatches = 262144;   // up to here the slab kernel keeps the winner's bit mask in LDS
int        big_mcap(int max_m);
size_t     big_ws_stride_dwords(int mcap);
hipError_t launch_filter_big(const FilterParams& p, int mcap, int n_workgroups, uint32_t* ws, hipStream_t stream);
// large pairs under the default flags (gms_kernel_band.hip): three-band 16-bit matrix in LDS; *flags_out marks the pairs
// left to launch_filter_big (bit 1)
size_t     band_ws_bytes_per_pair(int mcap, bool need_mask);
hipError_t launch_filter_band(const FilterParams& p, int mcap, void* ws, const uint32_t** flags_out, hipStream_t stream);
// large pairs with rotation / scale hypotheses (gms_kernel_band.hip): tiled 16-bit matrix, three launches per scale
size_t     tile_ws_bytes_per_pair(const FilterParams& p, int mcap, bool need_mask);
hipError_t launch_filter_tiles(const FilterParams& p, int mcap, void* ws, const uint32_t** flags_out, hipStream_t stream);
// pairs of 16 385 .. 65 536 matches, every flag combination (gms_kernel_stream.hip): the byte matrix with the matches streamed from a
// row-sorted workspace array, one workgroup per (pair, scale hypothesis); *flags_out marks the pairs left to launch_filter_big (bit 1)
hipError_t init_stream_kernels();
int        stream_max_matches();
size_t     stream_ws_bytes_per_pair(const FilterParams& p, int mcap, bool need_mask);
hipError_t launch_filter_stream(const FilterParams& p, int mcap, void* ws, const uint32_t** flags_out, hipStream_t stream);
// the same size class without scale hypotheses: one workgroup per pair, dense_pair() with the code words streamed from an L2-resident array
size_t     stream_dense_ws_bytes_per_pair(int mcap);
hipError_t launch_filter_stream_dense(const FilterParams& p, int mcap, void* ws, const uint32_t** flags_out, hipStream_t stream);
// brute-force descriptor matcher (bf_kernels.hip)
size_t     bf_prepared_bytes(int kind, int64_t total, int n_frames);
hipError_t launch_bf_prepare(int kind, const void* d_desc, const int64_t* d_frame_off, int n_frames, int64_t total, void* d_prep,
                             hipStream_t stream);
hipError_t launch_bf_match(int kind, const void* d_desc, const void* d_prep, int64_t total, const int64_t* d_frame_off, int n_frames,
                           const gms_pair* d_pairs, int n_pairs, int max_query, gms_dmatch* d_matches, hipStream_t stream);
// consumers of the filtered matches (consumer_kernels.hip)
hipError_t launch_disparity(const gms_keypoint* d_kp1, int n1, const gms_keypoint* d_kp2, int n2, const gms_dmatch* d_matches,
                            const int32_t* d_n_matches, int max_matches, int w, int h, const uint8_t* d_gt, int disp_ratio,
                            uint8_t* d_disparity, uint32_t* d_work, gms_disparity_stats* d_stats, hipStream_t stream);
hipError_t launch_gather_points(const gms_keypoint* d_kp1, int n1, const gms_keypoint* d_kp2, int n2, const gms_dmatch* d_matches,
                                const int32_t* d_n_matches, int max_matches, float* d_coords1, float* d_coords2, int32_t* d_status,
                                hipStream_t stream);
hipError_t launch_triangulate(const double* camera, const double* dist, const double* P1, const double* P2, const float* d_coords1,
                              const float* d_coords2, const int32_t* d_n_matches, int max_matches, double* d_points3d,
                              gms_triangulation_stats* d_stats, hipStream_t stream);
hipError_t launch_recover_pose(const double* camera, const double P[4][12], double dist_thresh, const float* d_coords1, const float* d_coords2,
                               const int32_t* d_n_matches, int max_matches, const uint8_t* d_in_mask, gms_pose* d_pose, uint8_t* d_out_mask,
                               void* d_work, hipStream_t stream);
// batched consumers (twoview_kernels.hip)
hipError_t launch_gather_batch(const gms_keypoint* d_kp, const int64_t* d_frame_off, int n_frames, const gms_pair* d_pairs, int n_pairs, int max_m,
                               const gms_dmatch* d_filtered, const gms_pair_result* d_results, float* d_coords1, float* d_coords2,
                               gms_two_view* d_tv, hipStream_t stream);
hipError_t launch_find_essential_batch(const gms_camera& cam, double prob, double threshold, int max_iters, const gms_pair* d_pairs, int n_pairs,
                                       const float* d_coords1, const float* d_coords2, uint8_t* d_mask, gms_two_view* d_tv, hipStream_t stream);
hipError_t launch_five_point_selftest(const double* d_pts, int n_samples, double* d_models, int* d_counts, hipStream_t stream);
hipError_t launch_recover_pose_batch(const gms_camera& cam, double dist_thresh, int use_in_mask, const gms_pair* d_pairs, int n_pairs,
                                     const float* d_coords1, const float* d_coords2, uint8_t* d_mask, gms_two_view* d_tv, hipStream_t stream);
hipError_t launch_triangulate_batch(const gms_camera& cam, const gms_pair* d_pairs, int n_pairs, const float* d_coords1, const float* d_coords2,
                                    const uint8_t* d_mask, double* d_points3d, gms_two_view* d_tv, hipStream_t stream);
hipError_t launch_disparity_batch(const gms_keypoint* d_kp, const int64_t* d_frame_off, const int32_t* d_wh, int n_frames, const gms_pair* d_pairs,
                                  int n_pairs, int max_m, const gms_dmatch* d_filtered, const gms_pair_result* d_results, const uint8_t* d_gt,
                                  int64_t gt_stride, int disp_ratio, uint8_t* d_disparity, int64_t map_stride, uint32_t* d_work,
                                  gms_disparity_stats* d_stats, hipStream_t stream);
// keypoint source (detect_kernels.hip)
size_t     detect_workspace_bytes(int w, int h, int n_images, int max_keypoints);
hipError_t launch_detect(const uint8_t* d_images, int n_images, int w, int h, int threshold, int max_keypoints, void* d_ws,
                         gms_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts, hipStream_t stream);
hipError_t launch_describe(const uint8_t* d_image, int w, int h, gms_keypoint* d_kp, int n, void* d_ws, uint8_t* d_desc, int32_t* d_status,
                           hipStream_t stream);
hipError_t launch_threshold(const int32_t* d_T, const int32_t* d_n, const int32_t* d_score, double factor,
                            int count, uint8_t* d_out, hipStream_t stream);

}  // namespace gms
