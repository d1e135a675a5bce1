// gms_capi.cpp -- the C ABI declared in include/gms.h, on top of the HIP kernels.
//
// Host side of the drop-in for cv::xfeatures2d::matchGMS (FeatureMatchUtil.cpp:69,
// DisparityUtil.cpp:149,299 of the reference). There is deliberately no CPU implementation of the
// filter in this library: without a working HIP device every compute entry point returns an error.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "gms.h"
#include "gms_kernels.h"
#include "twoview_core.h"

static_assert(sizeof(gms_keypoint) == 28, "gms_keypoint must match cv::KeyPoint (stride 0x1c)");
static_assert(sizeof(gms_dmatch) == 16, "gms_dmatch must match cv::DMatch (stride 0x10)");
static_assert(sizeof(gms_pair) == 24, "gms_pair layout");
static_assert(sizeof(gms_pair_result) == 16, "gms_pair_result layout");
static_assert(sizeof(gms_two_view) == 232 && sizeof(gms_camera) == 72, "two-view records");

static thread_local int t_last_hip = 0;
#ifdef GMS_PHASE_TIMING
static unsigned long long* g_diag = nullptr;
extern "C" int gms_diag_set_buffer(void* d_buf) { g_diag = (unsigned long long*)d_buf; return 0; }
#endif

#define GMS_HIP(call)                         \
    do {                                      \
        hipError_t e_ = (call);               \
        if (e_ != hipSuccess) {               \
            t_last_hip = (int)e_;             \
            return GMS_ERR_HIP;               \
        }                                     \
    } while (0)
#define GMS_TRY(call)                \
    do {                             \
        const int rc_ = (call);      \
        if (rc_ != GMS_OK) return rc_; \
    } while (0)

namespace {

// Diagnostic switches, read from the environment ONCE per process (never on the call path):
//   GMS_DENSE=0          keep every pair on the hashed path
//   GMS_BAND=0           keep large pairs on the HBM-slab kernel alone
//   GMS_STREAM=0         pairs of 16 385 .. 65 536 matches: the 16-bit band / tile kernels instead of the streamed byte-matrix kernels
//   GMS_PROBE_NIBBLE=m   which scale probes try four-bit entries first: 8 = the 28 x 28 grid, 16 = the 40 x 40 grid (default), 24 = both, 0 = none
//   GMS_STAGGER_US=n     spread of the first dispatch round's start times at 10k matches per pair (0 = off)
//   GMS_BAND_WS_BYTES=n  budget of the large-pair workspace (default 4 GiB); a batch is filtered in slices that fit it
//   GMS_DEAL=0|1         never / always deal the matches to the lanes of the byte-matrix kernel (default: what the probe saw)
//   GMS_SCALE_PROBE=0|1  never / always bound the finer scale hypotheses' inlier counts first (default: while it pays, see below)
//   GMS_CHECK_PAIRS=0|1  never / always validate the pair table behind a launch (default: the first launch and every sixteenth)
//   GMS_PREFETCH=t[,a]   byte-matrix kernel: before grid type t (0..3, default 3; -1 = never) a workgroup touches the match records of
//                        the pair a places ahead (default: the number of CUs = the pair its CU's next workgroup takes)
struct Knobs {
    bool dense_on = true, band_on = true, stream_on = true;
    int probe_nibble = 16;  // which fine right grids (bit 3: 28 x 28, bit 4: 40 x 40) a scale probe tries with four-bit entries first when the probe is forced on
    bool probe_nibble_set = false;  // GMS_PROBE_NIBBLE given: also a limit on what the library chooses by itself
    int stagger_us = -1, deal = -1, scale_probe = -1, check_pairs = -1;
    int prefetch_type = 3, prefetch_ahead = 0;
    size_t band_ws_budget = (size_t)4 << 30;
};
const Knobs& knobs()
{
    static const Knobs k = [] {
        Knobs v;
        if (const char* e = std::getenv("GMS_DENSE")) v.dense_on = std::atoi(e) != 0;
        if (const char* e = std::getenv("GMS_BAND")) v.band_on = std::atoi(e) != 0;
        if (const char* e = std::getenv("GMS_STREAM")) v.stream_on = std::atoi(e) != 0;
        if (const char* e = std::getenv("GMS_PROBE_NIBBLE")) { v.probe_nibble = (int)std::strtol(e, nullptr, 0) & 24; v.probe_nibble_set = true; }
        if (const char* e = std::getenv("GMS_STAGGER_US")) v.stagger_us = std::atoi(e);
        if (const char* e = std::getenv("GMS_DEAL")) v.deal = std::atoi(e) != 0 ? 1 : 0;
        if (const char* e = std::getenv("GMS_SCALE_PROBE")) v.scale_probe = std::atoi(e) != 0 ? 1 : 0;
        if (const char* e = std::getenv("GMS_CHECK_PAIRS")) v.check_pairs = std::atoi(e) != 0 ? 1 : 0;
        if (const char* e = std::getenv("GMS_PREFETCH")) {
            v.prefetch_type = std::atoi(e);
            if (const char* comma = std::strchr(e, ',')) v.prefetch_ahead = std::atoi(comma + 1);
        }
        if (const char* e = std::getenv("GMS_BAND_WS_BYTES")) {
            const long long b = std::atoll(e);
            if (b > 0) v.band_ws_budget = (size_t)b;
        }
        return v;
    }();
    return k;
}

// mScaleRatios (DLL .data 0x1802c5008) and setScale's cvRound (DLL@0x180048c10, cvtsd2si = lrint).
void right_grids(int rw[5], int rh[5])
{
    const double ratio[5] = {1.0, 1.0 / 2, 1.0 / std::sqrt(2.0), std::sqrt(2.0), 2.0};
    for (int s = 0; s < 5; ++s) {
        rw[s] = (int)std::lrint(gms::kLeftW * ratio[s]);
        rh[s] = (int)std::lrint(gms::kLeftH * ratio[s]);
    }
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// page-locked host memory: what the copy engines read and write without a bounce through the runtime's own staging
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes < 65536 ? 65536 : bytes + bytes / 4;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

// What one gms_filter_device call needs beyond its arguments.
struct WsNeed {
    size_t partial = 0, big = 0, band = 0;
    size_t slice = 0, per_pair = 0;  // large pairs: pairs per slice of the batch, workspace bytes per pair
    int kpt = 0, mcap = 0;
    bool stream = false;             // large pairs up to 65 536 matches: the streamed byte-matrix kernels
};

}  // namespace

// One lane of the host-pointer paths: pinned staging in both directions, the device arenas they mirror, a stream.
struct Lane {
    hipStream_t stream = nullptr;
    PinBuf hin, hout;
    DevBuf din, dout;
    hipEvent_t ev_res = nullptr, ev_out = nullptr;  // gms_filter_host_batch: a chunk's results are on the host / its survivors are
};
namespace { class CopyPool; }

struct gms_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::mutex mu;  // every entry point that touches the context's buffers holds it
    DevBuf aux, big_ws, band_ws, partial_ws, pose_ws;
    // What recent launches saw, as words in pinned host memory that small kernels write now and then and the host reads without
    // waiting: verdict[0] "recent batches came in spatial order" (order_probe_kernel; picks the byte-matrix kernel's instantiation),
    // verdict[1] "probing scale hypotheses pays" (probe_verdict_kernel, below)
    uint32_t* verdict = nullptr;
    unsigned dense_launches = 0;
    // verdict[0] and verdict[1] are never read while a kernel may be writing them: a verdict kernel is followed by an event, and the
    // first launch (or gms_ctx_synchronize) that finds the event complete ADOPTS the words; launches run on the adopted values in
    // between. verdict[2] (below) is different: a counter the streamed kernels add to, read without waiting -- it only steers which of
    // two bit-identical paths the next launches of large pairs take (and with it which workspace such a launch asks for).
    // last_* = what the most recent launch ran with (gms_ctx_query).
    int use_dealt = 0, use_probe = 0x1D | (16 << 8), last_dealt = 0, last_probe = 0, last_kpt = 0, last_stagger_ticks = 0;
    hipEvent_t verdict_event = nullptr;
    bool verdict_pending = false;
    unsigned filter_launches = 0;  // every launch of the context (pair-table validation every sixteenth)
    // verdict[2] counts the pairs the streamed byte-matrix kernels had to hand on (an entry above 255, a cell above 65 535): when it has
    // moved since the last look, the next 64 launches of large pairs take the 16-bit band / tile kernels instead, then the streamed ones get another try
    uint32_t overflow_seen = 0;
    int stream_penalty = 0;
    int opt_deal = -1, opt_probe = -1;  // gms_ctx_set_option: -1 = the library's own choice (and the environment switches), 0 / 1 = forced
    // Scale hypotheses: the kernels can bound a scale's inlier count before evaluating it (gms_kernels.hip, PROBE) and skip the
    // scale when it cannot win -- a gain when at least half of the probes let a scale skip, a loss otherwise. The kernels count
    // both in probe_stats (device); every sixteenth launch with scale hypotheses probes whatever the verdict and is followed by a
    // one-thread kernel that turns the counts into verdict[1], which the launches in between follow.
    DevBuf probe_stats;
    unsigned scale_launches = 0;
    // the workspaces above are shared by every launch of the context: the last launch that used them, and where
    hipEvent_t ws_event = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_pending = false;
    Lane lane[3];     // one-shot calls use lane 0 (on the context's stream); gms_filter_host_batch rotates through all three
    CopyPool* pool = nullptr;  // gms_filter_host_batch: the threads that stage its chunks (created with the first call)
    DevBuf tab_kp, tab_pts, tab_small;  // gms_filter_host_batch: the call's frame table
    int n_cus = 256;  // multiProcessorCount of the device
};

namespace {

int plan_workspace(const gms_ctx* c, int n_pairs, int max_m, bool rot, bool scale, bool need_mask_ws, bool allow_stream, WsNeed* w)
{
    *w = WsNeed();
    w->kpt = gms::filter_pick_kpt(max_m);  // 0: too large for the register + LDS kernel
    if (w->kpt) {
        if (scale && knobs().dense_on) w->partial = (size_t)n_pairs * gms::kPartialStrideDw * 4;
        return GMS_OK;
    }
    if (max_m > gms::kBigMaxMatches) return GMS_ERR_CAPACITY;
    w->mcap = gms::big_mcap(max_m);
    const int n_wg = n_pairs < c->n_cus ? n_pairs : c->n_cus;  // one persistent workgroup per CU at most
    w->big = (size_t)n_wg * gms::big_ws_stride_dwords(w->mcap) * 4;
    if (knobs().band_on) {
        gms::FilterParams p{};
        p.with_rotation = rot;
        p.with_scale = scale;
        right_grids(p.right_w, p.right_h);
        // pairs up to 65 536 matches: the streamed byte-matrix kernels -- one workgroup per (pair, scale, grid type, band) with scale
        // hypotheses, one per pair without (tools/config4_bench.py has both against the 16-bit band / tile kernels)
        w->stream = knobs().stream_on && max_m <= gms::stream_max_matches() && allow_stream;
        w->per_pair = w->stream ? std::max(gms::stream_ws_bytes_per_pair(p, w->mcap, need_mask_ws), scale ? (size_t)0 : gms::stream_dense_ws_bytes_per_pair(w->mcap))
                                : (!rot && !scale) ? gms::band_ws_bytes_per_pair(w->mcap, need_mask_ws)
                                                   : gms::tile_ws_bytes_per_pair(p, w->mcap, need_mask_ws);
        size_t slice = knobs().band_ws_budget / w->per_pair;
        if (slice < 1) slice = 1;
        if (slice > (size_t)n_pairs) slice = (size_t)n_pairs;
        w->slice = slice;
        w->band = slice * w->per_pair;
    }
    return GMS_OK;
}

// Grows the context's workspaces to `w`. Growing frees the old block, so everything that may still use it is waited for first;
// inside a stream capture that is impossible (and so is the allocation): GMS_ERR_NOT_RESERVED.
int grow_workspace(gms_ctx* c, const WsNeed& w, hipStream_t st)
{
    if (w.partial <= c->partial_ws.cap && w.big <= c->big_ws.cap && w.band <= c->band_ws.cap) return GMS_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return GMS_ERR_NOT_RESERVED;
    GMS_HIP(hipStreamSynchronize(st));
    if (c->ws_pending) {
        GMS_HIP(hipEventSynchronize(c->ws_event));
        c->ws_pending = false;
    }
    GMS_HIP(c->partial_ws.reserve(w.partial));
    GMS_HIP(c->big_ws.reserve(w.big));
    GMS_HIP(c->band_ws.reserve(w.band));
    return GMS_OK;
}

// The scratch of gms_recover_pose_device (vote bytes + counters): grown like the filter's workspaces -- never inside a stream
// capture, and only after everything that may still use the old block has finished.
int grow_pose_ws(gms_ctx* c, size_t bytes, hipStream_t st)
{
    if (bytes <= c->pose_ws.cap) return GMS_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return GMS_ERR_NOT_RESERVED;
    GMS_HIP(hipStreamSynchronize(st));
    GMS_HIP(hipDeviceSynchronize());  // (an earlier call may have run on another stream of the caller's)
    GMS_HIP(c->pose_ws.reserve(bytes));
    return GMS_OK;
}

// The filter launches of one call on stream `st`. The caller holds c->mu and has selected the device.
int filter_launch(gms_ctx* c, hipStream_t st, const float* d_pts, const int64_t* d_frame_off, int n_frames,
                  const gms_pair* d_pairs, int n_pairs, int max_m, const gms_dmatch* d_matches,
                  int with_rotation, int with_scale, double threshold_factor,
                  gms_dmatch* d_out, gms_pair_result* d_results, uint8_t* d_mask, bool validate_pairs = true)
{
    if (n_pairs < 0 || max_m < 0 || n_frames < 0) return GMS_ERR_BAD_ARG;
    if (n_pairs == 0) return GMS_OK;
    if (!d_frame_off || !d_pairs || !d_results) return GMS_ERR_BAD_ARG;
    if (max_m > 0 && (!d_matches || !d_out || !d_pts)) return GMS_ERR_BAD_ARG;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    if (gms::filter_pick_kpt(max_m) == 0 && !capturing) {
        // large pairs: the streamed kernels' report of pairs they could not keep (read without waiting: it only steers which of two
        // bit-identical paths runs)
        const uint32_t now = ((volatile uint32_t*)c->verdict)[2];
        if (now != c->overflow_seen) {
            c->overflow_seen = now;
            c->stream_penalty = 64;
        } else if (c->stream_penalty > 0) {
            --c->stream_penalty;
        }
    }
    WsNeed w;
    GMS_TRY(plan_workspace(c, n_pairs, max_m, with_rotation != 0, with_scale != 0, d_mask == nullptr, c->stream_penalty == 0, &w));
    GMS_TRY(grow_workspace(c, w, st));
    const bool uses_ws = w.partial || w.big || w.band;
    if (uses_ws && !capturing && c->ws_pending && c->ws_stream != st) GMS_HIP(hipStreamWaitEvent(st, c->ws_event, 0));

    gms::FilterParams p{};
    const int kpt = w.kpt;
    p.table_slots = kpt ? gms::filter_table_slots(kpt) : 0;
    p.region_shift = kpt ? gms::filter_region_shift(kpt) : 0;
    p.pts = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(d_pts) + gms::kTableHeaderBytes);  // behind the table's header
    p.frame_off = d_frame_off;
    p.n_frames = n_frames;
    p.pairs = d_pairs;
    p.n_pairs = n_pairs;
    p.stagger_blocks = c->n_cus;
    p.prefetch_type = knobs().prefetch_type;
    p.prefetch_ahead = knobs().prefetch_ahead > 0 ? knobs().prefetch_ahead : c->n_cus;
    p.matches = d_matches;
    p.out = d_out;
    p.results = d_results;
    p.mask = d_mask;
    p.with_rotation = with_rotation ? 1 : 0;
    p.with_scale = with_scale ? 1 : 0;
    p.threshold_factor = threshold_factor;
    p.pair_flags = nullptr;
    p.partial = nullptr;
    p.dealt = 0;
    right_grids(p.right_w, p.right_h);
    // the byte-matrix path is tried first whenever there are no scale hypotheses (the reference's default flags,
    // DisparityUtil.cpp:149,299)
    p.dense = (knobs().dense_on && !with_scale) ? 1 : 0;
    if (c->verdict_pending && !capturing && hipEventQuery(c->verdict_event) == hipSuccess) {
        c->use_dealt = c->verdict[0] != 0u ? 1 : 0;
        c->use_probe = (int)c->verdict[1];
        c->verdict_pending = false;
    }
    const int force_deal = c->opt_deal >= 0 ? c->opt_deal : knobs().deal, force_probe = c->opt_probe >= 0 ? c->opt_probe : knobs().scale_probe;
    const bool scales_matrix = kpt && with_scale && knobs().dense_on;  // (the byte-matrix kernel of the scale hypotheses deals too)
    p.dealt = ((p.dense || scales_matrix) && kpt && (force_deal >= 0 ? force_deal != 0 : c->use_dealt != 0)) ? 1 : 0;
    // First-round stagger: the spread is about two thirds of a pair's duration on the path the launch will mostly take -- 16 us (byte
    // matrix) / 72 us (hashed) at 10k matches, in proportion to max_m -- in ticks of the 100 MHz wall clock. Only launches of
    // at least four dispatch rounds are staggered, and none with scale hypotheses on the byte matrix: a pair takes ten times as
    // long there and its loads are a small share of it, so the spread only delays (1.12 ms against 1.18 per 2048 pairs without).
    const bool scales_on_matrix = with_scale && knobs().dense_on;
    p.stagger_ticks = 0;
    if (n_pairs >= 4 * p.stagger_blocks && kpt && max_m >= 2048) {  // (measured neutral at 4k matches, -1 % at 500: tools/measure_misc.py)
        const double us10k = knobs().stagger_us >= 0 ? (double)knobs().stagger_us : (scales_on_matrix ? 0.0 : p.dense ? 16.0 : 72.0);
        p.stagger_ticks = (int)(us10k * 100.0 * max_m / 10000.0);
    }
#ifdef GMS_PHASE_TIMING
    p.diag = g_diag;
#endif
    p.probe_scales = 0;
    p.probe_stats = nullptr;
    {
        void* dflag = nullptr;
        p.overflow_events = hipHostGetDevicePointer(&dflag, c->verdict, 0) == hipSuccess ? (uint32_t*)dflag + 2 : nullptr;
    }
    if (kpt && with_scale) {
        // the scales finer than 20 x 20 and the 14 x 14 one are probed (2, 3, 4); the measuring launches are left out of stream captures
        const bool measuring = force_probe < 0 && !capturing && (c->scale_launches++ & 15u) == 0u;
        // which scales are probed (every scale but 1, which the byte-matrix kernel evaluates first) and which of the two fine ones try
        // four-bit entries first: forced, or everything in a measuring launch, or what the last verdict said paid -- scale by scale
        const int all = 0x1D | (24 << 8);
        const int mask = force_probe >= 0 ? (force_probe != 0 ? 0x1D | (knobs().probe_nibble << 8) : 0) : (measuring ? all : c->use_probe);
        p.probe_scales = mask & 0x1D;
        p.probe_nibble = (mask >> 8) & 24 & (knobs().probe_nibble_set ? knobs().probe_nibble : 24);
        p.probe_stats = measuring ? (uint32_t*)c->probe_stats.p : nullptr;
    }
    if (kpt && with_scale && knobs().dense_on) {
        // scale hypotheses: scales 0..3 on the byte matrix, the last on the hashed path (two launches, one record per pair)
        p.partial = (uint32_t*)c->partial_ws.p;
        GMS_HIP(gms::launch_filter_scales(p, kpt, n_pairs, st));
        if (!capturing && p.probe_stats == nullptr && (c->dense_launches++ & 15u) == 0u) {  // the spatial-order probe of this batch, for later launches (one verdict event: not beside a measuring launch)
            void* dflag = nullptr;
            GMS_HIP(hipHostGetDevicePointer(&dflag, c->verdict, 0));
            GMS_HIP(gms::launch_order_probe(p, (uint32_t*)dflag, st));
            GMS_HIP(hipEventRecord(c->verdict_event, st));
            c->verdict_pending = true;
        }
        if (p.probe_stats != nullptr) {
            void* dflag = nullptr;
            GMS_HIP(hipHostGetDevicePointer(&dflag, c->verdict, 0));
            GMS_HIP(gms::launch_probe_verdict(p.probe_stats, (uint32_t*)dflag + 1, st));
            GMS_HIP(hipEventRecord(c->verdict_event, st));
            c->verdict_pending = true;
        }
    } else if (kpt) {
        GMS_HIP(gms::launch_filter(p, kpt, n_pairs, st));
        // every sixteenth byte-matrix launch (and the first) is followed by the spatial-order probe of its batch, for later launches
        if (p.dense && !capturing && (c->dense_launches++ & 15u) == 0u) {
            void* dflag = nullptr;
            GMS_HIP(hipHostGetDevicePointer(&dflag, c->verdict, 0));
            GMS_HIP(gms::launch_order_probe(p, (uint32_t*)dflag, st));
            GMS_HIP(hipEventRecord(c->verdict_event, st));
            c->verdict_pending = true;
        }
    } else if (knobs().band_on) {
        // Large pairs on the LDS kernels, a slice of the batch at a time so that the per-pair workspace (lists, histogram,
        // masks) stays bounded: three bands of rows for the default flags, tiles of left cells with three launches per
        // scale hypothesis otherwise. Pairs they flag (a cell above 65 535 matches) fall through to the HBM-slab kernel (a
        // fixed crew of persistent workgroups), which looks at flagged pairs only.
        const bool plain = !with_rotation && !with_scale;
        for (int s0 = 0; s0 < n_pairs; s0 += (int)w.slice) {
            gms::FilterParams ps = p;
            ps.pairs = d_pairs + s0;
            ps.results = d_results + s0;
            ps.n_pairs = (n_pairs - s0 < (int)w.slice) ? n_pairs - s0 : (int)w.slice;
            const uint32_t* flags = nullptr;
            // streamed byte matrix: one workgroup per (pair, scale, grid type, band) with scale hypotheses -- and without them when the
            // slice has fewer pairs than the chip has CUs (four workgroups per pair then) --, else one workgroup per pair
            if (w.stream && (with_scale || ps.n_pairs < c->n_cus)) GMS_HIP(gms::launch_filter_stream(ps, w.mcap, c->band_ws.p, &flags, st));
            else if (w.stream) GMS_HIP(gms::launch_filter_stream_dense(ps, w.mcap, c->band_ws.p, &flags, st));
            else if (plain) GMS_HIP(gms::launch_filter_band(ps, w.mcap, c->band_ws.p, &flags, st));
            else GMS_HIP(gms::launch_filter_tiles(ps, w.mcap, c->band_ws.p, &flags, st));
            ps.pair_flags = flags;
            const int wg = ps.n_pairs < c->n_cus ? ps.n_pairs : c->n_cus;
            GMS_HIP(gms::launch_filter_big(ps, w.mcap, wg, (uint32_t*)c->big_ws.p, st));
        }
    } else {
        const int n_wg = n_pairs < c->n_cus ? n_pairs : c->n_cus;
        GMS_HIP(gms::launch_filter_big(p, w.mcap, n_wg, (uint32_t*)c->big_ws.p, st));
    }
    c->last_dealt = p.dealt;
    c->last_probe = p.probe_scales | (p.probe_scales ? p.probe_nibble << 8 : 0);
    c->last_kpt = kpt;
    c->last_stagger_ticks = p.stagger_ticks;
    // pair-table validation (ranges [match_off, match_off + m) must be disjoint: include/gms.h), behind the filter: offenders'
    // status becomes GMS_ERR_BAD_ARG. The first launch of a context and every sixteenth; never inside a stream capture.
    if (validate_pairs && n_pairs > 1 && !capturing && knobs().check_pairs != 0 && (knobs().check_pairs == 1 || (c->filter_launches & 15u) == 0u))
        GMS_HIP(gms::launch_check_pairs(d_pairs, n_pairs, d_results, (uint32_t*)c->probe_stats.p + 16 + (c->filter_launches & 7u), st));  // (a flag word of its own per launch: launches of one context may run on different streams)
    if (!capturing) ++c->filter_launches;
    if (uses_ws && !capturing) {
        GMS_HIP(hipEventRecord(c->ws_event, st));
        c->ws_stream = st;
        c->ws_pending = true;
    }
    return GMS_OK;
}

// ---- the host-pointer paths --------------------------------------------------------------------------------------------
// Staging block of a chunk, host (pinned) and device alike:  [ pairs (24 B each, 16-aligned) | matches (16 B each) ]
// and coming back:                                            [ results (16 B each) | out (16 B each) ]
size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// Host-side staging: the copies between the caller's (pageable) arrays and the pinned blocks the copy engines work from. One core
// moves 25-35 GB/s here, four move 65-110 (tools/ubench/host_copy_rate.cpp), the copy engines 30-55 GB/s per direction: the copies of
// a chunk are cut into parts of 1 MB that a PERSISTENT pool of threads (created with the context's first host batch, not per chunk)
// takes from a shared counter; the calling thread works along.
struct CopyJob {
    void* dst;
    const void* src;
    size_t bytes;   // of dst
    int pack_xy;    // 0: memcpy; 1: src = gms_keypoint records, dst = (pt.x, pt.y) float pairs, 8 bytes each (bytes % 8 == 0)
};

// (pt.x, pt.y) of n keypoints, 8 bytes each: all the filter reads of a cv::KeyPoint (DLL@0x1800485d4) and all that has to
// cross PCIe. Pure data movement; the divide by the image size happens on the GPU (normalize_kernel).
void pack_xy(const gms_keypoint* kp, size_t n, float* dst)
{
    for (size_t i = 0; i < n; ++i) {
        dst[2 * i] = kp[i].x;
        dst[2 * i + 1] = kp[i].y;
    }
}

class CopyPool {
public:
    ~CopyPool() { shutdown(); }
    void shutdown()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (std::thread& t : threads_) t.join();
        threads_.clear();
        stop_ = false;
    }
    // Runs the jobs (disjoint destinations) and returns when every byte has landed.
    void run(const std::vector<CopyJob>& jobs)
    {
        if (jobs.empty()) return;
        starts_.resize(jobs.size() + 1);
        starts_[0] = 0;
        for (size_t i = 0; i < jobs.size(); ++i) starts_[i + 1] = starts_[i] + jobs[i].bytes;
        const size_t total = starts_.back();
        if (total == 0) return;
        const unsigned n_parts = (unsigned)((total + kPart - 1) / kPart);
        if (n_parts <= 1) {
            part(jobs, 0);
            return;
        }
        if (threads_.empty()) start();
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_ = &jobs;
            n_parts_ = n_parts;
            next_.store(0, std::memory_order_relaxed);
            done_ = 0;
            ++gen_;
        }
        cv_work_.notify_all();
        unsigned mine = 0;
        for (unsigned i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < n_parts; ++mine) part(jobs, i);
        std::unique_lock<std::mutex> lk(mu_);
        done_ += mine;
        cv_done_.wait(lk, [&] { return done_ == n_parts_; });
        jobs_ = nullptr;
    }

private:
    static constexpr size_t kPart = (size_t)1 << 20;
    void start()
    {
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned n = std::min(hw ? hw : 4u, 8u) - 1;   // plus the calling thread
        for (unsigned t = 0; t < n; ++t) threads_.emplace_back([this] { worker(); });
    }
    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            const std::vector<CopyJob>* jobs;
            unsigned n_parts;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                jobs = jobs_;
                n_parts = n_parts_;
            }
            if (!jobs) continue;
            unsigned mine = 0;
            for (unsigned i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < n_parts; ++mine) part(*jobs, i);
            std::lock_guard<std::mutex> lk(mu_);
            done_ += mine;
            if (done_ == n_parts_) cv_done_.notify_all();
        }
    }
    // bytes [i, i + 1) * kPart of the concatenated destinations
    void part(const std::vector<CopyJob>& jobs, unsigned i) const
    {
        const size_t lo = (size_t)i * kPart, hi = std::min(lo + kPart, starts_.back());
        size_t j = (size_t)(std::upper_bound(starts_.begin(), starts_.end(), lo) - starts_.begin()) - 1;
        for (; j < jobs.size() && starts_[j] < hi; ++j) {
            const size_t a = std::max(lo, starts_[j]) - starts_[j], b = std::min(hi, starts_[j + 1]) - starts_[j];
            if (a >= b) continue;
            if (jobs[j].pack_xy) pack_xy((const gms_keypoint*)jobs[j].src + a / 8, (b - a) / 8, (float*)((char*)jobs[j].dst + a));
            else std::memcpy((char*)jobs[j].dst + a, (const char*)jobs[j].src + a, b - a);
        }
    }
    std::vector<std::thread> threads_;
    std::vector<size_t> starts_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    const std::vector<CopyJob>* jobs_ = nullptr;
    unsigned n_parts_ = 0, done_ = 0;
    std::atomic<unsigned> next_{0};
    uint64_t gen_ = 0;
    bool stop_ = false;
};

}  // namespace

extern "C" {

const char* gms_version(void) { return "mi355-gms 0.2 (gfx950)"; }

int gms_last_hip_error(void) { return t_last_hip; }

const char* gms_error_string(int code)
{
    switch (code) {
    case GMS_OK: return "ok";
    case GMS_ERR_BAD_ARG: return "bad argument";
    case GMS_ERR_DOMAIN: return "input outside the domain on which the reference is defined";
    case GMS_ERR_HIP: return "HIP runtime error";
    case GMS_ERR_NO_DEVICE: return "no usable HIP device";
    case GMS_ERR_CAPACITY: return "too many matches per pair for this build";
    case GMS_ERR_IO: return "dataset file: cannot open, truncated, or not a GMSFRM01 file";
    case GMS_ERR_NO_MODEL: return "two-view stage: too few correspondences, or no essential matrix / pose found";
    case GMS_ERR_NOT_RESERVED: return "workspace not reserved for this shape (stream capture in progress)";
    default: return "unknown error";
    }
}

int gms_max_matches(void) { return gms::kBigMaxMatches; }

int gms_ctx_create(int device, gms_ctx** out_ctx)
{
    if (!out_ctx) return GMS_ERR_BAD_ARG;
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        t_last_hip = (int)e;
        return GMS_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) return GMS_ERR_BAD_ARG;
    GMS_HIP(hipSetDevice(device));
    gms_ctx* c = new (std::nothrow) gms_ctx;
    if (!c) return GMS_ERR_BAD_ARG;
    c->device = device;
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->lane[1].stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->lane[2].stream, hipStreamNonBlocking);
    for (Lane& l : c->lane) {
        if (e == hipSuccess) e = hipEventCreateWithFlags(&l.ev_res, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&l.ev_out, hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ws_event, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->verdict_event, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->verdict, 64, hipHostMallocDefault);
    if (e == hipSuccess) {
        c->verdict[0] = 0;
        c->verdict[1] = 0x1D | (16 << 8);
        c->verdict[2] = 0;
    }
    if (e == hipSuccess) e = c->probe_stats.reserve(128);  // [0..13] probe counters (probe_verdict_kernel), [16..23] flag words of the pair-table check
    if (e == hipSuccess) e = hipMemset(c->probe_stats.p, 0, 128);
    if (e == hipSuccess) e = gms::init_filter_kernels();
    if (e == hipSuccess) e = gms::init_band_kernels();
    if (e == hipSuccess) e = gms::init_stream_kernels();
    if (e == hipSuccess) e = gms::init_big_kernels();
    if (e != hipSuccess) {
        t_last_hip = (int)e;
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
        if (c->lane[1].stream) (void)hipStreamDestroy(c->lane[1].stream);
        if (c->lane[2].stream) (void)hipStreamDestroy(c->lane[2].stream);
        for (Lane& l : c->lane) {
            if (l.ev_res) (void)hipEventDestroy(l.ev_res);
            if (l.ev_out) (void)hipEventDestroy(l.ev_out);
        }
        if (c->ws_event) (void)hipEventDestroy(c->ws_event);
        if (c->verdict_event) (void)hipEventDestroy(c->verdict_event);
        if (c->verdict) (void)hipHostFree(c->verdict);
        c->probe_stats.release();
        delete c;
        return GMS_ERR_HIP;
    }
    c->stream = c->own_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
    *out_ctx = c;
    return GMS_OK;
}

int gms_ctx_destroy(gms_ctx* c)
{
    if (!c) return GMS_OK;
    {
        std::lock_guard<std::mutex> lock(c->mu);
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->own_stream);
        (void)hipStreamSynchronize(c->lane[1].stream);
        (void)hipStreamSynchronize(c->lane[2].stream);
        if (c->ws_pending) (void)hipEventSynchronize(c->ws_event);
        DevBuf* bufs[] = {&c->aux, &c->big_ws, &c->band_ws, &c->partial_ws, &c->pose_ws, &c->probe_stats, &c->tab_kp, &c->tab_pts, &c->tab_small};
        if (c->verdict) (void)hipHostFree(c->verdict);
        for (DevBuf* b : bufs) b->release();
        for (Lane& l : c->lane) {
            l.hin.release();
            l.hout.release();
            l.din.release();
            l.dout.release();
            (void)hipEventDestroy(l.ev_res);
            (void)hipEventDestroy(l.ev_out);
        }
        delete c->pool;
        c->pool = nullptr;
        (void)hipStreamDestroy(c->lane[1].stream);
        (void)hipStreamDestroy(c->lane[2].stream);
        (void)hipStreamDestroy(c->own_stream);
        (void)hipEventDestroy(c->ws_event);
        (void)hipEventDestroy(c->verdict_event);
    }
    delete c;
    return GMS_OK;
}

int gms_ctx_set_stream(gms_ctx* c, void* hip_stream)
{
    if (!c) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return GMS_OK;
}

int gms_ctx_synchronize(gms_ctx* c)
{
    if (!c) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(hipStreamSynchronize(c->stream));
    if (c->verdict_pending && hipEventQuery(c->verdict_event) == hipSuccess) {  // (recorded on another stream: maybe not yet)
        c->use_dealt = c->verdict[0] != 0u ? 1 : 0;
        c->use_probe = (int)c->verdict[1];
        c->verdict_pending = false;
    }
    return GMS_OK;
}

int gms_ctx_set_option(gms_ctx* c, int option, int value)
{
    if (!c || value < -1 || value > 1) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    switch (option) {
    case GMS_OPTION_DEAL: c->opt_deal = value; return GMS_OK;
    case GMS_OPTION_SCALE_PROBE: c->opt_probe = value; return GMS_OK;
    default: return GMS_ERR_BAD_ARG;
    }
}

int gms_ctx_query(gms_ctx* c, int what, int64_t* value)
{
    if (!c || !value) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    switch (what) {
    case GMS_QUERY_LAST_DEALT: *value = c->last_dealt; return GMS_OK;
    case GMS_QUERY_LAST_SCALE_PROBE: *value = c->last_probe; return GMS_OK;
    case GMS_QUERY_LAST_KPT: *value = c->last_kpt; return GMS_OK;
    case GMS_QUERY_LAUNCHES: *value = (int64_t)c->filter_launches; return GMS_OK;
    case GMS_QUERY_CUS: *value = c->n_cus; return GMS_OK;
    case GMS_QUERY_PREFETCH_TYPE: *value = knobs().prefetch_type; return GMS_OK;
    case GMS_QUERY_PREFETCH_AHEAD: *value = knobs().prefetch_ahead > 0 ? knobs().prefetch_ahead : c->n_cus; return GMS_OK;
    case GMS_QUERY_STAGGER_TICKS: *value = c->last_stagger_ticks; return GMS_OK;
    default: return GMS_ERR_BAD_ARG;
    }
}

int gms_ctx_reserve(gms_ctx* c, int n_pairs, int max_m, int with_rotation, int with_scale)
{
    if (!c || n_pairs < 0 || max_m < 0) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    WsNeed w;
    // with and without a caller-provided mask array, on the streamed and on the band / tile kernels (the library picks per launch): the largest
    GMS_TRY(plan_workspace(c, n_pairs, max_m, with_rotation != 0, with_scale != 0, true, true, &w));
    for (int variant = 1; variant < 4; ++variant) {
        WsNeed w2;
        GMS_TRY(plan_workspace(c, n_pairs, max_m, with_rotation != 0, with_scale != 0, (variant & 1) == 0, (variant & 2) == 0, &w2));
        w.band = std::max(w.band, w2.band);
    }
    GMS_TRY(grow_workspace(c, w, c->stream));
    return grow_pose_ws(c, (size_t)max_m + 48, c->stream);  // gms_recover_pose_device on a pair of this size
}

// header (16) | points (8 each) | lcode, rcode (2 each) | scode (4 each) | 16 spare bytes (the staging copies read whole uint4s)
int64_t gms_frame_table_bytes(int64_t total_kp) { return total_kp < 0 ? 0 : gms::kTableHeaderBytes + total_kp * 16 + 16; }

int gms_normalize_device(gms_ctx* c, const gms_keypoint* d_kp, const int64_t* d_frame_off,
                         const int32_t* d_wh, int n_frames, int64_t total_kp, float* d_pts)
{
    if (!c || n_frames < 0 || total_kp < 0) return GMS_ERR_BAD_ARG;
    if (total_kp == 0 || n_frames == 0) return GMS_OK;
    if (!d_kp || !d_frame_off || !d_wh || !d_pts) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_normalize(d_kp, (int)sizeof(gms_keypoint), d_frame_off, d_wh, n_frames, total_kp, d_pts, c->stream));
    return GMS_OK;
}

int gms_filter_device(gms_ctx* c, const float* d_pts, const int64_t* d_frame_off, int n_frames,
                      const gms_pair* d_pairs, int n_pairs, int max_m, const gms_dmatch* d_matches,
                      int with_rotation, int with_scale, double threshold_factor,
                      gms_dmatch* d_out, gms_pair_result* d_results, uint8_t* d_mask)
{
    if (!c) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    return filter_launch(c, c->stream, d_pts, d_frame_off, n_frames, d_pairs, n_pairs, max_m, d_matches, with_rotation,
                         with_scale, threshold_factor, d_out, d_results, d_mask);
}

int gms_match_ctx(gms_ctx* c, const gms_keypoint* kp1, int n1, int w1, int h1,
                  const gms_keypoint* kp2, int n2, int w2, int h2, const gms_dmatch* matches, int m,
                  int with_rotation, int with_scale, double threshold_factor,
                  gms_dmatch* out, int* n_out, gms_pair_result* result)
{
    if (n_out) *n_out = 0;
    if (result) *result = gms_pair_result{0, -1, -1, GMS_OK};
    if (!c || !n_out) return GMS_ERR_BAD_ARG;
    if (n1 < 0 || n2 < 0 || m < 0 || w1 <= 0 || h1 <= 0 || w2 <= 0 || h2 <= 0) return GMS_ERR_BAD_ARG;
    if ((n1 > 0 && !kp1) || (n2 > 0 && !kp2) || (m > 0 && (!matches || !out))) return GMS_ERR_BAD_ARG;
    if (m > gms_max_matches()) return GMS_ERR_CAPACITY;

    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    Lane& L = c->lane[0];
    // Everything the call sends travels as ONE block through pinned memory (one copy-engine transfer, no bounce through the
    // runtime's pageable staging):  header (64 B) | matches (16 B each) | (pt.x, pt.y) of both frames (8 B each)
    struct alignas(8) CallHeader {
        int64_t foff[3];
        int32_t wh[4];
        gms_pair pair;
    };
    static_assert(sizeof(CallHeader) == 64 && offsetof(CallHeader, wh) == 24 && offsetof(CallHeader, pair) == 40, "header layout");
    const size_t nkp = (size_t)n1 + (size_t)n2;
    const size_t off_m = sizeof(CallHeader), off_xy = off_m + (size_t)m * sizeof(gms_dmatch);
    const size_t in_bytes = off_xy + nkp * 8;
    const size_t out_bytes = sizeof(gms_pair_result) + (size_t)m * sizeof(gms_dmatch);  // result | out
    GMS_HIP(L.hin.reserve(in_bytes));
    GMS_HIP(L.hout.reserve(out_bytes));
    // (a DevBuf that grows frees its old block: earlier work of this lane is complete -- every host-pointer call ends
    // synchronised -- unless the caller switched streams in between, which the synchronize below covers)
    if (in_bytes > L.din.cap || out_bytes > L.dout.cap || (size_t)gms_frame_table_bytes((int64_t)nkp) > c->tab_pts.cap) GMS_HIP(hipDeviceSynchronize());
    GMS_HIP(L.din.reserve(in_bytes));
    GMS_HIP(L.dout.reserve(out_bytes));
    GMS_HIP(c->tab_pts.reserve((size_t)gms_frame_table_bytes((int64_t)nkp)));
    char* hin = (char*)L.hin.p;
    const CallHeader hdr = {{0, n1, (int64_t)n1 + n2}, {w1, h1, w2, h2}, {0, 1, m, 0, 0}};
    std::memcpy(hin, &hdr, sizeof hdr);
    if (m) std::memcpy(hin + off_m, matches, (size_t)m * sizeof(gms_dmatch));
    pack_xy(kp1, (size_t)n1, (float*)(hin + off_xy));
    pack_xy(kp2, (size_t)n2, (float*)(hin + off_xy) + 2 * (size_t)n1);
    // Small calls (up to 16k matches) skip the copy engines altogether: the kernels read the pinned block and write the pinned
    // result block directly over PCIe (every input byte is read once -- the records stay in registers -- and every output byte
    // written once), which saves the fixed cost of two DMA operations. Larger calls stage through device memory.
    const bool zero_copy = m <= 16384;
    const char* din = (const char*)L.din.p;
    char* dout = (char*)L.dout.p;
    if (zero_copy) {
        void *dev_in = nullptr, *dev_out = nullptr;
        GMS_HIP(hipHostGetDevicePointer(&dev_in, L.hin.p, 0));
        GMS_HIP(hipHostGetDevicePointer(&dev_out, L.hout.p, 0));
        din = (const char*)dev_in;
        dout = (char*)dev_out;
    } else {
        GMS_HIP(hipMemcpyAsync(L.din.p, hin, in_bytes, hipMemcpyHostToDevice, st));
    }
    const int64_t* d_foff = (const int64_t*)din;
    const int32_t* d_wh = (const int32_t*)(din + offsetof(CallHeader, wh));
    const gms_pair* d_pair = (const gms_pair*)(din + offsetof(CallHeader, pair));
    gms_pair_result* d_res = (gms_pair_result*)dout;
    gms_dmatch* d_out = (gms_dmatch*)(dout + sizeof(gms_pair_result));
    if (nkp) GMS_HIP(gms::launch_normalize(din + off_xy, 8, d_foff, d_wh, 2, (int64_t)nkp, (float*)c->tab_pts.p, st));
    GMS_TRY(filter_launch(c, st, (const float*)c->tab_pts.p, d_foff, 2, d_pair, 1, m, (const gms_dmatch*)(din + off_m),
                          with_rotation, with_scale, threshold_factor, d_out, d_res, nullptr));
    // Staged calls fetch the result record first and then exactly the survivors.
    if (!zero_copy) GMS_HIP(hipMemcpyAsync(L.hout.p, L.dout.p, sizeof(gms_pair_result), hipMemcpyDeviceToHost, st));
    GMS_HIP(hipStreamSynchronize(st));
    const gms_pair_result r = *(const gms_pair_result*)L.hout.p;
    if (result) *result = r;
    if (r.status != GMS_OK) return r.status;
    if (r.n_inliers < 0 || r.n_inliers > m) return GMS_ERR_HIP;  // (cannot happen: the kernels count what they wrote)
    const size_t keep_bytes = (size_t)r.n_inliers * sizeof(gms_dmatch);
    if (!zero_copy && keep_bytes) {
        GMS_HIP(hipMemcpyAsync((char*)L.hout.p + sizeof(gms_pair_result), d_out, keep_bytes, hipMemcpyDeviceToHost, st));
        GMS_HIP(hipStreamSynchronize(st));
    }
    // only the survivors reach the caller's array: what lies beyond *n_out is left untouched
    if (keep_bytes) std::memcpy(out, (const char*)L.hout.p + sizeof(gms_pair_result), keep_bytes);
    *n_out = r.n_inliers;
    return GMS_OK;
}

int gms_match(const gms_keypoint* kp1, int n1, int w1, int h1, const gms_keypoint* kp2, int n2, int w2, int h2,
              const gms_dmatch* matches, int m, int with_rotation, int with_scale, double threshold_factor,
              gms_dmatch* out, int* n_out)
{
    static std::mutex mu;
    static gms_ctx* def = nullptr;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!def) {
            int rc = gms_ctx_create(0, &def);
            if (rc != GMS_OK) {
                if (n_out) *n_out = 0;
                return rc;
            }
        }
    }
    return gms_match_ctx(def, kp1, n1, w1, h1, kp2, n2, w2, h2, matches, m, with_rotation, with_scale,
                         threshold_factor, out, n_out, nullptr);
}

int gms_filter_host_batch(gms_ctx* c, const gms_keypoint* kp, const int64_t* frame_off, const int32_t* wh, int n_frames,
                          const gms_pair* pairs, int n_pairs, const gms_dmatch* matches,
                          int with_rotation, int with_scale, double threshold_factor,
                          gms_dmatch* out, gms_pair_result* results)
{
    if (!c || n_frames < 0 || n_pairs < 0) return GMS_ERR_BAD_ARG;
    if (n_pairs == 0) return GMS_OK;
    if (!frame_off || !wh || !pairs || !results || n_frames == 0) return GMS_ERR_BAD_ARG;
    const int64_t total_kp = frame_off[n_frames];
    if (total_kp < 0 || (total_kp > 0 && !kp)) return GMS_ERR_BAD_ARG;
    int max_m = 0;
    bool in_order = true;
    for (int i = 0; i < n_pairs; ++i) {
        if (pairs[i].m < 0 || pairs[i].match_off < 0) return GMS_ERR_BAD_ARG;
        if (pairs[i].m > gms_max_matches()) return GMS_ERR_CAPACITY;
        max_m = std::max(max_m, pairs[i].m);
        if (i + 1 < n_pairs && pairs[i].match_off + pairs[i].m > pairs[i + 1].match_off) in_order = false;
    }
    if (max_m > 0 && (!matches || !out)) return GMS_ERR_BAD_ARG;
    if (!in_order) {  // the pairs' match ranges must be disjoint (include/gms.h): sort the non-empty ones by start and look at the neighbours
        std::vector<std::pair<int64_t, int64_t>> r;
        r.reserve((size_t)n_pairs);
        for (int i = 0; i < n_pairs; ++i)
            if (pairs[i].m > 0) r.emplace_back(pairs[i].match_off, pairs[i].match_off + pairs[i].m);
        std::sort(r.begin(), r.end());
        for (size_t i = 1; i < r.size(); ++i)
            if (r[i - 1].second > r[i].first) return GMS_ERR_BAD_ARG;
    }

    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    c->lane[0].stream = c->own_stream;
    // everything earlier on the context's streams and workspaces is complete before buffers are regrown and reused
    GMS_HIP(hipStreamSynchronize(c->stream));
    GMS_HIP(hipStreamSynchronize(c->lane[0].stream));
    GMS_HIP(hipStreamSynchronize(c->lane[1].stream));
    GMS_HIP(hipStreamSynchronize(c->lane[2].stream));

    // ---- chunks: runs of consecutive pairs of at most kChunkMatches matches (a pair larger than that is a chunk of its own)
    const size_t kChunkMatches = (size_t)2 << 20;  // 32 MB of match records per chunk
    const int kChunkPairs = 8192;
    struct Chunk { int first, count; size_t matches; int max_m; };
    std::vector<Chunk> chunks;
    for (int i = 0; i < n_pairs;) {
        Chunk ch{i, 0, 0, 0};
        while (i < n_pairs && ch.count < kChunkPairs && (ch.count == 0 || ch.matches + (size_t)pairs[i].m <= kChunkMatches)) {
            ch.matches += (size_t)pairs[i].m;
            ch.max_m = std::max(ch.max_m, pairs[i].m);
            ++ch.count;
            ++i;
        }
        chunks.push_back(ch);
    }
    size_t cap_m = 0, cap_p = 0;
    for (const Chunk& ch : chunks) {
        cap_m = std::max(cap_m, ch.matches);
        cap_p = std::max(cap_p, (size_t)ch.count);
    }
    // a lane's blocks, host (pinned) and device alike:   in  [ pairs (24 B each, 16-aligned) | matches (16 B each) ]
    //                                                    out [ results (16 B each) | survivor total (16 B) | out (16 B each) ]
    // and, once a chunk is filtered, its survivors packed back to back over the chunk's own match records in the device's in-block
    const size_t in_pairs_bytes = align16(cap_p * sizeof(gms_pair));
    const size_t in_bytes = std::max(in_pairs_bytes + cap_m * sizeof(gms_dmatch), (size_t)total_kp * 8);
    const size_t out_res_bytes = cap_p * sizeof(gms_pair_result) + 16;
    const size_t out_bytes = out_res_bytes + cap_m * sizeof(gms_dmatch);
    for (Lane& L : c->lane) {
        GMS_HIP(L.hin.reserve(&L == &c->lane[0] ? in_bytes : in_pairs_bytes + cap_m * sizeof(gms_dmatch)));
        GMS_HIP(L.hout.reserve(out_bytes));
        GMS_HIP(L.din.reserve(in_pairs_bytes + cap_m * sizeof(gms_dmatch)));
        GMS_HIP(L.dout.reserve(out_bytes));
    }
    {   // the workspaces for the largest chunk, once, so that no launch below has to grow them mid-pipeline
        WsNeed w;
        GMS_TRY(plan_workspace(c, (int)cap_p, max_m, with_rotation != 0, with_scale != 0, true, true, &w));
        WsNeed w2;
        GMS_TRY(plan_workspace(c, (int)cap_p, max_m, with_rotation != 0, with_scale != 0, true, false, &w2));
        w.band = std::max(w.band, w2.band);
        GMS_TRY(grow_workspace(c, w, c->lane[0].stream));
    }
    if (!c->pool) c->pool = new (std::nothrow) CopyPool;
    if (!c->pool) return GMS_ERR_BAD_ARG;
    CopyPool& pool = *c->pool;
    std::vector<CopyJob> jobs;

    // ---- the frame table: (pt.x, pt.y) of every keypoint packed into pinned memory by the pool, one transfer, normalizePoints on the GPU
    const size_t small_bytes = align16((size_t)(n_frames + 1) * 8) + (size_t)n_frames * 8;
    GMS_HIP(c->tab_kp.reserve((size_t)total_kp * 8));
    GMS_HIP(c->tab_pts.reserve((size_t)gms_frame_table_bytes(total_kp)));
    GMS_HIP(c->tab_small.reserve(small_bytes));
    {
        hipStream_t st = c->lane[0].stream;
        GMS_HIP(hipMemcpyAsync(c->tab_small.p, frame_off, (size_t)(n_frames + 1) * 8, hipMemcpyHostToDevice, st));
        GMS_HIP(hipMemcpyAsync((char*)c->tab_small.p + align16((size_t)(n_frames + 1) * 8), wh, (size_t)n_frames * 8,
                               hipMemcpyHostToDevice, st));
        if (total_kp) {
            jobs.push_back(CopyJob{c->lane[0].hin.p, kp, (size_t)total_kp * 8, 1});
            pool.run(jobs);
            jobs.clear();
            GMS_HIP(hipMemcpyAsync(c->tab_kp.p, c->lane[0].hin.p, (size_t)total_kp * 8, hipMemcpyHostToDevice, st));
            GMS_HIP(gms::launch_normalize(c->tab_kp.p, 8, (const int64_t*)c->tab_small.p,
                                          (const int32_t*)((char*)c->tab_small.p + align16((size_t)(n_frames + 1) * 8)),
                                          n_frames, total_kp, (float*)c->tab_pts.p, st));
        }
        GMS_HIP(hipStreamSynchronize(st));  // (frame_off / wh are the caller's pageable memory; lane 0's pinned block is reused below)
    }
    const int64_t* d_foff = (const int64_t*)c->tab_small.p;

    // ---- the pipeline, three chunks deep. Chunk k rides lane k % 3:
    //        iteration k     the pool copies chunk k into the lane's pinned block (and chunk k - 2's survivors out of its lane's:
    //                        one batch of jobs), then: upload, filter, pack the survivors back to back on the device
    //                        (compact_survivors_kernel), download the result records + the survivor total
    //        iteration k + 1 the records are on the host: download exactly the survivors (K records, not the m slots)
    //        iteration k + 2 the survivors are on the host: copy them to the caller's array at the pairs' offsets
    //      so that the copy engines (both directions) and the GPU work on chunks k - 1 and k while the host threads stage k + 1.
    auto in_matches = [&](Lane& L) { return (gms_dmatch*)((char*)L.din.p + in_pairs_bytes); };
    const size_t n_chunks = chunks.size();
    for (size_t k = 0; k < n_chunks + 2; ++k) {
        jobs.clear();
        if (k >= 2) {  // chunk k - 2: its survivors are (about to be) in the lane's pinned block
            const Chunk& ch = chunks[k - 2];
            Lane& L = c->lane[(k - 2) % 3];
            GMS_HIP(hipEventSynchronize(L.ev_out));
            const gms_pair_result* res = (const gms_pair_result*)L.hout.p;
            const gms_dmatch* o = (const gms_dmatch*)((const char*)L.hout.p + out_res_bytes);
            std::memcpy(results + ch.first, res, (size_t)ch.count * sizeof(gms_pair_result));
            size_t at = 0;
            for (int i = 0; i < ch.count; ++i) {
                if (res[i].status != GMS_OK || res[i].n_inliers <= 0) continue;
                jobs.push_back(CopyJob{out + pairs[ch.first + i].match_off, o + at, (size_t)res[i].n_inliers * sizeof(gms_dmatch), 0});
                at += (size_t)res[i].n_inliers;
            }
        }
        if (k < n_chunks) {  // chunk k: pair table rebased to the chunk, match records gathered
            const Chunk& ch = chunks[k];
            Lane& L = c->lane[k % 3];
            gms_pair* hp = (gms_pair*)L.hin.p;
            gms_dmatch* hm = (gms_dmatch*)((char*)L.hin.p + in_pairs_bytes);
            size_t local = 0;
            for (int i = 0; i < ch.count; ++i) {
                const gms_pair& pr = pairs[ch.first + i];
                hp[i] = gms_pair{pr.frame_a, pr.frame_b, pr.m, 0, (int64_t)local};
                if (pr.m) jobs.push_back(CopyJob{hm + local, matches + pr.match_off, (size_t)pr.m * sizeof(gms_dmatch), 0});
                local += (size_t)pr.m;
            }
        }
        pool.run(jobs);
        if (k < n_chunks) {
            const Chunk& ch = chunks[k];
            Lane& L = c->lane[k % 3];
            gms_pair_result* d_res = (gms_pair_result*)L.dout.p;
            int64_t* d_total = (int64_t*)((char*)L.dout.p + out_res_bytes - 16);
            gms_dmatch* d_out = (gms_dmatch*)((char*)L.dout.p + out_res_bytes);
            GMS_HIP(hipMemcpyAsync(L.din.p, L.hin.p, in_pairs_bytes + ch.matches * sizeof(gms_dmatch), hipMemcpyHostToDevice, L.stream));
            GMS_TRY(filter_launch(c, L.stream, (const float*)c->tab_pts.p, d_foff, n_frames, (const gms_pair*)L.din.p, ch.count, ch.max_m,
                                  in_matches(L), with_rotation, with_scale, threshold_factor, d_out, d_res, nullptr,
                                  false /* the ranges were validated on the host above; the device check's scratch word is per context, not per lane */));
            GMS_HIP(gms::launch_compact_survivors((const gms_pair*)L.din.p, d_res, ch.count, d_out, in_matches(L), d_total, L.stream));
            GMS_HIP(hipMemcpyAsync(L.hout.p, L.dout.p, out_res_bytes, hipMemcpyDeviceToHost, L.stream));
            GMS_HIP(hipEventRecord(L.ev_res, L.stream));
        }
        if (k >= 1 && k - 1 < n_chunks) {  // chunk k - 1: its records are (about to be) on the host; fetch its survivors
            const Chunk& ch = chunks[k - 1];
            Lane& L = c->lane[(k - 1) % 3];
            GMS_HIP(hipEventSynchronize(L.ev_res));
            const int64_t kept = *(const int64_t*)((const char*)L.hout.p + out_res_bytes - 16);
            if (kept < 0 || (size_t)kept > ch.matches) return GMS_ERR_HIP;  // (cannot happen: the kernels count what they wrote)
            if (kept)
                GMS_HIP(hipMemcpyAsync((char*)L.hout.p + out_res_bytes, in_matches(L), (size_t)kept * sizeof(gms_dmatch), hipMemcpyDeviceToHost, L.stream));
            GMS_HIP(hipEventRecord(L.ev_out, L.stream));
        }
    }
    return GMS_OK;
}

int64_t gms_bf_prepared_bytes(int desc_kind, int64_t total_desc, int n_frames)
{
    return (int64_t)gms::bf_prepared_bytes(desc_kind, total_desc, n_frames);
}

int gms_bf_prepare_device(gms_ctx* c, int desc_kind, const void* d_desc, const int64_t* d_frame_off, int n_frames,
                          int64_t total_desc, void* d_prepared)
{
    if (!c || n_frames < 0 || total_desc < 0) return GMS_ERR_BAD_ARG;
    if (desc_kind != GMS_DESC_HAMMING256 && desc_kind != GMS_DESC_L2_F32X128) return GMS_ERR_BAD_ARG;
    if (total_desc == 0 || n_frames == 0) return GMS_OK;
    if (!d_desc || !d_frame_off || !d_prepared) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_bf_prepare(desc_kind, d_desc, d_frame_off, n_frames, total_desc, d_prepared, c->stream));
    return GMS_OK;
}

int gms_bfmatch_device(gms_ctx* c, int desc_kind, const void* d_desc, const void* d_prepared, int64_t total_desc,
                       const int64_t* d_frame_off, int n_frames, const gms_pair* d_pairs, int n_pairs, int max_query,
                       gms_dmatch* d_matches)
{
    if (!c || n_frames < 0 || n_pairs < 0 || max_query < 0 || total_desc < 0) return GMS_ERR_BAD_ARG;
    if (desc_kind != GMS_DESC_HAMMING256 && desc_kind != GMS_DESC_L2_F32X128) return GMS_ERR_BAD_ARG;
    if (n_pairs == 0 || max_query == 0) return GMS_OK;
    if (!d_desc || !d_frame_off || !d_pairs || !d_matches) return GMS_ERR_BAD_ARG;
    if (desc_kind == GMS_DESC_L2_F32X128 && !d_prepared) return GMS_ERR_BAD_ARG;
    if (max_query > (1 << 22)) return GMS_ERR_CAPACITY;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_bf_match(desc_kind, d_desc, d_prepared, total_desc, d_frame_off, n_frames, d_pairs, n_pairs, max_query,
                                 d_matches, c->stream));
    return GMS_OK;
}

int gms_disparity_device(gms_ctx* c, const gms_keypoint* d_kp1, int n1, const gms_keypoint* d_kp2, int n2,
                         const gms_dmatch* d_matches, const int32_t* d_n_matches, int max_matches, int width, int height,
                         const uint8_t* d_gt, int disp_ratio, uint8_t* d_disparity, uint32_t* d_work, gms_disparity_stats* d_stats)
{
    if (!c || n1 < 0 || n2 < 0 || max_matches < 0 || width <= 0 || height <= 0) return GMS_ERR_BAD_ARG;
    if (!d_n_matches || !d_disparity || !d_work || !d_stats || (d_gt && disp_ratio == 0)) return GMS_ERR_BAD_ARG;
    if (max_matches > 0 && (!d_kp1 || !d_kp2 || !d_matches)) return GMS_ERR_BAD_ARG;
    if (max_matches >= (1 << 24)) return GMS_ERR_CAPACITY;  // the scatter key keeps the match index in 24 bits
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_disparity(d_kp1, n1, d_kp2, n2, d_matches, d_n_matches, max_matches, width, height, d_gt, disp_ratio,
                                  d_disparity, d_work, d_stats, c->stream));
    return GMS_OK;
}

int gms_gather_points_device(gms_ctx* c, const gms_keypoint* d_kp1, int n1, const gms_keypoint* d_kp2, int n2,
                             const gms_dmatch* d_matches, const int32_t* d_n_matches, int max_matches,
                             float* d_coords1, float* d_coords2, int32_t* d_status)
{
    if (!c || n1 < 0 || n2 < 0 || max_matches < 0 || !d_n_matches || !d_status) return GMS_ERR_BAD_ARG;
    if (max_matches > 0 && (!d_kp1 || !d_kp2 || !d_matches || !d_coords1 || !d_coords2)) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_gather_points(d_kp1, n1, d_kp2, n2, d_matches, d_n_matches, max_matches, d_coords1, d_coords2, d_status,
                                      c->stream));
    return GMS_OK;
}

int gms_triangulate_device(gms_ctx* c, const double camera[4], const double dist[5], const double P1[12], const double P2[12],
                           const float* d_coords1, const float* d_coords2, const int32_t* d_n_matches, int max_matches,
                           double* d_points3d, gms_triangulation_stats* d_stats)
{
    if (!c || !camera || !P1 || !P2 || !d_n_matches || !d_stats || max_matches < 0) return GMS_ERR_BAD_ARG;
    if (max_matches > 0 && (!d_coords1 || !d_coords2 || !d_points3d)) return GMS_ERR_BAD_ARG;
    if (camera[0] == 0.0 || camera[1] == 0.0) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_triangulate(camera, dist, P1, P2, d_coords1, d_coords2, d_n_matches, max_matches, d_points3d, d_stats, c->stream));
    return GMS_OK;
}

int gms_recover_pose_device(gms_ctx* c, const double E[9], const double camera[4], const float* d_coords1, const float* d_coords2,
                            const int32_t* d_n_matches, int max_matches, const uint8_t* d_in_mask, gms_pose* d_pose, uint8_t* d_out_mask)
{
    if (!c || !E || !camera || !d_n_matches || !d_pose || max_matches < 0) return GMS_ERR_BAD_ARG;
    if (max_matches > 0 && (!d_coords1 || !d_coords2)) return GMS_ERR_BAD_ARG;
    if (camera[0] == 0.0 || camera[1] == 0.0) return GMS_ERR_BAD_ARG;
    double R1[9], R2[9], t[3];
    if (!gms::tv::decompose_essential(E, R1, R2, t)) return GMS_ERR_BAD_ARG;  // cv::decomposeEssentialMat (twoview_core.h)
    double P[4][12];
    for (int h = 0; h < 4; ++h) {
        const double* R = (h & 1) ? R2 : R1;
        const double sg = h < 2 ? 1.0 : -1.0;
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < 3; ++k) P[h][4 * r + k] = R[3 * r + k];
            P[h][4 * r + 3] = sg * t[r];
        }
    }
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_TRY(grow_pose_ws(c, (size_t)max_matches + 48, c->stream));
    GMS_HIP(gms::launch_recover_pose(camera, P, 50.0, d_coords1, d_coords2, d_n_matches, max_matches, d_in_mask, d_pose, d_out_mask,
                                     c->pose_ws.p, c->stream));
    return GMS_OK;
}

// ---- the batched consumers (twoview_kernels.hip): nothing but launches on the context's stream --------------------------------
static bool camera_ok(const gms_camera* cam) { return cam && cam->fx != 0.0 && cam->fy != 0.0; }

int gms_gather_points_batch_device(gms_ctx* c, const gms_keypoint* d_kp, const int64_t* d_frame_off, int n_frames, const gms_pair* d_pairs,
                                   int n_pairs, int max_m, const gms_dmatch* d_filtered, const gms_pair_result* d_results, float* d_coords1,
                                   float* d_coords2, gms_two_view* d_tv)
{
    if (!c || n_frames < 0 || n_pairs < 0 || max_m < 0) return GMS_ERR_BAD_ARG;
    if (n_pairs == 0) return GMS_OK;
    if (!d_frame_off || !d_pairs || !d_results || !d_tv) return GMS_ERR_BAD_ARG;
    if (max_m > 0 && (!d_kp || !d_filtered || !d_coords1 || !d_coords2)) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_gather_batch(d_kp, d_frame_off, n_frames, d_pairs, n_pairs, max_m, d_filtered, d_results, d_coords1, d_coords2, d_tv,
                                     c->stream));
    return GMS_OK;
}

int gms_find_essential_batch_device(gms_ctx* c, const gms_camera* camera, double prob, double threshold, int max_iters, const gms_pair* d_pairs,
                                    int n_pairs, const float* d_coords1, const float* d_coords2, uint8_t* d_mask, gms_two_view* d_tv)
{
    if (!c || n_pairs < 0 || !camera_ok(camera)) return GMS_ERR_BAD_ARG;
    if (!(prob > 0.0 && prob < 1.0) || !(threshold > 0.0) || max_iters < 1) return GMS_ERR_BAD_ARG;  // (CV_Assert(confidence > 0 && confidence < 1))
    if (n_pairs == 0) return GMS_OK;
    if (!d_pairs || !d_coords1 || !d_coords2 || !d_mask || !d_tv) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_find_essential_batch(*camera, prob, threshold, max_iters, d_pairs, n_pairs, d_coords1, d_coords2, d_mask, d_tv, c->stream));
    return GMS_OK;
}

int gms_recover_pose_batch_device(gms_ctx* c, const gms_camera* camera, int use_in_mask, const gms_pair* d_pairs, int n_pairs,
                                  const float* d_coords1, const float* d_coords2, uint8_t* d_mask, gms_two_view* d_tv)
{
    if (!c || n_pairs < 0 || !camera_ok(camera)) return GMS_ERR_BAD_ARG;
    if (n_pairs == 0) return GMS_OK;
    if (!d_pairs || !d_coords1 || !d_coords2 || !d_mask || !d_tv) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_recover_pose_batch(*camera, 50.0, use_in_mask ? 1 : 0, d_pairs, n_pairs, d_coords1, d_coords2, d_mask, d_tv, c->stream));
    return GMS_OK;
}

int gms_triangulate_batch_device(gms_ctx* c, const gms_camera* camera, const gms_pair* d_pairs, int n_pairs, const float* d_coords1,
                                 const float* d_coords2, const uint8_t* d_mask, double* d_points3d, gms_two_view* d_tv)
{
    if (!c || n_pairs < 0 || !camera_ok(camera)) return GMS_ERR_BAD_ARG;
    if (n_pairs == 0) return GMS_OK;
    if (!d_pairs || !d_coords1 || !d_coords2 || !d_points3d || !d_tv) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_triangulate_batch(*camera, d_pairs, n_pairs, d_coords1, d_coords2, d_mask, d_points3d, d_tv, c->stream));
    return GMS_OK;
}

int gms_two_view_batch_device(gms_ctx* c, const gms_camera* camera, double prob, double threshold, int max_iters, const gms_keypoint* d_kp,
                              const int64_t* d_frame_off, int n_frames, const gms_pair* d_pairs, int n_pairs, int max_m,
                              const gms_dmatch* d_filtered, const gms_pair_result* d_results, float* d_coords1, float* d_coords2,
                              uint8_t* d_mask, double* d_points3d, gms_two_view* d_tv)
{
    GMS_TRY(gms_gather_points_batch_device(c, d_kp, d_frame_off, n_frames, d_pairs, n_pairs, max_m, d_filtered, d_results, d_coords1, d_coords2, d_tv));
    GMS_TRY(gms_find_essential_batch_device(c, camera, prob, threshold, max_iters, d_pairs, n_pairs, d_coords1, d_coords2, d_mask, d_tv));
    GMS_TRY(gms_recover_pose_batch_device(c, camera, 1, d_pairs, n_pairs, d_coords1, d_coords2, d_mask, d_tv));
    return gms_triangulate_batch_device(c, camera, d_pairs, n_pairs, d_coords1, d_coords2, d_mask, d_points3d, d_tv);
}

int gms_disparity_batch_device(gms_ctx* c, const gms_keypoint* d_kp, const int64_t* d_frame_off, const int32_t* d_wh, int n_frames,
                               const gms_pair* d_pairs, int n_pairs, int max_m, const gms_dmatch* d_filtered, const gms_pair_result* d_results,
                               const uint8_t* d_gt, int64_t gt_stride, int disp_ratio, uint8_t* d_disparity, int64_t map_stride, uint32_t* d_work,
                               gms_disparity_stats* d_stats)
{
    if (!c || n_frames < 0 || n_pairs < 0 || max_m < 0 || map_stride <= 0 || gt_stride < 0) return GMS_ERR_BAD_ARG;
    if (n_pairs == 0) return GMS_OK;
    if (!d_frame_off || !d_wh || !d_pairs || !d_results || !d_disparity || !d_work || !d_stats || (d_gt && disp_ratio == 0)) return GMS_ERR_BAD_ARG;
    if (max_m > 0 && (!d_kp || !d_filtered)) return GMS_ERR_BAD_ARG;
    if (max_m >= (1 << 24)) return GMS_ERR_CAPACITY;  // the scatter key keeps the match index in 24 bits
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_disparity_batch(d_kp, d_frame_off, d_wh, n_frames, d_pairs, n_pairs, max_m, d_filtered, d_results, d_gt, gt_stride, disp_ratio,
                                        d_disparity, map_stride, d_work, d_stats, c->stream));
    return GMS_OK;
}

size_t gms_detect_workspace_bytes(int width, int height, int n_images, int max_keypoints)
{
    return gms::detect_workspace_bytes(width, height, n_images, max_keypoints);
}

static bool detect_image_ok(int width, int height)
{
    return width > 2 * GMS_DETECT_BORDER && height > 2 * GMS_DETECT_BORDER && width <= 65535 && height <= 65535;
}

int gms_detect_batch_device(gms_ctx* c, const uint8_t* d_images, int n_images, int width, int height, int threshold, int max_keypoints,
                            void* d_workspace, size_t workspace_bytes, gms_keypoint* d_keypoints, uint8_t* d_descriptors, int32_t* d_counts)
{
    if (!c || n_images < 0 || max_keypoints < 0 || threshold < 0 || threshold > 254 || !detect_image_ok(width, height)) return GMS_ERR_BAD_ARG;
    if (n_images == 0) return GMS_OK;
    if (!d_images || !d_workspace || !d_counts || (max_keypoints > 0 && (!d_keypoints || !d_descriptors))) return GMS_ERR_BAD_ARG;
    if (workspace_bytes < gms::detect_workspace_bytes(width, height, n_images, max_keypoints)) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_detect(d_images, n_images, width, height, threshold, max_keypoints, d_workspace, d_keypoints, d_descriptors, d_counts,
                               c->stream));
    return GMS_OK;
}

int gms_describe_device(gms_ctx* c, const uint8_t* d_image, int width, int height, gms_keypoint* d_keypoints, int n,
                        void* d_workspace, size_t workspace_bytes, uint8_t* d_descriptors, int32_t* d_status)
{
    if (!c || n < 0 || !detect_image_ok(width, height) || !d_image || !d_workspace || !d_status) return GMS_ERR_BAD_ARG;
    if (n > 0 && (!d_keypoints || !d_descriptors)) return GMS_ERR_BAD_ARG;
    if (workspace_bytes < gms::detect_workspace_bytes(width, height, 1, 0)) return GMS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_describe(d_image, width, height, d_keypoints, n, d_workspace, d_descriptors, d_status, c->stream));
    return GMS_OK;
}

int gms_selftest_five_point(gms_ctx* c, const double* pts, int n_samples, double* models, int32_t* counts)
{
    if (!c || n_samples < 0 || (n_samples > 0 && (!pts || !models || !counts))) return GMS_ERR_BAD_ARG;
    if (n_samples == 0) return GMS_OK;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    const size_t b_pts = (size_t)n_samples * 20 * 8, b_models = (size_t)n_samples * 90 * 8, b_counts = (size_t)n_samples * 4;
    GMS_HIP(hipStreamSynchronize(c->stream));
    GMS_HIP(c->aux.reserve(b_pts + b_models + b_counts));
    double* d_pts = (double*)c->aux.p;
    double* d_models = d_pts + (size_t)n_samples * 20;
    int* d_counts = (int*)(d_models + (size_t)n_samples * 90);
    GMS_HIP(hipMemcpyAsync(d_pts, pts, b_pts, hipMemcpyHostToDevice, c->stream));
    GMS_HIP(gms::launch_five_point_selftest(d_pts, n_samples, d_models, d_counts, c->stream));
    GMS_HIP(hipMemcpyAsync(models, d_models, b_models, hipMemcpyDeviceToHost, c->stream));
    GMS_HIP(hipMemcpyAsync(counts, d_counts, b_counts, hipMemcpyDeviceToHost, c->stream));
    GMS_HIP(hipStreamSynchronize(c->stream));
    return GMS_OK;
}

int gms_selftest_threshold(gms_ctx* c, const int32_t* T, const int32_t* n, const int32_t* score, double factor,
                           int count, uint8_t* out)
{
    if (!c || count < 0 || (count > 0 && (!T || !n || !score || !out))) return GMS_ERR_BAD_ARG;
    if (count == 0) return GMS_OK;
    std::lock_guard<std::mutex> lock(c->mu);
    GMS_HIP(hipSetDevice(c->device));
    const size_t nb = (size_t)count * 4;
    GMS_HIP(c->aux.reserve(3 * nb + (size_t)count));
    int32_t* dT = (int32_t*)c->aux.p;
    int32_t* dn = dT + count;
    int32_t* ds = dn + count;
    uint8_t* dout = (uint8_t*)(ds + count);
    GMS_HIP(hipMemcpyAsync(dT, T, nb, hipMemcpyHostToDevice, c->stream));
    GMS_HIP(hipMemcpyAsync(dn, n, nb, hipMemcpyHostToDevice, c->stream));
    GMS_HIP(hipMemcpyAsync(ds, score, nb, hipMemcpyHostToDevice, c->stream));
    GMS_HIP(gms::launch_threshold(dT, dn, ds, factor, count, dout, c->stream));
    GMS_HIP(hipMemcpyAsync(out, dout, (size_t)count, hipMemcpyDeviceToHost, c->stream));
    GMS_HIP(hipStreamSynchronize(c->stream));
    return GMS_OK;
}

}  // extern "C"
