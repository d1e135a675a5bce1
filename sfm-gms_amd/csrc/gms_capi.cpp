// gms_capi.cpp -- the C ABI declared in include/gms.h, on top of the HIP kernels.
//
// Host side of the drop-in for cv::xfeatures2d::matchGMS (FeatureMatchUtil.cpp:69,
// DisparityUtil.cpp:149,299 of the reference). There is deliberately no CPU implementation of the
// filter in this library: without a working HIP device every compute entry point returns an error.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "gms.h"
#include "gms_kernels.h"

static_assert(sizeof(gms_keypoint) == 28, "gms_keypoint must match cv::KeyPoint (stride 0x1c)");
static_assert(sizeof(gms_dmatch) == 16, "gms_dmatch must match cv::DMatch (stride 0x10)");
static_assert(sizeof(gms_pair) == 24, "gms_pair layout");
static_assert(sizeof(gms_pair_result) == 16, "gms_pair_result layout");

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

namespace {

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

}  // namespace

struct gms_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::mutex mu;  // serialises the one-shot path's scratch buffers
    DevBuf kp, foff, wh, pts, pair, matches, out, result, aux, big_ws, band_ws, partial_ws;
    int n_cus = 256;  // multiProcessorCount of the device
};

extern "C" {

const char* gms_version(void) { return "mi355-gms 0.1 (gfx950)"; }

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
    if (e != hipSuccess) {
        t_last_hip = (int)e;
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
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->kp, &c->foff, &c->wh, &c->pts, &c->pair, &c->matches, &c->out, &c->result, &c->aux, &c->big_ws, &c->band_ws, &c->partial_ws};
    for (DevBuf* b : bufs) b->release();
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return GMS_OK;
}

int gms_ctx_set_stream(gms_ctx* c, void* hip_stream)
{
    if (!c) return GMS_ERR_BAD_ARG;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return GMS_OK;
}

int gms_ctx_synchronize(gms_ctx* c)
{
    if (!c) return GMS_ERR_BAD_ARG;
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(hipStreamSynchronize(c->stream));
    return GMS_OK;
}

int gms_normalize_device(gms_ctx* c, const gms_keypoint* d_kp, const int64_t* d_frame_off,
                         const int32_t* d_wh, int n_frames, int64_t total_kp, float* d_pts)
{
    if (!c || n_frames < 0 || total_kp < 0) return GMS_ERR_BAD_ARG;
    if (total_kp == 0 || n_frames == 0) return GMS_OK;
    if (!d_kp || !d_frame_off || !d_wh || !d_pts) return GMS_ERR_BAD_ARG;
    GMS_HIP(hipSetDevice(c->device));
    GMS_HIP(gms::launch_normalize(d_kp, d_frame_off, d_wh, n_frames, total_kp, d_pts, c->stream));
    return GMS_OK;
}

int gms_filter_device(gms_ctx* c, const float* d_pts, const int64_t* d_frame_off, int n_frames,
                      const gms_pair* d_pairs, int n_pairs, int max_m, const gms_dmatch* d_matches,
                      int with_rotation, int with_scale, double threshold_factor,
                      gms_dmatch* d_out, gms_pair_result* d_results, uint8_t* d_mask)
{
    if (!c || n_pairs < 0 || max_m < 0 || n_frames < 0) return GMS_ERR_BAD_ARG;
    if (n_pairs == 0) return GMS_OK;
    if (!d_frame_off || !d_pairs || !d_results) return GMS_ERR_BAD_ARG;
    if (max_m > 0 && (!d_matches || !d_out || !d_pts)) return GMS_ERR_BAD_ARG;
    gms::FilterParams p;
    const int kpt = gms::filter_pick_kpt(max_m);  // 0: too large for the register + LDS kernel
    if (!kpt && max_m > gms::kBigMaxMatches) return GMS_ERR_CAPACITY;
    p.table_slots = kpt ? gms::filter_table_slots(kpt) : 0;
    p.region_shift = kpt ? gms::filter_region_shift(kpt) : 0;
    if (const char* e = std::getenv("GMS_REGION_SHIFT")) p.region_shift = std::max(p.region_shift, std::atoi(e));  // tuning knob
    p.pts = reinterpret_cast<const float2*>(d_pts);
    p.frame_off = d_frame_off;
    p.n_frames = n_frames;
    p.pairs = d_pairs;
    p.n_pairs = n_pairs;
    // experimental, off by default (measured slower at 10k matches/pair): GMS_PREFETCH_STRIDE=256 touches the
    // match array of the pair one dispatch round ahead
    static const int prefetch_stride = [] {
        const char* e = std::getenv("GMS_PREFETCH_STRIDE");
        return e ? std::atoi(e) : 0;
    }();
    p.prefetch_stride = prefetch_stride;
    // GMS_STAGGER_US: spread of the first dispatch round's start times (0 turns it off); by default about one
    // pair's duration at the benchmark shape on the path the launch will mostly take
    static const int stagger_us = [] {
        const char* e = std::getenv("GMS_STAGGER_US");
        return e ? std::atoi(e) : -1;
    }();
    p.stagger_blocks = 256;
    static const int stagger_mode = [] {
        const char* e = std::getenv("GMS_STAGGER_MODE");
        return e ? std::atoi(e) : 0;
    }();
    p.stagger_mode = stagger_mode;
    p.matches = d_matches;
    p.out = d_out;
    p.results = d_results;
    p.mask = d_mask;
    p.with_rotation = with_rotation ? 1 : 0;
    p.with_scale = with_scale ? 1 : 0;
    p.threshold_factor = threshold_factor;
    p.pair_flags = nullptr;
    p.partial = nullptr;
    right_grids(p.right_w, p.right_h);
    // GMS_DENSE=0 keeps every pair on the hashed path (diagnostics); by default the byte-matrix path is tried first
    // whenever there are no scale hypotheses (the reference's default flags, DisparityUtil.cpp:149,299)
    static const bool dense_on = [] {
        const char* e = std::getenv("GMS_DENSE");
        return !e || std::atoi(e) != 0;
    }();
    p.dense = (dense_on && !with_scale) ? 1 : 0;
    p.stagger_cycles = (n_pairs >= 4 * p.stagger_blocks) ? (stagger_us >= 0 ? stagger_us : (p.dense ? 26 : 72)) * 2400 : 0;
#ifdef GMS_PHASE_TIMING
    p.diag = g_diag;
#endif
    GMS_HIP(hipSetDevice(c->device));
    static const bool use_occ2 = [] {  // GMS_OCC2=1: the two-workgroups-per-CU kernel for pairs it can hold
        const char* e = std::getenv("GMS_OCC2");
        return e && std::atoi(e) != 0;
    }();
    const int kpt2 = use_occ2 ? gms::occ2_pick_kpt(max_m) : 0;
    if (kpt2) {
        p.table_slots = gms::occ2_table_slots(kpt2);
        GMS_HIP(gms::launch_filter_occ2(p, kpt2, n_pairs, c->stream));
    } else if (kpt && with_scale && dense_on) {
        // scale hypotheses: scales 0..2 on the byte matrix, the rest on the hashed path (two launches, one record per pair)
        const size_t need = (size_t)n_pairs * gms::kPartialStrideDw * 4;
        if (need > c->partial_ws.cap) {
            GMS_HIP(hipStreamSynchronize(c->stream));
            GMS_HIP(c->partial_ws.reserve(need));
        }
        p.partial = (uint32_t*)c->partial_ws.p;
        GMS_HIP(gms::launch_filter_scales(p, kpt, n_pairs, c->stream));
    } else if (kpt) {
        GMS_HIP(gms::launch_filter(p, kpt, n_pairs, c->stream));
    } else {
        // Large pairs: a fixed crew of persistent workgroups, each with an HBM slab for the pair's code words
        // and table. The slab is (re)allocated here when it has to grow -- this branch is not stream-capturable.
        const int mcap = gms::big_mcap(max_m);
        const int n_wg = n_pairs < c->n_cus ? n_pairs : c->n_cus;  // one persistent workgroup per CU at most
        const size_t need = (size_t)n_wg * gms::big_ws_stride_dwords(mcap) * 4;
        if (need > c->big_ws.cap) {
            GMS_HIP(hipStreamSynchronize(c->stream));
            GMS_HIP(c->big_ws.reserve(need));
        }
        p.stagger_cycles = 0;
        // GMS_BAND=0 keeps large pairs on the HBM-slab kernel alone (diagnostics)
        static const bool band_on = [] {
            const char* e = std::getenv("GMS_BAND");
            return !e || std::atoi(e) != 0;
        }();
        if (band_on) {
            // The LDS kernels for large pairs, a slice of the batch at a time so that the per-pair workspace (lists, histogram,
            // masks) stays bounded: three bands of rows for the default flags, tiles of left cells with three launches per
            // scale hypothesis otherwise. Pairs they flag (a cell above 65 535 matches) fall through to the HBM-slab kernel,
            // which looks at flagged pairs only.
            const bool plain = !with_rotation && !with_scale;
            const size_t per_pair = plain ? gms::band_ws_bytes_per_pair(mcap, d_mask == nullptr)
                                          : gms::tile_ws_bytes_per_pair(p, mcap, d_mask == nullptr);
            const size_t budget = (size_t)4 << 30;
            size_t slice = budget / per_pair;
            if (slice < 1) slice = 1;
            if (slice > (size_t)n_pairs) slice = (size_t)n_pairs;
            if (slice * per_pair > c->band_ws.cap) {
                GMS_HIP(hipStreamSynchronize(c->stream));
                GMS_HIP(c->band_ws.reserve(slice * per_pair));
            }
            for (int s0 = 0; s0 < n_pairs; s0 += (int)slice) {
                gms::FilterParams ps = p;
                ps.pairs = d_pairs + s0;
                ps.results = d_results + s0;
                ps.n_pairs = (n_pairs - s0 < (int)slice) ? n_pairs - s0 : (int)slice;
                const uint32_t* flags = nullptr;
                if (plain) GMS_HIP(gms::launch_filter_band(ps, mcap, c->band_ws.p, &flags, c->stream));
                else GMS_HIP(gms::launch_filter_tiles(ps, mcap, c->band_ws.p, &flags, c->stream));
                ps.pair_flags = flags;
                const int wg = ps.n_pairs < c->n_cus ? ps.n_pairs : c->n_cus;
                GMS_HIP(gms::launch_filter_big(ps, mcap, wg, (uint32_t*)c->big_ws.p, c->stream));
            }
        } else {
            GMS_HIP(gms::launch_filter_big(p, mcap, n_wg, (uint32_t*)c->big_ws.p, c->stream));
        }
    }
    return GMS_OK;
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
    // the three small tables of the call travel as one 64-byte block: frame_off[3] | wh[4] | the pair
    struct alignas(8) CallHeader {
        int64_t foff[3];
        int32_t wh[4];
        gms_pair pair;
    };
    static_assert(sizeof(CallHeader) == 64 && offsetof(CallHeader, wh) == 24 && offsetof(CallHeader, pair) == 40, "header layout");
    const CallHeader hdr = {{0, n1, (int64_t)n1 + n2}, {w1, h1, w2, h2}, {0, 1, m, 0, 0}};
    const size_t nkp = (size_t)n1 + (size_t)n2;
    GMS_HIP(c->kp.reserve(nkp * sizeof(gms_keypoint)));
    GMS_HIP(c->pts.reserve(nkp * 2 * sizeof(float)));
    GMS_HIP(c->foff.reserve(sizeof hdr));
    GMS_HIP(c->matches.reserve((size_t)m * sizeof(gms_dmatch)));
    GMS_HIP(c->out.reserve((size_t)m * sizeof(gms_dmatch)));
    GMS_HIP(c->result.reserve(sizeof(gms_pair_result)));
    if (n1) GMS_HIP(hipMemcpyAsync(c->kp.p, kp1, (size_t)n1 * sizeof(gms_keypoint), hipMemcpyHostToDevice, st));
    if (n2)
        GMS_HIP(hipMemcpyAsync((gms_keypoint*)c->kp.p + n1, kp2, (size_t)n2 * sizeof(gms_keypoint),
                               hipMemcpyHostToDevice, st));
    GMS_HIP(hipMemcpyAsync(c->foff.p, &hdr, sizeof hdr, hipMemcpyHostToDevice, st));
    if (m) GMS_HIP(hipMemcpyAsync(c->matches.p, matches, (size_t)m * sizeof(gms_dmatch), hipMemcpyHostToDevice, st));
    const int64_t* d_foff = (const int64_t*)c->foff.p;
    const int32_t* d_wh = (const int32_t*)((const char*)c->foff.p + offsetof(CallHeader, wh));
    const gms_pair* d_pair = (const gms_pair*)((const char*)c->foff.p + offsetof(CallHeader, pair));
    // the staging copies above read caller/stack memory: finish them before anything can go out of scope
    GMS_HIP(hipStreamSynchronize(st));

    int rc = gms_normalize_device(c, (const gms_keypoint*)c->kp.p, d_foff, d_wh, 2, (int64_t)nkp, (float*)c->pts.p);
    if (rc != GMS_OK) return rc;
    rc = gms_filter_device(c, (const float*)c->pts.p, d_foff, 2, d_pair, 1, m, (const gms_dmatch*)c->matches.p, with_rotation,
                           with_scale, threshold_factor, (gms_dmatch*)c->out.p, (gms_pair_result*)c->result.p, nullptr);
    if (rc != GMS_OK) return rc;
    gms_pair_result r;
    GMS_HIP(hipMemcpyAsync(&r, c->result.p, sizeof r, hipMemcpyDeviceToHost, st));
    // Small calls get the output in the same round trip as the result: all m slots are copied (the caller's buffer has room
    // for m by contract; what lies beyond *n_out is unspecified) instead of waiting for the count first. Measured: pays up to
    // about 2k matches (32 KB); at 10k the extra bytes cost more than the saved synchronisation.
    const bool eager = m > 0 && m <= 2048;
    if (eager) GMS_HIP(hipMemcpyAsync(out, c->out.p, (size_t)m * sizeof(gms_dmatch), hipMemcpyDeviceToHost, st));
    GMS_HIP(hipStreamSynchronize(st));
    if (result) *result = r;
    if (r.status != GMS_OK) return r.status;
    if (!eager && r.n_inliers > 0) {
        GMS_HIP(hipMemcpyAsync(out, c->out.p, (size_t)r.n_inliers * sizeof(gms_dmatch), hipMemcpyDeviceToHost, st));
        GMS_HIP(hipStreamSynchronize(st));
    }
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
