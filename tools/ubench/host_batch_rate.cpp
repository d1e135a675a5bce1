// Diagnostic (not product): gms_filter_host_batch called straight through the C ABI from C++ (what mi355::matchGMSBatch does),
// 2048 pairs x 10k matches in pageable host memory, with the caller's output array touched beforehand and -- second figure --
// freshly allocated per call (first-touch page faults of 327 MB land inside the call then).
// Build + run on the GPU box:
//   hipcc -O2 -pthread -Iinclude tools/ubench/host_batch_rate.cpp -Lsfm-gms_amd/csrc -lgms_hip -Wl,-rpath,$PWD/sfm-gms_amd/csrc -o /tmp/hbr && /tmp/hbr
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gms.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

int main(int argc, char** argv)
{
    const int n_frames = 72, n_kp = 10000, n_pairs = argc > 1 ? atoi(argv[1]) : 2048, w = 1920, h = 1080;
    std::vector<gms_keypoint> kp((size_t)n_frames * n_kp);
    std::vector<int64_t> foff(n_frames + 1);
    std::vector<int32_t> wh(2 * n_frames);
    for (int f = 0; f <= n_frames; ++f) foff[f] = (int64_t)f * n_kp;
    for (int f = 0; f < n_frames; ++f) {
        wh[2 * f] = w;
        wh[2 * f + 1] = h;
        for (int i = 0; i < n_kp; ++i) {   // one scene under a slow drift: keypoint i of every frame is the same scene point
            const uint64_t r = mix((uint64_t)i);
            gms_keypoint k{};
            k.x = 0.15f * w + 0.7f * w * (float)(r & 0xFFFFFF) / 16777216.0f + 0.5f * f;
            k.y = 0.15f * h + 0.7f * h * (float)((r >> 24) & 0xFFFFFF) / 16777216.0f + 0.3f * f;
            k.size = 31;
            k.angle = -1;
            k.class_id = -1;
            kp[(size_t)f * n_kp + i] = k;
        }
    }
    std::vector<gms_pair> pairs(n_pairs);
    std::vector<gms_dmatch> matches((size_t)n_pairs * n_kp);
    int a = 0, b = 1;
    for (int p = 0; p < n_pairs; ++p) {
        pairs[p] = gms_pair{a, b, n_kp, 0, (int64_t)p * n_kp};
        for (int i = 0; i < n_kp; ++i) {
            const uint64_t r = mix(((uint64_t)p << 20) ^ (uint64_t)i);
            matches[(size_t)p * n_kp + i] = gms_dmatch{i, (r & 1) ? i : (int)((r >> 8) % n_kp), 0, (float)(r >> 40)};
        }
        if (++b == n_frames) { ++a; b = a + 1; }
    }
    gms_ctx* ctx = nullptr;
    if (gms_ctx_create(0, &ctx) != GMS_OK) return 1;
    std::vector<gms_pair_result> res(n_pairs);
    const size_t total = matches.size();
    gms_dmatch* out = (gms_dmatch*)malloc(total * sizeof(gms_dmatch));
    memset(out, 0, total * sizeof(gms_dmatch));
    printf("{");
    for (int flags = 0; flags < 2; ++flags) {
        std::vector<double> t;
        for (int rep = 0; rep < 5; ++rep) {
            const double t0 = now();
            const int rc = gms_filter_host_batch(ctx, kp.data(), foff.data(), wh.data(), n_frames, pairs.data(), n_pairs, matches.data(), flags, flags, 6.0, out, res.data());
            t.push_back(now() - t0);
            if (rc != GMS_OK) return 2;
        }
        std::sort(t.begin() + 1, t.end());
        const double med = t[1 + (t.size() - 1) / 2];
        long long kept = 0;
        for (int p = 0; p < n_pairs; ++p) kept += res[p].n_inliers;
        // the same with an output array nobody has touched yet (a caller's fresh std::vector would be touched by its constructor)
        gms_dmatch* fresh = (gms_dmatch*)malloc(total * sizeof(gms_dmatch));
        const double t0 = now();
        gms_filter_host_batch(ctx, kp.data(), foff.data(), wh.data(), n_frames, pairs.data(), n_pairs, matches.data(), flags, flags, 6.0, fresh, res.data());
        const double t_fresh = now() - t0;
        free(fresh);
        printf("%s\"host_batch_%dx10k_rot%d_scale%d\": {\"pairs_per_s\": %.0f, \"ms\": %.2f, \"GB_per_s_in\": %.2f, \"GB_per_s_out\": %.2f, \"first_call_ms\": %.2f, "
               "\"ms_with_untouched_output_array\": %.2f, \"mean_kept\": %.1f}",
               flags ? ", " : "", n_pairs, flags, flags, n_pairs / med, med * 1e3, total * 16.0 / med / 1e9, kept * 16.0 / med / 1e9, t[0] * 1e3, t_fresh * 1e3,
               (double)kept / n_pairs);
    }
    printf("}\n");
    free(out);
    gms_ctx_destroy(ctx);
    return 0;
}
