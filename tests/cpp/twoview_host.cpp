// tests/cpp/twoview_host.cpp -- TEST BUILD of sfm-gms_amd/csrc/twoview_core.h for the CPU (g++): the per-lane arithmetic of the
// two-view kernels (five-point solver, cv::RNG, RANSACUpdateNumIters, the error, decomposeEssentialMat) behind a few extern "C"
// entry points, so that tests/test_twoview_core.py can check it against the numpy restatement (oracle/sfm_ref.py) without a GPU.
// Not part of the product library: libgms_hip.so runs this arithmetic in twoview_kernels.hip only. The RANSAC loop below mirrors
// find_essential_kernel's round structure (sixteen samples per round, models replayed in order) with the same core calls.
#include <cstdint>
#include <cstring>
#include <vector>

#include "twoview_core.h"

using namespace gms::tv;

namespace {
struct Plain {
    double* base;
    double& operator()(int i) const { return base[i]; }
};
}  // namespace

extern "C" {

int tvh_five_point(const double* x1, const double* y1, const double* x2, const double* y2, double* models /* [90] */)
{
    double A[200], basis[36], work[60];
    FivePointMem<Plain> mem{Plain{A}, Plain{basis}, Plain{work}};
    Plain out{models};
    return five_point(x1, y1, x2, y2, mem, out);
}

void tvh_rng(uint64_t seed, int count, int n_samples, int* out /* [n_samples][5] */)
{
    CvRng r;
    r.seed(seed);
    for (int i = 0; i < n_samples; ++i) r.sample5(count, out + 5 * i);
}

int tvh_update_iters(double p, double ep, int model_points, int max_iters) { return ransac_update_num_iters(p, ep, model_points, max_iters); }

int tvh_decompose(const double* E, double* R1, double* R2, double* t) { return decompose_essential(E, R1, R2, t) ? 1 : 0; }

void tvh_errors(const double* E, const double* x1, const double* x2, int n, float* err)
{
    for (int i = 0; i < n; ++i) err[i] = sampson_error(E, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1]);
}

// find_essential_kernel's control flow on the host: uv in pixels (float), camera = fx, fy, cx, cy. Returns the inlier count.
int tvh_find_essential(const float* uv1, const float* uv2, int n, const double* camera, double prob, double threshold, int max_iters,
                       double* E_out, uint8_t* mask, int* iters_out)
{
    const double fx = camera[0], fy = camera[1], cx = camera[2], cy = camera[3];
    const double thr = threshold / ((fx + fy) / 2.0);
    const float t = (float)(thr * thr);
    std::vector<double> X1(2 * (size_t)n), X2(2 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        X1[2 * i] = ((double)uv1[2 * i] - cx) / fx;
        X1[2 * i + 1] = ((double)uv1[2 * i + 1] - cy) / fy;
        X2[2 * i] = ((double)uv2[2 * i] - cx) / fx;
        X2[2 * i + 1] = ((double)uv2[2 * i + 1] - cy) / fy;
    }
    std::memset(mask, 0, (size_t)n);
    for (int k = 0; k < 9; ++k) E_out[k] = 0.0;
    *iters_out = 0;
    if (n < 5) return 0;
    double best[9] = {0};
    int best_count = 0, niters = max_iters > 1 ? max_iters : 1, it_done = 0;
    auto solve = [&](const int* idx, double* models) {
        double x1[5], y1[5], x2[5], y2[5];
        for (int j = 0; j < 5; ++j) {
            x1[j] = X1[2 * idx[j]];
            y1[j] = X1[2 * idx[j] + 1];
            x2[j] = X2[2 * idx[j]];
            y2[j] = X2[2 * idx[j] + 1];
        }
        return tvh_five_point(x1, y1, x2, y2, models);
    };
    if (n == 5) {
        const int idx[5] = {0, 1, 2, 3, 4};
        double models[90];
        if (solve(idx, models) > 0) {
            std::memcpy(best, models, sizeof best);
            best_count = 5;
        }
    } else {
        CvRng rng;
        rng.seed(0xFFFFFFFFFFFFFFFFull);
        const int R = 16;
        for (int it0 = 0;; it0 += R) {
            int samples[R * 5], nm[R];
            double models[R * 90];
            unsigned counts[R * 10] = {0};
            for (int k = 0; k < R; ++k) rng.sample5(n, samples + 5 * k);
            const int live = niters - it0 < R ? niters - it0 : R;
            for (int k = 0; k < R; ++k) nm[k] = k < live ? solve(samples + 5 * k, models + 90 * k) : 0;
            for (int k = 0; k < live; ++k)
                for (int j = 0; j < nm[k]; ++j)
                    for (int i = 0; i < n; ++i)
                        counts[k * 10 + j] += sampson_error(models + (k * 10 + j) * 9, X1[2 * i], X1[2 * i + 1], X2[2 * i], X2[2 * i + 1]) <= t;
            int it = it0;
            for (int k = 0; k < R && it < niters; ++k, ++it)
                for (int j = 0; j < nm[k]; ++j) {
                    const int good = (int)counts[k * 10 + j];
                    if (good > (best_count > 4 ? best_count : 4)) {
                        best_count = good;
                        std::memcpy(best, models + (k * 10 + j) * 9, sizeof best);
                        niters = ransac_update_num_iters(prob, (double)(n - good) / n, 5, niters);
                    }
                }
            it_done = it;
            if (it_done >= niters) break;
        }
    }
    *iters_out = it_done;
    if (best_count > 0) {
        for (int i = 0; i < n; ++i)
            mask[i] = n == 5 ? 1 : (sampson_error(best, X1[2 * i], X1[2 * i + 1], X2[2 * i], X2[2 * i + 1]) <= t ? 1 : 0);
        canonical_sign(best);
        std::memcpy(E_out, best, sizeof best);
    }
    return best_count;
}

}  // extern "C"
