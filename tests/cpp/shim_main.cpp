// Exercises the C++ shim the way the reference's callers use matchGMS (FeatureMatchUtil.cpp:66-69:
// M = N1 matches with queryIdx = i, then matchGMS(size1, size2, kp1, kp2, matches, out, true, true)).
// Deterministic LCG data; prints "n_out checksum" for the two flag combinations the reference uses.
#include <cstdint>
#include <cstdio>
#include <exception>
#include <utility>

#include "mi355_gms.hpp"

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

int main()
{
    const int w = 1280, h = 720, n = 4000;
    std::vector<mi355::KeyPoint> kp1(n), kp2(n);
    std::vector<mi355::DMatch> matches(n), out;
    uint32_t s = 12345u;
    for (int i = 0; i < n; ++i) {
        kp1[i].pt.x = (float)(lcg(s) % ((w - 1) * 16)) / 16.0f;
        kp1[i].pt.y = (float)(lcg(s) % ((h - 1) * 16)) / 16.0f;
        kp2[i].pt.x = kp1[i].pt.x * 0.98f + 7.25f;   // a mild zoom + shift keeps true matches consistent
        kp2[i].pt.y = kp1[i].pt.y * 0.98f + 3.5f;
        matches[i].queryIdx = i;
        matches[i].trainIdx = (lcg(s) % 100 < 55) ? i : (int)(lcg(s) % n);
        matches[i].imgIdx = i % 3;
        matches[i].distance = (float)(lcg(s) % 1024) / 4.0f;
    }
    out.resize(17);  // the reference's callers pass a non-empty vector again (main.cpp:39): it must be cleared
    try {
        for (int flags = 0; flags < 2; ++flags) {
            mi355::matchGMS(mi355::Size(w, h), mi355::Size(w, h), kp1, kp2, matches, out, flags != 0, flags != 0);
            uint64_t sum = 1469598103934665603ull;
            for (const auto& m : out) {
                uint32_t bits;
                __builtin_memcpy(&bits, &m.distance, 4);
                for (uint32_t v : {(uint32_t)m.queryIdx, (uint32_t)m.trainIdx, (uint32_t)m.imgIdx, bits})
                    sum = (sum ^ v) * 1099511628211ull;
            }
            std::printf("%zu %llu\n", out.size(), (unsigned long long)sum);
        }
        // the batch form: the same pair three times (forwards, with the roles of the frames swapped is another input, so: twice
        // forwards with different flags would need two calls -- one call, one flag set) plus an empty pair in the middle
        std::vector<std::vector<mi355::KeyPoint>> frames = {kp1, kp2};
        std::vector<mi355::Size> sizes = {mi355::Size(w, h), mi355::Size(w, h)};
        std::vector<std::pair<int, int>> prs = {{0, 1}, {0, 1}, {0, 1}};
        std::vector<std::vector<mi355::DMatch>> in = {matches, {}, matches}, outs;
        in[2].resize(n / 2);
        std::vector<bool> ok;
        mi355::matchGMSBatch(sizes, frames, prs, in, outs, true, true, 6.0, &ok);
        for (size_t p = 0; p < outs.size(); ++p) {
            uint64_t sum = 1469598103934665603ull;
            for (const auto& m : outs[p]) {
                uint32_t bits;
                __builtin_memcpy(&bits, &m.distance, 4);
                for (uint32_t v : {(uint32_t)m.queryIdx, (uint32_t)m.trainIdx, (uint32_t)m.imgIdx, bits})
                    sum = (sum ^ v) * 1099511628211ull;
            }
            std::printf("%zu %llu %d\n", outs[p].size(), (unsigned long long)sum, (int)ok[p]);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 3;
    }
    return 0;
}
