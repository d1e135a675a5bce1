// Diagnostic (not product): what the host side of gms_filter_host_batch can move on this box.
//   memcpy pageable -> pinned and pinned -> pageable with 1..16 threads (32 MB pieces, the chunk size of the host batch);
//   hipMemcpyAsync pinned H2D, D2H, and both directions at once on two streams.
// Build + run on the GPU box:  hipcc -O2 -pthread tools/ubench/host_copy_rate.cpp -o /tmp/host_copy_rate && /tmp/host_copy_rate
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_copy(char* dst, const char* src, size_t bytes, int n_thr)
{
    std::vector<std::thread> pool;
    for (int t = 1; t < n_thr; ++t) pool.emplace_back([=] { std::memcpy(dst + bytes * t / n_thr, src + bytes * t / n_thr, bytes * (t + 1) / n_thr - bytes * t / n_thr); });
    std::memcpy(dst, src, bytes / n_thr);
    for (auto& th : pool) th.join();
}

int main()
{
    const size_t piece = (size_t)32 << 20, total = (size_t)512 << 20;
    char* pageable = (char*)malloc(total);
    char* pageable2 = (char*)malloc(total);
    memset(pageable, 1, total);
    memset(pageable2, 2, total);
    char *pin_a, *pin_b, *dev_a, *dev_b;
    hipHostMalloc((void**)&pin_a, piece, hipHostMallocDefault);
    hipHostMalloc((void**)&pin_b, piece, hipHostMallocDefault);
    hipMalloc((void**)&dev_a, piece);
    hipMalloc((void**)&dev_b, piece);
    memset(pin_a, 3, piece);
    memset(pin_b, 4, piece);
    printf("{\"hardware_concurrency\": %u", std::thread::hardware_concurrency());
    for (int n_thr : {1, 2, 4, 8, 16}) {
        double t0 = now();
        for (size_t off = 0; off < total; off += piece) par_copy(pin_a, pageable + off, piece, n_thr);
        double t1 = now();
        for (size_t off = 0; off < total; off += piece) par_copy(pageable2 + off, pin_a, piece, n_thr);
        double t2 = now();
        printf(", \"memcpy_%dthr_pageable_to_pinned_GBps\": %.2f, \"memcpy_%dthr_pinned_to_pageable_GBps\": %.2f", n_thr, total / (t1 - t0) / 1e9, n_thr,
               total / (t2 - t1) / 1e9);
    }
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    const int reps = 16;
    hipMemcpyAsync(dev_a, pin_a, piece, hipMemcpyHostToDevice, s0);
    hipStreamSynchronize(s0);
    double t0 = now();
    for (int i = 0; i < reps; ++i) hipMemcpyAsync(dev_a, pin_a, piece, hipMemcpyHostToDevice, s0);
    hipStreamSynchronize(s0);
    double t1 = now();
    for (int i = 0; i < reps; ++i) hipMemcpyAsync(pin_b, dev_b, piece, hipMemcpyDeviceToHost, s1);
    hipStreamSynchronize(s1);
    double t2 = now();
    for (int i = 0; i < reps; ++i) {
        hipMemcpyAsync(dev_a, pin_a, piece, hipMemcpyHostToDevice, s0);
        hipMemcpyAsync(pin_b, dev_b, piece, hipMemcpyDeviceToHost, s1);
    }
    hipStreamSynchronize(s0);
    hipStreamSynchronize(s1);
    double t3 = now();
    printf(", \"hipMemcpy_pinned_H2D_GBps\": %.2f, \"hipMemcpy_pinned_D2H_GBps\": %.2f, \"both_directions_at_once_GBps_each\": %.2f}\n", reps * piece / (t1 - t0) / 1e9,
           reps * piece / (t2 - t1) / 1e9, reps * piece / (t3 - t2) / 1e9);
    return 0;
}
