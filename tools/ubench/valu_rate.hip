// Diagnostic: VALU issue rate on gfx950 for the integer ops the GMS kernel uses, as a function of waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int MODE>
__global__ void k(unsigned long long* cyc, uint32_t* sink, int iters)
{
    uint32_t a = threadIdx.x * 2654435761u + 1, b = a ^ 0x5bd1e995u, c = a + 77, d = b + 99;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) { a = a ^ (b >> 3); b = b + (c & 0x7ff); c = c ^ (d << 5); d = d + (a | 1); }         // 8 simple int ops
            if (MODE == 1) { a = (a > b) ? c : a; b = (b > c) ? d : b; c = (c > d) ? a : c; d = (d > a) ? b : d; } // 4 cmp + 4 cndmask
            if (MODE == 2) { a = __umul24(a, b) + c; b = __umul24(b, c) + d; c = __umul24(c, d) + a; d = __umul24(d, a) + b; } // mad_u24
            if (MODE == 3) { a = a * b + c; b = b * c + d; c = c * d + a; d = d * a + b; }                       // 32-bit mul
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if ((a ^ b ^ c ^ d) == 0x12345678u) sink[0] = a;
}
template <int MODE>
void run(const char* name, int threads, int ops_per_iter)
{
    unsigned long long* d; uint32_t* sink;
    hipMalloc(&d, 256 * 8); hipMalloc(&sink, 64);
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v; s /= 256;
    const double instr_per_wave = (double)iters * 16 * ops_per_iter;
    const int waves_per_simd = threads / 64 / 4;
    printf("%-22s %2d waves/SIMD: %.2f cycles per wave-instruction per SIMD\n", name, waves_per_simd,
           s / (instr_per_wave * (waves_per_simd ? waves_per_simd : 1)));
    hipFree(d); hipFree(sink);
}
int main()
{
    for (int threads : {256, 512, 1024}) {
        run<0>("int xor/add/shift", threads, 12);
        run<1>("cmp + cndmask", threads, 8);
        run<2>("mad_u24", threads, 4);
        run<3>("mul_lo_u32 + add", threads, 8);
    }
    return 0;
}
