// Diagnostic microbenchmark (not product): LDS throughput for the access shapes the GMS kernel uses.
// One 1024-thread workgroup per CU; every thread issues N ops on a 16384-dword LDS table; cycles by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int S = 16384;
constexpr int N = 40;

template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned long long* cyc, uint32_t* sink)
{
    __shared__ uint32_t tab[S];
    const int tid = threadIdx.x;
    for (int i = tid; i < S; i += 1024) tab[i] = (MODE == 2) ? 0xFFFFFFFFu : 0;
    __syncthreads();
    uint32_t x = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        x = x * 1664525u + 1013904223u;
        uint32_t a;
        if (MODE == 1 || MODE == 5) a = (uint32_t)((tid + i * 1024) & (S - 1));      // lane-linear
        else a = (x >> 10) & (S - 1);                                                 // random
        if (MODE == 0 || MODE == 1) atomicAdd(&tab[a], 1u);                            // no-return add
        else if (MODE == 2) acc += atomicCAS(&tab[a], 0xFFFFFFFFu, x);                 // returning CAS
        else if (MODE == 3) acc += atomicAdd(&tab[a], 1u);                             // returning add
        else if (MODE == 4 || MODE == 5) acc += tab[a];                                // b32 read
        else if (MODE == 6) { uint4 v = *reinterpret_cast<uint4*>(&tab[a & ~3u]); acc += v.x ^ v.y ^ v.z ^ v.w; }  // b128 read
        else if (MODE == 7) atomicAdd(&tab[a & 1023], 1u);                             // random over 1024 addrs
        else if (MODE == 8) atomicAdd(&tab[(a & 0x1FF)], 1u);                          // hot 512
        else if (MODE == 9) atomicMax(&tab[a & 511], x);
    }
    __syncthreads();
    unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc + tab[tid];
}

template <int MODE>
void run(const char* name)
{
    unsigned long long* d; uint32_t* sink;
    hipMalloc(&d, 256 * 8); hipMalloc(&sink, 4096);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v; s /= 256;
    // 16 waves x N wave-instructions per workgroup
    printf("%-28s %9.0f cycles/WG  %7.1f cycles per wave-instruction (16 waves share the CU)\n", name, s, s / (16.0 * N));
    hipFree(d); hipFree(sink);
}

int main()
{
    run<0>("atomicAdd noret random");
    run<1>("atomicAdd noret linear");
    run<2>("atomicCAS ret random");
    run<3>("atomicAdd ret random");
    run<4>("read b32 random");
    run<5>("read b32 linear");
    run<6>("read b128 random");
    run<7>("atomicAdd noret rand1024");
    run<8>("atomicAdd noret hot512");
    run<9>("atomicMax noret hot512");
    return 0;
}
