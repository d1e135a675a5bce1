// Diagnostic: how fast ONE workgroup that owns its CU (1024 threads, the whole LDS) gets a pair's match records into registers, in the
// filter kernel's pattern -- ten 16-byte loads per thread, lane-consecutive -- while the other CUs are somewhere else in their pairs
// (every workgroup idles ~40k cycles after its load, like the kernel's compute phases). Variants: the whole record (dwordx4), the first
// eight bytes of every record (dwordx2 at a 16-byte stride), half the records; from HBM (every workgroup its own 160 KB) or from a
// footprint that stays in L2. Prints mean shader cycles from the first load to the last byte.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/load_rate.hip -o /tmp/load_rate && /tmp/load_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(1024) k(const uint4* __restrict__ src, size_t region_u4, int n_regions, int idle_sleeps, uint32_t* sink,
                                          unsigned long long* cyc)
{
    extern __shared__ uint32_t smem[];
    const int tid = threadIdx.x;
    const uint4* base = src + (size_t)(blockIdx.x % n_regions) * region_u4;
    uint4 r[10];
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) r[i] = base[i * 1024 + tid];
    } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const uint2 v = *reinterpret_cast<const uint2*>(base + i * 1024 + tid);
            r[i] = make_uint4(v.x, v.y, 0, 0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 5; ++i) r[i] = base[i * 1024 + tid];
#pragma unroll
        for (int i = 5; i < 10; ++i) r[i] = make_uint4(0, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) acc += r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
    smem[tid] = acc;
    for (int i = 0; i < idle_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    if (acc == 0x12345678u) sink[0] = acc + smem[(tid + 1) & 1023];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, const uint4* d, size_t region_u4, int n_regions, int n_wg, uint32_t* sink, unsigned long long* dcyc)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<unsigned long long> h(n_wg);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<MODE>, dim3(n_wg), dim3(1024), 160 * 1024, 0, d, region_u4, n_regions, 5, sink, dcyc);  // 5 x 127 x 64 clocks ~ 40k cycles
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), dcyc, n_wg * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0, s2 = 0;
    for (int i = 256; i < n_wg; ++i) s += (double)h[i];      // (the first dispatch round starts in lockstep: left out)
    for (int i = 0; i < 256; ++i) s2 += (double)h[i];
    printf("%-44s %8.0f cycles (first round, in lockstep: %8.0f)\n", name, s / (n_wg - 256), s2 / 256);
}

int main()
{
    const int n_wg = 4096;
    const size_t region_u4 = 10 * 1024;  // 160 KB
    uint4* d;
    hipMalloc(&d, (size_t)n_wg * region_u4 * sizeof(uint4));
    hipMemset(d, 1, (size_t)n_wg * region_u4 * sizeof(uint4));
    uint32_t* sink;
    unsigned long long* dcyc;
    hipMalloc(&sink, 64);
    hipMalloc(&dcyc, n_wg * sizeof(unsigned long long));
    run<0>("HBM, whole records (160 KB, dwordx4)", d, region_u4, n_wg, n_wg, sink, dcyc);
    run<1>("HBM, first 8 bytes of every record (dwordx2)", d, region_u4, n_wg, n_wg, sink, dcyc);
    run<2>("HBM, half the records (80 KB, dwordx4)", d, region_u4, n_wg, n_wg, sink, dcyc);
    run<0>("L2 (8 regions), whole records", d, region_u4, 8, n_wg, sink, dcyc);
    run<1>("L2 (8 regions), first 8 bytes", d, region_u4, 8, n_wg, sink, dcyc);
    run<2>("L2 (8 regions), half the records", d, region_u4, 8, n_wg, sink, dcyc);
    return 0;
}
