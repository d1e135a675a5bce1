// Diagnostic (not product): calibrates rocprofv3's FETCH_SIZE on gfx950 for the access shapes the GMS filter
// uses, on buffers far larger than the 256 MiB Infinity Cache (so every byte really comes from HBM):
//   k_stride8 : 8 bytes per lane at a 16-byte stride  (the (queryIdx, trainIdx) read of 16-byte DMatch records)
//   k_vec16   : 16 bytes per lane, contiguous          (the DMatch re-read of the copy-out; also the guide's case)
// Every 128-byte line of the buffer is touched exactly once by both, so the true HBM read is the buffer size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_stride8(const uint4* __restrict__ p, size_t n, uint32_t* sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += stride) {
        const int2 v = *reinterpret_cast<const int2*>(&p[i]);
        acc += (uint32_t)(v.x ^ v.y);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_vec16(const uint4* __restrict__ p, size_t n, uint32_t* sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += stride) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main()
{
    const size_t bytes = (size_t)2 << 30;  // 2 GiB
    uint4* d; uint32_t* sink;
    hipMalloc(&d, bytes); hipMalloc(&sink, 64);
    hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    const size_t n = bytes / 16;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stride8, dim3(4096), dim3(256), 0, 0, d, n, sink);
        hipLaunchKernelGGL(k_vec16, dim3(4096), dim3(256), 0, 0, d, n, sink);
    }
    hipDeviceSynchronize();
    printf("buffer bytes %zu\n", bytes);
    return 0;
}
