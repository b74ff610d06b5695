// read_bw.hip -- what a READ-ONLY streaming kernel reaches on this device (the ceiling the cohort kernel,
// gd_sums_stream.hpp, should be measured against): hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip
//   A  grid-stride, 16 bytes per lane, 4 loads in flight
//   B  the cohort kernel's shape: a wave walks 16 consecutive 1 KB pieces of TWO arrays (2 pieces ahead)
//   C  as B with 64 pieces per wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void kA(const v4u* __restrict__ a, size_t n, unsigned* out)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (; i + 3 * st < n; i += 4 * st) {
        v4u x0 = a[i], x1 = a[i + st], x2 = a[i + 2 * st], x3 = a[i + 3 * st];
        acc += x0.x ^ x0.y ^ x0.z ^ x0.w ^ x1.x ^ x1.y ^ x1.z ^ x1.w ^ x2.x ^ x2.y ^ x2.z ^ x2.w ^ x3.x ^ x3.y ^ x3.z ^ x3.w;
    }
    if (acc == 0x12345678u) *out = acc;
}

template <int PIECES>
__global__ __launch_bounds__(256) void kB(const v4u* __restrict__ a, const v4u* __restrict__ b, size_t n_waves, unsigned* out)
{
    const int lane = threadIdx.x & 63;
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    const v4u* pa = a + w * PIECES * 64 + lane;
    const v4u* pb = b + w * PIECES * 64 + lane;
    unsigned acc = 0;
    v4u x0 = pa[0], y0 = pb[0], x1 = pa[64], y1 = pb[64];
#pragma unroll 1
    for (int k = 0; k < PIECES; k += 2) {
        v4u x2 = x0, y2 = y0, x3 = x1, y3 = y1;
        if (k + 2 < PIECES) { x2 = pa[(k + 2) * 64]; y2 = pb[(k + 2) * 64]; x3 = pa[(k + 3) * 64]; y3 = pb[(k + 3) * 64]; }
        acc += x0.x ^ x0.y ^ x0.z ^ x0.w ^ y0.x ^ y0.y ^ y0.z ^ y0.w;
        acc += x1.x ^ x1.y ^ x1.z ^ x1.w ^ y1.x ^ y1.y ^ y1.z ^ y1.w;
        x0 = x2; y0 = y2; x1 = x3; y1 = y3;
    }
    if (acc == 0x12345678u) *out = acc;
}

int main()
{
    const size_t bytes = (size_t)20 << 30;               // two arrays of 20 GB
    void *a, *b; unsigned* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, double gb, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("%-40s %8.3f ms  %6.2f TB/s\n", name, ms, gb / ms / 1e9 * 1e3 / 1e3);
    };
    const size_t n16 = bytes / 16;
    for (int blocks : {2048, 8192, 32768})
        timeit(blocks == 2048 ? "A grid-stride 2048 blocks" : blocks == 8192 ? "A grid-stride 8192 blocks" : "A grid-stride 32768 blocks",
               (double)bytes, [&] { hipLaunchKernelGGL(kA, dim3(blocks), dim3(256), 0, 0, (const v4u*)a, n16, out); });
    {
        const size_t nw = n16 / (16 * 64);
        timeit("B 16 x 1 KB per wave, two arrays", 2.0 * bytes, [&] { hipLaunchKernelGGL(kB<16>, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, 0, (const v4u*)a, (const v4u*)b, nw, out); });
    }
    {
        const size_t nw = n16 / (64 * 64);
        timeit("C 64 x 1 KB per wave, two arrays", 2.0 * bytes, [&] { hipLaunchKernelGGL(kB<64>, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, 0, (const v4u*)a, (const v4u*)b, nw, out); });
    }
    {
        const size_t nw = n16 / (256 * 64);
        timeit("D 256 x 1 KB per wave, two arrays", 2.0 * bytes, [&] { hipLaunchKernelGGL(kB<256>, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, 0, (const v4u*)a, (const v4u*)b, nw, out); });
    }
    return 0;
}
