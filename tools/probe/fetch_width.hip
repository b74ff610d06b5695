// fetch_width.hip -- calibration of rocprofv3's FETCH_SIZE by LOAD WIDTH on this device (VERDICT
// round 3 item 6c).  MI355X_MICROARCH.md: FETCH_SIZE reports exactly half the bytes of a wide (16 B per lane) coalesced
// streaming read on gfx950 and "other access widths are uncalibrated".  The cohort kernel (gd_sums_stream.hpp) mixes
// 16-byte, 8-byte and 4-byte loads, and doubling all of its FETCH_SIZE gave 1.18x its algorithmic bytes.  Here ONE array of
// a known size is read exactly once by kernels that differ only in the width of their loads -- each lane 4, 8 or 16
// bytes per instruction, fully coalesced, grid-stride -- and, as the cohort kernel does for flag / MAPQ, 8 / 4 bytes per
// FOUR lanes' worth of records (a quarter / an eighth of a wave's lanes active).  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- ./fetch_width
// and compare the counter of each kernel with the bytes it read (tools/probe/fetch_width.sh prints the table).
//   hipcc --offload-arch=gfx950 -O3 -o fetch_width fetch_width.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

#define SINK(acc) if ((acc) == 0x12345678u) *out = (acc)

__global__ __launch_bounds__(256) void fw_b128(const v4u* __restrict__ a, size_t n, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const v4u x = a[i]; acc += x.x ^ x.y ^ x.z ^ x.w; }
    SINK(acc);
}
__global__ __launch_bounds__(256) void fw_b64(const v2u* __restrict__ a, size_t n, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const v2u x = a[i]; acc += x.x ^ x.y; }
    SINK(acc);
}
__global__ __launch_bounds__(256) void fw_b32(const unsigned* __restrict__ a, size_t n, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += a[i];
    SINK(acc);
}
// one wave-instruction covers 1 KB of the array like fw_b128, but as 4 instructions of 4 bytes per lane (what a
// kernel that loads a dword per record does)
__global__ __launch_bounds__(256) void fw_b32x4(const unsigned* __restrict__ a, size_t n, unsigned* out)
{
    unsigned acc = 0;
    const size_t st = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * st < n; i += 4 * st) acc += a[i] ^ a[i + st] ^ a[i + 2 * st] ^ a[i + 3 * st];
    SINK(acc);
}
// a 2-byte and a 1-byte array read as the cohort kernel reads flag / MAPQ: ONE 8-byte (4-byte) load per lane covers
// that lane's four records -- i.e. 16-bit / 8-bit elements, fully coalesced, 8 (4) bytes per lane
__global__ __launch_bounds__(256) void fw_u16x4(const v2u* __restrict__ a, size_t n, unsigned* out)
{
    unsigned acc = 0;
    const size_t st = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + st < n; i += 2 * st) { const v2u x = a[i], y = a[i + st]; acc += x.x ^ x.y ^ y.x ^ y.y; }
    SINK(acc);
}
// a gather: every lane loads one dword 64 bytes from its neighbour's (the first op of a read: CSR offsets apart)
__global__ __launch_bounds__(256) void fw_gather64(const unsigned* __restrict__ a, size_t n_lines, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * 256) acc += a[i * 16];
    SINK(acc);
}

int main()
{
    const size_t bytes = (size_t)8 << 30;                // 8 GiB: 32x the Infinity Cache
    void* a; unsigned* out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, bytes);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 16384;
    auto run = [&](const char* name, double read_bytes, auto launch) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-12s read_bytes %.0f  %8.3f ms  %6.2f TB/s\n", name, read_bytes, ms, read_bytes / ms / 1e9);
    };
    for (int rep = 0; rep < 2; ++rep) {
        run("fw_b128", (double)bytes, [&] { fw_b128<<<grid, 256>>>((const v4u*)a, bytes / 16, out); });
        run("fw_b64", (double)bytes, [&] { fw_b64<<<grid, 256>>>((const v2u*)a, bytes / 8, out); });
        run("fw_b32", (double)bytes, [&] { fw_b32<<<grid, 256>>>((const unsigned*)a, bytes / 4, out); });
        run("fw_b32x4", (double)bytes, [&] { fw_b32x4<<<grid, 256>>>((const unsigned*)a, bytes / 4, out); });
        run("fw_u16x4", (double)bytes, [&] { fw_u16x4<<<grid, 256>>>((const v2u*)a, bytes / 8, out); });
        run("fw_gather64", (double)bytes / 16 * 4, [&] { fw_gather64<<<grid, 256>>>((const unsigned*)a, bytes / 64, out); });
    }
    return 0;
}
