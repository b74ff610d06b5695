// tile_overlap.hip -- the experiment DESIGN.md section 7 named for the tile kernels: does a PERSISTENT workgroup that
// issues tile t+1's loads before it stores tile t move the byte mix of a tile kernel (13 KB read + 16 KB written
// per 4096-position tile) faster than one workgroup per tile with seven of them resident per CU?
//
// A tile here does what bounds the real kernels and nothing else: every lane loads L 16-byte vectors (records + ops:
// L = 3 is 12 KB per tile, the short-read kernel on raw records reads 13), the workgroup passes them through LDS
// with two barriers (phase A marks -> phase B scan: a tile's stores cannot start before all of its loads have
// landed), WORK dependent VALU instructions per stored vector stand for the scan / window / class arithmetic, and
// every lane stores four 16-byte vectors (non-temporal; 16 KB per tile, the per-base vector).
//   variant 0  one workgroup per tile, 22 KB of LDS (seven per CU)                       -- the shipped shape
//   variant 1  persistent workgroups (4 per CU, 38 KB of LDS: two tiles' arrays), the loads of tile t+1 are issued
//              BEFORE the stores of tile t and consumed after them
//   variant 2  persistent workgroups, same residency as 1, no software pipelining (control)
// hipcc --offload-arch=gfx950 -O3 -o tile_overlap tile_overlap.hip && ./tile_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int T = 4096;

template <int L>
__device__ __forceinline__ void load_tile(const v4u* __restrict__ in, size_t tile, int tid, v4u (&x)[L])
{
#pragma unroll
    for (int k = 0; k < L; ++k) x[k] = in[(tile * L + k) * 256 + tid];
}

template <int L, int WORK>
__device__ __forceinline__ void process_store(int32_t* s_diff, v4u (&x)[L], int32_t* __restrict__ out, size_t tile, int tid)
{
    // phase A: the loaded records become marks in the tile's LDS array
#pragma unroll
    for (int k = 0; k < L; ++k) {
        atomicAdd(&s_diff[(x[k].x + 4u * tid) & (T - 1)], 1);
        atomicAdd(&s_diff[(x[k].y + x[k].z) & (T - 1)], -1);
    }
    __syncthreads();
    // phase B: every lane owns 16 positions, "scans" them and stores
    v4i v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = *reinterpret_cast<const v4i*>(&s_diff[(r * 256 + tid) * 4]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<v4i*>(&s_diff[(r * 256 + tid) * 4]) = v4i{0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int a = v[r].x + v[r].y, b = v[r].z + v[r].w;
#pragma unroll
        for (int i = 0; i < WORK; ++i) { a = (a ^ (b >> 3)) + i; b = (b + (a << 1)) ^ i; }
        v[r].x += a; v[r].w += b;
        __builtin_nontemporal_store(v[r], reinterpret_cast<v4i*>(out + tile * T + (size_t)(r * 256 + tid) * 4));
    }
}

template <int L, int WORK>
__global__ __launch_bounds__(256) void k_one(const v4u* __restrict__ in, int32_t* __restrict__ out, size_t n_tiles)
{
    __shared__ __attribute__((aligned(16))) int32_t s_diff[T + 1400];      // 22 KB like the shipped kernel
    const int tid = threadIdx.x;
    const size_t per = (n_tiles + 7) >> 3;
    const size_t tile = (size_t)(blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    v4u x[L];
    load_tile<L>(in, tile, tid, x);
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<v4i*>(&s_diff[(r * 256 + tid) * 4]) = v4i{0, 0, 0, 0};
    __syncthreads();
    process_store<L, WORK>(s_diff, x, out, tile, tid);
}

template <int L, int WORK, bool PIPE>
__global__ __launch_bounds__(256) void k_persist(const v4u* __restrict__ in, int32_t* __restrict__ out, size_t n_tiles)
{
    __shared__ __attribute__((aligned(16))) int32_t s_diff[2][T + 768];    // 38 KB: four workgroups per CU
    const int tid = threadIdx.x;
    // every workgroup owns a contiguous run of tiles (neighbouring tiles share their look-back reads in the real kernel)
    const size_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
    const size_t t0 = (size_t)blockIdx.x * per, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
    if (t0 >= t1) return;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<v4i*>(&s_diff[b][(r * 256 + tid) * 4]) = v4i{0, 0, 0, 0};
    __syncthreads();
    v4u x[L], y[L];
    load_tile<L>(in, t0, tid, x);
    int buf = 0;
    for (size_t t = t0; t < t1; ++t) {
        if (PIPE && t + 1 < t1) load_tile<L>(in, t + 1, tid, y);          // in flight while tile t is marked, scanned, stored
        process_store<L, WORK>(s_diff[buf], x, out, t, tid);
        if (!PIPE && t + 1 < t1) load_tile<L>(in, t + 1, tid, y);
#pragma unroll
        for (int k = 0; k < L; ++k) x[k] = y[k];
        buf ^= 1;
    }
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int L, int WORK>
void run(size_t n_tiles, const v4u* in, int32_t* out, int cus)
{
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const double bytes = (double)n_tiles * (L * 4096.0 + T * 4.0);
    for (int variant = 0; variant < 3; ++variant) {
        for (int wg_per_cu : {4, 8}) {
            if (variant == 0 && wg_per_cu != 4) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 6; ++rep) {
                CHK(hipEventRecord(a));
                if (variant == 0)
                    hipLaunchKernelGGL((k_one<L, WORK>), dim3((unsigned)(((n_tiles + 7) / 8) * 8)), dim3(256), 0, 0, in, out, n_tiles);
                else if (variant == 1)
                    hipLaunchKernelGGL((k_persist<L, WORK, true>), dim3(cus * wg_per_cu), dim3(256), 0, 0, in, out, n_tiles);
                else
                    hipLaunchKernelGGL((k_persist<L, WORK, false>), dim3(cus * wg_per_cu), dim3(256), 0, 0, in, out, n_tiles);
                CHK(hipEventRecord(b));
                CHK(hipEventSynchronize(b));
                float ms;
                CHK(hipEventElapsedTime(&ms, a, b));
                if (rep && ms < best) best = ms;
            }
            printf("L=%d (%.0f KB read + 16 KB written per tile) WORK=%d  %-44s %7.3f ms  %6.2f TB/s\n", L, L * 4.0, WORK,
                   variant == 0 ? "one workgroup per tile (7 per CU)" :
                   variant == 1 ? (wg_per_cu == 4 ? "persistent, next tile's loads first, grid 4/CU" : "persistent, next tile's loads first, grid 8/CU") :
                                  (wg_per_cu == 4 ? "persistent, no pipelining, grid 4/CU" : "persistent, no pipelining, grid 8/CU"),
                   best, bytes / best / 1e9);
        }
    }
}

int main()
{
    const size_t n_tiles = 755785;                                         // a 30x hg19 genome
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    v4u* in; int32_t* out;
    CHK(hipMalloc(&in, n_tiles * 4 * 4096));
    CHK(hipMalloc(&out, n_tiles * T * 4));
    CHK(hipMemset(in, 1, n_tiles * 4 * 4096));
    printf("%s, %d CUs, %zu tiles\n", p.name, p.multiProcessorCount, n_tiles);
    run<3, 0>(n_tiles, in, out, p.multiProcessorCount);
    run<3, 24>(n_tiles, in, out, p.multiProcessorCount);
    run<3, 64>(n_tiles, in, out, p.multiProcessorCount);
    run<2, 24>(n_tiles, in, out, p.multiProcessorCount);
    return 0;
}
