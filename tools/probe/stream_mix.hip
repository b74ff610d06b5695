// stream_mix.hip -- which ingredient of the cohort kernel (gd_sums_stream.hpp) costs the read bandwidth?  The
// read-only streaming shape (a wave walks 16 consecutive 1 KB pieces of two arrays, 2 pieces in flight) reaches
// 6.4 TB/s on MI355X (read_bw.hip); the real kernel 4.4.  Ingredients are added one at a time (template flags):
//   OPS    third array read as 4 dword loads per lane, 16-byte lane stride (the first-op gather)
//   VALU   ~N dependent integer VALU instructions per piece
//   SCAN   4 wave prefix sums (DPP) + 3 bpermutes per piece
//   ATOM   3 global 64-bit atomics per piece from ~6 lanes each, addresses following the stream
//   PRO    a chain of 3 dependent global loads before the stream starts
// hipcc --offload-arch=gfx950 -O3 -o stream_mix stream_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int wscan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

template <int PIECES, bool OPS, int VALU, bool SCAN, bool ATOM, bool PRO>
__global__ __launch_bounds__(256) void kmix(const v4u* __restrict__ a, const v4u* __restrict__ b, const unsigned* __restrict__ c,
                                            const unsigned* __restrict__ chain, unsigned long long* wsum, size_t n_waves, unsigned* out)
{
    const int lane = threadIdx.x & 63;
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    unsigned salt = 0;
    if (PRO) {
        unsigned i = chain[(w * 7) & 1023];
        i = chain[i & 1023];
        i = chain[(i + 1) & 1023];
        salt = i & 1u;
    }
    const v4u* pa = a + w * PIECES * 64 + lane + salt * 0;
    const v4u* pb = b + w * PIECES * 64 + lane;
    const unsigned* pc = c + w * PIECES * 256 + lane * 4;
    unsigned acc = salt;
    v4u x0 = pa[0], y0 = pb[0], x1 = pa[64], y1 = pb[64];
    unsigned o0[4] = {0, 0, 0, 0};
    if (OPS) { o0[0] = pc[0]; o0[1] = pc[1]; o0[2] = pc[2]; o0[3] = pc[3]; }
#pragma unroll 1
    for (int k = 0; k < PIECES; ++k) {
        v4u x2 = x1, y2 = y1;
        if (k + 2 < PIECES) { x2 = pa[(k + 2) * 64]; y2 = pb[(k + 2) * 64]; }
        unsigned o1[4] = {0, 0, 0, 0};
        if (OPS && k + 1 < PIECES) { const unsigned* q = pc + (k + 1) * 256; o1[0] = q[0]; o1[1] = q[1]; o1[2] = q[2]; o1[3] = q[3]; }
        unsigned t = x0.x ^ x0.y ^ x0.z ^ x0.w ^ y0.x ^ y0.y ^ y0.z ^ y0.w ^ o0[0] ^ o0[1] ^ o0[2] ^ o0[3];
#pragma unroll
        for (int i = 0; i < VALU; ++i) t = (t ^ (t >> 3)) + (unsigned)i;        // 2 VALU each, dependent
        if (SCAN) {
            int s0 = wscan((int)t), s1 = wscan((int)(t >> 1)), s2 = wscan((int)(t >> 2)), s3 = wscan((int)(t >> 3));
            const int pt = (lane * 7) & 63;
            t = (unsigned)(__shfl(s0, pt, 64) ^ __shfl(s1, pt, 64) ^ __shfl(s2, pt, 64) ^ s3);
        }
        if (ATOM) {
            if ((lane & 7) == 3) {
                const size_t win = (w * PIECES + k) * 5 + (lane >> 3);           // ~5 windows per piece, neighbours overlap
                atomicAdd(&wsum[win], (unsigned long long)(t & 0xff));
                atomicAdd(&wsum[win + 1], (unsigned long long)(t & 0xf));
                atomicAdd(&wsum[win + 2], (unsigned long long)(t & 0x3));
            }
        }
        acc += t;
        x0 = x1; y0 = y1; x1 = x2; y1 = y2;
#pragma unroll
        for (int u = 0; u < 4; ++u) o0[u] = o1[u];
    }
    if (acc == 0x12345678u) *out = acc;
}

int main()
{
    const size_t bytes = (size_t)16 << 30;               // three arrays of 16 GB
    void *a, *b, *c; unsigned* out; unsigned* chain; unsigned long long* wsum;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&c, bytes); (void)hipMalloc(&out, 4);
    (void)hipMalloc(&chain, 4096); (void)hipMalloc(&wsum, (size_t)1 << 30);
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 2, bytes); (void)hipMemset(c, 3, bytes); (void)hipMemset(chain, 0, 4096);
    (void)hipMemset(wsum, 0, (size_t)1 << 30);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    constexpr int P = 16;
    const size_t nw = bytes / 16 / (P * 64);
    const dim3 grid((unsigned)((nw + 3) / 4));
#define RUN(NAME, GB, ...)                                                                                          \
    {                                                                                                               \
        auto launch = [&] { hipLaunchKernelGGL((kmix<P, __VA_ARGS__>), grid, dim3(256), 0, 0, (const v4u*)a, (const v4u*)b, \
                                               (const unsigned*)c, chain, wsum, nw, out); };                        \
        launch(); (void)hipDeviceSynchronize();                                                                     \
        (void)hipEventRecord(e0); for (int i = 0; i < 3; ++i) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); \
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;                                                  \
        printf("%-52s %8.3f ms  %6.2f TB/s\n", NAME, ms, (GB) * (double)bytes / ms / 1e9);                          \
    }
    RUN("two arrays", 2.0, false, 0, false, false, false)
    RUN("+ ops gather (3 arrays)", 3.0, true, 0, false, false, false)
    RUN("+ ops + 100 VALU", 3.0, true, 50, false, false, false)
    RUN("+ ops + 200 VALU", 3.0, true, 100, false, false, false)
    RUN("+ ops + 300 VALU", 3.0, true, 150, false, false, false)
    RUN("+ ops + 100 VALU + scans", 3.0, true, 50, true, false, false)
    RUN("+ ops + 100 VALU + scans + atomics", 3.0, true, 50, true, true, false)
    RUN("+ ops + 100 VALU + scans + atomics + prologue", 3.0, true, 50, true, true, true)
    RUN("two arrays + scans", 2.0, false, 0, true, false, false)
    RUN("two arrays + atomics", 2.0, false, 0, false, true, false)
    RUN("two arrays + prologue", 2.0, false, 0, false, false, true)
    RUN("two arrays + 200 VALU", 2.0, false, 100, false, false, false)
    return 0;
}
