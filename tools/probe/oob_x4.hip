// Probe: does a raw buffer_load_dwordx4 that straddles num_records return the in-range dwords (per-dword
// range check) or all zeros (whole-access check) on gfx950?  hipcc --offload-arch=gfx950 -o oob_x4 oob_x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned nrec_bytes, unsigned* out)
{
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), (short)0, (int)nrec_bytes, 0x00020000);
    v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, 0, 0);
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main()
{
    unsigned *d, *o;
    std::vector<unsigned> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    for (unsigned n : {64u, 61u, 62u, 63u}) {                  // records (dwords) in range; +1 offset: unaligned base
        for (int shift : {0, 1}) {
            hipMemset(o, 0xff, 1024);
            k<<<1, 64>>>(d + shift, n * 4, o);
            std::vector<unsigned> r(256);
            hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
            printf("n=%u shift=%d: lane15 = %u %u %u %u | lane16 = %u %u %u %u\n", n, shift, r[60], r[61], r[62], r[63], r[64], r[65], r[66], r[67]);
        }
    }
    return 0;
}
