// inflate_emul.cpp -- TEST INFRASTRUCTURE (tests/test_inflate_emul.py): the device inflate kernels compiled for the
// HOST and run as fibers, one per lane, so that the state machine of goleft_amd/csrc/gd_inflate.hpp can be fuzzed
// against zlib on the CPU box -- tens of thousands of streams, the rare paths (stored blocks, a member's first and last
// 16 bytes, periodic matches, block boundaries) included -- before a GPU ever sees a change.  The kernel SOURCE is the
// product's; what is replaced is the machine under it: `__shared__` memory is a static array, the 64 lanes of a
// workgroup are 64 fibers, `__syncthreads` and `__ballot` are barriers between them, the few gfx9 builtins the kernel uses are
// spelled out.  Nothing here is linked into the product.
//   clang++ -O2 -std=c++17 -shared -fPIC -o inflate_emul.so inflate_emul.cpp     (clang: ext_vector_type)
#include <algorithm>
#include "emul_machine.hpp"
#include "../../goleft_amd/csrc/gd_inflate.hpp"

// ---- a launch: the workgroups one after another ----------------------------------------------------------------------
static const gd::InflateJob* g_job = nullptr;
static void body_inflate() { gd::gd_inflate_kernel(*g_job); }
static void body_inflate_wave() { gd::gd_inflate_wave_kernel<gd::INF_WAVE_NW>(*g_job); }
static int g_kernel = 0;                                   // GD_OPT_INFLATE_KERNEL: 0 the lane-per-member kernel alone; 1 the workgroup-per-member kernel first
static uint32_t g_fallbacks = 0;
extern "C" void emul_inflate_kernel(int k) { g_kernel = k; }
extern "C" uint32_t emul_inflate_fallbacks() { return g_fallbacks; }   // members the last launch left to the lane-per-member kernel
static void body_crc() { gd::gd_inflate_crc_kernel(*g_job); }
static void body_crc_wave() { gd::gd_inflate_crc_wave_kernel(*g_job); }

// The same with every buffer ending exactly INF_SLACK bytes (what the device buffers are allocated beyond their contents)
// before an inaccessible page: a read or write past what the kernel may touch ends the process.
extern "C" int emul_inflate(const uint8_t* comp, const uint64_t* in_off, const uint32_t* in_len, const uint64_t* out_off,
                            const uint32_t* out_len, const uint32_t* crc, uint8_t* out, uint32_t* status, uint32_t n);

extern "C" int emul_inflate_guarded(const uint8_t* comp, uint64_t comp_bytes, const uint64_t* in_off, const uint32_t* in_len,
                                    const uint64_t* out_off, const uint32_t* out_len, const uint32_t* crc, uint8_t* out,
                                    uint64_t out_bytes, uint32_t* status, uint32_t n)
{
    emul::Guarded c(comp_bytes + gd::INF_SLACK), o(out_bytes + gd::INF_SLACK);
    memcpy(c.p, comp, comp_bytes);
    memset(c.p + comp_bytes, 0x5a, gd::INF_SLACK);
    memset(o.p, 0xee, out_bytes + gd::INF_SLACK);
    const int rc = emul_inflate(c.p, in_off, in_len, out_off, out_len, crc, o.p, status, n);
    for (size_t k = 0; k < gd::INF_SLACK; ++k)
        if (o.p[out_bytes + k] != 0xee) return -2;         // wrote past the last member
    memcpy(out, o.p, out_bytes);
    return rc;
}

extern "C" int emul_inflate(const uint8_t* comp, const uint64_t* in_off, const uint32_t* in_len, const uint64_t* out_off,
                            const uint32_t* out_len, const uint32_t* crc, uint8_t* out, uint32_t* status, uint32_t n)
{
    gd::InflateJob job{};
    job.comp = comp; job.in_off = in_off; job.in_len = in_len; job.out_off = out_off; job.out_len = out_len;
    job.crc = crc; job.out = out; job.status = status; job.n = n;
    g_job = &job;
    g_fallbacks = 0;
    if (g_kernel == 1) {
        for (unsigned b = 0; b < n; ++b) emul::run(body_inflate_wave, 64u * gd::INF_WAVE_NW, b);
        for (uint32_t i = 0; i < n; ++i) g_fallbacks += status[i] == gd::WV_FALLBACK;
        job.only_status = gd::WV_FALLBACK;
    }
    for (unsigned b = 0; b < (n + gd::INF_LANES - 1) / gd::INF_LANES; ++b) emul::run(body_inflate, gd::INF_LANES, b);
    if (crc) {
        // both CRC kernels: the wave-per-member one on a copy of the status words (a small grid: the grid-stride loop runs),
        // then the lane-per-member one -- and they must agree on every member
        std::vector<uint32_t> before(status, status + n);
        gridDim.x = n > 40 ? 3 : 1;
        for (unsigned b = 0; b < gridDim.x; ++b) emul::run(body_crc_wave, 256u, b);
        std::vector<uint32_t> wave(status, status + n);
        memcpy(status, before.data(), n * sizeof(uint32_t));
        for (unsigned b = 0; b < (n + 255u) / 256u; ++b) emul::run(body_crc, 256u, b);
        for (uint32_t i = 0; i < n; ++i)
            if (wave[i] != status[i]) return -3;
    }
    return 0;
}
