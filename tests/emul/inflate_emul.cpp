// inflate_emul.cpp -- TEST INFRASTRUCTURE (tests/test_inflate_emul.py): the device inflate kernels compiled for the
// HOST and run as fibers, one per lane, so that the state machine of goleft_amd/csrc/gd_inflate.hpp can be fuzzed
// against zlib on the CPU box -- tens of thousands of streams, the rare paths (stored blocks, a member's first and last
// 16 bytes, periodic matches, block boundaries) included -- before a GPU ever sees a change.  The kernel SOURCE is the
// product's; what is replaced is the machine under it: `__shared__` memory is a static array, the 64 lanes of a
// workgroup are 64 fibers, `__syncthreads` and `__ballot` are barriers between them, the few gfx9 builtins the kernel uses are
// spelled out.  Nothing here is linked into the product.
//   clang++ -O2 -std=c++17 -shared -fPIC -o inflate_emul.so inflate_emul.cpp     (clang: ext_vector_type)
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

// ---- the machine ------------------------------------------------------------------------------------------------
// The lanes of a workgroup are FIBERS of one OS thread (ucontext): a lane runs until it meets a barrier (a ballot is
// one), then the next lane runs; when the last one arrives they all go on.  Deterministic, and a barrier costs
// a context switch instead of a futex.
#include <ucontext.h>

struct Dim3 { unsigned x = 1, y = 1, z = 1; };
static Dim3 threadIdx, blockIdx;                           // of the running fiber (set at every switch)
namespace emul {
struct Lane { ucontext_t ctx; std::vector<char> stack; bool done = false; unsigned gen = 0, ballots = 0; };
static std::vector<Lane> lanes;
static ucontext_t sched_ctx;
static unsigned cur = 0, bar_gen = 0;
static uint64_t ballot_acc[3] = {0, 0, 0};
static void (*s_body)() = nullptr;

static void trampoline() { s_body(); lanes[cur].done = true; }
// all lanes that are still running must call this together
static void barrier()
{
    const unsigned my = bar_gen;
    lanes[cur].gen = my + 1;
    while (bar_gen == my) swapcontext(&lanes[cur].ctx, &sched_ctx);   // the scheduler opens the barrier when everyone is here
}
static void run(void (*body)(), unsigned n, unsigned block)
{
    lanes.assign(n, Lane());
    bar_gen = 0; ballot_acc[0] = ballot_acc[1] = ballot_acc[2] = 0;
    s_body = body;
    for (unsigned t = 0; t < n; ++t) {
        Lane& l = lanes[t];
        l.stack.resize(256 * 1024);
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = l.stack.data();
        l.ctx.uc_stack.ss_size = l.stack.size();
        l.ctx.uc_link = &sched_ctx;
        makecontext(&l.ctx, trampoline, 0);
    }
    for (;;) {
        unsigned alive = 0, waiting = 0;
        for (unsigned t = 0; t < n; ++t)
            if (!lanes[t].done) { ++alive; if (lanes[t].gen == bar_gen + 1) ++waiting; }
        if (alive == 0) break;
        if (waiting == alive) ++bar_gen;                    // everybody still running is at the barrier: open it
        for (unsigned t = 0; t < n; ++t) {
            Lane& l = lanes[t];
            if (l.done || l.gen == bar_gen + 1) continue;   // finished, or waiting at a barrier that is still closed
            cur = t;
            threadIdx.x = t;
            blockIdx.x = block;
            swapcontext(&sched_ctx, &l.ctx);
        }
    }
}
}  // namespace emul

static inline void emul_syncthreads() { emul::barrier(); }
static inline uint64_t emul_ballot(bool p)
{
    // Three accumulators in turn and ONE barrier per ballot.  Ballot k uses accumulator k % 3 and clears (k + 1) % 3 on
    // the way in: that one was last read in ballot k - 2, and every lane has read it before it arrived at barrier
    // k - 1, which is behind whoever enters ballot k; nobody adds to it before passing barrier k, i.e. before every
    // lane has entered ballot k and cleared it.
    const unsigned k = emul::lanes[emul::cur].ballots++;
    emul::ballot_acc[(k + 1u) % 3u] = 0;
    if (p) emul::ballot_acc[k % 3u] |= 1ull << threadIdx.x;
    emul::barrier();
    return emul::ballot_acc[k % 3u];
}
static inline uint32_t emul_alignbyte(uint32_t hi, uint32_t lo, uint32_t s)
{
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (s & 3u)));
}
static inline uint32_t emul_perm(uint32_t s0, uint32_t s1, uint32_t sel)   // V_PERM_B32: selectors 0-3 bytes of s1, 4-7 of s0, 12: 0x00
{
    const uint64_t src = ((uint64_t)s0 << 32) | s1;
    uint32_t out = 0;
    for (int b = 0; b < 4; ++b) {
        const uint32_t k = (sel >> (8 * b)) & 0xffu;
        uint32_t v;
        if (k < 8u) v = (uint32_t)(src >> (8 * k)) & 0xffu;
        else if (k == 12u) v = 0u;
        else if (k >= 13u) v = 0xffu;
        else v = ((src >> (16 * (k - 8u) + 15)) & 1u) ? 0xffu : 0u;   // 8-11: sign of a 16-bit half (not used by the kernel)
        out |= v << (8 * b);
    }
    return out;
}
static inline uint32_t emul_brev(uint32_t x)
{
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    return __builtin_bswap32(x);
}

#define __global__
#define __device__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(n)
#define __syncthreads emul_syncthreads
#define __ballot emul_ballot
#define __popcll __builtin_popcountll
#define __brev emul_brev
#define __builtin_amdgcn_alignbyte emul_alignbyte
#define __builtin_amdgcn_perm emul_perm
typedef int hipStream_t;
struct dim3 { unsigned x; dim3(unsigned a) : x(a) {} };
#define hipLaunchKernelGGL(...) ((void)0)

#include "../../goleft_amd/csrc/gd_inflate.hpp"

// ---- a launch: the workgroups one after another ----------------------------------------------------------------------
static const gd::InflateJob* g_job = nullptr;
static void body_inflate() { gd::gd_inflate_kernel(*g_job); }
static void body_crc() { gd::gd_inflate_crc_kernel(*g_job); }

// The same with every buffer ending exactly INF_SLACK bytes (what the device buffers are allocated beyond their contents)
// before an inaccessible page: a read or write past what the kernel may touch ends the process.
#include <sys/mman.h>
namespace {
struct Guarded {
    uint8_t* map = nullptr; size_t len = 0; uint8_t* p = nullptr;
    Guarded(size_t n)
    {
        const size_t page = 4096, body = (n + page - 1) / page * page;
        len = body + page;
        map = static_cast<uint8_t*>(mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        mprotect(map + body, page, PROT_NONE);
        p = map + (body - n);
    }
    ~Guarded() { if (map) munmap(map, len); }
};
}
extern "C" int emul_inflate(const uint8_t* comp, const uint64_t* in_off, const uint32_t* in_len, const uint64_t* out_off,
                            const uint32_t* out_len, const uint32_t* crc, uint8_t* out, uint32_t* status, uint32_t n);

extern "C" int emul_inflate_guarded(const uint8_t* comp, uint64_t comp_bytes, const uint64_t* in_off, const uint32_t* in_len,
                                    const uint64_t* out_off, const uint32_t* out_len, const uint32_t* crc, uint8_t* out,
                                    uint64_t out_bytes, uint32_t* status, uint32_t n)
{
    Guarded c(comp_bytes + gd::INF_SLACK), o(out_bytes + gd::INF_SLACK);
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
    for (unsigned b = 0; b < (n + gd::INF_LANES - 1) / gd::INF_LANES; ++b) emul::run(body_inflate, gd::INF_LANES, b);
    if (crc)
        for (unsigned b = 0; b < (n + 255u) / 256u; ++b) emul::run(body_crc, 256u, b);
    return 0;
}
