// emul_machine.hpp -- TEST INFRASTRUCTURE: what a HIP kernel of this repository needs in order to compile for the host and
// run lane by lane (tests/emul/*.cpp).  Include this, then the kernel header.
#pragma once
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
static Dim3 gridDim, blockDim;                             // set by whoever launches a kernel that asks for them
namespace emul {
// A lane waits for a generation counter to move on: the workgroup's (a barrier) or its wave's (a ballot -- the 64 lanes of a
// wave are in step on the machine, the waves of a workgroup are not: a kernel whose waves run different loops -- the
// decoder / writer pair of the round-5 experiment, commits 886121a and fdd3ff3 in the history -- ballots per wave).
struct Lane { ucontext_t ctx; std::vector<char> stack; bool done = false; const unsigned* wait_ctr = nullptr; unsigned wait_val = 0, ballots = 0; };
constexpr unsigned MAX_WAVES = 8, WAVE = 64;
static std::vector<Lane> lanes;
static ucontext_t sched_ctx;
static unsigned cur = 0, bar_gen = 0, wbar_gen[MAX_WAVES];
static uint64_t ballot_acc[MAX_WAVES][3];
static void (*s_body)() = nullptr;

static void trampoline() { s_body(); lanes[cur].done = true; }
static void wait_on(const unsigned* ctr)
{
    Lane& l = lanes[cur];
    l.wait_ctr = ctr;
    l.wait_val = *ctr;
    while (*ctr == l.wait_val) swapcontext(&l.ctx, &sched_ctx);   // the scheduler opens the barrier when everyone is here
    l.wait_ctr = nullptr;
}
// all lanes of the workgroup that are still running must call this together
static void barrier() { wait_on(&bar_gen); }
// all lanes of the caller's wave that are still running must call this together
static void wave_barrier() { wait_on(&wbar_gen[cur / WAVE]); }
static void run(void (*body)(), unsigned n, unsigned block)
{
    lanes.assign(n, Lane());
    bar_gen = 0;
    for (unsigned w = 0; w < MAX_WAVES; ++w) { wbar_gen[w] = 0; ballot_acc[w][0] = ballot_acc[w][1] = ballot_acc[w][2] = 0; }
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
        unsigned alive = 0, at_bar = 0, w_alive[MAX_WAVES] = {}, w_at[MAX_WAVES] = {};
        for (unsigned t = 0; t < n; ++t) {
            const Lane& l = lanes[t];
            if (l.done) continue;
            ++alive; ++w_alive[t / WAVE];
            if (l.wait_ctr == &bar_gen && *l.wait_ctr == l.wait_val) ++at_bar;
            if (l.wait_ctr == &wbar_gen[t / WAVE] && *l.wait_ctr == l.wait_val) ++w_at[t / WAVE];
        }
        if (alive == 0) break;
        if (at_bar == alive) ++bar_gen;                     // everybody still running is at the barrier: open it
        for (unsigned w = 0; w < MAX_WAVES; ++w)
            if (w_alive[w] != 0 && w_at[w] == w_alive[w]) ++wbar_gen[w];
        for (unsigned t = 0; t < n; ++t) {
            Lane& l = lanes[t];
            if (l.done || (l.wait_ctr && *l.wait_ctr == l.wait_val)) continue;   // finished, or waiting at a barrier that is still closed
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
    // Per WAVE.  Three accumulators in turn and ONE barrier per ballot.  Ballot k uses accumulator k % 3 and clears
    // (k + 1) % 3 on the way in: that one was last read in ballot k - 2, and every lane has read it before it arrived at
    // barrier k - 1, which is behind whoever enters ballot k; nobody adds to it before passing barrier k, i.e. before
    // every lane has entered ballot k and cleared it.
    const unsigned k = emul::lanes[emul::cur].ballots++, w = emul::cur / emul::WAVE;
    emul::ballot_acc[w][(k + 1u) % 3u] = 0;
    if (p) emul::ballot_acc[w][k % 3u] |= 1ull << (threadIdx.x % emul::WAVE);
    emul::wave_barrier();
    return emul::ballot_acc[w][k % 3u];
}
// v_readlane / a shuffle: every lane of the wave publishes its value, then reads another lane's (all running lanes of the wave call together)
namespace emul { static uint32_t xchg[MAX_WAVES][WAVE]; }
static inline uint32_t emul_readlane(uint32_t v, uint32_t l)
{
    const unsigned w = emul::cur / emul::WAVE;
    emul::xchg[w][threadIdx.x % emul::WAVE] = v;
    emul::wave_barrier();
    const uint32_t r = emul::xchg[w][l % emul::WAVE];
    emul::wave_barrier();
    return r;
}
static inline uint32_t emul_shfl_up(uint32_t v, uint32_t d)
{
    const unsigned w = emul::cur / emul::WAVE, lane = threadIdx.x % emul::WAVE;
    emul::xchg[w][lane] = v;
    emul::wave_barrier();
    const uint32_t r = lane >= d ? emul::xchg[w][lane - d] : v;
    emul::wave_barrier();
    return r;
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
#define __launch_bounds__(...)
#define __syncthreads emul_syncthreads
#define __ballot emul_ballot
#define __popcll __builtin_popcountll
#define __popc __builtin_popcount
#define __brev emul_brev
#define __builtin_amdgcn_alignbyte emul_alignbyte
#define __builtin_amdgcn_perm emul_perm
#define __builtin_amdgcn_alignbit(hi, lo, s) ((uint32_t)((((uint64_t)(hi) << 32) | (uint32_t)(lo)) >> ((s) & 31u)))
#define __builtin_amdgcn_ubfe(v, off, w) (((w) & 31u) == 0u ? 0u : (((uint32_t)(v) >> ((off) & 31u)) & ((1u << ((w) & 31u)) - 1u)))
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define GD_EMUL_HOST 1
typedef int hipStream_t;
struct dim3 { unsigned x; dim3(unsigned a) : x(a) {} };
#define hipLaunchKernelGGL(...) ((void)0)


// A buffer of n bytes that ends exactly at an inaccessible page: a read or write past it ends the process.
#include <sys/mman.h>
namespace emul {
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
