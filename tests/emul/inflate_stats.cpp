// inflate_stats.cpp -- MEASUREMENT TOOL (tools/inflate_model.py), not a test and not product code: the inflate kernel of
// goleft_amd/csrc/gd_inflate.hpp under the host emulation with its probe macro counting, per workgroup (= one wave, 64
// members), what the lanes do in every iteration of the kernel's loop -- the numbers a GPU profiler does not give: how many
// of a wave's lane-iterations decode a literal or start a match, copy a chunk, wait for the other lanes' block headers, or
// sit finished while the wave's longest member runs on; how often the header path runs and for how many lanes; how many
// chunk sources come from memory and from the ring.
//   clang++ -O2 -std=c++17 -shared -fPIC -o inflate_stats.so inflate_stats.cpp
#include <cstdint>
#include <cstring>
struct InfStats {
    uint64_t iters;            // loop iterations of the wave
    uint64_t lane_mode[4];     // lane-iterations by mode at the top of the iteration: DECODE, COPY, HDR, DONE
    uint64_t hdr_runs;         // iterations in which the header path ran
    uint64_t hdr_lanes;        // lanes served by those runs
    uint64_t chunk_mem, chunk_ring, win_refill;    // lane-iterations with a chunk loaded from memory / read from the ring / an input slot loaded
    uint64_t mem_iters;        // iterations in which at least one lane loaded a chunk from memory
    // what a per-lane cache of the last aligned block(s) a chunk source was loaded from would catch: chunk loads from memory
    // whose 16 bytes lie inside a block the lane loaded from before -- [block size 64 / 128 / 256][1, 2 or 4 blocks kept]
    uint64_t src_hit[3][3];
    uint64_t mem_iters_after[3][3];   // iterations that would still have a lane loading from memory
    uint64_t dist_le[8];              // chunk loads from memory whose source lies at most 128, 256, 512, 1 K, 2 K, 4 K, 8 K, 32 K bytes back
};
static InfStats g_st;
static uint64_t g_hdr_calls = 0;
static bool g_mem_seen = false;
static unsigned probe_lane();
// (the fibers of a wave run in lane order between two barriers, and every lane stays in the loop until the wave leaves it:
// lane 0's probe at the top of an iteration closes the iteration before)
static uint32_t g_blk[64][3][3][4];                        // [lane][size][ways]: block numbers kept, most recent first
static bool g_miss_seen[3][3];
static inline void inf_probe(int what, uint32_t v)
{
    if (what == 5) {
        if (v == 0) return;
        static const uint32_t lim[8] = {128, 256, 512, 1024, 2048, 4096, 8192, 32768};
        for (int k = 0; k < 8; ++k) if (v <= lim[k]) g_st.dist_le[k]++;
        return;
    }
    if (what == 4) {
        if (v == 0xffffffffu) return;
        const unsigned lane = probe_lane();
        for (int z = 0; z < 3; ++z)
            for (int w = 0; w < 3; ++w) {
                const uint32_t bs = 64u << z, ways = 1u << w;
                const uint32_t b0 = v / bs, b1 = (v + 15u) / bs;          // the chunk's 16 bytes touch one or two blocks
                uint32_t* keep = g_blk[lane][z][w];
                auto has = [&](uint32_t b) { for (uint32_t k = 0; k < ways; ++k) if (keep[k] == b) return true; return false; };
                auto put = [&](uint32_t b) { if (has(b)) return; for (uint32_t k = ways - 1; k > 0; --k) keep[k] = keep[k - 1]; keep[0] = b; };
                if (has(b0) && has(b1)) g_st.src_hit[z][w]++;
                else g_miss_seen[z][w] = true;
                put(b0); put(b1);
            }
        return;
    }
    if (what == 0) {
        if (probe_lane() == 0) {
            g_st.iters++;
            if (g_mem_seen) g_st.mem_iters++;
            g_mem_seen = false;
            for (int z = 0; z < 3; ++z)
                for (int w = 0; w < 3; ++w) { if (g_miss_seen[z][w]) g_st.mem_iters_after[z][w]++; g_miss_seen[z][w] = false; }
        }
        g_st.lane_mode[v & 3u]++;
    } else if (what == 1) {
        g_hdr_calls++;
        g_st.hdr_lanes += v ? 1 : 0;
    } else if (what == 2) {
        if (v == 1) { g_st.chunk_mem++; g_mem_seen = true; }
        else if (v == 2) g_st.chunk_ring++;
    } else if (what == 3) {
        if (v) g_st.win_refill++;
    }
}
#define GD_INFLATE_PROBE(what, value) inf_probe(what, (uint32_t)(value))
#include "emul_machine.hpp"
static unsigned probe_lane() { return threadIdx.x; }
#include "../../goleft_amd/csrc/gd_inflate.hpp"

static const gd::InflateJob* g_job = nullptr;
static void body_inflate() { gd::gd_inflate_kernel(*g_job); }

// one workgroup per call: members [first, first + 64) of the job
extern "C" int emul_inflate_stats(const uint8_t* comp, const uint64_t* in_off, const uint32_t* in_len, const uint64_t* out_off,
                                  const uint32_t* out_len, uint8_t* out, uint32_t* status, uint32_t n, uint32_t block, InfStats* st)
{
    gd::InflateJob job{};
    job.comp = comp; job.in_off = in_off; job.in_len = in_len; job.out_off = out_off; job.out_len = out_len;
    job.crc = nullptr; job.out = out; job.status = status; job.n = n;
    g_job = &job;
    g_st = InfStats{};
    g_hdr_calls = 0;
    g_mem_seen = false;
    memset(g_blk, 0xff, sizeof g_blk);
    memset(g_miss_seen, 0, sizeof g_miss_seen);
    emul::run(body_inflate, gd::INF_LANES, block);
    if (g_mem_seen) g_st.mem_iters++;
    g_st.hdr_runs = g_hdr_calls / gd::INF_LANES;
    *st = g_st;
    return 0;
}
