// rccl_stub.cpp -- TEST INFRASTRUCTURE (tests/test_gpu_gather_world.py): the eight RCCL entry points the library's native gather
// binds (goleft_amd/csrc/gd_api_comm.inc), between PROCESSES that share ONE GPU, over POSIX shared memory.
//
// RCCL itself refuses two ranks on one device, and the development boxes have one -- so the grouped ncclSend / ncclRecv of
// gd_gather_export (the offsets of the root's receives, the 128-byte ncclUniqueId passed BY VALUE, the order between a
// compute, its gather and the next compute, two alternating send buffers) had never run with more than one rank before the
// driver's 8-GPU run.  This stand-in gives those calls a world: a rank's send goes device -> a mailbox in shared memory
// -> the root's receive buffer.  It is synchronous where RCCL is asynchronous (ncclGroupEnd drains the stream, copies, and
// waits for its peers), which is within what the ABI promises a caller (gd_gather_wait may return at once).
// Nothing of this is linked into the product; the library loads it only when GOLEFT_RCCL_LIB names it.
//   hipcc -O2 -shared -fPIC -o librccl_stub.so rccl_stub.cpp -lrt
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
constexpr size_t kBox = 24u << 20;                         // bytes a mailbox holds (one send)
constexpr int kMaxRanks = 8;
struct Box { std::atomic<uint64_t> written, taken; uint64_t bytes; uint64_t pad[5]; };
struct Shm {
    std::atomic<uint32_t> arrived;                         // ncclCommInitRank is collective
    uint32_t nranks;
    Box box[kMaxRanks][kMaxRanks];                          // [to][from]
    // the mailboxes' bytes follow: kBox each, [to][from]
};
struct Comm { int rank, nranks; Shm* shm; uint8_t* data; size_t map_bytes; char name[64]; };
struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local std::vector<Op> g_ops;
thread_local int g_depth = 0;

double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
template <class F> bool wait_for(F f, double seconds = 60.0)
{
    const double t0 = now();
    while (!f()) { if (now() - t0 > seconds) return false; usleep(50); }
    return true;
}
uint8_t* box_bytes(Comm* c, int to, int from) { return c->data + ((size_t)to * kMaxRanks + (size_t)from) * kBox; }

int run(std::vector<Op>& ops)
{
    // every send of the group first (a mailbox takes one message: the receiver of the previous step has emptied it), then
    // the receives -- the root sends to itself inside the same group
    for (const Op& o : ops) {
        if (!o.send) continue;
        if (o.bytes > kBox) return 4;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;       // what the stream computed is what is sent
        Box& b = o.comm->shm->box[o.peer][o.comm->rank];
        if (!wait_for([&] { return b.taken.load() == b.written.load(); })) return 6;
        if (hipMemcpy(box_bytes(o.comm, o.peer, o.comm->rank), o.buf, o.bytes, hipMemcpyDefault) != hipSuccess) return 1;
        b.bytes = o.bytes;
        b.written.fetch_add(1);
    }
    for (const Op& o : ops) {
        if (o.send) continue;
        Box& b = o.comm->shm->box[o.comm->rank][o.peer];
        if (!wait_for([&] { return b.written.load() > b.taken.load(); })) return 6;
        if (b.bytes != o.bytes) return 4;                                  // a receive must match its send
        if (hipMemcpy(o.buf, box_bytes(o.comm, o.comm->rank, o.peer), o.bytes, hipMemcpyDefault) != hipSuccess) return 1;
        b.taken.fetch_add(1);
    }
    return 0;
}
}  // namespace

extern "C" {
struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/gdstub_%d_%08x", (int)getpid(), (unsigned)(now() * 1e6));
    return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank)     // the id BY VALUE, as rccl.h declares it
{
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks || id.internal[0] != '/') return 4;
    const size_t bytes = sizeof(Shm) + (size_t)kMaxRanks * kMaxRanks * kBox;
    int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 2;
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return 2; }          // (sparse: only the mailboxes in use are touched)
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    Comm* c = new Comm{rank, nranks, static_cast<Shm*>(p), static_cast<uint8_t*>(p) + sizeof(Shm), bytes, {0}};
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->shm->nranks = (uint32_t)nranks;
    c->shm->arrived.fetch_add(1);
    if (!wait_for([&] { return c->shm->arrived.load() >= (uint32_t)nranks; })) { munmap(p, bytes); delete c; return 6; }
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm)
{
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return 4;
    if (c->rank == 0) shm_unlink(c->name);
    munmap(c->shm, c->map_bytes);
    delete c;
    return 0;
}

int ncclGroupStart() { ++g_depth; return 0; }
int ncclGroupEnd()
{
    if (g_depth <= 0) return 4;
    if (--g_depth) return 0;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run(ops);
}
static int size_of(int dtype) { return dtype == 4 || dtype == 5 || dtype == 8 ? 8 : dtype == 2 || dtype == 3 || dtype == 7 ? 4 : dtype == 6 || dtype == 9 ? 2 : 1; }
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream)
{
    Op o{true, const_cast<void*>(buf), count * (size_t)size_of(dtype), peer, static_cast<Comm*>(comm), stream};
    if (!o.comm || peer < 0 || peer >= o.comm->nranks) return 4;
    g_ops.push_back(o);
    if (g_depth == 0) { std::vector<Op> ops; ops.swap(g_ops); return run(ops); }
    return 0;
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream)
{
    Op o{false, buf, count * (size_t)size_of(dtype), peer, static_cast<Comm*>(comm), stream};
    if (!o.comm || peer < 0 || peer >= o.comm->nranks) return 4;
    g_ops.push_back(o);
    if (g_depth == 0) { std::vector<Op> ops; ops.swap(g_ops); return run(ops); }
    return 0;
}
const char* ncclGetErrorString(int e)
{
    switch (e) {
    case 0: return "no error";
    case 1: return "stub: a HIP call failed";
    case 2: return "stub: shared memory";
    case 4: return "stub: invalid argument (or a message larger than a mailbox, or a receive that does not match its send)";
    case 6: return "stub: a peer did not arrive within a minute";
    default: return "stub: error";
    }
}
}  // extern "C"
