// Stand-in for librccl in tests (JTP_RCCL_LIB): the eight entry points libjtprop.so binds, for several processes
// that share ONE GPU (RCCL refuses two ranks on one device, and the test box has a single GPU): real kernels, real
// exchange schedule, real message arena - only the transport differs.  Test infrastructure only.
//
// Round 3: the transport is STREAM ORDERED and asynchronous like the real one - nothing here synchronises a stream or
// blocks the host.  Every ordered pair of ranks (src, dst) has a mailbox in /dev/shm, mapped by both processes and
// registered with HIP (hipHostRegisterMapped): a ring of slots, each a payload area plus a FULL word and an ACK word.
//   ncclSend of message s (slot = s % SLOTS, round = s / SLOTS + 1), enqueued on the caller's stream:
//       kernel: wait until ACK[slot] >= round - 1 (the receiver has emptied the slot)  ->  hipMemcpyAsync device -> payload
//       ->  kernel: FULL[slot] = round   (system scope)
//   ncclRecv: kernel: wait until FULL[slot] >= round  ->  hipMemcpyAsync payload -> device  ->  kernel: ACK[slot] = round
// so a message moves only after the kernels that produce it (earlier on the sender's stream), and the kernels that
// consume it (later on the receiver's stream) start only after it has arrived: an exchange the engine forgot to order
// against its kernels would read or overwrite the wrong bytes here exactly as it would over RCCL.  A group's sends
// are enqueued before its receives (ncclGroupEnd), as RCCL progresses both sides of a group together.  Every wait is
// bounded (20 s of the 100 MHz clock): a wait that gives up sets the communicator's error word, which every later
// call returns as a failure - a lost message fails the test, it does not hang the GPU.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

#define SLOTS 8
#define SLOT_BYTES (1u << 20)        // largest message: 64 partial copies of a 2^11-entry separator

struct Box {
    uint32_t full[SLOTS];
    uint32_t ack[SLOTS];
    uint32_t pad[48];
    char payload[SLOTS][SLOT_BYTES];
};

struct Mapped {
    Box *host = nullptr;
    Box *dev = nullptr;
    long seq = 0;                    // messages issued so far on this side
};

struct MockComm {
    int rank = 0, size = 1;
    std::string dir;
    std::map<std::pair<int, int>, Mapped> box;      // (src, dst)
    uint32_t *err = nullptr;                        // pinned: set by a wait that gave up
};
struct Op { int send; void *buf; size_t bytes; int peer; MockComm *comm; hipStream_t stream; };
static thread_local int depth = 0;
static thread_local std::vector<Op> queue;

__global__ void mock_wait_ge(const uint32_t *word, uint32_t want, uint32_t *err) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        __builtin_amdgcn_s_sleep(64);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000000ull) {        // 20 s
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
}

__global__ void mock_signal(uint32_t *word, uint32_t value) {
    __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static Mapped *mailbox(MockComm *c, int src, int dst) {
    auto it = c->box.find({src, dst});
    if (it != c->box.end()) return &it->second;
    char path[512];
    snprintf(path, sizeof path, "%s/box_%d_%d", c->dir.c_str(), src, dst);
    int fd = open(path, O_RDWR | O_CREAT, 0600);
    if (fd < 0) return nullptr;
    if (ftruncate(fd, sizeof(Box)) != 0) { close(fd); return nullptr; }       // (same size from both sides: contents stay)
    void *p = mmap(nullptr, sizeof(Box), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return nullptr;
    if (hipHostRegister(p, sizeof(Box), hipHostRegisterMapped) != hipSuccess) return nullptr;
    Mapped m;
    m.host = (Box *)p;
    if (hipHostGetDevicePointer((void **)&m.dev, p, 0) != hipSuccess) return nullptr;
    return &(c->box[{src, dst}] = m);
}

static int run(const Op &op) {
    MockComm *c = op.comm;
    if (op.bytes > SLOT_BYTES) return 1;
    if (op.send) {
        Mapped *m = mailbox(c, c->rank, op.peer);
        if (!m) return 1;
        const long s = m->seq++;
        const int slot = (int)(s % SLOTS);
        const uint32_t round = (uint32_t)(s / SLOTS) + 1u;
        hipLaunchKernelGGL(mock_wait_ge, dim3(1), dim3(1), 0, op.stream, (const uint32_t *)&m->dev->ack[slot], round - 1u, c->err);
        if (hipMemcpyAsync(m->host->payload[slot], op.buf, op.bytes, hipMemcpyDeviceToHost, op.stream) != hipSuccess) return 1;
        hipLaunchKernelGGL(mock_signal, dim3(1), dim3(1), 0, op.stream, &m->dev->full[slot], round);
    } else {
        Mapped *m = mailbox(c, op.peer, c->rank);
        if (!m) return 1;
        const long s = m->seq++;
        const int slot = (int)(s % SLOTS);
        const uint32_t round = (uint32_t)(s / SLOTS) + 1u;
        hipLaunchKernelGGL(mock_wait_ge, dim3(1), dim3(1), 0, op.stream, (const uint32_t *)&m->dev->full[slot], round, c->err);
        if (hipMemcpyAsync(op.buf, m->host->payload[slot], op.bytes, hipMemcpyHostToDevice, op.stream) != hipSuccess) return 1;
        hipLaunchKernelGGL(mock_signal, dim3(1), dim3(1), 0, op.stream, &m->dev->ack[slot], round);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

static int flush() {
    int rc = 0;
    for (const Op &op : queue) if (op.send) rc |= run(op);     // all sends first: no cyclic waits
    for (const Op &op : queue) if (!op.send) rc |= run(op);
    queue.clear();
    return rc;
}

static int failed(MockComm *c) { return c && c->err && *(volatile uint32_t *)c->err != 0; }

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
int ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "jtmock_%d_%ld", (int)getpid(), (long)time(nullptr));
    return 0;
}
int ncclCommInitRank(MockComm **comm, int n, ncclUniqueId id, int rank) {
    MockComm *c = new MockComm();
    c->rank = rank;
    c->size = n;
    c->dir = std::string("/dev/shm/") + id.internal;
    mkdir(c->dir.c_str(), 0700);
    if (hipHostMalloc((void **)&c->err, 64, hipHostMallocMapped) != hipSuccess) return 1;
    *c->err = 0;
    for (int peer = 0; peer < n; ++peer)                  // every mailbox this rank is a party to, mapped up front
        if (peer != rank && (!mailbox(c, rank, peer) || !mailbox(c, peer, rank))) return 1;
    *comm = c;
    return 0;
}
int ncclCommDestroy(MockComm *c) {
    if (!c) return 0;
    (void)hipDeviceSynchronize();
    const int bad = failed(c);
    if (bad) fprintf(stderr, "mock rccl: rank %d: a wait for a message gave up\n", c->rank);
    for (auto &kv : c->box) {
        (void)hipHostUnregister(kv.second.host);
        munmap(kv.second.host, sizeof(Box));
    }
    if (c->err) (void)hipHostFree(c->err);
    if (c->rank == 0) {
        std::string cmd = "rm -rf '" + c->dir + "'";
        if (system(cmd.c_str()) != 0) { /* leftovers in /dev/shm are harmless */ }
    }
    delete c;
    return bad;
}
int ncclCommCount(const MockComm *c, int *n) { *n = c->size; return 0; }
int ncclCommUserRank(const MockComm *c, int *r) { *r = c->rank; return 0; }
int ncclGroupStart(void) { ++depth; return 0; }
int ncclGroupEnd(void) { return --depth == 0 ? flush() : 0; }
int ncclSend(const void *buf, size_t count, int dtype, int peer, MockComm *c, hipStream_t s) {
    if (dtype != 8 || failed(c)) return 1;                 // ncclFloat64 is all libjtprop sends
    queue.push_back({1, const_cast<void *>(buf), count * 8, peer, c, s});
    return depth == 0 ? flush() : 0;
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, MockComm *c, hipStream_t s) {
    if (dtype != 8 || failed(c)) return 1;
    queue.push_back({0, buf, count * 8, peer, c, s});
    return depth == 0 ? flush() : 0;
}
const char *ncclGetErrorString(int) { return "mock rccl: transfer failed (a wait gave up, or a mailbox could not be mapped)"; }
}
