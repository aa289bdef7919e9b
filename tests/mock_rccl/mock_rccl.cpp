// Stand-in for librccl in tests (JTP_RCCL_LIB): the eight entry points libjtprop.so binds, moving the
// bytes through files in /dev/shm instead of xGMI, so that several processes sharing ONE GPU can run a
// multi-rank plan end to end - real kernels, real exchange schedule, real message arena - on a box that
// has no second GPU.  Sends and receives of a group are carried out at ncclGroupEnd, on the host, after
// synchronising the stream: correct data movement, no claim about timing.  Test infrastructure only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

struct MockComm {
    int rank = 0, size = 1;
    std::string dir;
    std::map<std::pair<int, int>, long> seq;      // (src, dst) -> messages so far
};
struct Op { int send; void *buf; size_t bytes; int peer; MockComm *comm; hipStream_t stream; };
static thread_local int depth = 0;
static thread_local std::vector<Op> queue;

static int run(const Op &op) {
    MockComm *c = op.comm;
    char path[512], tmp[512];
    if (op.send) {
        if (hipStreamSynchronize(op.stream) != hipSuccess) return 1;
        std::vector<char> host(op.bytes);
        if (hipMemcpy(host.data(), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        long s = c->seq[{c->rank, op.peer}]++;
        snprintf(path, sizeof path, "%s/%d_%d_%ld", c->dir.c_str(), c->rank, op.peer, s);
        snprintf(tmp, sizeof tmp, "%s.tmp", path);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(host.data(), 1, op.bytes, f) != op.bytes) return 1;
        fclose(f);
        if (rename(tmp, path) != 0) return 1;
        return 0;
    }
    long s = c->seq[{op.peer, c->rank}]++;
    snprintf(path, sizeof path, "%s/%d_%d_%ld", c->dir.c_str(), op.peer, c->rank, s);
    std::vector<char> host(op.bytes);
    for (int tries = 0; tries < 60000; ++tries) {          // up to 60 s
        FILE *f = fopen(path, "rb");
        if (f) {
            size_t got = fread(host.data(), 1, op.bytes, f);
            fclose(f);
            if (got != op.bytes) return 1;
            unlink(path);
            if (hipStreamSynchronize(op.stream) != hipSuccess) return 1;
            return hipMemcpy(op.buf, host.data(), op.bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
        }
        usleep(1000);
    }
    return 1;
}

static int flush() {
    int rc = 0;
    for (const Op &op : queue) if (op.send) rc |= run(op);     // all sends first: no cyclic waits
    for (const Op &op : queue) if (!op.send) rc |= run(op);
    queue.clear();
    return rc;
}

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
    *comm = c;
    return 0;
}
int ncclCommDestroy(MockComm *c) {
    if (c && c->rank == 0) {
        std::string cmd = "rm -rf '" + c->dir + "'";
        if (system(cmd.c_str()) != 0) { /* leftovers in /dev/shm are harmless */ }
    }
    delete c;
    return 0;
}
int ncclGroupStart(void) { ++depth; return 0; }
int ncclGroupEnd(void) { return --depth == 0 ? flush() : 0; }
int ncclSend(const void *buf, size_t count, int dtype, int peer, MockComm *c, hipStream_t s) {
    if (dtype != 8) return 1;                              // ncclFloat64 is all libjtprop sends
    queue.push_back({1, const_cast<void *>(buf), count * 8, peer, c, s});
    return depth == 0 ? flush() : 0;
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, MockComm *c, hipStream_t s) {
    if (dtype != 8) return 1;
    queue.push_back({0, buf, count * 8, peer, c, s});
    return depth == 0 ? flush() : 0;
}
const char *ncclGetErrorString(int) { return "mock rccl: transfer failed"; }
}
