// C ABI of libjtprop.so (declared in include/jtprop.h): device memory, launch schedule,
// RCCL point-to-point exchange at subtree cuts, host<->device layout conversion.
// Plain HIP runtime + RCCL; no PyTorch, no Triton.
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "jtp_kernels.hip.h"
#include "jtp_plan.h"

// ------------------------------------------------------------------------------------------ errors

static thread_local std::string g_err;

static int set_err(int code, const char *fmt, ...) {
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return set_err(JTP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------------------------------ RCCL (lazy)

namespace rccl {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclFloat64 = 8 };
typedef int (*GetUniqueId_t)(ncclUniqueId *);
typedef int (*CommInitRank_t)(ncclComm_t *, int, ncclUniqueId, int);
typedef int (*CommDestroy_t)(ncclComm_t);
typedef int (*Send_t)(const void *, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*Recv_t)(void *, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*Group_t)(void);
typedef const char *(*ErrStr_t)(int);
typedef int (*CommQuery_t)(const ncclComm_t, int *);

static void *lib = nullptr;
static GetUniqueId_t GetUniqueId;
static CommInitRank_t CommInitRank;
static CommDestroy_t CommDestroy;
static Send_t Send;
static Recv_t Recv;
static Group_t GroupStart, GroupEnd;
static ErrStr_t GetErrorString;
static CommQuery_t CommCount = nullptr, CommUserRank = nullptr, CommCuDevice = nullptr;      // (optional: what the communicator itself says)
static ncclComm_t comm = nullptr;
static int comm_rank = 0, comm_size = 1;

static int load() {
    if (lib) return JTP_OK;
    // JTP_RCCL_LIB: another library with the same eight entry points (tests/mock_rccl: several processes on
    // one GPU exchanging through /dev/shm, to exercise the multi-rank path where there is no second GPU)
    const char *names[] = {getenv("JTP_RCCL_LIB") ? getenv("JTP_RCCL_LIB") : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    if (!lib) return set_err(JTP_ECOMM, "cannot load librccl: %s", dlerror());
#define SYM(var, name)                                                            \
    var = (decltype(var))dlsym(lib, name);                                        \
    if (!var) return set_err(JTP_ECOMM, "librccl lacks symbol %s", name);
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    CommCount = (CommQuery_t)dlsym(lib, "ncclCommCount");
    CommUserRank = (CommQuery_t)dlsym(lib, "ncclCommUserRank");
    CommCuDevice = (CommQuery_t)dlsym(lib, "ncclCommCuDevice");
    return JTP_OK;
}
}  // namespace rccl

// ------------------------------------------------------------------------------------------ roctx ranges (lazy, optional)
// phases show up as named ranges in rocprofv3 --marker-trace.  The library is looked up at the
// first propagate of a plan created with JTP_ROCTX=1 in the environment; without it (or without the library) the
// calls are no-ops.
namespace roctx {
typedef int (*Push_t)(const char *);
typedef int (*Pop_t)(void);
static Push_t Push = nullptr;
static Pop_t Pop = nullptr;
static int state = 0;                   // 0 not looked up, 1 available, -1 absent
static void load() {
    if (state != 0) return;
    state = -1;
    for (const char *n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
        void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!h) continue;
        Push = (Push_t)dlsym(h, "roctxRangePushA");
        Pop = (Pop_t)dlsym(h, "roctxRangePop");
        if (Push && Pop) {
            state = 1;
            return;
        }
    }
}
struct Range {
    bool on;
    Range(bool enabled, const char *name) : on(enabled && state == 1) { if (on) Push(name); }
    ~Range() { if (on) Pop(); }
};
}  // namespace roctx

#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        int _r = (expr);                                                                        \
        if (_r != rccl::ncclSuccess)                                                            \
            return set_err(JTP_ECOMM, "%s failed: %s", #expr, rccl::GetErrorString(_r));         \
    } while (0)

// ------------------------------------------------------------------------------------------ plan object


static inline int mixk(const HostPlan &hp) { return hp.tmix ? (hp.tmix_compact ? 2 : 1) : 0; }

static bool flow_both() {
    static const bool on = !(getenv("JTP_FLOW_BOTH") && atoi(getenv("JTP_FLOW_BOTH")) == 0);
    return on;
}

template <typename T>
struct KernelTable {
    typedef void (*fn)(const JtTask *, const JtBlock *, const int *, const T *, T *, double *, JtFlow);
    // (tmix: 0 no mixed-radix rows, 1 one row per step, 2 the compact form - two rows per step, HostPlan::tmix_compact: mixk())
    static fn get(int variant, int tmix) {
        if (tmix == 2) {
            if (variant >= JT_K_COLLECT0 && variant <= JT_K_COLLECT3) return jt_collect_level_mix<T, true>;
            if (variant >= JT_K_DIST_P0C0 && variant <= JT_K_DIST_P1C3) return jt_distribute_level_mix<T, true>;
            if (variant == JT_K_COLLECT_LEVEL) return jt_collect_level_mix<T, true>;
            if (variant == JT_K_DISTRIBUTE_LEVEL) return jt_distribute_level_mix<T, true>;
            if (variant == JT_K_SINGLE || variant == JT_K_MARGINALS) return jt_single_mix<T, true>;
        }
        if (tmix) {                 // plans with a mixed-radix thread part: one kernel per launch style (they dispatch on the task)
            if (variant >= JT_K_COLLECT0 && variant <= JT_K_COLLECT3) return jt_collect_level_mix<T, false>;
            if (variant >= JT_K_DIST_P0C0 && variant <= JT_K_DIST_P1C3) return jt_distribute_level_mix<T, false>;
            if (variant == JT_K_COLLECT_LEVEL) return jt_collect_level_mix<T, false>;
            if (variant == JT_K_DISTRIBUTE_LEVEL) return jt_distribute_level_mix<T, false>;
            if (variant == JT_K_SINGLE || variant == JT_K_MARGINALS) return jt_single_mix<T, false>;
        }
        switch (variant) {
            case JT_K_COLLECT0: return jt_collect<T, 0>;
            case JT_K_COLLECT1: return jt_collect<T, 1>;
            case JT_K_COLLECT2: return jt_collect<T, 2>;
            case JT_K_COLLECT3: return jt_collect<T, 3>;
            case JT_K_DIST_P0C0: return jt_distribute<T, 0, 0>;
            case JT_K_DIST_P0C1: return jt_distribute<T, 0, 1>;
            case JT_K_DIST_P0C2: return jt_distribute<T, 0, 2>;
            case JT_K_DIST_P0C3: return jt_distribute<T, 0, 3>;
            case JT_K_DIST_P1C0: return jt_distribute<T, 1, 0>;
            case JT_K_DIST_P1C1: return jt_distribute<T, 1, 1>;
            case JT_K_DIST_P1C2: return jt_distribute<T, 1, 2>;
            case JT_K_DIST_P1C3: return jt_distribute<T, 1, 3>;
            case JT_K_COLLECT_LEVEL: return jt_collect_level<T>;
            case JT_K_DISTRIBUTE_LEVEL: return jt_distribute_level<T>;
            case JT_K_REDUCE_LEVEL: return jt_reduce_level<T>;
            case JT_K_MULTI_COLLECT: return jt_multi_flow<T>;
            case JT_K_MULTI_DISTRIBUTE: return jt_multi_flow<T>;
            case JT_K_SINGLE: return jt_single<T>;
            case JT_K_MARGINALS: return jt_marginals<T>;
            case JT_K_LEAN_SINGLE: return jt_lean_single<T>;
        }
        return nullptr;
    }
    static fn get_flow(int phase, bool chain, int tmix, bool marg) {
        if (tmix == 2) return phase == 0 ? jt_collect_flow_mix<T, true> : jt_distribute_flow_mix<T, true>;
        if (tmix) return phase == 0 ? jt_collect_flow_mix<T, false> : jt_distribute_flow_mix<T, false>;      // (never merged: jtp_plan.cpp finish())
        // (marg: the plan has marginal tasks folded into its distribute phase - the build of the kernel that can run them)
        if (phase == 2) return marg ? jt_propagate_flow_marg<T> : jt_propagate_flow<T>;          // both phases in one launch
        // The kernel that runs both phases dispatches on the task's mode, so it serves a distribute segment alone as well - and its
        // build of the distribute pass is the faster one (round 5, A/B by environment on one box: config 3 in two launches 8.35 -> 8.13 ms,
        // the whole gain of "one launch"; a rank's share of config 4 at 8 ranks 178 -> 176 us).  JTP_FLOW_BOTH=0: jt_distribute_flow as before.
        if (phase == 1 && !chain && flow_both()) return marg ? jt_propagate_flow_marg<T> : jt_propagate_flow<T>;
        return phase == 0 ? jt_collect_flow<T> : (chain ? jt_distribute_flow_chain<T> : jt_distribute_flow<T>);
    }
};

static const char *k_names[JT_K_COUNT] = {
    "jt_collect<T, 0>", "jt_collect<T, 1>", "jt_collect<T, 2>", "jt_collect<T, 3>",
    "jt_distribute<T, 0, 0>", "jt_distribute<T, 0, 1>", "jt_distribute<T, 0, 2>", "jt_distribute<T, 0, 3>",
    "jt_distribute<T, 1, 0>", "jt_distribute<T, 1, 1>", "jt_distribute<T, 1, 2>", "jt_distribute<T, 1, 3>",
    "jt_collect_level<T>", "jt_distribute_level<T>", "jt_collect_flow<T>", "jt_distribute_flow<T>", "jt_reduce_level<T>",
    "jt_multi_flow<T>", "jt_multi_flow<T>", "jt_single<T>", "jt_propagate_flow<T>", "jt_marginals<T>", "jt_lean_single<T>",
};

struct BatchBuffers {
    void *psi = nullptr;
    void *bel = nullptr;
    double *msg = nullptr;
    double *fix = nullptr;          // fixed arena: the static tables of unit cliques (HostPlan::statics; shared like psi)
    uint32_t *ev = nullptr;         // hard evidence: (mask, value) per planner node, or null (jtp_set_evidence)
    bool ev_any = false;            // ... and it observes something: the kernels get a null table otherwise (single-set plans: the lean
                                    // unit pass takes that for "no evidence anywhere", jt_unit_collect)
    uint32_t *sync = nullptr;       // dataflow launches: abort flag and ticket counters
    uint32_t epoch = 0;             // propagates enqueued so far; its parity selects the message arena half
    uint32_t flow_runs = 0;         // of which dataflow
    uint32_t ticket_runs = 0;       // of which in ticket order: the segments' ticket counters only grow, by one launch's workgroups
                                    // per such run (NOT per dataflow run: a plan changes between blockIdx and ticket order as other
                                    // plans come and go)
    bool unchecked = false;         // a dataflow propagate was enqueued and its abort flag not looked at yet
    int64_t cur_off(int64_t half) const { return (epoch & 1u) ? half : 0; }     // half in use by the last propagate
    // JtFlow::fix_shift of a launch that reads this propagate's half: fixed arena - (message arena + cur_off), in doubles
    int64_t fix_shift(int64_t cur) const { return fix ? (int64_t)(((intptr_t)fix - (intptr_t)msg) / 8) - cur : 0; }
};

// device tables of one list of marginal requests (jtp_get_marginals), kept for the next call
struct MargBatch {
    int lean_nblocks = 0, lean_lds = 0;  // the first workgroups of the unit list have a lean record (jt_lean_single)
    // the list is the one the plan was made with (jtp_tree_desc.fold_*) and every request on a clique without a table was folded into
    // the propagate: `d_descs_fold` says where the propagate left them; the unit launches are then skipped (jtp_get_marginals)
    bool folded = false;
    JtMargDesc *d_descs_fold = nullptr;
    std::vector<JtTask> h_tasks;         // multi-set plans with active lists: the records as planned (readout_redirect patches copies of them)
    std::vector<int32_t> key;            // n, cliques, var_off, var_ids
    JtTask *d_tasks = nullptr;
    JtBlock *d_blocks = nullptr;
    int *d_itab = nullptr;
    JtMargDesc *d_descs = nullptr;
    double *scratch = nullptr, *stage = nullptr;
    int n = 0, nblocks = 0, lds = 0, max_grid_x = 1;
    // requests on UNIT cliques (no belief table): psi x every incoming table marginalised directly (kernel jt_single); their
    // workgroup records follow the others' in d_blocks
    int unit_nblocks = 0, unit_lds = 0;
    int64_t total_out = 0;
    std::vector<int64_t> elems;          // host entries of each request
    void release() {
        if (d_tasks) (void)hipFree(d_tasks);
        if (d_blocks) (void)hipFree(d_blocks);
        if (d_itab) (void)hipFree(d_itab);
        if (d_descs) (void)hipFree(d_descs);
        if (d_descs_fold) (void)hipFree(d_descs_fold);
        if (scratch) (void)hipFree(scratch);
        if (stage) (void)hipFree(stage);
    }
};

// Plans with a dataflow propagate enqueued and not yet synchronised, per device: dataflow kernels of two plans running
// at once need ticket order (see jtp_propagate).  A plan that merely EXISTS costs the others nothing (round 2 counted
// live plans: a library user with two junction trees paid the ticket round trip - +10 % on config 4 - on every propagate).
static std::atomic<int> g_inflight[64];

// dynamic LDS above 64 KiB must be allowed per kernel function: remember what each function was raised to
// (per device: the attribute belongs to the function on the CURRENT device; under a lock: plans may be created from
//  several host threads)
static std::map<std::pair<int, const void *>, int> g_lds_raised;
static std::mutex g_lds_mutex;
static hipError_t raise_lds(const void *func, int bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_lds_mutex);
    int &have = g_lds_raised[std::make_pair(dev, func)];
    if (have >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

struct jtp_plan {
    HostPlan hp;
    bool device = false;
    bool widened = false;           // asked for float32 tables, made with float64 ones (jtp_plan_create)
    bool inflight = false;          // counted in g_inflight: a dataflow propagate of this plan may still be running
    int launch_mode = 0;            // of the last propagate: 0 one launch per level, 1 dataflow in blockIdx order, 2 dataflow, ticket order
    int tickets_used = 0;           // propagates (per evidence set) that ran in ticket order
    int foreign_seen = 0;           // propagates that found ANOTHER PROCESS with a dataflow propagate in flight on the device
    double device_bytes = 0;        // device memory allocated at plan creation (arenas, message arenas, tables)
    int flow_propagates = 0;        // propagates (per evidence set) that ran as dataflow launches
    uint32_t flow_debug = 0;        // JTP_FLOW_DEBUG at plan creation, or jtp_debug_set(plan, "flow_debug", v)
    bool env_tickets = false;       // JTP_FLOW_TICKETS at plan creation
    bool roctx = false;             // JTP_ROCTX at plan creation: named ranges around the phases of a propagate
    // multi-set plans (JTP_MULTISET): evidence sets in groups of JT_MSETS, one allocation each for all sets'
    // message arenas, evidence tables and sync areas (bufs[b] point into them; bufs[b].psi/.bel are shared)
    bool multiset = false;
    int n_groups = 0;
    double *msg_all = nullptr;
    uint32_t *ev_all = nullptr, *sync_all = nullptr;
    int64_t set_stride = 0;         // doubles between consecutive sets' arenas (both halves)
    uint32_t ev_stride = 0;         // uint32 per set's evidence table
    // read-out of multi-set plans: belief task of each clique, built on first use
    std::vector<uint32_t> ev_host;  // host copy of ev_all (which tasks may sum their elements first depends on it)
    // evidence-free subtrees: the first JT_MSETS arena slots are not the caller's (the caller's set b is slot set0 + b); slot 0 runs
    // every collect task without evidence, and a set takes from it the upward message of every clique below which it observes nothing
    int set0 = 0;
    // Round 6: per TASK, not per group - the active list of a collect task holds the arena slots of the sets that observe something below
    // its clique (slot 0, the evidence-free set, first); the list of a downward task every caller's slot (rebuild_active).
    std::vector<uint8_t> member_host;     // [task * cap + slot] != 0: the slot is on the task's list
    std::vector<uint16_t> act_ids_host;   // [task * cap + j]
    std::vector<int32_t> act_n_host;      // [task]
    std::vector<uint8_t> esum_oct_host;   // [task * n_groups + g]: entries 8 g .. 8 g + 7 of the list observe nothing on the clique's element bits
    uint8_t *d_member = nullptr, *d_esum_oct = nullptr;
    uint16_t *d_act_ids = nullptr;
    int32_t *d_act_n = nullptr;
    bool act_dirty = false;
    JtFanout *d_fanout = nullptr;
    int n_fanout = 0, cap_fanout = 0;
    struct BeliefTask { JtTask *d_task = nullptr; JtBlock *d_blk = nullptr; int *d_tab = nullptr; int nblocks = 0, lds = 0; JtTask h_task; };
    std::vector<BeliefTask> belief_tasks;
    std::vector<hipStream_t> streams;
    std::vector<BatchBuffers> bufs;
    JtTask *d_tasks = nullptr;
    JtBlock *d_blocks = nullptr;
    JtBlock *d_init[2] = {nullptr, nullptr};      // HostPlan::init_blocks on the device (mixed-radix plans)
    int *d_itab = nullptr;
    void *stage = nullptr;          // device staging buffer for host<->device conversion
    size_t stage_bytes = 0;
    // uploads (jtp_set_potential): two device staging buffers used in turn, an event each - a call waits only for
    // the pack kernel that last read ITS buffer (two calls back), not for the stream
    void *up_stage[2] = {nullptr, nullptr};
    size_t up_bytes[2] = {0, 0};
    hipEvent_t up_ev[2] = {nullptr, nullptr};
    bool up_busy[2] = {false, false};
    unsigned up_cursor = 0;
    hipEvent_t region_ev[2] = {nullptr, nullptr};      // jtp_region_begin / jtp_region_end
    bool region_open = false;
    int prof_steps = 0;             // 0: off; else ring of this many event sets
    std::vector<hipEvent_t> ev;     // prof_steps x (2 per launch)
    int prof_cursor = 0;            // propagates recorded since profiling was switched on
    int prof_stride = 1;            // every how many propagates one is timed (jtp_set_profiling_stride)
    unsigned prof_calls = 0;        // propagates since profiling was switched on, timed or not
    bool prof_per_launch = false;   // event pair per launch instead of three per propagate
    bool flow = true;               // dataflow launches (one per phase) instead of one per level
    bool chain = false;             // the plan is made of latency-bound levels (JtTask::settle): distribute runs the build without spills
    bool marg_tasks = false;        // some marginal request was folded into the propagate (HostPlan::folded): jt_propagate_flow_marg
    uint32_t *host_abort = nullptr; // pinned: set by a workgroup that gave up waiting
    int flow_fallbacks = 0;         // times that happened (then: one launch per level from there on)
    int fake_comm = 0;              // JTP_FAKE_COMM: 1 = what a rank would receive is filled with ones, what it would send goes nowhere;
                                    // 2 = the exchange steps run as REAL RCCL groups in loop-back (every ncclSend / ncclRecv of the step
                                    // addressed to this rank itself, on the plan's stream, between the launches as in a sharded run)
    bool esum_dirty = false;        // multi-set plans: JtTask::esum_groups changed on the host since the last upload
    bool psi_dirty = false;         // shared potentials were written (on stream 0) since the last propagate
    std::vector<MargBatch *> marg_cache;
    // factor tables and records on their way to jt_eval_batch: slices of one buffer handed out in turn, so that
    // evaluate calls following each other need no synchronisation until the buffer wraps
    char *eval_stage = nullptr;          // device
    char *eval_host = nullptr;           // pinned mirror: the caller's tables are copied here before the call returns
    size_t eval_bytes = 0, eval_cursor = 0;
    void *unit_scratch = nullptr;        // scratch arena in which the belief of a unit clique is formed on demand (jtp_get_belief)
    hipStream_t eval_stream = nullptr;   // stream whose kernels may still read the buffer
    bool eval_pending = false;
    int esize = 4;
};

// ... and the same across PROCESSES (round 4): every process using this library on a device keeps its count of in-flight
// dataflow propagates in a slot of a small shared-memory board, /dev/shm/jtprop_flight_<PCI bus id>; a process that finds
// another LIVE process's count above zero launches in ticket order, as it does for a second plan of its own.  Round 3 left
// that case to an environment variable (JTP_FLOW_TICKETS) and to the 2 s time-out with its fall-back to level launches.
// Processes that do not share /dev/shm (containers) still cannot see each other: for them the time-out stands.
namespace board {
struct Slot { std::atomic<int32_t> pid, count; };
constexpr int SLOTS = 64;
// Trust model: the board is advisory.  It is world-writable (any local user's process on the device must be able to publish), so
// a hostile local user could pin every process to ticket order (10 % slower) or hide itself - never corrupt a result: a process that
// is not seen falls under the 2 s time-out and its fall-back to level launches.  Liveness is `kill(pid, 0)`: processes in different
// PID namespaces that share /dev/shm cannot check each other and treat every published count as live.
struct Board { Slot *slots = nullptr; int mine = -1; bool tried = false; int32_t owner_pid = 0; };
static Board g_board[64];
static std::mutex g_mutex;

static bool alive(int32_t pid) { return pid > 0 && (kill((pid_t)pid, 0) == 0 || errno == EPERM); }

static Board &open_board(int device) {
    Board &b = g_board[device & 63];
    std::lock_guard<std::mutex> lock(g_mutex);
    if (b.tried && b.owner_pid == (int32_t)getpid()) return b;
    if (b.tried) {                                      // a forked child: the parent's mapping is there, its SLOT is not ours
        b.mine = -1;
        b.owner_pid = (int32_t)getpid();
        if (!b.slots) return b;
    } else {
    b.tried = true;
    b.owner_pid = (int32_t)getpid();
    char bus[64] = "unknown";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) return b;
    for (char *c = bus; *c; ++c)
        if (*c == ':' || *c == '.') *c = '_';
    char name[128];
    snprintf(name, sizeof name, "/jtprop_flight_%s", bus);
    // an existing board is opened as it is (O_CREAT on another user's file fails under fs.protected_regular); a new one is made
    // exclusively and opened up with fchmod - the process umask is never touched (other threads may be creating files)
    int fd = shm_open(name, O_RDWR, 0);
    if (fd < 0 && errno == ENOENT) {
        fd = shm_open(name, O_RDWR | O_CREAT | O_EXCL, 0600);
        if (fd >= 0) (void)fchmod(fd, 0666);
        else if (errno == EEXIST) fd = shm_open(name, O_RDWR, 0);          // (somebody else was first)
    }
    if (fd < 0) return b;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || ((size_t)sb.st_size < sizeof(Slot) * SLOTS && ftruncate(fd, sizeof(Slot) * SLOTS) != 0)) { close(fd); return b; }
    void *m = mmap(nullptr, sizeof(Slot) * SLOTS, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return b;
    b.slots = static_cast<Slot *>(m);
    }
    const int32_t me = (int32_t)getpid();
    for (int pass = 0; pass < 2 && b.mine < 0; ++pass)
        for (int i = 0; i < SLOTS && b.mine < 0; ++i) {
            int32_t owner = b.slots[i].pid.load();
            if (owner == me) { b.mine = i; break; }                     // (a forked child inherits nothing useful: it has its own pid)
            if (owner != 0 && (pass == 0 || alive(owner))) continue;    // pass 0: free slots only; pass 1: slots of dead processes too
            if (b.slots[i].pid.compare_exchange_strong(owner, me)) {
                b.slots[i].count.store(0);
                b.mine = i;
            }
        }
    return b;
}

// this process has `n` dataflow propagates in flight on the device; returns whether another live process has any
static bool publish(int device, int n) {
    Board &b = open_board(device);
    if (!b.slots || b.mine < 0) return false;
    b.slots[b.mine].count.store(n);
    bool others = false;
    for (int i = 0; i < SLOTS; ++i) {
        if (i == b.mine || b.slots[i].count.load() <= 0) continue;
        const int32_t owner = b.slots[i].pid.load();
        if (alive(owner)) others = true;
        else {                                             // left behind by a process that died in flight: release the slot FIRST, and
            int32_t expect = owner;                        // clear its count only if that release was ours (a new owner may have published)
            if (owner != 0 && b.slots[i].pid.compare_exchange_strong(expect, 0)) b.slots[i].count.store(0);
        }
    }
    return others;
}
}  // namespace board

// Dataflow launches in blockIdx order are safe only while no OTHER dataflow kernel can be resident on the device at the
// same time (jtp_propagate).  A plan enters the count at its first dataflow propagate and leaves it when the host has
// seen all its streams idle (jtp_sync, a read-out's settle, jtp_plan_destroy).
static bool enter_flight(jtp_plan *pl) {          // returns whether ANOTHER plan - of this process or of another - is in flight on the device
    std::atomic<int> &g = g_inflight[pl->hp.device & 63];
    bool mine = false;
    if (!pl->inflight) {
        pl->inflight = true;
        mine = g.fetch_add(1) > 0;
    } else
        mine = g.load() > 1;
    const bool foreign = board::publish(pl->hp.device, g.load());
    if (foreign) pl->foreign_seen++;
    return mine || foreign;
}
static void leave_flight(jtp_plan *pl) {
    if (!pl->inflight) return;
    for (const auto &b : pl->bufs)
        if (b.unchecked) return;                   // some evidence set's stream has not been waited for yet
    pl->inflight = false;
    const int left = --g_inflight[pl->hp.device & 63];
    (void)board::publish(pl->hp.device, left);
}

static int ensure_stage(jtp_plan *pl, size_t bytes) {
    if (pl->stage_bytes >= bytes) return JTP_OK;
    if (pl->stage) HIP_TRY(hipFree(pl->stage));
    pl->stage = nullptr;
    pl->stage_bytes = 0;
    HIP_TRY(hipMalloc(&pl->stage, bytes));
    pl->stage_bytes = bytes;
    return JTP_OK;
}

// messages and marginals are plain bit fields: complete the physical part of their layout record
static void bitfield_desc(JtPackDesc &d) {
    for (int i = 0; i < d.nvars; ++i) {
        d.dstride[i] = 1u << d.pos[i];
        d.dmod[i] = 1 << d.nb[i];
    }
    d.phys_elems = (int64_t)1 << d.nbits;
    d.low_bits = d.nbits;
    d.row_elems = 0;
    d.split_var = -1;
}

template <typename T, typename S>
static void launch_pack(const JtPackDesc &d, const S *stage, T *arena, hipStream_t s) {
    const int64_t n = d.phys_elems;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL((jt_pack<T, S, 0>), dim3(grid), dim3(256), 0, s, d, stage, arena, 0ull, 0.0);
}

extern "C" {

const char *jtp_last_error(void) { return g_err.c_str(); }
// JTP_SOURCE_ID: digest of the library's sources, passed in by junction-tree_amd/build.py; profiles/ files carry the id of
// the build they were measured on, and bench.py quotes a traffic figure only from a file whose id matches the running one
#ifndef JTP_SOURCE_ID
#define JTP_SOURCE_ID "unknown"
#endif
const char *jtp_version(void) { return "jtprop 0.6.0 (gfx950, HIP, RCCL p2p) src:" JTP_SOURCE_ID; }

int jtp_host_alloc(void **ptr, size_t bytes) {
    if (!ptr) return set_err(JTP_EINVAL, "null argument");
    *ptr = nullptr;
    hipError_t e = hipHostMalloc(ptr, std::max<size_t>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) return set_err(e == hipErrorOutOfMemory ? JTP_ENOMEM : JTP_EHIP, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return JTP_OK;
}

int jtp_host_free(void *ptr) {
    if (!ptr) return JTP_OK;
    HIP_TRY(hipHostFree(ptr));
    return JTP_OK;
}

int jtp_device_count(int32_t *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return set_err(JTP_EHIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return JTP_OK;
}

int jtp_device_memory(int32_t device, uint64_t *free_bytes, uint64_t *total_bytes) {
    size_t f = 0, t = 0;
    int before = 0;
    HIP_TRY(hipGetDevice(&before));
    HIP_TRY(hipSetDevice(device));
    const hipError_t e = hipMemGetInfo(&f, &t);
    (void)hipSetDevice(before);                    // (the caller's current device stays what it was)
    if (e != hipSuccess) return set_err(JTP_EHIP, "hipMemGetInfo failed: %s", hipGetErrorString(e));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return JTP_OK;
}

const char *jtp_kernel_name(int32_t variant) {
    if (variant < 0 || variant >= JT_K_COUNT) return nullptr;
    return k_names[variant];
}

// ------------------------------------------------------------------------------------------ lifetime

void jtp_plan_destroy(jtp_plan *pl) {
    if (!pl) return;
    if (pl->device) {
        (void)hipSetDevice(pl->hp.device);
        for (auto s : pl->streams) (void)hipStreamSynchronize(s);
        if (pl->inflight) {
            pl->inflight = false;
            (void)board::publish(pl->hp.device, --g_inflight[pl->hp.device & 63]);
        }
        if (pl->multiset) {
            if (!pl->bufs.empty()) {
                if (pl->bufs[0].psi) (void)hipFree(pl->bufs[0].psi);
                if (pl->bufs[0].bel) (void)hipFree(pl->bufs[0].bel);
            }
            if (pl->msg_all) (void)hipFree(pl->msg_all);
            if (pl->d_member) (void)hipFree(pl->d_member);
            if (pl->d_esum_oct) (void)hipFree(pl->d_esum_oct);
            if (pl->d_act_ids) (void)hipFree(pl->d_act_ids);
            if (pl->d_act_n) (void)hipFree(pl->d_act_n);
            if (pl->d_fanout) (void)hipFree(pl->d_fanout);
            if (pl->ev_all) (void)hipFree(pl->ev_all);
            if (pl->sync_all) (void)hipFree(pl->sync_all);
        } else
        for (auto &b : pl->bufs) {
            if (b.psi && (&b == &pl->bufs[0] || b.psi != pl->bufs[0].psi)) (void)hipFree(b.psi);
            if (b.bel) (void)hipFree(b.bel);
            if (b.msg) (void)hipFree(b.msg);
            if (b.fix && (&b == &pl->bufs[0] || b.fix != pl->bufs[0].fix)) (void)hipFree(b.fix);
            if (b.ev) (void)hipFree(b.ev);
            if (b.sync) (void)hipFree(b.sync);
        }
        if (pl->host_abort) (void)hipHostFree(pl->host_abort);
        for (auto &bt : pl->belief_tasks) {
            if (bt.d_task) (void)hipFree(bt.d_task);
            if (bt.d_blk) (void)hipFree(bt.d_blk);
            if (bt.d_tab) (void)hipFree(bt.d_tab);
        }
        for (MargBatch *mb : pl->marg_cache) {
            mb->release();
            delete mb;
        }
        if (pl->d_tasks) (void)hipFree(pl->d_tasks);
        if (pl->d_blocks) (void)hipFree(pl->d_blocks);
        for (int m = 0; m < 2; ++m)
            if (pl->d_init[m]) (void)hipFree(pl->d_init[m]);
        if (pl->d_itab) (void)hipFree(pl->d_itab);
        if (pl->stage) (void)hipFree(pl->stage);
        for (int i = 0; i < 2; ++i) {
            if (pl->up_stage[i]) (void)hipFree(pl->up_stage[i]);
            if (pl->up_ev[i]) (void)hipEventDestroy(pl->up_ev[i]);
        }
        if (pl->eval_stage) (void)hipFree(pl->eval_stage);
        if (pl->eval_host) (void)hipHostFree(pl->eval_host);
        if (pl->unit_scratch) (void)hipFree(pl->unit_scratch);
        for (auto e : pl->ev) (void)hipEventDestroy(e);
        for (auto e : pl->region_ev)
            if (e) (void)hipEventDestroy(e);
        for (auto s : pl->streams) (void)hipStreamDestroy(s);
    }
    delete pl;
}

static int zero_padding(jtp_plan *pl, double *msg, int nsets, hipStream_t s, int halves = 3);

int jtp_plan_create(const jtp_tree_desc *desc, jtp_plan **out) {
    if (!out) return set_err(JTP_EINVAL, "null output pointer");
    *out = nullptr;
    jtp_plan *pl = new jtp_plan();
    std::string err;
    int rc = jtp_build_plan(desc, pl->hp, err);
    // Fall-backs of the planner's defaults, tried in turn while the structure is "unsupported":
    //  - multi-set plans: sub-boxes too large for one evidence set's LDS region under the default bit order -> the
    //    order with the smallest sub-boxes;
    //  - a table that cannot be cut into workgroups without splitting a variable stored at its true cardinality ->
    //    every table padded to powers of two (the round-1 layout).
    for (int attempt = 1; attempt < 4 && rc == JTP_EUNSUPPORTED && desc; ++attempt) {
        jtp_tree_desc again = *desc;
        const bool relayout = (attempt & 1) && (desc->flags & JTP_MULTISET) && desc->layout_policy == 0;
        const bool pad = (attempt & 2) && !(desc->flags & JTP_NO_COMPACT);
        if (!relayout && !pad) continue;
        if ((attempt & 1) && !relayout) continue;
        if ((attempt & 2) && !pad) continue;
        if (relayout) again.layout_policy = 2;
        if (pad) again.flags |= JTP_NO_COMPACT;
        std::string err2;
        delete pl;
        pl = new jtp_plan();
        const int rc2 = jtp_build_plan(&again, pl->hp, err2);
        if (rc2 == JTP_OK) rc = rc2;
        else if (rc2 != JTP_EUNSUPPORTED) rc = rc2, err = err2;
    }
    //  - float32 storage means 1024-element rows: a clique of few rows with four or more neighbours whose separators are
    //    nearly the whole clique then needs more LDS than a CU has ("message sub-boxes do not fit in LDS").  The same tree in
    //    float64 storage (512-element rows) plans: the plan is then made with double tables - twice the device bytes of
    //    what was asked for, the host interface unchanged (every call names its host type) - and says so in
    //    jtp_stats.storage_dtype.  (Round 3 did this in the Python layer only, keyed on the message text.)
    if (rc == JTP_EUNSUPPORTED && desc && desc->dtype == JTP_F32 && !(desc->flags & JTP_MULTISET)) {
        for (int attempt = 0; attempt < 2 && rc == JTP_EUNSUPPORTED; ++attempt) {
            jtp_tree_desc again = *desc;
            again.dtype = JTP_F64;
            if (attempt == 1) {
                if (desc->flags & JTP_NO_COMPACT) break;
                again.flags |= JTP_NO_COMPACT;
            }
            std::string err2;
            delete pl;
            pl = new jtp_plan();
            const int rc2 = jtp_build_plan(&again, pl->hp, err2);
            if (rc2 == JTP_OK) rc = rc2, pl->widened = true;
            else if (rc2 != JTP_EUNSUPPORTED) rc = rc2, err = err2;
        }
    }
    //  - a clique that keeps no table stages the product of its factors like one more message: where that does not fit (or the
    //    tighter layout rules of such cliques cannot be met) the tree is planned with every table materialised, as rounds 1-4 did
    if (rc == JTP_EUNSUPPORTED && desc && desc->cover_off && !(desc->flags & JTP_MULTISET)) {
        jtp_tree_desc again = *desc;
        again.cover_off = again.cover_ids = nullptr;
        jtp_plan *fresh = nullptr;
        const int rc2 = jtp_plan_create(&again, &fresh);
        delete pl;
        if (rc2 == JTP_OK) {
            fresh->hp.lean_refused = err.empty() ? std::string("unsupported") : err;      // (jtp_stats.lean_refused, jtp_plan_describe)
            *out = fresh;
        }
        return rc2;
    }
    if (rc != JTP_OK) {
        delete pl;
        return set_err(rc, "%s", err.c_str());
    }
    HostPlan &hp = pl->hp;
    pl->esize = hp.dtype == JTP_F32 ? 4 : 8;
    pl->multiset = hp.multiset;
    if (pl->multiset && !(hp.flags & JTP_SHARE_POTENTIALS)) {
        delete pl;
        return set_err(JTP_EINVAL, "JTP_MULTISET needs JTP_SHARE_POTENTIALS (the evidence sets of a group read one table)");
    }
    // one launch per level when asked for, for per-shape launches and for the JTP_DEBUG experiments
    pl->flow = !(hp.flags & (JTP_LEVEL_LAUNCHES | JTP_SPLIT_VARIANTS)) && !(hp.knobs.debug & 1) && !hp.knobs.force_level_launches;
    // Sub-boxes so large that one or two workgroups fill a CU (config 3: 121 KB): a waiting workgroup
    // then idles a whole CU, and staging is a large share of the traffic, which per-level launches read
    // through L2 while a dataflow launch has to read through to memory.  Measured 39.6 vs 46.9 ms.
    if (hp.max_lds > 64 * 1024 && !pl->multiset && !hp.knobs.force_flow) pl->flow = false;
    pl->chain = hp.chain_plan;
    for (const HostPlan::FoldReq &fr : hp.folded) pl->marg_tasks = pl->marg_tasks || fr.task >= 0;
    if (hp.flags & JTP_PLAN_ONLY) {
        *out = pl;
        return JTP_OK;
    }
    // JTP_FAKE_COMM=1 (development aid): run ONE rank's share of a multi-rank plan on its own; what
    // it would receive is filled with ones, what it would send goes nowhere.  Timing only.
    pl->fake_comm = hp.n_ranks > 1 ? hp.knobs.fake_comm : 0;
    if (pl->fake_comm == 2 && (!rccl::comm || rccl::comm_size != 1)) {
        delete pl;
        return set_err(JTP_ECOMM, "JTP_FAKE_COMM=2 (exchange steps as RCCL groups in loop-back) needs a communicator of ONE rank: jtp_comm_init(0, 1, ...)");
    }
    if (hp.n_ranks > 1 && !pl->fake_comm && (!rccl::comm || rccl::comm_size != hp.n_ranks || rccl::comm_rank != hp.rank)) {
        delete pl;
        return set_err(JTP_ECOMM, "n_ranks=%d but jtp_comm_init was not called with a matching communicator", hp.n_ranks);
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        delete pl;
        return set_err(JTP_EHIP, "no HIP device available (%s); libjtprop has no CPU fallback",
                       e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    }
#define CREATE_TRY(expr)                                                                                 \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            set_err(_e == hipErrorOutOfMemory ? JTP_ENOMEM : JTP_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
            jtp_plan_destroy(pl);                                                                        \
            return _e == hipErrorOutOfMemory ? JTP_ENOMEM : JTP_EHIP;                                    \
        }                                                                                                \
    } while (0)
    pl->device = true;
    pl->flow_debug = hp.knobs.flow_debug;
    pl->env_tickets = hp.knobs.flow_tickets != 0;
    pl->roctx = hp.knobs.roctx != 0;
    if (pl->roctx) roctx::load();
    CREATE_TRY(hipSetDevice(hp.device));
    const int nstreams = pl->multiset ? 1 : std::min(hp.n_batch, 16);
    pl->streams.resize(nstreams);
    for (auto &s : pl->streams) CREATE_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    pl->bufs.resize(hp.n_batch);
    const size_t abytes = (size_t)std::max<int64_t>(hp.arena_elems, 256) * pl->esize;
    // two halves, used by alternate propagates (jtp_internal.h: JT_UNWRITTEN)
    const size_t mbytes = (size_t)std::max<int64_t>(hp.msg_doubles, 2) * 8 * 2;
    const bool share_psi = (hp.flags & JTP_SHARE_POTENTIALS) != 0;     // one potential arena for all evidence sets
    if (pl->multiset) {
        // (group 0: the evidence-free sets the others take their untouched upward messages from - one more group through the collect
        //  pass, which pays from eight groups on: measured 64 sets 4.97 -> 4.8 ms, 8 sets 0.91 -> 1.3; JTP_EF_SHARE=1 / JTP_NO_EF_SHARE=1 force it)
        const bool ef = hp.knobs.no_ef_share == 0 && ((hp.n_batch + JT_MSETS - 1) / JT_MSETS >= 8 || hp.knobs.no_ef_share < 0);
        pl->set0 = ef ? JT_MSETS : 0;
        pl->n_groups = (hp.n_batch + JT_MSETS - 1) / JT_MSETS + (pl->set0 ? 1 : 0);
        const size_t nsets = (size_t)pl->n_groups * JT_MSETS;          // (the last group is padded with evidence-free sets)
        pl->set_stride = (int64_t)(mbytes / 8);
        pl->ev_stride = (uint32_t)(2 * hp.pn.size());
        void *psi = nullptr, *bel = nullptr;
        CREATE_TRY(hipMalloc(&psi, abytes));
        pl->bufs[0].psi = psi;
        CREATE_TRY(hipMalloc(&bel, abytes));                           // scratch: one belief table at a time, on demand
        pl->bufs[0].bel = bel;
        CREATE_TRY(hipMemsetAsync(psi, 0, abytes, pl->streams[0]));
        CREATE_TRY(hipMemsetAsync(bel, 0, abytes, pl->streams[0]));
        CREATE_TRY(hipMalloc((void **)&pl->msg_all, mbytes * nsets));
        CREATE_TRY(hipMemsetD32Async((hipDeviceptr_t)pl->msg_all, (int)(uint32_t)(JT_UNWRITTEN & 0xffffffffu), mbytes * nsets / 4, pl->streams[0]));
        CREATE_TRY(hipMalloc((void **)&pl->ev_all, (size_t)pl->ev_stride * 4 * nsets));
        CREATE_TRY(hipMemsetAsync(pl->ev_all, 0, (size_t)pl->ev_stride * 4 * nsets, pl->streams[0]));
        CREATE_TRY(hipMalloc((void **)&pl->sync_all, (size_t)hp.sync_words * 4 * pl->n_groups));
        CREATE_TRY(hipMemsetAsync(pl->sync_all, 0, (size_t)hp.sync_words * 4 * pl->n_groups, pl->streams[0]));
        for (int b = 0; b < hp.n_batch; ++b) {
            BatchBuffers &bb = pl->bufs[b];
            bb.psi = psi;
            bb.bel = bel;
            bb.msg = pl->msg_all + (int64_t)(pl->set0 + b) * pl->set_stride;
            bb.ev = pl->ev_all + (size_t)(pl->set0 + b) * pl->ev_stride;
            bb.sync = pl->sync_all + (size_t)((pl->set0 + b) / JT_MSETS) * hp.sync_words;
        }
        pl->belief_tasks.resize(hp.pn.size());
        if (pl->set0) {
            // (the active lists are made by the first propagate: rebuild_active)
            const size_t cap = (size_t)pl->n_groups * JT_MSETS, nt = hp.tasks.size();
            CREATE_TRY(hipMalloc((void **)&pl->d_member, nt * cap));
            CREATE_TRY(hipMalloc((void **)&pl->d_act_ids, nt * cap * sizeof(uint16_t)));
            CREATE_TRY(hipMalloc((void **)&pl->d_act_n, nt * sizeof(int32_t)));
            CREATE_TRY(hipMalloc((void **)&pl->d_esum_oct, nt * (size_t)pl->n_groups));
            pl->act_dirty = true;
        }
    } else
    for (auto &b : pl->bufs) {
        if (share_psi && &b != &pl->bufs[0]) b.psi = pl->bufs[0].psi;
        else CREATE_TRY(hipMalloc(&b.psi, abytes));
        CREATE_TRY(hipMalloc(&b.bel, abytes));
        CREATE_TRY(hipMalloc((void **)&b.msg, mbytes));
        if (hp.fix_doubles > 0) {
            if (share_psi && &b != &pl->bufs[0]) b.fix = pl->bufs[0].fix;
            else {
                CREATE_TRY(hipMalloc((void **)&b.fix, (size_t)hp.fix_doubles * 8));
                CREATE_TRY(hipMemsetAsync(b.fix, 0, (size_t)hp.fix_doubles * 8, pl->streams[0]));
            }
        }
        CREATE_TRY(hipMemsetAsync(b.psi, 0, abytes, pl->streams[0]));
        CREATE_TRY(hipMemsetAsync(b.bel, 0, abytes, pl->streams[0]));
        CREATE_TRY(hipMemsetD32Async((hipDeviceptr_t)b.msg, (int)(uint32_t)(JT_UNWRITTEN & 0xffffffffu), mbytes / 4, pl->streams[0]));
        CREATE_TRY(hipMalloc((void **)&b.sync, (size_t)hp.sync_words * 4));
        CREATE_TRY(hipMemsetAsync(b.sync, 0, (size_t)hp.sync_words * 4, pl->streams[0]));
    }
    CREATE_TRY(hipHostMalloc((void **)&pl->host_abort, 64, hipHostMallocMapped));
    *pl->host_abort = 0;

    for (auto &b : pl->bufs) {
        if (&b != &pl->bufs[0] && b.psi == pl->bufs[0].psi) continue;          // shared tables: filled once
        for (const VirtualFill &vf : hp.virtual_fills) {
            const int64_t n = vf.d.phys_elems;
            const int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
            if (hp.dtype == JTP_F32) hipLaunchKernelGGL((jt_pack<float, float, 2>), dim3(grid), dim3(256), 0, pl->streams[0], vf.d, (const float *)nullptr, (float *)b.psi, 0ull, 1.0);
            else hipLaunchKernelGGL((jt_pack<double, double, 2>), dim3(grid), dim3(256), 0, pl->streams[0], vf.d, (const double *)nullptr, (double *)b.psi, 0ull, 1.0);
        }
        CREATE_TRY(hipGetLastError());
    }
    if (pl->multiset) {
        pl->ev_host.assign((size_t)pl->ev_stride * pl->n_groups * JT_MSETS, 0u);
        for (JtTask &tk : hp.tasks)
            if (tk.esum & 1) tk.esum |= 2, tk.esum_groups = ~0ull;      // no evidence yet
    }
    if (!hp.tasks.empty()) {
        CREATE_TRY(hipMalloc((void **)&pl->d_tasks, hp.tasks.size() * sizeof(JtTask)));
        CREATE_TRY(hipMemcpy(pl->d_tasks, hp.tasks.data(), hp.tasks.size() * sizeof(JtTask), hipMemcpyHostToDevice));
    }
    if (!hp.blocks.empty()) {
        CREATE_TRY(hipMalloc((void **)&pl->d_blocks, hp.blocks.size() * sizeof(JtBlock)));
        CREATE_TRY(hipMemcpy(pl->d_blocks, hp.blocks.data(), hp.blocks.size() * sizeof(JtBlock), hipMemcpyHostToDevice));
    }
    if (!hp.itab.empty()) {
        CREATE_TRY(hipMalloc((void **)&pl->d_itab, hp.itab.size() * sizeof(int32_t)));
        CREATE_TRY(hipMemcpy(pl->d_itab, hp.itab.data(), hp.itab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    for (int m = 0; m < 2; ++m)
        if (!hp.init_blocks[m].empty()) {
            CREATE_TRY(hipMalloc((void **)&pl->d_init[m], hp.init_blocks[m].size() * sizeof(JtBlock)));
            CREATE_TRY(hipMemcpy(pl->d_init[m], hp.init_blocks[m].data(), hp.init_blocks[m].size() * sizeof(JtBlock), hipMemcpyHostToDevice));
        }
    // dynamic LDS beyond 64 KiB has to be allowed per kernel function (raise_lds remembers what each one has)
    auto kfunc = [&](int v) { return hp.dtype == JTP_F32 ? (const void *)KernelTable<float>::get(v, mixk(hp)) : (const void *)KernelTable<double>::get(v, mixk(hp)); };
    if (pl->multiset) {
        CREATE_TRY(raise_lds(kfunc(JT_K_MULTI_COLLECT), JT_RING_BYTES + JT_MSETS * (JT_MSETS > 8 ? JT_SETB_SMALL : JT_SETB_LARGE)));
    } else if (hp.max_lds > 64 * 1024) {
        for (int v = 0; v < JT_K_COUNT; ++v) {
            if (kfunc(v) == nullptr) continue;                 // (the dataflow kernels: below)
            CREATE_TRY(raise_lds(kfunc(v), hp.max_lds));
        }
        for (int ph = 0; ph < 3; ++ph) {
            const void *f = hp.dtype == JTP_F32 ? (const void *)KernelTable<float>::get_flow(ph, pl->chain, mixk(hp), pl->marg_tasks) : (const void *)KernelTable<double>::get_flow(ph, pl->chain, mixk(hp), pl->marg_tasks);
            CREATE_TRY(raise_lds(f, hp.max_lds));
        }
    }
    {   // what the plan holds on the device from now on (the plan cache of the Python layer budgets with it)
        double b = 0;
        if (pl->multiset) {
            const size_t nsets = (size_t)pl->n_groups * JT_MSETS;
            b = 2.0 * abytes + (double)mbytes * nsets + (double)pl->ev_stride * 4 * nsets + (double)hp.sync_words * 4 * pl->n_groups;
        } else {
            const size_t npsi = share_psi ? 1 : pl->bufs.size();
            b = (double)abytes * (npsi + pl->bufs.size()) + ((double)mbytes + (double)hp.sync_words * 4) * pl->bufs.size() + (double)hp.fix_doubles * 8 * npsi;
        }
        b += (double)hp.tasks.size() * sizeof(JtTask) + (double)hp.blocks.size() * sizeof(JtBlock) + (double)hp.itab.size() * 4;
        pl->device_bytes = b;
    }
    if (pl->multiset) {
        if (int rc = zero_padding(pl, pl->msg_all, pl->n_groups * JT_MSETS, pl->streams[0])) {
            jtp_plan_destroy(pl);
            return rc;
        }
    } else
        for (auto &b : pl->bufs)
            if (int rc = zero_padding(pl, b.msg, 1, pl->streams[0])) {
                jtp_plan_destroy(pl);
                return rc;
            }
    CREATE_TRY(hipStreamSynchronize(pl->streams[0]));
#undef CREATE_TRY
    *out = pl;
    return JTP_OK;
}

const char *jtp_plan_describe(jtp_plan *pl) {
    if (!pl) return "";
    if (pl->hp.json.empty()) jtp_plan_to_json(pl->hp, pl->hp.tasks.size() <= 20000);
    return pl->hp.json.c_str();
}

// ------------------------------------------------------------------------------------------ data in

// potentials are written through evidence set 0 when the plan shares them
static int check_writable(jtp_plan *pl, int batch) {
    if ((pl->hp.flags & JTP_SHARE_POTENTIALS) && batch != 0)
        return set_err(JTP_EINVAL, "the plan shares its potentials between evidence sets: set them through evidence set 0");
    return JTP_OK;
}

static int check_ready(jtp_plan *pl, int batch) {
    if (!pl) return set_err(JTP_EINVAL, "null plan");
    if (!pl->device) return set_err(JTP_EHIP, "plan was created with JTP_PLAN_ONLY: no device work possible");
    if (batch < 0 || batch >= pl->hp.n_batch) return set_err(JTP_EINVAL, "batch %d out of range [0,%d)", batch, pl->hp.n_batch);
    return JTP_OK;
}

int jtp_set_potential(jtp_plan *pl, int32_t batch, int32_t node, const void *host, const int64_t *shape,
                      int32_t host_dtype) {
    int rc = check_ready(pl, batch);
    if (rc) return rc;
    rc = check_writable(pl, batch);
    if (rc) return rc;
    pl->psi_dirty = true;
    HostPlan &hp = pl->hp;
    if (node < 0 || node >= hp.n_cliques) return set_err(JTP_EINVAL, "node %d is not a clique", node);
    if (!(hp.pn[node].owner == hp.rank || hp.pn[node].owner == hp.n_ranks)) return set_err(JTP_EINVAL, "clique %d belongs to rank %d", node, hp.pn[node].owner);
    if (host_dtype != JTP_F32 && host_dtype != JTP_F64) return set_err(JTP_EINVAL, "bad host dtype");
    // a unit clique (jtp_tree_desc.cover_*) keeps its potential as a static table over the covered variables: the other axes
    // of the host array must have length 1, as the reference's evaluate leaves them (junctiontree.py:52-61)
    const bool unit = hp.pn[node].unit;
    if (unit && hp.pn[node].stat < 0) {
        bool one = true;
        for (size_t i = 0; i < hp.node_vars[node].size(); ++i) one = one && (!shape || shape[i] == 1 || hp.card[hp.node_vars[node][i]] == 1);
        if (!shape) for (int v : hp.node_vars[node]) one = one && hp.card[v] == 1;
        const double v = !host ? 0.0 : (host_dtype == JTP_F32 ? (double)*(const float *)host : *(const double *)host);
        if (!one || v != 1.0)
            return set_err(JTP_EINVAL, "clique %d was described as depending on none of its variables (jtp_tree_desc.cover_*): its potential is 1", node);
        return JTP_OK;
    }
    JtPackDesc d = unit ? hp.stat_pack[node] : hp.pack[node];
    int64_t stride = 1;
    for (int i = d.nvars - 1; i >= 0; --i) {
        const int64_t len = shape ? shape[i] : d.card[i];
        if (len != d.card[i] && len != 1) {
            if (unit && len == hp.card[hp.node_vars[node][i]])
                return set_err(JTP_EINVAL, "clique %d axis %d has length %lld, but the clique was described as not depending on that variable (jtp_tree_desc.cover_*)", node, i, (long long)len);
            return set_err(JTP_EINVAL, "clique %d axis %d has length %lld, expected %d or 1", node, i, (long long)len, d.card[i]);
        }
        d.hstride[i] = (len == 1) ? 0 : stride;
        stride *= len;
    }
    const size_t hbytes = (size_t)stride * (host_dtype == JTP_F32 ? 4 : 8);
    HIP_TRY(hipSetDevice(hp.device));
    const int ui = (int)(pl->up_cursor++ & 1u);
    if (!pl->up_ev[ui]) HIP_TRY(hipEventCreateWithFlags(&pl->up_ev[ui], hipEventDisableTiming));
    if (pl->up_busy[ui]) {                                  // the pack kernel that read this buffer two calls ago
        HIP_TRY(hipEventSynchronize(pl->up_ev[ui]));
        pl->up_busy[ui] = false;
    }
    if (pl->up_bytes[ui] < hbytes) {
        if (pl->up_stage[ui]) HIP_TRY(hipFree(pl->up_stage[ui]));
        pl->up_stage[ui] = nullptr;
        pl->up_bytes[ui] = 0;
        HIP_TRY(hipMalloc(&pl->up_stage[ui], std::max<size_t>(hbytes, 256)));
        pl->up_bytes[ui] = std::max<size_t>(hbytes, 256);
    }
    void *stage = pl->up_stage[ui];
    hipStream_t s = pl->streams[batch % pl->streams.size()];
    // (from pageable memory the copy returns once the runtime has staged the caller's bytes; from page-locked
    //  memory - jtp_host_alloc - it is asynchronous and the caller must keep the array alive until jtp_sync)
    HIP_TRY(hipMemcpyAsync(stage, host, hbytes, hipMemcpyHostToDevice, s));
    BatchBuffers &b = pl->bufs[batch];
    if (unit) {
        if (host_dtype == JTP_F32) launch_pack<double, float>(d, (const float *)stage, b.fix, s);
        else launch_pack<double, double>(d, (const double *)stage, b.fix, s);
    } else if (hp.dtype == JTP_F32) {
        if (host_dtype == JTP_F32) launch_pack<float, float>(d, (const float *)stage, (float *)b.psi, s);
        else launch_pack<float, double>(d, (const double *)stage, (float *)b.psi, s);
    } else {
        if (host_dtype == JTP_F32) launch_pack<double, float>(d, (const float *)stage, (double *)b.psi, s);
        else launch_pack<double, double>(d, (const double *)stage, (double *)b.psi, s);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(pl->up_ev[ui], s));
    pl->up_busy[ui] = true;
    return JTP_OK;
}

// CliqueGraph.evaluate (junctiontree.py:203-226) for a list of cliques: ONE host-to-device copy of every factor table and
// of the kernel's records, ONE launch of jt_eval_batch over all the cliques (plus one per further JT_EVAL_MAX_F factors of
// the clique with the most).  Round 3 ran a copy and a launch per clique, and a kernel that decoded every element.
int jtp_set_potential_products(jtp_plan *pl, int32_t batch, int32_t n, const int32_t *cliques, const int32_t *factor_off,
                               const jtp_factor *factors) {
    int rc = check_ready(pl, batch);
    if (rc) return rc;
    rc = check_writable(pl, batch);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!cliques || !factor_off))) return set_err(JTP_EINVAL, "null argument");
    if (n == 0) return JTP_OK;
    HostPlan &hp = pl->hp;
    if (factor_off[0] < 0) return set_err(JTP_EINVAL, "bad factor list");
    const int32_t f0 = factor_off[0], nfact = factor_off[n] - f0;
    if (nfact < 0 || (nfact > 0 && !factors)) return set_err(JTP_EINVAL, "bad factor list");
    // the tables in the staging buffer: 8-byte slots so that f32 and f64 tables can mix
    std::vector<int64_t> offs((size_t)nfact), elems((size_t)nfact);
    size_t tbytes = 0;
    int npass = 1;
    std::vector<char> listed((size_t)hp.n_cliques, 0);
    for (int i = 0; i < n; ++i) {
        const int clique = cliques[i];
        if (clique < 0 || clique >= hp.n_cliques) return set_err(JTP_EINVAL, "node %d is not a clique", clique);
        // (every listed clique is formed by workgroups of ONE launch: a clique listed twice would be written by two of them)
        if (listed[clique]) return set_err(JTP_EINVAL, "clique %d is listed twice", clique);
        listed[clique] = 1;
        if (!(hp.pn[clique].owner == hp.rank || hp.pn[clique].owner == hp.n_ranks)) return set_err(JTP_EINVAL, "clique %d belongs to rank %d", clique, hp.pn[clique].owner);
        if (factor_off[i + 1] < factor_off[i]) return set_err(JTP_EINVAL, "bad factor list");
        const std::vector<int> &cvars = hp.node_vars[clique];
        npass = std::max(npass, (factor_off[i + 1] - factor_off[i] + JT_EVAL_MAX_F - 1) / JT_EVAL_MAX_F);
        for (int f = factor_off[i]; f < factor_off[i + 1]; ++f) {
            const jtp_factor &ft = factors[f];
            const int fi = f - factor_off[i];
            if (ft.n_vars < 0 || ft.n_vars > JT_MAX_VARS) return set_err(JTP_EINVAL, "factor %d: bad variable count", fi);
            if (ft.dtype != JTP_F32 && ft.dtype != JTP_F64) return set_err(JTP_EINVAL, "factor %d: bad dtype", fi);
            if (!ft.host || (ft.n_vars > 0 && !ft.var_ids)) return set_err(JTP_EINVAL, "factor %d: null argument", fi);
            int64_t ne = 1;
            for (int j = 0; j < ft.n_vars; ++j) {
                const int v = ft.var_ids[j];
                bool found = false;
                for (int cv : cvars) found = found || cv == v;
                if (!found) return set_err(JTP_EINVAL, "factor %d: variable %d is not in clique %d", fi, v, clique);
                const int64_t len = ft.shape ? ft.shape[j] : hp.card[v];
                if (len != hp.card[v] && len != 1) return set_err(JTP_EINVAL, "factor %d axis %d has length %lld, expected %d or 1", fi, j, (long long)len, hp.card[v]);
                if (hp.pn[clique].unit && len != 1) {
                    bool covered = false;
                    for (int cv : hp.pn[clique].cover) covered = covered || cv == v;
                    if (!covered) return set_err(JTP_EINVAL, "factor %d: clique %d was described as not depending on variable %d (jtp_tree_desc.cover_*)", fi, clique, v);
                }
                ne *= len;
            }
            if (hp.pn[clique].unit && hp.pn[clique].stat < 0)
                return set_err(JTP_EINVAL, "clique %d was described as depending on none of its variables (jtp_tree_desc.cover_*): it takes no factor", clique);
            elems[f - f0] = ne;
            offs[f - f0] = (int64_t)(tbytes / 8);
            tbytes += (size_t)((ne * (ft.dtype == JTP_F32 ? 4 : 8) + 7) / 8) * 8;
        }
    }
    pl->psi_dirty = true;
    HIP_TRY(hipSetDevice(hp.device));
    hipStream_t s = pl->streams[batch % pl->streams.size()];
    BatchBuffers &b = pl->bufs[batch];
    // the kernel's records: per pass the clique records, then the workgroup prefix sums
    // (lists [0, npass): cliques that keep a table, formed in the potential arena in its storage type; [npass, 2 npass): static
    //  tables of unit cliques, plain bit fields of doubles in the fixed arena)
    std::vector<std::vector<JtEvalTask>> tasks((size_t)npass * 2);
    std::vector<JtEvalVar> fvars;
    int lds_doubles = 0;
    for (int i = 0; i < n; ++i) {
        const int clique = cliques[i];
        const std::vector<int> &cvars = hp.node_vars[clique];
        const PNode &p = hp.pn[clique];
        const int nfc = factor_off[i + 1] - factor_off[i];
        int done = 0, pass = 0;
        if (p.unit && p.stat < 0) continue;                 // all ones, nothing stored
        do {
            JtEvalTask tk;
            memset(&tk, 0, sizeof tk);
            tk.clique = p.unit ? hp.stat_pack[clique] : hp.pack[clique];
            if (p.unit) tk.clique.low_bits = std::min(tk.clique.nbits, 9);      // (rows of at most 512 doubles: a 16-byte vector per thread)
            tk.accumulate = done > 0;
            tk.nf = std::min(nfc - done, JT_EVAL_MAX_F);
            tk.row_len = tk.clique.row_elems > 0 ? tk.clique.row_elems : 1 << tk.clique.low_bits;
            tk.n_rows = (int32_t)(tk.clique.phys_elems / tk.row_len);
            // the variable with a digit part inside the row and one above it
            tk.straddle = tk.clique.row_elems > 0 ? tk.clique.split_var : -1;
            if (tk.clique.row_elems == 0)
                for (int j = 0; j < tk.clique.nvars; ++j)
                    if (tk.clique.pos[j] < tk.clique.low_bits && tk.clique.pos[j] + tk.clique.nb[j] > tk.clique.low_bits) tk.straddle = j;
            int used = 0;
            for (int k = 0; k < tk.nf; ++k) {
                const int f = factor_off[i] + done + k;
                const jtp_factor &ft = factors[f];
                tk.fnv[k] = ft.n_vars;
                tk.fis64[k] = ft.dtype == JTP_F64;
                tk.foff[k] = ft.dtype == JTP_F64 ? offs[f - f0] : offs[f - f0] * 2;     // in elements of its own type
                tk.felems[k] = (int32_t)std::min<int64_t>(elems[f - f0], INT32_MAX);
                tk.flds[k] = -1;
                if (used + elems[f - f0] <= JT_EVAL_LDS_DOUBLES) {
                    tk.flds[k] = used;
                    used += (int)((elems[f - f0] + 1) & ~(int64_t)1);
                }
                tk.fv_off[k] = (int32_t)fvars.size();
                fvars.resize(fvars.size() + (size_t)ft.n_vars);
                JtEvalVar *fv = fvars.data() + tk.fv_off[k];
                int64_t stride = 1;
                for (int j = ft.n_vars - 1; j >= 0; --j) {
                    const int v = ft.var_ids[j];
                    int pos = 0;
                    while (cvars[pos] != v) ++pos;
                    const JtPackDesc &cd = tk.clique;
                    fv[j].ds = cd.dstride[pos];
                    fv[j].mod = cd.dmod[pos];
                    fv[j].kind = cd.row_elems > 0 && cd.pos[pos] < cd.low_bits ? (pos == cd.split_var ? 2 : 1) : 0;
                    const int64_t len = ft.shape ? ft.shape[j] : hp.card[v];
                    fv[j].stride = (len == 1) ? 0 : (int32_t)stride;
                    stride *= len;
                }
            }
            lds_doubles = std::max(lds_doubles, used);
            tasks[(p.unit ? npass : 0) + pass].push_back(tk);
            done += tk.nf;
            ++pass;
        } while (done < nfc);
        (void)p;
    }
    const int nlists = 2 * npass;
    std::vector<size_t> task_at((size_t)nlists), blk_at((size_t)nlists);
    size_t bytes = 0;
    for (int k = 0; k < nlists; ++k) {
        task_at[k] = bytes;
        bytes += (tasks[k].size() * sizeof(JtEvalTask) + 255) & ~(size_t)255;
        blk_at[k] = bytes;
        bytes += ((tasks[k].size() + 1) * sizeof(int32_t) + 255) & ~(size_t)255;
    }
    const size_t fvars_at = bytes;
    bytes += std::max<size_t>((fvars.size() * sizeof(JtEvalVar) + 255) & ~(size_t)255, 256);
    const size_t tables_at = bytes;
    bytes += std::max<size_t>((tbytes + 255) & ~(size_t)255, 256);
    if (pl->eval_pending && pl->eval_stream != s) {        // another evidence set's kernels may still read the buffer
        HIP_TRY(hipStreamSynchronize(pl->eval_stream));
        pl->eval_pending = false;
        pl->eval_cursor = 0;
    }
    if (pl->eval_cursor + bytes > pl->eval_bytes) {
        if (pl->eval_pending) HIP_TRY(hipStreamSynchronize(pl->eval_stream));
        pl->eval_pending = false;
        pl->eval_cursor = 0;
        if (bytes > pl->eval_bytes) {
            if (pl->eval_stage) HIP_TRY(hipFree(pl->eval_stage));
            if (pl->eval_host) HIP_TRY(hipHostFree(pl->eval_host));
            pl->eval_stage = pl->eval_host = nullptr;
            pl->eval_bytes = 0;
            const size_t want = std::max<size_t>(bytes, (size_t)8 << 20);
            HIP_TRY(hipMalloc((void **)&pl->eval_stage, want));
            HIP_TRY(hipHostMalloc((void **)&pl->eval_host, want, hipHostMallocDefault));
            pl->eval_bytes = want;
        }
    }
    char *stage = pl->eval_stage + pl->eval_cursor;
    char *hstage = pl->eval_host + pl->eval_cursor;
    pl->eval_cursor += bytes;
    pl->eval_stream = s;
    pl->eval_pending = true;
    for (int f = 0; f < nfact; ++f)
        memcpy(hstage + tables_at + offs[f] * 8, factors[f0 + f].host, (size_t)elems[f] * (factors[f0 + f].dtype == JTP_F32 ? 4 : 8));
    if (!fvars.empty()) memcpy(hstage + fvars_at, fvars.data(), fvars.size() * sizeof(JtEvalVar));
    std::vector<int> grid((size_t)nlists, 0);
    for (int k = 0; k < nlists; ++k) {
        memcpy(hstage + task_at[k], tasks[k].data(), tasks[k].size() * sizeof(JtEvalTask));
        int32_t *bs = reinterpret_cast<int32_t *>(hstage + blk_at[k]);
        int64_t at = 0;
        for (size_t t = 0; t < tasks[k].size(); ++t) {
            bs[t] = (int32_t)at;
            at += (tasks[k][t].n_rows + JT_EVAL_ROWS - 1) / JT_EVAL_ROWS;
        }
        bs[tasks[k].size()] = (int32_t)at;
        if (at > INT32_MAX) return set_err(JTP_EUNSUPPORTED, "too many rows in one evaluate call");
        grid[k] = (int)at;
    }
    HIP_TRY(hipMemcpyAsync(stage, hstage, bytes, hipMemcpyHostToDevice, s));
    const int lds = lds_doubles * 8;
    for (int k = 0; k < nlists; ++k) {
        if (grid[k] == 0) continue;
        const JtEvalTask *dt = reinterpret_cast<const JtEvalTask *>(stage + task_at[k]);
        const int32_t *bs = reinterpret_cast<const int32_t *>(stage + blk_at[k]);
        const JtEvalVar *fvp = reinterpret_cast<const JtEvalVar *>(stage + fvars_at);
        if (k >= npass) hipLaunchKernelGGL((jt_eval_batch<double>), dim3(grid[k]), dim3(256), lds, s, dt, bs, (int)tasks[k].size(), fvp, (const char *)(stage + tables_at), b.fix);
        else if (hp.dtype == JTP_F32) hipLaunchKernelGGL((jt_eval_batch<float>), dim3(grid[k]), dim3(256), lds, s, dt, bs, (int)tasks[k].size(), fvp, (const char *)(stage + tables_at), (float *)b.psi);
        else hipLaunchKernelGGL((jt_eval_batch<double>), dim3(grid[k]), dim3(256), lds, s, dt, bs, (int)tasks[k].size(), fvp, (const char *)(stage + tables_at), (double *)b.psi);
    }
    HIP_TRY(hipGetLastError());
    return JTP_OK;                        // (the caller's tables were copied to pinned memory above)
}

int jtp_set_potential_product(jtp_plan *pl, int32_t batch, int32_t clique, int32_t n_factors, const jtp_factor *factors) {
    if (n_factors < 0 || (n_factors > 0 && !factors)) return set_err(JTP_EINVAL, "bad factor list");
    const int32_t off[2] = {0, n_factors};
    return jtp_set_potential_products(pl, batch, 1, &clique, off, factors);
}

static uint64_t host_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int jtp_fill_synthetic(jtp_plan *pl, int32_t batch, uint64_t seed, const double *scale) {
    int rc = check_ready(pl, batch);
    if (rc) return rc;
    rc = check_writable(pl, batch);
    if (rc) return rc;
    pl->psi_dirty = true;
    HostPlan &hp = pl->hp;
    HIP_TRY(hipSetDevice(hp.device));
    hipStream_t s = pl->streams[batch % pl->streams.size()];
    BatchBuffers &b = pl->bufs[batch];
    for (int c = 0; c < hp.n_cliques; ++c) {
        if (!(hp.pn[c].owner == hp.rank || hp.pn[c].owner == hp.n_ranks)) continue;
        if (hp.pn[c].unit && hp.pn[c].stat < 0) continue;          // all ones, nothing stored
        const JtPackDesc &d = hp.pn[c].unit ? hp.stat_pack[c] : hp.pack[c];
        const uint64_t key = host_splitmix64(seed * 0x100000001B3ull + (uint64_t)c);
        const double sc = scale ? scale[c] : 1.0;
        const int64_t n = d.phys_elems;
        const int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
        if (hp.pn[c].unit)       // (the static table: the same counter-based values over the covered shape)
            hipLaunchKernelGGL((jt_pack<double, double, 1>), dim3(grid), dim3(256), 0, s, d, (const double *)nullptr, b.fix, key, sc);
        else if (hp.dtype == JTP_F32)
            hipLaunchKernelGGL((jt_pack<float, float, 1>), dim3(grid), dim3(256), 0, s, d, (const float *)nullptr, (float *)b.psi, key, sc);
        else
            hipLaunchKernelGGL((jt_pack<double, double, 1>), dim3(grid), dim3(256), 0, s, d, (const double *)nullptr, (double *)b.psi, key, sc);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return JTP_OK;
}

// ------------------------------------------------------------------------------------------ compute

static int launch_variant(jtp_plan *pl, int variant, int nblocks, int lds, hipStream_t s, const JtTask *tasks,
                          const JtBlock *blocks, const int *itab, void *psi, void *bel, double *msg, const JtFlow &fl) {
    if (pl->hp.dtype == JTP_F32) {
        auto f = KernelTable<float>::get(variant, mixk(pl->hp));
        hipLaunchKernelGGL(f, dim3(nblocks), dim3(JT_THREADS), lds, s, tasks, blocks, itab, (const float *)psi, (float *)bel, msg, fl);
    } else {
        auto f = KernelTable<double>::get(variant, mixk(pl->hp));
        hipLaunchKernelGGL(f, dim3(nblocks), dim3(JT_THREADS), lds, s, tasks, blocks, itab, (const double *)psi, (double *)bel, msg, fl);
    }
    return JTP_OK;
}

// The chunks whose own digits do not exist (a digit beyond a variable's cardinality, a padding bit set) are not in the block
// lists of a single-set plan (HostPlan::init_blocks): whatever their incoming messages, all they would write is their partial
// copies of the outgoing messages, all zeros.  Those zeros are written HERE, once per arena half, on `s`, after the arena was set
// to "unwritten": the entries carry no marker from then on (nobody re-arms them), every propagate finds them written.
__global__ __launch_bounds__(256) void jt_zero_copies(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk, double *__restrict__ msg,
                                                      int64_t cur_off, int64_t set_stride) {
    const JtBlock &bk = blk[blockIdx.x];
    const JtTask &tk = tasks[bk.task];
    msg += (int64_t)blockIdx.y * set_stride;               // (multi-set plans: one arena per evidence set)
    for (int j = 0; j < tk.n_out; ++j) {
        const JtMsg &m = tk.msg[JT_MAX_IN + j];
        // (where a workgroup's flush puts entry s of its sub-box: jt_pass / jt_mpass, "flush outgoing sub-boxes")
        const int64_t at = cur_off + m.off + (int64_t)bk.pnum[j] * m.pstride + bk.gbase[JT_MAX_IN + j];
        const int n = 1 << m.nfree;
        for (int s = threadIdx.x; s < n; s += 256) {
            uint32_t idx = 0;
            for (int b = 0; b < m.nfree; ++b) idx += (((uint32_t)s >> b) & 1u) << m.free_pos[b];
            msg[at + idx] = 0.0;
        }
    }
}

// (`msg`, `nsets`: one evidence set's arena, or - multi-set plans - all of them, set_stride doubles apart; `halves`: bit h = arena half h)
static int zero_padding(jtp_plan *pl, double *msg, int nsets, hipStream_t s, int halves) {
    const HostPlan &hp = pl->hp;
    const int64_t half = std::max<int64_t>(hp.msg_doubles, 2);
    for (int m = 0; m < 2; ++m) {
        if (hp.init_blocks[m].empty() || !pl->d_init[m]) continue;
        for (int h = 0; h < 2; ++h)
            if ((halves >> h) & 1)
                hipLaunchKernelGGL(jt_zero_copies, dim3((unsigned)hp.init_blocks[m].size(), (unsigned)nsets), dim3(256), 0, s, pl->d_tasks, pl->d_init[m], msg,
                                   h ? half : (int64_t)0, pl->set_stride);
    }
    HIP_TRY(hipGetLastError());
    return JTP_OK;
}

// Called wherever the host has just synchronised with the plan's streams.  A dataflow launch whose
// workgroups gave up waiting (it would take workgroups dispatched out of order, or a stuck device;
// never observed) has left that propagate unfinished: mark the whole arena unwritten again, switch the
// plan to one launch per level for good, and run the affected evidence sets again that way.
// `synced`: the evidence set whose stream the caller has just synchronised (-1: all of them).
static int check_flow(jtp_plan *pl, int synced = -1) {
    if (!pl->host_abort) return JTP_OK;
    if (*(volatile uint32_t *)pl->host_abort == 0) {
        // only what has actually finished is known to be good (sets sharing the stream finished with it)
        for (size_t i = 0; i < pl->bufs.size(); ++i)
            if (synced < 0 || i % pl->streams.size() == (size_t)synced % pl->streams.size()) pl->bufs[i].unchecked = false;
        leave_flight(pl);
        return JTP_OK;
    }
    *(volatile uint32_t *)pl->host_abort = 0;
    pl->flow = false;
    pl->flow_fallbacks++;
    if (pl->hp.n_ranks > 1) {
        // the other ranks have moved on with whatever this rank sent them: no local repair is possible
        for (auto s : pl->streams) (void)hipStreamSynchronize(s);
        for (auto &b : pl->bufs) b.unchecked = false;
        leave_flight(pl);
        return set_err(JTP_EHIP, "a dataflow launch of rank %d timed out waiting for a message (is the GPU shared with other "
                                 "work? then set JTP_FLOW_TICKETS=1); the results of this propagate are invalid on every rank; "
                                 "this plan launches per level from now on", pl->hp.rank);
    }
    for (auto s : pl->streams) HIP_TRY(hipStreamSynchronize(s));
    // only the sets that are run again lose their messages: a set whose propagate was already checked keeps
    // its arena (its separator beliefs are read from there), and no later launch of this plan waits on markers
    if (pl->multiset) {                                     // (all sets run together, the padding sets of the last group too)
        const size_t mbytes = (size_t)std::max<int64_t>(pl->hp.msg_doubles, 2) * 16;
        // (on the plan's stream, like the zeros that follow: that stream does not synchronise with the null stream)
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)pl->msg_all, (int)(uint32_t)(JT_UNWRITTEN & 0xffffffffu), mbytes * pl->n_groups * JT_MSETS / 4, pl->streams[0]));
        if (int rc = zero_padding(pl, pl->msg_all, pl->n_groups * JT_MSETS, pl->streams[0])) return rc;
        HIP_TRY(hipStreamSynchronize(pl->streams[0]));
        for (auto &b : pl->bufs) b.epoch = 0, b.flow_runs = 0, b.ticket_runs = 0;
    } else
    for (size_t i = 0; i < pl->bufs.size(); ++i) {
        BatchBuffers &b = pl->bufs[i];
        if (!b.unchecked) continue;
        HIP_TRY(hipMemsetAsync(b.sync, 0, (size_t)pl->hp.sync_words * 4, pl->streams[0]));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)b.msg, (int)(uint32_t)(JT_UNWRITTEN & 0xffffffffu), (size_t)std::max<int64_t>(pl->hp.msg_doubles, 2) * 4, pl->streams[0]));
        if (int rc = zero_padding(pl, b.msg, 1, pl->streams[0])) return rc;
        HIP_TRY(hipStreamSynchronize(pl->streams[0]));
        b.epoch = 0;
        b.flow_runs = 0;
        b.ticket_runs = 0;
    }
    if (pl->multiset) {
        bool any = false;
        for (auto &b : pl->bufs) any = any || b.unchecked, b.unchecked = false;
        HIP_TRY(hipMemset(pl->sync_all, 0, (size_t)pl->hp.sync_words * 4 * pl->n_groups));
        if (any) {
            int rc = jtp_propagate(pl, 0, pl->hp.n_batch);
            if (rc) return rc;
        }
    } else
    for (size_t i = 0; i < pl->bufs.size(); ++i) {
        if (!pl->bufs[i].unchecked) continue;
        pl->bufs[i].unchecked = false;
        int rc = jtp_propagate(pl, (int32_t)i, (int32_t)i + 1);
        if (rc) return rc;
    }
    for (auto s : pl->streams) HIP_TRY(hipStreamSynchronize(s));
    leave_flight(pl);
    return JTP_OK;
}

// Before anything is read out: if a dataflow propagate of this evidence set has not been checked yet, wait
// for it and look at the abort flag FIRST, so that a propagate that had to be run again per level is run
// again before the read-out kernels copy anything (they used to copy the aborted propagate's data).
static int settle(jtp_plan *pl, int batch) {
    if (!pl->bufs[batch].unchecked) return JTP_OK;
    HIP_TRY(hipStreamSynchronize(pl->streams[batch % pl->streams.size()]));
    return check_flow(pl, batch);
}

int jtp_debug_set(jtp_plan *pl, const char *knob, int64_t value) {
    if (!pl || !knob) return set_err(JTP_EINVAL, "null argument");
    if (!strcmp(knob, "flow_debug")) pl->flow_debug = (uint32_t)value;      // fault injection (tests): see JtFlow::dbg
    else if (!strcmp(knob, "flow")) pl->flow = value != 0 && !pl->hp.segments.empty();
    else return set_err(JTP_EINVAL, "unknown knob %s", knob);
    return JTP_OK;
}

int jtp_set_evidence(jtp_plan *pl, int32_t batch, int32_t n, const int32_t *var_ids, const int32_t *states) {
    int rc = check_ready(pl, batch);
    if (rc) return rc;
    HostPlan &hp = pl->hp;
    if (n < 0 || (n > 0 && (!var_ids || !states))) return set_err(JTP_EINVAL, "null argument");
    BatchBuffers &b = pl->bufs[batch];
    std::vector<uint32_t> ev(2 * hp.pn.size(), 0u);
    std::vector<char> seen(hp.n_vars, 0);
    for (int i = 0; i < n; ++i) {
        const int v = var_ids[i];
        if (v < 0 || v >= hp.n_vars) return set_err(JTP_EINVAL, "evidence %d: variable %d out of range", i, v);
        if (states[i] < 0 || states[i] >= hp.card[v]) return set_err(JTP_EINVAL, "evidence %d: state %d of variable %d (cardinality %d)", i, states[i], v, hp.card[v]);
        if (seen[v]) return set_err(JTP_EINVAL, "variable %d observed twice", v);
        seen[v] = 1;
        // the indicator goes into ONE clique that contains the variable: the first in the caller's
        // numbering (every rank makes the same choice; the owner applies it)
        int host = -1;
        for (int c = 0; c < hp.n_cliques && host < 0; ++c)
            for (int u : hp.pn[c].vars)
                if (u == v) host = c;
        if (host < 0) return set_err(JTP_EINVAL, "variable %d is in no clique", v);
        const PNode &p = hp.pn[host];
        for (size_t j = 0; j < p.vars.size(); ++j)
            if (p.vars[j] == v) {
                ev[2 * host] |= ((1u << p.nb[j]) - 1u) << p.pos[j];
                ev[2 * host + 1] |= (uint32_t)states[i] << p.pos[j];
            }
    }
    HIP_TRY(hipSetDevice(hp.device));
    hipStream_t s = pl->streams[batch % pl->streams.size()];
    HIP_TRY(hipStreamSynchronize(s));                      // a propagate in flight may still read the old table
    if (!b.ev) HIP_TRY(hipMalloc((void **)&b.ev, ev.size() * sizeof(uint32_t)));      // (multi-set plans: a slice of ev_all)
    HIP_TRY(hipMemcpy(b.ev, ev.data(), ev.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    b.ev_any = n > 0;
    if (pl->multiset) {
        // a group of evidence sets may sum the elements of a vector before the message product on a clique while none of ITS
        // sets observes a variable on that clique's element bits (JtTask::esum_groups; bit b stands for the groups g = b mod 64)
        const int iset = pl->set0 + batch;                       // the set's place in the allocation (group 0: evidence-free sets)
        std::copy(ev.begin(), ev.end(), pl->ev_host.begin() + (size_t)iset * pl->ev_stride);
        if (pl->set0) pl->act_dirty = true;                       // (the tasks' active lists follow the evidence: rebuilt by the next propagate)
        const uint32_t emask = (1u << hp.EB) - 1u;
        const int bit = (iset / JT_MSETS) & 63;
        std::vector<char> on_e(hp.pn.size(), 0);
        const size_t nsets = pl->ev_host.size() / pl->ev_stride;
        for (size_t sidx = 0; sidx < nsets; ++sidx) {
            if ((int)((sidx / JT_MSETS) & 63) != bit) continue;
            for (size_t p = 0; p < hp.pn.size(); ++p)
                if (pl->ev_host[sidx * pl->ev_stride + 2 * p] & emask) on_e[p] = 1;
        }
        const bool always = hp.knobs.esum_always != 0;             // (timing experiment: wrong results)
        for (size_t t = 0; t < hp.tasks.size(); ++t) {
            JtTask &tk = hp.tasks[t];
            if (tk.kind != 0 || !(tk.esum & 1)) continue;
            const uint64_t want = (on_e[tk.pnode] && !always) ? tk.esum_groups & ~(1ull << bit) : tk.esum_groups | (1ull << bit);
            if (want != tk.esum_groups) {
                tk.esum_groups = want;
                tk.esum = 1 | (want == ~0ull ? 2 : 0);
                pl->esum_dirty = true;                               // uploaded in one copy by the next jtp_propagate
            }
        }
    }
    return JTP_OK;
}

// dynamic LDS of a multi-set launch: the ring plus one region per evidence set of the group (reduce tasks: none)
static int multiset_lds(const HostPlan &hp, const Launch &L) {
    int lds = 0;
    for (int t : L.tasks) lds = std::max(lds, hp.tasks[t].kind == 0 ? hp.tasks[t].lds_bytes : 0);
    return lds;
}

// Multi-set plans with an evidence-free set (round 6): which evidence sets every task serves.  The upward message of a clique below
// which a set observes NOTHING is the evidence-free one, whatever the set observes elsewhere; with 16 observations per set on the
// width-20 tree that is four collect tasks in five, per SET - round 5 skipped a task only where all eight sets of a fixed group
// agreed, one in three.  So the sets of a workgroup are no longer "group g" but entries 8 g .. 8 g + 7 of the TASK's list:
//   collect task of clique c (and its reduce task): arena slot 0 - the evidence-free set - and every caller's set with an observed
//     variable in the subtree below c;
//   downward task: every caller's slot (and the padding slots behind them, which exist: the last group as before).
// A consumer stages an upward message of slot s from s's own arena where s is on the producer's list, from slot 0 where it is not
// (JtFlow::skip = member) - and so does the read-out (readout_redirect): nobody copies slot 0's messages into the other sets' arenas
// (round 5 and the first form of this round did, behind every propagate: 7 % of a 64-set step).  The entries of a (task, slot) off the
// lists stay "unwritten" in both arena halves; those of a pair that LEAVES a list are set back to that, once, here.
static int rebuild_active(jtp_plan *pl, hipStream_t s) {
    const HostPlan &hp = pl->hp;
    const int cap = pl->n_groups * JT_MSETS, set0 = pl->set0, S = hp.n_batch;
    const size_t nt = hp.tasks.size(), np = hp.pn.size();
    const std::vector<uint8_t> was = pl->member_host;       // the lists of the last propagate (empty: none yet)
    pl->member_host.assign(nt * cap, 0);
    pl->act_ids_host.assign(nt * cap, 0);
    pl->act_n_host.assign(nt, 0);
    pl->esum_oct_host.assign(nt * (size_t)pl->n_groups, 0);
    // below[slot * np + p]: the set in that slot observes a variable hosted by clique p or by a clique below it
    std::vector<uint8_t> below((size_t)cap * np, 0);
    std::vector<int> order(np);
    for (size_t p = 0; p < np; ++p) order[p] = (int)p;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return hp.pn[a].depth > hp.pn[b].depth; });
    for (int slot = set0; slot < set0 + S; ++slot) {
        uint8_t *bl = &below[(size_t)slot * np];
        const uint32_t *ev = &pl->ev_host[(size_t)slot * pl->ev_stride];
        for (size_t p = 0; p < np; ++p) bl[p] = ev[2 * p] != 0;
        for (int p : order)
            if (bl[p] && hp.pn[p].parent >= 0) bl[hp.pn[p].parent] = 1;
    }
    auto put = [&](int t, const std::vector<uint16_t> &list) {
        if (t < 0) return;
        pl->act_n_host[t] = (int32_t)list.size();
        for (size_t j = 0; j < list.size(); ++j) {
            pl->act_ids_host[(size_t)t * cap + j] = list[j];
            pl->member_host[(size_t)t * cap + list[j]] = 1;
        }
    };
    std::vector<uint16_t> everyone;
    for (int slot = set0; slot < cap; ++slot) everyone.push_back((uint16_t)slot);
    // What leaves a list is reset: the entries of a (collect task, slot) that was on the task's list for the last propagate and is not
    // now hold that propagate's values - in the halves' turn the task would find them "written" when the slot comes back (its reduce
    // task sums the partial copies it finds without a marker).  Both halves of such messages, partial copies included, are marked
    // "unwritten" ONCE, here; entries of pairs that stay off the lists are never read (consumers and read-out go to slot 0) nor written.
    std::vector<JtFanout> fan;
    auto reset = [&](int64_t off, int64_t count, const std::vector<uint16_t> &slots) {
        for (size_t i = 0; i < slots.size(); i += JT_MSETS) {
            JtFanout f;
            memset(&f, 0, sizeof f);
            f.off = off, f.count = (int32_t)count, f.flags = JT_FANOUT_RESET;
            for (int j = 0; j < JT_MSETS; ++j) f.slot[j] = i + j < slots.size() ? slots[i + j] : (uint16_t)0xffffu;
            fan.push_back(f);
        }
    };
    for (size_t p = 0; p < np; ++p) {
        const PNode &pn = hp.pn[p];
        if (pn.collect_task >= 0) {
            std::vector<uint16_t> list(1, (uint16_t)0), left;
            for (int slot = set0; slot < set0 + S; ++slot) {
                const bool on = below[(size_t)slot * np + p] != 0;
                if (on) list.push_back((uint16_t)slot);
                else if (!was.empty() && was[(size_t)pn.collect_task * cap + slot]) left.push_back((uint16_t)slot);
            }
            put(pn.collect_task, list);
            const PSep &sp = hp.ps[pn.psep];
            put(sp.up_red_task, list);
            if (!left.empty()) {
                reset(sp.up_roff, ((int64_t)sp.up_rnpart) << sp.nbits, left);
                if (sp.up_red_task >= 0) reset(sp.up_off, ((int64_t)sp.up_npart) << sp.nbits, left);
            }
        }
    }
    for (const PSep &sp : hp.ps) {
        put(sp.dn_task, everyone);
        put(sp.dn_red_task, everyone);
    }
    const uint32_t emask = (1u << hp.EB) - 1u;
    for (size_t t = 0; t < nt; ++t) {
        const JtTask &tk = hp.tasks[t];
        if (tk.kind != 0 || !(tk.esum & 1)) continue;
        const int n = pl->act_n_host[t];
        for (int g = 0; g * JT_MSETS < n; ++g) {
            bool free_e = true;
            for (int j = g * JT_MSETS; j < std::min(n, (g + 1) * JT_MSETS); ++j)
                if (pl->ev_host[(size_t)pl->act_ids_host[t * cap + j] * pl->ev_stride + 2 * tk.pnode] & emask) free_e = false;
            pl->esum_oct_host[t * (size_t)pl->n_groups + g] = (free_e || hp.knobs.esum_always) ? 1 : 0;
        }
    }
    if ((int)fan.size() > pl->cap_fanout) {
        if (pl->d_fanout) HIP_TRY(hipFree(pl->d_fanout));
        pl->d_fanout = nullptr;
        pl->cap_fanout = 0;
        HIP_TRY(hipMalloc((void **)&pl->d_fanout, fan.size() * sizeof(JtFanout)));
        pl->cap_fanout = (int)fan.size();
    }
    HIP_TRY(hipMemcpyAsync(pl->d_member, pl->member_host.data(), pl->member_host.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(pl->d_act_ids, pl->act_ids_host.data(), pl->act_ids_host.size() * sizeof(uint16_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(pl->d_act_n, pl->act_n_host.data(), pl->act_n_host.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(pl->d_esum_oct, pl->esum_oct_host.data(), pl->esum_oct_host.size(), hipMemcpyHostToDevice, s));
    if (!fan.empty()) HIP_TRY(hipMemcpyAsync(pl->d_fanout, fan.data(), fan.size() * sizeof(JtFanout), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));                    // (the sources are host vectors)
    pl->n_fanout = (int)fan.size();
    pl->act_dirty = false;
    if (pl->n_fanout > 0) {
        JtFlow fl;
        memset(&fl, 0, sizeof fl);
        fl.set_stride = pl->set_stride;
        fl.oth_off = std::max<int64_t>(hp.msg_doubles, 2);          // (the second half starts here: the pass marks both)
        hipLaunchKernelGGL(jt_multi_fanout, dim3(pl->n_fanout), dim3(256), 0, s, pl->d_fanout, pl->msg_all, fl);
        HIP_TRY(hipGetLastError());
        // (the marks cover the partial copies of chunks that do not exist, which nobody writes again: set back to their zeros)
        if (pl->d_init[0] || pl->d_init[1])
            if (int rc = zero_padding(pl, pl->msg_all, pl->n_groups * JT_MSETS, s)) return rc;
    }
    return JTP_OK;
}

int jtp_propagate(jtp_plan *pl, int32_t batch_begin, int32_t batch_end) {
    int rc = check_ready(pl, batch_begin);
    if (rc) return rc;
    HostPlan &hp = pl->hp;
    if (batch_end <= batch_begin || batch_end > hp.n_batch) return set_err(JTP_EINVAL, "bad batch range [%d,%d)", batch_begin, batch_end);
    HIP_TRY(hipSetDevice(hp.device));
    roctx::Range whole(pl->roctx, pl->multiset ? "jtp_propagate (multi-set: collect + distribute)" : "jtp_propagate (collect + distribute)");
    if ((hp.flags & JTP_SHARE_POTENTIALS) && pl->psi_dirty) {
        HIP_TRY(hipStreamSynchronize(pl->streams[0]));        // the shared tables were written on stream 0
        pl->psi_dirty = false;
    }
    if (pl->multiset) {
        if (batch_begin != 0 || batch_end != hp.n_batch)
            return set_err(JTP_EINVAL, "a multi-set plan propagates all its evidence sets together: pass [0, %d)", hp.n_batch);
        if (pl->prof_per_launch) return set_err(JTP_EINVAL, "per-launch profiling is not available for multi-set plans");
        hipStream_t s = pl->streams[0];
        const bool prof = pl->prof_steps > 0 && (pl->prof_calls++ % (unsigned)pl->prof_stride) == 0;
        if (prof) while (pl->ev.size() < 3 * (size_t)pl->prof_steps) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            pl->ev.push_back(e);
        }
        const size_t ev_base = prof ? 3 * (size_t)(pl->prof_cursor % pl->prof_steps) : 0;
        const int64_t half = std::max<int64_t>(hp.msg_doubles, 2);
        for (auto &bb : pl->bufs) bb.epoch++;
        BatchBuffers &b0 = pl->bufs[0];
        JtFlow fl;
        memset(&fl, 0, sizeof fl);
        fl.sync = pl->sync_all;
        fl.host_abort = pl->host_abort;
        fl.cur_off = b0.cur_off(half);
        fl.oth_off = half - fl.cur_off;                      // the kernel waits on markers in every launch mode
        fl.dbg = pl->flow_debug;
        fl.ev = pl->ev_all;
        fl.set_stride = pl->set_stride;
        fl.ev_stride = pl->ev_stride;
        fl.sync_stride = (uint32_t)hp.sync_words;
        if (pl->act_dirty) {
            if (int rc2 = rebuild_active(pl, s)) return rc2;
        }
        fl.skip = pl->d_member;
        fl.act_ids = pl->d_act_ids;
        fl.act_n = pl->d_act_n;
        fl.esum_oct = pl->d_esum_oct;
        fl.cap = (uint32_t)(pl->n_groups * JT_MSETS);
        fl.n_tasks = (uint32_t)hp.tasks.size();
        if (pl->esum_dirty) {
            // which groups may sum a vector's elements first on which task (jtp_set_evidence): the fields of ALL tasks in one
            // strided copy, ordered before the launches below on the plan's stream
            HIP_TRY(hipMemcpy2DAsync(&pl->d_tasks[0].esum_groups, sizeof(JtTask), &hp.tasks[0].esum_groups, sizeof(JtTask), sizeof(uint64_t),
                                     hp.tasks.size(), hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpy2DAsync(&pl->d_tasks[0].esum, sizeof(JtTask), &hp.tasks[0].esum, sizeof(JtTask), sizeof(int32_t),
                                     hp.tasks.size(), hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));                    // (the source is the plan's own task table: pageable)
            pl->esum_dirty = false;
        }
        const bool flow = pl->flow;
        if (flow) {
            b0.flow_runs++;
            for (auto &bb : pl->bufs) bb.unchecked = true;
        }
        const bool others = flow ? enter_flight(pl) : false;
        const bool tickets = (hp.flags & JTP_FLOW_TICKETS) != 0 || pl->env_tickets || others;
        pl->launch_mode = flow ? (tickets ? 2 : 1) : 0;
        if (flow) pl->flow_propagates += hp.n_batch, pl->tickets_used += tickets ? hp.n_batch : 0;
        const uint32_t ticket_run = b0.ticket_runs;            // ticket-ordered runs before this one
        if (flow && tickets) b0.ticket_runs++;
        bool mid_done = false;
        if (prof) HIP_TRY(hipEventRecord(pl->ev[ev_base + 0], s));
        auto launch = [&](int phase, int64_t blk_off, int nblocks, int lds, int ticket_idx, uint32_t ticket_base) {
            fl.ticket_idx = ticket_idx >= 0 ? (uint32_t)ticket_idx : 0xffffffffu;
            fl.ticket_base = ticket_base;
            fl.blk_base = (uint32_t)blk_off;
            fl.n_groups = (uint32_t)pl->n_groups;
            fl.n_blocks = (uint32_t)nblocks;
            // (1-D grid: eight records of group 0, the same eight of group 1, ... - see jt_multi_flow)
            const unsigned grid = (unsigned)((nblocks + 7) / 8) * 8u * (unsigned)pl->n_groups;
            if (hp.dtype == JTP_F32)
                hipLaunchKernelGGL(jt_multi_flow<float>, dim3(grid), dim3(JT_THREADS), lds, s, pl->d_tasks, pl->d_blocks + blk_off,
                                   pl->d_itab, (const float *)b0.psi, (float *)b0.bel, pl->msg_all, fl);
            else
                hipLaunchKernelGGL(jt_multi_flow<double>, dim3(grid), dim3(JT_THREADS), lds, s, pl->d_tasks, pl->d_blocks + blk_off,
                                   pl->d_itab, (const double *)b0.psi, (double *)b0.bel, pl->msg_all, fl);
        };
        for (const Step &st : (flow ? hp.flow_steps : hp.steps)) {
            if (st.kind != 0) continue;
            const int phase = flow ? hp.segments[st.first].phase : hp.launches[st.first].phase;
            if (prof && !mid_done && phase == 1) {
                HIP_TRY(hipEventRecord(pl->ev[ev_base + 1], s));
                mid_done = true;
            }
            if (flow) {
                const Segment &sg = hp.segments[st.first];
                int lds = 0;
                for (int i = sg.first_launch; i < sg.first_launch + sg.n_launch; ++i) lds = std::max(lds, multiset_lds(hp, hp.launches[i]));
                launch(sg.phase, sg.blk_off, sg.nblocks, lds, tickets ? sg.ticket_idx : -1, ticket_run * (uint32_t)sg.nblocks);
            } else {
                const Launch &L = hp.launches[st.first];
                launch(L.phase, L.blk_off, L.nblocks, multiset_lds(hp, L), -1, 0u);
            }
        }
        if (prof) {
            if (!mid_done) HIP_TRY(hipEventRecord(pl->ev[ev_base + 1], s));
            HIP_TRY(hipEventRecord(pl->ev[ev_base + 2], s));
            pl->prof_cursor++;
        }
        HIP_TRY(hipGetLastError());
        return JTP_OK;
    }
    const bool prof = pl->prof_steps > 0 && (pl->prof_calls++ % (unsigned)pl->prof_stride) == 0;
    const size_t ev_per_step = pl->prof_per_launch ? 2 * hp.launches.size() : 3;
    if (prof) {
        size_t need = ev_per_step * (size_t)pl->prof_steps;
        while (pl->ev.size() < need) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            pl->ev.push_back(e);
        }
    }
    const size_t ev_base = prof ? ev_per_step * (size_t)(pl->prof_cursor % pl->prof_steps) : 0;
    for (int b = batch_begin; b < batch_end; ++b) {
        hipStream_t s = pl->streams[b % pl->streams.size()];
        BatchBuffers &bb = pl->bufs[b];
        const bool pb = prof && b == batch_begin;
        const bool per_launch = pb && pl->prof_per_launch;
        const bool per_phase = pb && !pl->prof_per_launch;
        bool mid_done = false;
        if (per_phase) HIP_TRY(hipEventRecord(pl->ev[ev_base + 0], s));
        const bool flow = pl->flow && !per_launch;
        const int64_t half = std::max<int64_t>(hp.msg_doubles, 2);
        bb.epoch++;
        JtFlow fl;
        memset(&fl, 0, sizeof fl);
        fl.sync = bb.sync;
        fl.host_abort = pl->host_abort;
        fl.cur_off = bb.cur_off(half);
        // (a plan that launches per level never waits on entries: it need not mark the other half)
        fl.oth_off = pl->flow ? half - fl.cur_off : -1;
        fl.dbg = pl->flow_debug;
        fl.ev = bb.ev_any ? bb.ev : nullptr;
        fl.fix_shift = bb.fix_shift(fl.cur_off);
        if (flow) {
            bb.flow_runs++;
            bb.unchecked = true;
        }
        // Several evidence sets = several dataflow kernels on the device at once.  In blockIdx order that can
        // deadlock: kernel A's waiting workgroups fill the XCD on which kernel B's lowest unfinished
        // workgroup should start, and the other way round (seen: --batch 4 hit the 2 s time-out).  A
        // ticket is drawn by a workgroup that is already running, so the lowest unfinished record of every
        // kernel is always being worked on, whatever else shares the device.
        // The same holds for two plans of one process whose propagates overlap (plan_for caches plans, each on
        // its own stream), hence tickets whenever another plan of this process has a dataflow propagate IN FLIGHT on
        // this device (round 2: whenever another plan existed).  The plan that was there first keeps blockIdx order:
        // the newcomer's ticket-ordered workgroups always make progress and drain, so it cannot be starved for good.
        // (JTP_FLOW_TICKETS=1 in the environment: for processes that share their GPU with other processes)
        const bool others = flow ? enter_flight(pl) : false;
        const bool tickets = (hp.flags & JTP_FLOW_TICKETS) != 0 || pl->streams.size() > 1 || pl->env_tickets || others;
        pl->launch_mode = flow ? (tickets ? 2 : 1) : 0;
        if (flow) pl->flow_propagates++, pl->tickets_used += tickets ? 1 : 0;
        const uint32_t ticket_run = bb.ticket_runs;            // ticket-ordered runs of this evidence set before this one
        if (flow && tickets) bb.ticket_runs++;
        for (const Step &st : (flow ? hp.flow_steps : hp.steps)) {
            if (st.kind == 0 && flow) {
                const Segment &sg = hp.segments[st.first];
                if (per_phase && !mid_done && sg.phase >= 1) {       // (a merged launch counts as the second phase)
                    HIP_TRY(hipEventRecord(pl->ev[ev_base + 1], s));
                    mid_done = true;
                }
                fl.ticket_idx = tickets ? (uint32_t)sg.ticket_idx : 0xffffffffu;
                fl.blk_base = (uint32_t)sg.blk_off;
                fl.ticket_base = ticket_run * (uint32_t)sg.nblocks;
                if (hp.dtype == JTP_F32)
                    hipLaunchKernelGGL(KernelTable<float>::get_flow(sg.phase, pl->chain, mixk(hp), pl->marg_tasks), dim3(sg.nblocks), dim3(JT_THREADS), sg.lds_bytes, s, pl->d_tasks,
                                       pl->d_blocks + sg.blk_off, pl->d_itab, (const float *)bb.psi, (float *)bb.bel, bb.msg, fl);
                else
                    hipLaunchKernelGGL(KernelTable<double>::get_flow(sg.phase, pl->chain, mixk(hp), pl->marg_tasks), dim3(sg.nblocks), dim3(JT_THREADS), sg.lds_bytes, s, pl->d_tasks,
                                       pl->d_blocks + sg.blk_off, pl->d_itab, (const double *)bb.psi, (double *)bb.bel, bb.msg, fl);
            } else if (st.kind == 0) {
                const Launch &L = hp.launches[st.first];
                if (per_phase && !mid_done && L.phase == 1) {
                    HIP_TRY(hipEventRecord(pl->ev[ev_base + 1], s));
                    mid_done = true;
                }
                if (per_launch) HIP_TRY(hipEventRecord(pl->ev[ev_base + 2 * st.first], s));
                fl.blk_base = (uint32_t)L.blk_off;
                launch_variant(pl, L.variant, L.nblocks, L.lds_bytes, s, pl->d_tasks, pl->d_blocks + L.blk_off, pl->d_itab, bb.psi, bb.bel, bb.msg, fl);
                if (per_launch) HIP_TRY(hipEventRecord(pl->ev[ev_base + 2 * st.first + 1], s));
            } else if (pl->fake_comm == 2) {
                // loop-back: the step's sends and receives as one RCCL group addressed to this rank itself (RCCL pairs the k-th
                // send to a peer with the k-th receive from it: a receive without a send of its own takes this rank's first
                // outgoing message, a send without a receive lands in a spare buffer) - the real cost of the group on this GPU,
                // without the wire
                std::vector<const CommOp *> sends, recvs;
                for (int i = st.first; i < st.first + st.count; ++i) (hp.comm[i].send ? sends : recvs).push_back(&hp.comm[i]);
                const size_t n = std::max(sends.size(), recvs.size());
                int64_t most = 0;
                for (int i = st.first; i < st.first + st.count; ++i) most = std::max(most, hp.comm[i].count);
                const int rc2 = ensure_stage(pl, (size_t)most * 8 * 2);
                if (rc2) return rc2;
                double *spare = (double *)pl->stage;
                NCCL_TRY(rccl::GroupStart());
                for (size_t k = 0; k < n; ++k) {
                    const CommOp *sd = k < sends.size() ? sends[k] : nullptr, *rv = k < recvs.size() ? recvs[k] : nullptr;
                    const int64_t cnt = rv ? rv->count : sd->count;
                    const double *src = sd && sd->count >= cnt ? bb.msg + fl.cur_off + sd->off : spare + most;
                    double *dst = rv ? bb.msg + fl.cur_off + rv->off : spare;
                    NCCL_TRY(rccl::Send(src, (size_t)cnt, rccl::ncclFloat64, rccl::comm_rank, rccl::comm, s));
                    NCCL_TRY(rccl::Recv(dst, (size_t)cnt, rccl::ncclFloat64, rccl::comm_rank, rccl::comm, s));
                }
                NCCL_TRY(rccl::GroupEnd());
            } else if (pl->fake_comm) {
                for (int i = st.first; i < st.first + st.count; ++i) {
                    const CommOp &op = hp.comm[i];
                    if (op.send) continue;
                    const int grid = (int)std::min<int64_t>((op.count + 255) / 256, 1024);
                    hipLaunchKernelGGL(jt_fill_value, dim3(grid), dim3(256), 0, s, bb.msg + fl.cur_off + op.off, op.count, 1.0);
                }
            } else {
                NCCL_TRY(rccl::GroupStart());
                for (int i = st.first; i < st.first + st.count; ++i) {
                    const CommOp &op = hp.comm[i];
                    if (op.send) NCCL_TRY(rccl::Send(bb.msg + fl.cur_off + op.off, (size_t)op.count, rccl::ncclFloat64, op.peer, rccl::comm, s));
                    else NCCL_TRY(rccl::Recv(bb.msg + fl.cur_off + op.off, (size_t)op.count, rccl::ncclFloat64, op.peer, rccl::comm, s));
                }
                NCCL_TRY(rccl::GroupEnd());
            }
        }
        if (per_phase) {
            if (!mid_done) HIP_TRY(hipEventRecord(pl->ev[ev_base + 1], s));
            HIP_TRY(hipEventRecord(pl->ev[ev_base + 2], s));
        }
        if (pb) pl->prof_cursor++;
    }
    HIP_TRY(hipGetLastError());
    return JTP_OK;
}

int jtp_sync(jtp_plan *pl) {
    if (!pl) return set_err(JTP_EINVAL, "null plan");
    if (!pl->device) return JTP_OK;
    HIP_TRY(hipSetDevice(pl->hp.device));
    for (auto s : pl->streams) HIP_TRY(hipStreamSynchronize(s));
    pl->eval_pending = false;
    pl->eval_cursor = 0;
    return check_flow(pl);
}

// ------------------------------------------------------------------------------------------ data out

// Multi-set plans with active lists (rebuild_active): the upward message of a (collect task, arena slot) that is NOT on the task's list
// exists in slot 0's arena only - nobody copies it into the set's own (round 5 did, after every propagate: 7 % of a 64-set step).  A
// read-out task of evidence set `batch` takes its inputs from that set's arena; an input formed by such a task has its offset moved
// back by the slot's distance, i.e. is read from slot 0.  `member_host` describes the lists the LAST propagate ran with.
static bool readout_redirect(const jtp_plan *pl, int batch, JtTask &tk) {
    if (!pl->multiset || pl->set0 == 0 || pl->member_host.empty()) return false;
    const size_t cap = (size_t)pl->n_groups * JT_MSETS, slot = (size_t)(pl->set0 + batch);
    bool any = false;
    for (int k = 0; k < tk.n_in; ++k) {
        const int t = tk.msg[k].src_task;
        if (t >= 0 && !pl->member_host[(size_t)t * cap + slot]) tk.msg[k].off -= (int64_t)slot * pl->set_stride, any = true;
    }
    return any;
}

int jtp_get_belief(jtp_plan *pl, int32_t batch, int32_t node, void *host, int32_t host_dtype) {
    int rc = check_ready(pl, batch);
    if (rc) return rc;
    HostPlan &hp = pl->hp;
    if (node < 0 || node >= hp.n_nodes) return set_err(JTP_EINVAL, "node %d out of range", node);
    if (host_dtype != JTP_F32 && host_dtype != JTP_F64) return set_err(JTP_EINVAL, "bad host dtype");
    HIP_TRY(hipSetDevice(hp.device));
    roctx::Range range(pl->roctx, "jtp_get_belief");
    rc = settle(pl, batch);
    if (rc) return rc;
    hipStream_t s = pl->streams[batch % pl->streams.size()];
    BatchBuffers &b = pl->bufs[batch];
    const size_t hsz = host_dtype == JTP_F32 ? 4 : 8;
    if (node < hp.n_cliques) {
        if (!(hp.pn[node].owner == hp.rank || hp.pn[node].owner == hp.n_ranks)) return set_err(JTP_EINVAL, "clique %d belongs to rank %d", node, hp.pn[node].owner);
        const JtPackDesc &d = hp.pack[node];
        rc = ensure_stage(pl, (size_t)d.host_elems * hsz);
        if (rc) return rc;
        const bool unit = hp.pn[node].unit;
        void *bel_src = b.bel;
        if (unit) {
            // a unit clique keeps no belief table either: formed now, into a scratch arena laid out as its table would be
            if (!pl->unit_scratch) {
                HIP_TRY(hipMalloc(&pl->unit_scratch, (size_t)hp.scratch_elems * pl->esize));
                HIP_TRY(hipMemsetAsync(pl->unit_scratch, 0, (size_t)hp.scratch_elems * pl->esize, s));
            }
            bel_src = pl->unit_scratch;
        }
        if (pl->multiset || unit) {
            // no belief tables are kept: form this clique's belief for this evidence set now, from the shared
            // table and the set's final messages (computation.py:216-224), into the scratch arena
            if (pl->belief_tasks.size() < hp.pn.size()) pl->belief_tasks.resize(hp.pn.size());
            jtp_plan::BeliefTask &bt = pl->belief_tasks[node];
            if (!bt.d_task) {
                JtTask tk;
                std::vector<int32_t> itab;
                std::vector<JtBlock> blocks;
                std::string err;
                rc = jtp_plan_belief_task(hp, node, tk, itab, blocks, err);
                if (rc) return set_err(rc, "%s", err.c_str());
                HIP_TRY(hipMalloc((void **)&bt.d_task, sizeof(JtTask)));
                HIP_TRY(hipMalloc((void **)&bt.d_blk, blocks.size() * sizeof(JtBlock)));
                HIP_TRY(hipMalloc((void **)&bt.d_tab, std::max<size_t>(itab.size(), 1) * sizeof(int32_t)));
                HIP_TRY(hipMemcpy(bt.d_task, &tk, sizeof tk, hipMemcpyHostToDevice));
                bt.h_task = tk;
                HIP_TRY(hipMemcpy(bt.d_blk, blocks.data(), blocks.size() * sizeof(JtBlock), hipMemcpyHostToDevice));
                if (!itab.empty()) HIP_TRY(hipMemcpy(bt.d_tab, itab.data(), itab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                bt.nblocks = (int)blocks.size();
                bt.lds = tk.lds_bytes;
            }
            if (pl->multiset && pl->set0) {              // (which inputs come from the evidence-free set's arena depends on the set)
                JtTask patched = bt.h_task;
                readout_redirect(pl, batch, patched);
                HIP_TRY(hipStreamSynchronize(s));        // (an earlier read-out's kernel may still read the record)
                HIP_TRY(hipMemcpy(bt.d_task, &patched, sizeof patched, hipMemcpyHostToDevice));
            }
            JtFlow one;
            memset(&one, 0, sizeof one);
            one.cur_off = b.cur_off(std::max<int64_t>(hp.msg_doubles, 2));
            one.oth_off = -1;
            one.ev = b.ev_any || pl->multiset ? b.ev : nullptr;
            one.fix_shift = b.fix_shift(one.cur_off);
            HIP_TRY(raise_lds(hp.dtype == JTP_F32 ? (const void *)KernelTable<float>::get(JT_K_SINGLE, mixk(hp)) : (const void *)KernelTable<double>::get(JT_K_SINGLE, mixk(hp)), bt.lds));
            launch_variant(pl, JT_K_SINGLE, bt.nblocks, bt.lds, s, bt.d_task, bt.d_blk, bt.d_tab, b.psi, bel_src, b.msg, one);
            HIP_TRY(hipGetLastError());
        }
        const int grid = (int)std::min<int64_t>((d.host_elems + 255) / 256, 4096);
        if (hp.dtype == JTP_F32) {
            if (host_dtype == JTP_F32) hipLaunchKernelGGL((jt_unpack<float, float>), dim3(grid), dim3(256), 0, s, d, (const float *)bel_src, (float *)pl->stage);
            else hipLaunchKernelGGL((jt_unpack<float, double>), dim3(grid), dim3(256), 0, s, d, (const float *)bel_src, (double *)pl->stage);
        } else {
            if (host_dtype == JTP_F32) hipLaunchKernelGGL((jt_unpack<double, float>), dim3(grid), dim3(256), 0, s, d, (const double *)bel_src, (float *)pl->stage);
            else hipLaunchKernelGGL((jt_unpack<double, double>), dim3(grid), dim3(256), 0, s, d, (const double *)bel_src, (double *)pl->stage);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host, pl->stage, (size_t)d.host_elems * hsz, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return check_flow(pl, batch);
    }
    const int si = hp.sep_of_node[node];
    if (si < 0) return set_err(JTP_EINVAL, "separator node %d is not part of the tree", node);
    const PSep &sp = hp.ps[si];
    if (sp.up_off < 0) return set_err(JTP_EINVAL, "separator node %d is not held by rank %d", node, hp.rank);
    JtPackDesc d;
    memset(&d, 0, sizeof d);
    d.nvars = (int)hp.node_vars[node].size();
    d.nbits = sp.nbits;
    int64_t stride = 1;
    for (int i = d.nvars - 1; i >= 0; --i) {
        const int v = hp.node_vars[node][i];
        int j = 0;
        while (sp.vars[j] != v) ++j;
        d.pos[i] = (uint8_t)sp.pos[j];
        d.nb[i] = (uint8_t)sp.nb[j];
        d.card[i] = hp.card[v];
        d.hstride[i] = stride;
        stride *= hp.card[v];
    }
    d.host_elems = stride;
    bitfield_desc(d);
    rc = ensure_stage(pl, (size_t)stride * hsz);
    if (rc) return rc;
    const int grid = (int)std::min<int64_t>((stride + 255) / 256, 4096);
    const int64_t pstride = (int64_t)1 << sp.nbits;
    const double *cur = b.msg + b.cur_off(std::max<int64_t>(hp.msg_doubles, 2));      // the half the last propagate wrote
    const double *cur_up = cur;
    if (pl->multiset && pl->set0 && !pl->member_host.empty() && sp.child >= 0 && hp.pn[sp.child].collect_task >= 0 &&
        !pl->member_host[(size_t)hp.pn[sp.child].collect_task * ((size_t)pl->n_groups * JT_MSETS) + (size_t)(pl->set0 + batch)])
        cur_up = pl->msg_all + b.cur_off(std::max<int64_t>(hp.msg_doubles, 2));       // (readout_redirect: the evidence-free set's upward message)
    if (host_dtype == JTP_F32)
        hipLaunchKernelGGL((jt_msg_unpack<float>), dim3(grid), dim3(256), 0, s, d, cur_up + sp.up_roff, sp.up_rnpart, cur + sp.dn_roff, sp.dn_rnpart, pstride, (float *)pl->stage);
    else
        hipLaunchKernelGGL((jt_msg_unpack<double>), dim3(grid), dim3(256), 0, s, d, cur_up + sp.up_roff, sp.up_rnpart, cur + sp.dn_roff, sp.dn_rnpart, pstride, (double *)pl->stage);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host, pl->stage, (size_t)stride * hsz, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return check_flow(pl, batch);
}

int jtp_get_marginals(jtp_plan *pl, int32_t batch, int32_t n, const int32_t *cliques, const int32_t *var_off,
                      const int32_t *var_ids, const int64_t *out_off, double *host);

// One marginal = a request list of one (its device tables are kept with the plan like any other list's:
// no allocation per call, nothing to leak on an error path).
int jtp_get_marginal(jtp_plan *pl, int32_t batch, int32_t clique, const int32_t *out_vars, int32_t n_out, double *host) {
    int rc = check_ready(pl, batch);
    if (rc) return rc;
    if (n_out < 0 || n_out > JT_MAX_VARS || (n_out > 0 && !out_vars) || !host) return set_err(JTP_EINVAL, "bad variable list");
    if (clique < 0 || clique >= pl->hp.n_cliques) return set_err(JTP_EINVAL, "node %d is not a clique", clique);
    int64_t elems = 1;
    for (int i = 0; i < n_out; ++i) {
        if (out_vars[i] < 0 || out_vars[i] >= pl->hp.n_vars) return set_err(JTP_EINVAL, "variable %d out of range", out_vars[i]);
        elems *= pl->hp.card[out_vars[i]];
    }
    const int32_t var_off[2] = {0, n_out};
    const int64_t out_off[2] = {0, elems};
    const int32_t none = 0;
    return jtp_get_marginals(pl, batch, 1, &clique, var_off, n_out > 0 ? out_vars : &none, out_off, host);
}

int jtp_get_marginals(jtp_plan *pl, int32_t batch, int32_t n, const int32_t *cliques, const int32_t *var_off,
                      const int32_t *var_ids, const int64_t *out_off, double *host) {
    int rc = check_ready(pl, batch);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!cliques || !var_off || !out_off || !host))) return set_err(JTP_EINVAL, "null argument");
    if (n == 0) return JTP_OK;
    if (n > 65535) {                                        // grid.y of the read-out launch
        for (int32_t i = 0; i < n; i += 65535) {
            rc = jtp_get_marginals(pl, batch, std::min(65535, n - i), cliques + i, var_off + i, var_ids, out_off + i, host);
            if (rc) return rc;
        }
        return JTP_OK;
    }
    HostPlan &hp = pl->hp;
    HIP_TRY(hipSetDevice(hp.device));
    roctx::Range range(pl->roctx, "jtp_get_marginals");
    rc = settle(pl, batch);
    if (rc) return rc;
    hipStream_t s = pl->streams[batch % pl->streams.size()];
    BatchBuffers &b = pl->bufs[batch];
    std::vector<int32_t> key;
    key.push_back(n);
    key.insert(key.end(), cliques, cliques + n);
    for (int i = 0; i <= n; ++i) key.push_back(var_off[i] - var_off[0]);
    key.insert(key.end(), var_ids + var_off[0], var_ids + var_off[n]);
    MargBatch *mb = nullptr;
    for (size_t i = 0; i < pl->marg_cache.size(); ++i)
        if (pl->marg_cache[i]->key == key) {                // most recently used last
            mb = pl->marg_cache[i];
            pl->marg_cache.erase(pl->marg_cache.begin() + i);
            pl->marg_cache.push_back(mb);
            break;
        }
    if (!mb) {
        std::vector<JtTask> tasks;
        std::vector<JtBlock> blocks, ublocks;              // passes over belief tables; passes of cliques that keep none
        std::vector<int32_t> itab;
        std::vector<JtMargDesc> descs((size_t)n);
        std::vector<int64_t> elems((size_t)n);
        int64_t scratch_doubles = 0, total_out = 0;
        int lds = 0, ulds = 0;
        // Requests on ONE clique share passes over its belief table, JT_MAX_OUT of them per pass (a pairwise model asks a
        // clique for two or three factor marginals: round 3 read the table once per request - config 3: 1831 reads of 878
        // tables, 2.1 x the bytes).  Multi-set plans marginalise psi x messages directly and keep one request per task.
        std::vector<char> lean_later;                        // per task: a unit clique's marginals (single-set plans)
        std::vector<std::vector<int>> groups;
        {
            std::map<int, int> open;                         // clique -> its group that still has room
            for (int i = 0; i < n; ++i) {
                const int clique = cliques[i];
                if (clique < 0 || clique >= hp.n_cliques) return set_err(JTP_EINVAL, "request %d: node %d is not a clique", i, clique);
                if (!(hp.pn[clique].owner == hp.rank || hp.pn[clique].owner == hp.n_ranks)) return set_err(JTP_EINVAL, "clique %d belongs to rank %d", clique, hp.pn[clique].owner);
                auto it = open.find(clique);
                // (multi-set plans: one request per pass; unit cliques of single-set plans share passes like everybody else)
                if (pl->multiset || it == open.end() || (int)groups[it->second].size() >= hp.knobs.marg_group) {
                    open[clique] = (int)groups.size();
                    groups.push_back(std::vector<int>());
                }
                groups[open[clique]].push_back(i);
            }
        }
        for (const std::vector<int> &grp : groups) {
            const int clique = cliques[grp[0]];
            std::vector<std::vector<int>> ovs;
            for (int i : grp) {
                const int n_out = var_off[i + 1] - var_off[i];
                if (n_out < 0 || n_out > JT_MAX_VARS) return set_err(JTP_EINVAL, "request %d: bad variable count", i);
                std::vector<int> ov(var_ids + var_off[i], var_ids + var_off[i + 1]);
                for (int a = 0; a < n_out; ++a) {
                    if (ov[a] < 0 || ov[a] >= hp.n_vars) return set_err(JTP_EINVAL, "request %d: variable %d out of range", i, ov[a]);
                    for (int c = 0; c < a; ++c)
                        if (ov[a] == ov[c]) return set_err(JTP_EINVAL, "request %d: variable %d requested twice", i, ov[a]);
                }
                ovs.push_back(ov);
            }
            JtTask tk;
            std::vector<int> out_bits, npart;
            std::vector<JtBlock> blk;
            std::vector<int32_t> tab;
            std::string err;
            const bool direct = pl->multiset || hp.pn[clique].unit;     // psi x incoming tables marginalised directly
            rc = jtp_plan_marginal_task(hp, clique, ovs, tk, tab, out_bits, npart, blk, err, direct);
            if (rc) return set_err(rc, "request %d: %s", grp[0], err.c_str());
            tk.itab_off = (int64_t)itab.size();
            if (tk.tmap_off >= 0) tk.tmap_off += tk.itab_off;      // (the clique's thread map travels behind the task's rows)
            itab.insert(itab.end(), tab.begin(), tab.end());
            for (JtBlock &bk : blk) {
                bk.task = (uint32_t)tasks.size();
                (direct ? ublocks : blocks).push_back(bk);
            }
            lean_later.push_back(direct && hp.pn[clique].unit && !pl->multiset);
            if (direct) ulds = std::max(ulds, tk.lds_bytes);
            else lds = std::max(lds, tk.lds_bytes);
            for (size_t j = 0; j < grp.size(); ++j) {
                const int i = grp[j];
                const std::vector<int> &ov = ovs[j];
                const int n_out = (int)ov.size();
                tk.msg[JT_MAX_IN + j].off = scratch_doubles;
                JtMargDesc md;
                memset(&md, 0, sizeof md);
                md.d.nvars = n_out;
                md.d.nbits = out_bits[j];
                int64_t stride = 1;
                int bit = 0;
                std::vector<int> pos(n_out);
                for (int a = n_out - 1; a >= 0; --a) {          // last requested variable = lowest bits
                    pos[a] = bit;
                    bit += hp.vbits[ov[a]];
                }
                for (int a = n_out - 1; a >= 0; --a) {
                    md.d.pos[a] = (uint8_t)pos[a];
                    md.d.nb[a] = (uint8_t)hp.vbits[ov[a]];
                    md.d.card[a] = hp.card[ov[a]];
                    md.d.hstride[a] = stride;
                    stride *= hp.card[ov[a]];
                }
                md.d.host_elems = stride;
                bitfield_desc(md.d);
                md.src_off = scratch_doubles;
                md.pstride = (int64_t)1 << out_bits[j];
                md.npart = npart[j];
                descs[i] = md;
                elems[i] = stride;
                scratch_doubles += md.pstride * npart[j];
            }
            tasks.push_back(tk);
        }
        for (int i = 0; i < n; ++i) {                          // results in request order
            descs[i].dst_off = total_out;
            total_out += elems[i];
        }
        // (round 6) marginals of unit cliques run the lean pass: the records are made once every output's place is known
        for (size_t t = 0; t < tasks.size(); ++t)
            if (lean_later[t]) jtp_make_lean(hp, tasks[t], itab, true);
        // (their workgroups first in the list of the cliques that keep no table: a launch of jt_lean_single, then jt_single for the rest)
        std::stable_partition(ublocks.begin(), ublocks.end(), [&](const JtBlock &bk) { return tasks[bk.task].lean_off > 0; });
        int n_lean_blocks = 0, lean_lds = 0;
        for (const JtBlock &bk : ublocks)
            if (tasks[bk.task].lean_off > 0) ++n_lean_blocks, lean_lds = std::max(lean_lds, tasks[bk.task].lds_bytes);
        mb = new MargBatch();
        mb->lean_nblocks = n_lean_blocks;
        mb->lean_lds = lean_lds;
        if (pl->multiset && pl->set0) mb->h_tasks = tasks;
        mb->key = key;
        mb->n = n;
        mb->nblocks = (int)blocks.size();
        mb->lds = lds;
        mb->unit_nblocks = (int)ublocks.size();
        mb->unit_lds = ulds;
        blocks.insert(blocks.end(), ublocks.begin(), ublocks.end());
        mb->total_out = total_out;
        mb->elems = elems;
        int64_t biggest = 1;
        for (int64_t e : elems) biggest = std::max(biggest, e);
        mb->max_grid_x = (int)std::min<int64_t>((biggest + 255) / 256, 64);
        hipError_t e = hipMalloc((void **)&mb->d_tasks, tasks.size() * sizeof(JtTask));
        if (e == hipSuccess) e = hipMalloc((void **)&mb->d_blocks, blocks.size() * sizeof(JtBlock));
        if (e == hipSuccess) e = hipMalloc((void **)&mb->d_itab, std::max<size_t>(itab.size(), 1) * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&mb->d_descs, descs.size() * sizeof(JtMargDesc));
        if (e == hipSuccess) e = hipMalloc((void **)&mb->scratch, (size_t)std::max<int64_t>(scratch_doubles, 1) * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&mb->stage, (size_t)std::max<int64_t>(total_out, 1) * 8);
        if (e == hipSuccess) e = hipMemcpy(mb->d_tasks, tasks.data(), tasks.size() * sizeof(JtTask), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(mb->d_blocks, blocks.data(), blocks.size() * sizeof(JtBlock), hipMemcpyHostToDevice);
        if (e == hipSuccess && !itab.empty()) e = hipMemcpy(mb->d_itab, itab.data(), itab.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(mb->d_descs, descs.data(), descs.size() * sizeof(JtMargDesc), hipMemcpyHostToDevice);
        // the plan's own list: where the folded tasks of the propagate leave these marginals
        if (e == hipSuccess && !hp.folded.empty() && key == hp.fold_key && !pl->multiset) {
            bool all = true;
            std::vector<JtMargDesc> fd = descs;
            for (int i = 0; i < n; ++i) {
                const bool direct = hp.pn[cliques[i]].unit;
                const HostPlan::FoldReq &fr = hp.folded[i];
                if (!direct) continue;                             // (a belief table: jt_marginals, as ever)
                if (fr.task < 0 || fr.out_bits != fd[i].d.nbits) {
                    all = false;
                    break;
                }
                fd[i].src_off = fr.off;
                fd[i].pstride = (int64_t)1 << fr.out_bits;
                fd[i].npart = fr.npart;
                fd[i].in_arena = 1;
            }
            if (all) {
                e = hipMalloc((void **)&mb->d_descs_fold, fd.size() * sizeof(JtMargDesc));
                if (e == hipSuccess) e = hipMemcpy(mb->d_descs_fold, fd.data(), fd.size() * sizeof(JtMargDesc), hipMemcpyHostToDevice);
                mb->folded = e == hipSuccess;
            }
        }
        if (e != hipSuccess) {
            mb->release();
            delete mb;
            return set_err(e == hipErrorOutOfMemory ? JTP_ENOMEM : JTP_EHIP, "marginal tables: %s", hipGetErrorString(e));
        }
        if (pl->marg_cache.size() >= 32) {                  // a model asks for a few lists (and Z); keep the last used
            pl->marg_cache.front()->release();
            delete pl->marg_cache.front();
            pl->marg_cache.erase(pl->marg_cache.begin());
        }
        pl->marg_cache.push_back(mb);
    }
    if (!mb->h_tasks.empty()) {                          // (multi-set plans with active lists: readout_redirect, per evidence set)
        std::vector<JtTask> patched = mb->h_tasks;
        for (JtTask &tk : patched) readout_redirect(pl, batch, tk);
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(mb->d_tasks, patched.data(), patched.size() * sizeof(JtTask), hipMemcpyHostToDevice));
    }
    // the kernels that are actually launched below must be allowed this much dynamic LDS
    if (mb->nblocks > 0)
        HIP_TRY(raise_lds(hp.dtype == JTP_F32 ? (const void *)KernelTable<float>::get(JT_K_MARGINALS, mixk(hp)) : (const void *)KernelTable<double>::get(JT_K_MARGINALS, mixk(hp)), mb->lds));
    if (mb->unit_nblocks > 0)
        HIP_TRY(raise_lds(hp.dtype == JTP_F32 ? (const void *)KernelTable<float>::get(JT_K_SINGLE, mixk(hp)) : (const void *)KernelTable<double>::get(JT_K_SINGLE, mixk(hp)), mb->unit_lds));
    JtFlow plain;
    memset(&plain, 0, sizeof plain);
    plain.oth_off = -1;
    // marginalise the BELIEF tables: each is the "potential" argument of a childless collect
    if (mb->nblocks > 0)
        launch_variant(pl, JT_K_MARGINALS, mb->nblocks, mb->lds, s, mb->d_tasks, mb->d_blocks, mb->d_itab, b.bel, b.bel, mb->scratch, plain);
    // Marginals the propagate formed itself (fold_marginals): valid when the last propagate of this evidence set ran them - a dataflow
    // launch whose distribute segment is jt_propagate_flow, or one launch per level - and the set observes nothing (a clique that hosts
    // an observed variable has no lean pass).  Then only the belief-table requests are computed here.
    bool use_fold = false;
    if (mb->folded && !b.ev_any && b.epoch > 0) {
        if (pl->launch_mode == 0) use_fold = true;
        else {
            use_fold = !hp.segments.empty();
            for (const Segment &sg : hp.segments)
                if (sg.phase == 1 && (pl->chain || hp.tmix || !flow_both())) use_fold = false;
        }
    }
    if (mb->unit_nblocks > 0 && !use_fold) {
        // cliques that keep no belief table (multi-set plans: all; else the unit cliques): psi * (the incoming tables)
        // marginalised directly - inputs from the set's message arena (and the fixed arena), outputs into the request list's
        // scratch buffer (JtFlow::out_shift)
        plain.cur_off = b.cur_off(std::max<int64_t>(hp.msg_doubles, 2));
        plain.ev = b.ev_any || pl->multiset ? b.ev : nullptr;
        plain.fix_shift = b.fix_shift(plain.cur_off);
        plain.out_shift = (int64_t)(((intptr_t)mb->scratch - (intptr_t)(b.msg + plain.cur_off)) / 8);
        // (round 6) the tasks with a lean record through jt_lean_single while the evidence set observes nothing
        const int n_lean = plain.ev == nullptr ? mb->lean_nblocks : 0;
        if (n_lean > 0) {
            HIP_TRY(raise_lds(hp.dtype == JTP_F32 ? (const void *)KernelTable<float>::get(JT_K_LEAN_SINGLE, 0) : (const void *)KernelTable<double>::get(JT_K_LEAN_SINGLE, 0), mb->lean_lds));
            launch_variant(pl, JT_K_LEAN_SINGLE, n_lean, mb->lean_lds, s, mb->d_tasks, mb->d_blocks + mb->nblocks, mb->d_itab, b.psi, b.bel, b.msg, plain);
        }
        if (mb->unit_nblocks > n_lean)
            launch_variant(pl, JT_K_SINGLE, mb->unit_nblocks - n_lean, mb->unit_lds, s, mb->d_tasks, mb->d_blocks + mb->nblocks + n_lean, mb->d_itab, b.psi, b.bel, b.msg, plain);
    }
    hipLaunchKernelGGL(jt_marg_unpack, dim3(mb->max_grid_x, mb->n), dim3(256), 0, s, use_fold ? mb->d_descs_fold : mb->d_descs, mb->scratch, mb->stage,
                       (const double *)(b.msg + b.cur_off(std::max<int64_t>(hp.msg_doubles, 2))));
    HIP_TRY(hipGetLastError());
    bool packed = true;
    for (int i = 0; i < n; ++i) packed = packed && out_off[i + 1] - out_off[i] == mb->elems[i];
    if (packed) {
        HIP_TRY(hipMemcpyAsync(host + out_off[0], mb->stage, (size_t)mb->total_out * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    } else {
        std::vector<double> tmp((size_t)mb->total_out);
        HIP_TRY(hipMemcpyAsync(tmp.data(), mb->stage, (size_t)mb->total_out * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        int64_t at = 0;
        for (int i = 0; i < n; ++i) {
            memcpy(host + out_off[i], tmp.data() + at, (size_t)mb->elems[i] * 8);
            at += mb->elems[i];
        }
    }
    return check_flow(pl, batch);
}

int jtp_get_z(jtp_plan *pl, int32_t batch, double *z) {
    if (!pl) return set_err(JTP_EINVAL, "null plan");
    if (pl->hp.pn[pl->hp.root].owner != pl->hp.rank && pl->hp.pn[pl->hp.root].owner != pl->hp.n_ranks)
        return set_err(JTP_EINVAL, "the root clique belongs to rank %d", pl->hp.pn[pl->hp.root].owner);
    return jtp_get_marginal(pl, batch, pl->hp.root, nullptr, 0, z);
}

// ------------------------------------------------------------------------------------------ instrumentation

int jtp_set_profiling(jtp_plan *pl, int32_t on) {
    if (!pl) return set_err(JTP_EINVAL, "null plan");
    pl->prof_steps = on > 0 ? std::min(on, 256) : 0;     // `on` = number of propagates to keep
    pl->prof_cursor = 0;
    pl->prof_calls = 0;
    return JTP_OK;
}

int jtp_set_profiling_stride(jtp_plan *pl, int32_t stride) {
    if (!pl) return set_err(JTP_EINVAL, "null plan");
    if (stride < 1) return set_err(JTP_EINVAL, "stride must be at least 1");
    pl->prof_stride = stride;
    pl->prof_calls = 0;
    pl->prof_cursor = 0;
    return JTP_OK;
}

int jtp_set_profiling_granularity(jtp_plan *pl, int32_t per_launch) {
    if (!pl) return set_err(JTP_EINVAL, "null plan");
    pl->prof_per_launch = per_launch != 0;
    pl->prof_cursor = 0;
    return JTP_OK;
}

// ONE event pair around a whole region of propagates (a benchmark's timed steps): the device time from the first launch of
// the region to the end of its last, nothing in between - per-propagate events cost 2-3 us of idle GPU each, and a span
// that contains them reads longer than the step it is meant to time.
int jtp_region_begin(jtp_plan *pl) {
    int rc = check_ready(pl, 0);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(pl->hp.device));
    for (int i = 0; i < 2; ++i)
        if (!pl->region_ev[i]) HIP_TRY(hipEventCreate(&pl->region_ev[i]));
    HIP_TRY(hipEventRecord(pl->region_ev[0], pl->streams[0]));
    pl->region_open = true;
    return JTP_OK;
}

int jtp_region_end(jtp_plan *pl, double *ms) {
    int rc = check_ready(pl, 0);
    if (rc) return rc;
    if (!pl->region_open || !ms) return set_err(JTP_EINVAL, "jtp_region_end without jtp_region_begin");
    HIP_TRY(hipSetDevice(pl->hp.device));
    HIP_TRY(hipEventRecord(pl->region_ev[1], pl->streams[0]));
    HIP_TRY(hipEventSynchronize(pl->region_ev[1]));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, pl->region_ev[0], pl->region_ev[1]));
    *ms = t;
    pl->region_open = false;
    return JTP_OK;
}

int jtp_get_stats(jtp_plan *pl, jtp_stats *st) {
    if (!pl || !st) return set_err(JTP_EINVAL, "null argument");
    HostPlan &hp = pl->hp;
    memset(st, 0, sizeof *st);
    st->struct_size = (int32_t)sizeof(jtp_stats);
    const bool flow = pl->flow && !pl->prof_per_launch;
    st->n_launches = (int32_t)(flow ? hp.segments.size() : hp.launches.size());
    st->n_messages = hp.n_messages;
    st->n_tasks = (int32_t)hp.tasks.size();
    // multi-set plans: a table is read once per GROUP of evidence sets, messages once per set
    st->algorithmic_bytes = hp.alg_bytes;
    if (pl->multiset) {
        // what THIS engine streams: a table once per pass and GROUP of evidence sets that runs the pass (group 0 - the evidence-free
        // sets - included; a (task, group) whose subtree meets no evidence copies group 0's message and streams nothing), messages per set
        double tb = 0;
        for (const Launch &L : hp.launches) {
            if (L.variant != JT_K_MULTI_COLLECT && L.variant != JT_K_MULTI_DISTRIBUTE) continue;
            for (int t : L.tasks) {
                const PNode &p = hp.pn[hp.tasks[t].pnode];
                const double table = hp.tasks[t].kind == 0 && p.real >= 0 ? (double)hp.pack[p.real].host_elems * pl->esize : 0.0;
                tb += table * (pl->act_n_host.empty() ? pl->n_groups : (pl->act_n_host[t] + JT_MSETS - 1) / JT_MSETS);
            }
        }
        st->algorithmic_bytes = tb + hp.alg_msg_bytes * hp.n_batch;
    }
    st->flow_fallbacks = pl->flow_fallbacks;
    st->launch_mode = pl->launch_mode;
    st->tickets_used = pl->tickets_used;
    st->flow_propagates = pl->flow_propagates;
    st->device_bytes = pl->device_bytes;
    st->storage_dtype = hp.dtype;
    st->foreign_seen = pl->foreign_seen;
    {
        const board::Board &bd = board::g_board[hp.device & 63];
        st->flight_board = !bd.tried ? -1 : (bd.slots && bd.mine >= 0 ? 1 : 0);
    }
    st->algorithmic_bytes_full = hp.alg_bytes_full;
    st->fixed_bytes = (double)hp.fix_doubles * 8;
    st->lean_refused = hp.lean_refused.empty() ? 0 : 1;
    for (int c = 0; c < hp.n_cliques; ++c) {
        st->n_unit_cliques += hp.pn[c].unit ? 1 : 0;
        st->n_static_tables += hp.pn[c].unit && hp.pn[c].stat >= 0 ? 1 : 0;
    }
    if (pl->multiset) {
        const int groups = pl->n_groups;
        // float64 operations of the element loop of jt_mpass, per thread and table row (VEC elements), G = JT_MSETS sets:
        //   elements summed first (JtTask::esum == 3): VEC - 1 additions, then per set (n_in - 1) multiplications and one
        //   fused multiply-add;  no message on the element bits: per set (n_in - 1) multiplications and VEC fused multiply-adds;
        //   else per set and element n_in multiplications and one fused multiply-add
        const int VEC = hp.VEC;
        double flops = 0, insts = 0;
        for (const Launch &L : hp.launches) {
            if (L.variant != JT_K_MULTI_COLLECT && L.variant != JT_K_MULTI_DISTRIBUTE) continue;
            for (int t : L.tasks) {
                const JtTask &tk = hp.tasks[t];
                if (tk.kind != 0) continue;
                bool edep = false;
                for (int k = 0; k < tk.n_in; ++k) edep = edep || tk.msg[k].e_dep != 0;
                const double nin1 = std::max(tk.n_in - 1, 0);
                const double rows = (double)JT_THREADS * (double)tk.total * (double)(1u << tk.nF);
                const int runs = pl->act_n_host.empty() ? pl->n_groups : (pl->act_n_host[t] + JT_MSETS - 1) / JT_MSETS;
                for (int g = 0; g < runs; ++g) {
                    double per, ins;
                    const bool sum_first = pl->act_n_host.empty() ? ((tk.esum_groups >> (g & 63)) & 1ull) != 0 : pl->esum_oct_host[(size_t)t * pl->n_groups + g] != 0;
                    if ((tk.esum & 1) && sum_first && tk.setb <= JT_SETB_SMALL)
                        per = (VEC - 1) + JT_MSETS * (nin1 + 2.0), ins = (VEC - 1) + JT_MSETS * (nin1 + 1.0);
                    else if (!edep) per = JT_MSETS * (nin1 + 2.0 * VEC), ins = JT_MSETS * (nin1 + VEC);
                    else per = JT_MSETS * VEC * (tk.n_in + 2.0), ins = JT_MSETS * VEC * (tk.n_in + 1.0);
                    flops += per * rows;
                    insts += ins * rows;
                }
            }
        }
        st->f64_flops = flops;
        st->f64_insts = insts;
        for (const Launch &L : hp.launches) {
            if (L.variant != JT_K_MULTI_COLLECT && L.variant != JT_K_MULTI_DISTRIBUTE) continue;
            double tb = 0, mb = 0;
            for (int t : L.tasks) {
                const PNode &p = hp.pn[hp.tasks[t].pnode];
                const double table = p.real >= 0 ? (double)hp.pack[p.real].host_elems * pl->esize : 0.0;
                tb += table;
            }
            mb = L.alg_bytes - tb;
            st->kernel_bytes[L.variant] += tb * groups + mb * hp.n_batch;
        }
        for (const Segment &sg : hp.segments) st->kernel_launches[sg.phase == 0 ? JT_K_MULTI_COLLECT : JT_K_MULTI_DISTRIBUTE] += flow ? 1 : sg.n_launch;
    } else if (flow) {
        for (const Segment &sg : hp.segments) {
            // (the kernel that actually runs: KernelTable::get_flow)
            const int v = sg.phase == 0 ? JT_K_COLLECT_FLOW : (sg.phase == 1 && (pl->chain || hp.tmix || !flow_both()) ? JT_K_DISTRIBUTE_FLOW : JT_K_BOTH_FLOW);
            for (int i = sg.first_launch; i < sg.first_launch + sg.n_launch; ++i) st->kernel_bytes[v] += hp.launches[i].alg_bytes;
            st->kernel_launches[v] += 1;
        }
    } else {
        for (size_t i = 0; i < hp.launches.size(); ++i) {
            const Launch &L = hp.launches[i];
            st->kernel_bytes[L.variant] += L.alg_bytes;
            st->kernel_launches[L.variant] += 1;
        }
    }
    if (pl->device && pl->prof_steps > 0 && pl->prof_cursor > 0) {
        HIP_TRY(hipSetDevice(hp.device));
        const int kept = std::min(pl->prof_cursor, pl->prof_steps);
        if (pl->prof_per_launch) {
            for (int k = 0; k < kept; ++k) {
                const size_t base = 2 * hp.launches.size() * (size_t)k;
                for (size_t i = 0; i < hp.launches.size(); ++i) {
                    const Launch &L = hp.launches[i];
                    HIP_TRY(hipEventSynchronize(pl->ev[base + 2 * i + 1]));
                    float ms = 0;
                    HIP_TRY(hipEventElapsedTime(&ms, pl->ev[base + 2 * i], pl->ev[base + 2 * i + 1]));
                    st->kernel_ms[L.variant] += ms / kept;      // mean per propagate
                    if (L.phase == 0) st->collect_ms += ms / kept;
                    else st->distribute_ms += ms / kept;
                }
            }
        } else {
            for (int k = 0; k < kept; ++k) {
                const size_t base = 3 * (size_t)k;
                HIP_TRY(hipEventSynchronize(pl->ev[base + 2]));
                float c = 0, d = 0;
                HIP_TRY(hipEventElapsedTime(&c, pl->ev[base + 0], pl->ev[base + 1]));
                HIP_TRY(hipEventElapsedTime(&d, pl->ev[base + 1], pl->ev[base + 2]));
                st->collect_ms += c / kept;
                st->distribute_ms += d / kept;
            }
            // with one kernel per phase (the default), the phase time is that kernel's time over
            // its back-to-back launches (gaps included)
            if (pl->multiset) {
                st->kernel_ms[JT_K_MULTI_COLLECT] = st->collect_ms;
                st->kernel_ms[JT_K_MULTI_DISTRIBUTE] = st->distribute_ms;
            } else if (flow) {
                bool merged = false;
                for (const Segment &sg : hp.segments) merged = merged || sg.phase == 2;
                if (merged && hp.segments.size() == 1) {
                    st->kernel_ms[JT_K_BOTH_FLOW] = st->collect_ms + st->distribute_ms;       // one launch: the whole propagate
                } else if (merged) {
                    // (sharded plans: a collect launch, the exchange, then the merged launch)
                    st->kernel_ms[JT_K_COLLECT_FLOW] = st->collect_ms;
                    st->kernel_ms[JT_K_BOTH_FLOW] = st->distribute_ms;
                } else {
                    st->kernel_ms[JT_K_COLLECT_FLOW] = st->collect_ms;
                    st->kernel_ms[(pl->chain || hp.tmix || !flow_both()) ? JT_K_DISTRIBUTE_FLOW : JT_K_BOTH_FLOW] = st->distribute_ms;
                }
            } else if (!(hp.flags & JTP_SPLIT_VARIANTS)) {
                st->kernel_ms[JT_K_COLLECT_LEVEL] = st->collect_ms;
                st->kernel_ms[JT_K_DISTRIBUTE_LEVEL] = st->distribute_ms;
            }
        }
    }
    return JTP_OK;
}

int jtp_debug_read_msg(jtp_plan *pl, int32_t batch, int64_t off, int64_t n, double *host) {
    int rc = check_ready(pl, batch);
    if (rc) return rc;
    if (off < 0 || n < 0 || off + n > pl->hp.msg_doubles) return set_err(JTP_EINVAL, "range outside the message arena");
    HIP_TRY(hipSetDevice(pl->hp.device));
    HIP_TRY(hipMemcpy(host, pl->bufs[batch].msg + off, (size_t)n * 8, hipMemcpyDeviceToHost));
    return JTP_OK;
}

int jtp_get_launch_ms(jtp_plan *pl, double *out, int32_t n) {
    if (!pl) return set_err(JTP_EINVAL, "null plan");
    HostPlan &hp = pl->hp;
    const int nl = (int)hp.launches.size();
    if (!(pl->device && pl->prof_steps > 0 && pl->prof_cursor > 0 && pl->prof_per_launch))
        return set_err(JTP_EINVAL, "per-launch profiling is off or nothing was recorded");
    HIP_TRY(hipSetDevice(hp.device));
    const int kept = std::min(pl->prof_cursor, pl->prof_steps);
    for (int i = 0; i < nl && i < n; ++i) out[i] = 0.0;
    for (int k = 0; k < kept; ++k) {
        const size_t base = 2 * hp.launches.size() * (size_t)k;
        for (int i = 0; i < nl && i < n; ++i) {
            HIP_TRY(hipEventSynchronize(pl->ev[base + 2 * i + 1]));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, pl->ev[base + 2 * i], pl->ev[base + 2 * i + 1]));
            out[i] += ms / kept;
        }
    }
    return nl;
}

// ------------------------------------------------------------------------------------------ multi-GPU

int jtp_comm_unique_id(void *id128) {
    int rc = rccl::load();
    if (rc) return rc;
    rccl::ncclUniqueId id;
    NCCL_TRY(rccl::GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return JTP_OK;
}

int jtp_comm_init(int32_t rank, int32_t n_ranks, const void *id128, int32_t device) {
    int rc = rccl::load();
    if (rc) return rc;
    if (rccl::comm) return set_err(JTP_ECOMM, "communicator already initialised");
    HIP_TRY(hipSetDevice(device));
    rccl::ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    NCCL_TRY(rccl::CommInitRank(&rccl::comm, n_ranks, id, rank));
    rccl::comm_rank = rank;
    rccl::comm_size = n_ranks;
    return JTP_OK;
}

// What the communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice; -1 where the library has no such
// entry point): a multi-rank benchmark line carries it, so that the reader sees RCCL saw N ranks.
int jtp_comm_info(int32_t *n_ranks, int32_t *rank, int32_t *device) {
    if (!rccl::comm) return set_err(JTP_ECOMM, "communicator not initialised");
    int v = -1;
    if (n_ranks) *n_ranks = (rccl::CommCount && rccl::CommCount(rccl::comm, &v) == rccl::ncclSuccess) ? v : -1;
    v = -1;
    if (rank) *rank = (rccl::CommUserRank && rccl::CommUserRank(rccl::comm, &v) == rccl::ncclSuccess) ? v : -1;
    v = -1;
    if (device) *device = (rccl::CommCuDevice && rccl::CommCuDevice(rccl::comm, &v) == rccl::ncclSuccess) ? v : -1;
    return JTP_OK;
}

int jtp_comm_selftest(int32_t n) {
    if (!rccl::comm) return set_err(JTP_ECOMM, "communicator not initialised");
    if (n <= 0) return set_err(JTP_EINVAL, "n must be positive");
    double *a = nullptr, *b = nullptr;
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void **)&a, (size_t)n * 8));
    HIP_TRY(hipMalloc((void **)&b, (size_t)n * 8));
    std::vector<double> h(n), back(n, -1.0);
    for (int i = 0; i < n; ++i) h[i] = 0.5 * i + 1.0;
    HIP_TRY(hipMemcpyAsync(a, h.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(b, 0, (size_t)n * 8, s));
    NCCL_TRY(rccl::GroupStart());
    NCCL_TRY(rccl::Send(a, (size_t)n, rccl::ncclFloat64, rccl::comm_rank, rccl::comm, s));
    NCCL_TRY(rccl::Recv(b, (size_t)n, rccl::ncclFloat64, rccl::comm_rank, rccl::comm, s));
    NCCL_TRY(rccl::GroupEnd());
    HIP_TRY(hipMemcpyAsync(back.data(), b, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    (void)hipFree(a);
    (void)hipFree(b);
    (void)hipStreamDestroy(s);
    for (int i = 0; i < n; ++i)
        if (back[i] != h[i]) return set_err(JTP_ECOMM, "self send/recv mismatch at %d: %g vs %g", i, back[i], h[i]);
    return JTP_OK;
}

int jtp_comm_destroy(void) {
    if (rccl::comm) {
        NCCL_TRY(rccl::CommDestroy(rccl::comm));
        rccl::comm = nullptr;
    }
    return JTP_OK;
}

}  // extern "C"
