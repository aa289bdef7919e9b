// CDNA4 (gfx950) kernels of the junction-tree message-passing hot path.
//
// One kernel body, `jt_pass`, covers the reference's per-clique work in both traversal
// directions (junctiontree/computation.py):
//   collect    (:47-96)   up[S_p]     = sum_{C \ S_p} psi * prod_k up_k
//   distribute (:140-224) down_k[S_k] = sum_{C \ S_k} psi * down_p * prod_{j != k} up_j
//                         belief[C]   = psi * down_p * prod_k up_k
// i.e. the einsum call sites K1..K5 and K7 of SURVEY.md section 2.1 fused into ONE pass
// over the clique table per phase.  The divide-out of `remove_message` (:99-136) does not
// exist here: the all-but-one products are formed directly.
//
// Bound: HBM.  Per clique element the kernel reads sizeof(T) bytes (collect) or reads and
// writes sizeof(T) (distribute); everything else (messages) lives in LDS.  See DESIGN.md.
//
// Work decomposition (jtp_internal.h): a workgroup of 256 threads owns one chunk (fixed F
// bits) of one clique.  Thread t loads VEC consecutive elements (16 bytes) at
//   x = xF + xA(a) + xR(r) + t*VEC
// so a wave reads 1 KiB contiguous per instruction.  Incoming messages are staged once per
// workgroup into LDS as the sub-box this chunk can touch (partial copies summed on the
// way); per element the message entry is one ds_read at  thread_offset + uniform_offset.
// Outgoing sums are kept in VEC registers per message over the R loop, then reduced
// in-thread (e bits), by wave shuffles (lane bits) and through the LDS sub-box (wave bits
// and A loop), and written once per workgroup as one partial copy.  No float atomics: results
// are bit-reproducible.
//
// Launch structure: by default ONE launch per phase (jt_collect_flow / jt_distribute_flow) over the
// block list of all tree levels; a workgroup that finds a message entry still marked unwritten
// waits for it (FLOW = true below).  The same body runs once per tree level in jt_*_level and once
// per level and clique shape in jt_collect<> / jt_distribute<> (timing aids, and the fallback).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "jtp_internal.h"

#ifndef JT_MIX_WAVES
#define JT_MIX_WAVES 4          // waves per SIMD the dataflow *_mix kernels are compiled for (3: no spills, and 6-16 % slower on cardinality 3 / 5 trees, 3 % faster on cardinality 6 - A/B on one box)
#endif
#ifndef JT_UT
#define JT_UT 4                 // ... of a mixed-radix plan (rows gathered into registers, *_mix kernels): 4 or 8 (8: 2-4 % slower, A/B on one box)
#endif
#ifndef JT_U
#define JT_U 4                  // loop iterations whose element loads are in flight
#endif

template <typename T> struct JtVec;
template <> struct JtVec<float> { using type = float4; };
template <> struct JtVec<double> { using type = double2; };

// LDS-DMA: 16 bytes per lane from global memory straight into LDS at `lds_dst` + lane * 16
// (wave-uniform byte address).  Not tracked by the compiler's s_waitcnt insertion: callers count
// vmcnt themselves (cdna_hip_programming.md section 5.7).
// `reused` (uniform): the row will be read again soon by workgroups of the same XCD - multi-set plans of several groups of
// evidence sets, whose grid puts the groups' workgroups for one record next to each other on one XCD: default cache policy
// there (16 sets 1.78 -> 1.73 ms, 64 sets 6.09 -> 6.04 ms, 8 sets = one group unchanged; A/B on one box); everywhere else a row is read once per pass: non-temporal.
__device__ __forceinline__ void jt_dma16(const void *gsrc, uint32_t lds_dst, bool reused = false) {
    unsigned keep;
    if (reused) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_dst)
                     : "memory");
        return;
    }
#ifndef JT_TABLE_NO_NT       // non-temporal policy on the table stream (rows are read once per phase, by one CU): config 4 0.610 ->
                            // 0.5975 ms, A/B on one box over three runs each; -DJT_TABLE_NO_NT builds the default-policy loads
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
    return;
#endif
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void jt_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ double jt_shfl_xor(double v, int laneMask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, laneMask, 64);
    hi = __shfl_xor(hi, laneMask, 64);
    return __hiloint2double(hi, lo);
}

// v of lane + N (N = 1, 2, 4, 8; inside a row of 16 lanes, 0.0 beyond it) without a trip through the LDS crossbar: two DPP moves.
// A sum over lane bit b only has to arrive in the lanes whose bit b is clear (the lanes that store it), and lane + 2^b is their
// partner of the butterfly: same operands, same order, same double as acc += shfl_xor(acc, 2^b) there.
template <int N>
__device__ __forceinline__ double jt_row_down(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x100 + N, 0xf, 0xf, true);       // row_shl:N
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x100 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// acc[e] += the partner's acc[e] over every lane bit of red_lane, all elements together (no per-element branch, one wait per bit)
template <int VEC>
__device__ __forceinline__ void jt_lane_sums(double (&a)[VEC], const int red_lane) {
    if (red_lane & 1) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] += jt_row_down<1>(a[e]);
    }
    if (red_lane & 2) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] += jt_row_down<2>(a[e]);
    }
    if (red_lane & 4) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] += jt_row_down<4>(a[e]);
    }
    if (red_lane & 8) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] += jt_row_down<8>(a[e]);
    }
#pragma unroll
    for (int b = 4; b < 6; ++b) {
        if ((red_lane >> b) & 1) {
            double t[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) t[e] = jt_shfl_xor(a[e], 1 << b);
#pragma unroll
            for (int e = 0; e < VEC; ++e) a[e] += t[e];
        }
    }
}

// Diagnostic time stamps (builds with -DJT_STAMPS only: `python junction-tree_amd/build.py --out stamps.so -DJT_STAMPS`, plans
// made with JTP_DEBUG=2): lane 0 of a workgroup stores the 100 MHz clock at stage boundaries into its JT_NSTAMP slots of the
// time-stamp region (JtTask::dbg_off).  In the product build the macro is empty: no registers, no branches, no stores.
// Slots: 0 entry, 1 first element loads issued, 2 end of the first staging attempt, 3 staged, 4 constants read,
// 5-8 after loop steps 0-3, 9 loop done, 10 epilogues done (barrier), 11 flush stores issued, 12 flush stores retired,
// 13 staging attempts (a count, not a time), 14 end of the last wait for a producer (0: never waited).
#define JT_NSTAMP 16
#ifdef JT_STAMPS
#define JT_STAMP_AT(slot, value) do { if ((dbg & 2) && threadIdx.x == 0) stamp_out[slot] = (double)(value); } while (0)
#else
#define JT_STAMP_AT(slot, value) do { } while (0)
#endif
#define JT_STAMP(slot) JT_STAMP_AT(slot, __builtin_amdgcn_s_memrealtime())

// message-index bit of sub-box index bit b (free_pos[] packed four per word)
#define JT_FPOS(fp, b) (((fp)[(b) >> 2] >> (8 * ((b) & 3))) & 0xffu)

// Dataflow launches (FLOW = true): ONE launch covers all levels of a phase, so a message may still be
// in the making (by workgroups of the same launch) when its consumer starts.  Every message entry
// carries its own "ready" state: the arena half of this propagate holds JT_UNWRITTEN markers until the
// producer stores the value (64-bit stores are single-copy atomic), so the consumer loads its sub-box,
// and while it finds a marker it waits on that entry and loads again.  Loads and stores of message
// entries are agent-scope atomics here (they go through to memory: the L2 caches of the eight XCDs
// are not coherent with each other inside a launch); no fences, no counters.
template <int N, typename F>
__device__ __forceinline__ void jt_static_for(F &&f) {
    if constexpr (N > 0) {
        jt_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ bool jt_unwritten(double v) { return (uint64_t)__double_as_longlong(v) == JT_UNWRITTEN; }

template <bool FLOW>
__device__ __forceinline__ double jt_msg_load(const double *p) {
    if constexpr (FLOW)
        return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else
        return *p;
}

// (FLOW) through to memory only where the producer may be running in this launch (JtMsg::same_launch)
template <bool FLOW>
__device__ __forceinline__ double jt_msg_load(const double *p, bool through) {
    if constexpr (FLOW) return through ? jt_msg_load<true>(p) : *p;
    else return *p;
}

template <bool FLOW>
__device__ __forceinline__ void jt_msg_store(double *p, double v) {
    if constexpr (FLOW)
        __hip_atomic_store(reinterpret_cast<long long *>(p), __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}

// (FLOW) From the second staging attempt on - the producer is known to be flushing: the polled entry has just
// arrived - a thread that finds one of ITS entries still unwritten loads that entry again for up to ~1.5 us
// instead of sending the whole workgroup round the poll-and-reload loop once more (measured: the loop then
// ends at the second attempt nearly always).
template <bool FLOW>
__device__ __forceinline__ double jt_msg_settle(const double *p, double v, bool through, int attempt) {
#ifndef JT_NO_SETTLE                 // (A/B builds: python junction-tree_amd/build.py --out x.so -DJT_NO_SETTLE)
    if constexpr (FLOW) {
        if (attempt > 0 && through) {
            for (int spins = 0; spins < 24 && jt_unwritten(v); ++spins) {
                __builtin_amdgcn_s_sleep(2);
                v = jt_msg_load<true>(p);
            }
        }
    }
#endif
    return v;
}


// Where entry i of a workgroup's sub-box of a message lives in the message: the sub-box index is a bit-deposit of i into the
// message's index (bit b of i lands on message bit free_pos[b]).  The deposit is linear over disjoint bit groups, so the flush
// splits it once per outgoing message into the part of a thread's own low eight index bits (jt_sub_lo: a vector value) and a
// 32-row table of the bits above them, row j in lane j (jt_sub_hi, read with v_readlane): an entry's address costs one add.
// (Round 3 also rebuilt the STAGING loop on this split, every message in lock step: fewer instructions by far, and slower on
//  every config - 2 % on config 4, 1.5 % on config 3, 12 % on config 2, A/B on one box - so staging keeps the loop of round 2.)
__device__ __forceinline__ uint32_t jt_sub_lo(const uint32_t (&fp)[4], int nfree, int i) {
    uint32_t g = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b)
        if (b < nfree) g += (((uint32_t)i >> b) & 1u) << JT_FPOS(fp, b);
    return g;
}
__device__ __forceinline__ uint32_t jt_sub_hi(const uint32_t (&fp)[4], int nfree, int lane) {
    uint32_t g = 0;
#pragma unroll
    for (int b = 8; b < JT_MAX_FREE; ++b)
        if (b < nfree) g += (((uint32_t)lane >> (b - 8)) & 1u) << JT_FPOS(fp, b);
    return g;
}

// UNIT (JtTask::unit): the clique keeps NO table - every entry that exists counts as 1 (a clique without factors, a virtual
// clique of the binarisation, or a clique whose factors cover few of its variables: their product is then one of the incoming
// tables, JtMsg::fixed, staged like a message).  No row is loaded and no element ring is kept: which entries of a row exist
// says the clique's thread map (read once), which rows exist the iteration table; the belief is stored only where the task
// names a place for it (read-out tasks).  The reference never materialises such axes either (junctiontree.py:52-61).
// BEL (unit tasks only): the belief is stored - by read-out tasks; the passes of a propagate leave it out at compile time (the
// conversions and the store were a tenth of the distribute step's vector instructions).
template <typename T, int NIN, int NOUT, int MODE, bool FLOW = false, bool EARLY_FLUSH = (MODE == 0), bool TMIX = false, bool KEEP = false, bool UNIT = false, bool BEL = !UNIT, bool VG = false>
__device__ __forceinline__ void jt_pass(const JtTask &tk, const JtBlock &bk, const int *__restrict__ itab,
                                        const T *__restrict__ psi_arena, T *__restrict__ bel_arena,
                                        double *__restrict__ msg_arena, const JtFlow &fl,
                                        uint32_t bindex, uint32_t *flow_ctl = nullptr, uint64_t t_entry = 0) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int EB = (VEC == 4) ? 2 : 1;
    constexpr int NMSG = NIN + NOUT;
    constexpr int NPAR = NIN - NOUT * (MODE == 1);   // distribute: leading inputs that are not children
    constexpr int U = JT_U;                          // element loads in flight per wave
    // (mixed-radix rows are gathered into registers, a few hundred bytes each; JT_UT = 8 - twice the rows in flight - was tried: slower)
    constexpr int UT = TMIX ? JT_UT : U;
    static_assert((U == 4 || U == 8) && (1 << JT_MIN_ITER_LOG2) == 4 && JT_RING_BYTES == U * 4096, "the loop groups below are written out for four or eight slots");
    using VT = typename JtVec<T>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // Mixed-radix rows, compact form (JtTask::vgroups == 2, round 5): at most 128 of the 256 logical threads of the thread part own
    // an entry that exists (cardinality 3: 81), so two waves serve a row and the workgroup works on TWO rows at a time - threads
    // 0..127 on rows 0, 2, 4, ..., threads 128..255 on rows 1, 3, 5, ...; which logical thread a thread stands for says the clique's
    // list behind its thread map (vt).  Everything that indexes by the thread part (message look-ups, thread map, evidence) uses
    // vt; staging and flush - plain copies - keep the physical thread; sums over thread-part variables go through LDS adds in wave
    // order (the lanes of a logical variable are no longer an XOR apart).
    // (VG: instantiated by the *_mix kernels for such tasks only - jt_collect_mix / jt_distribute_mix)
    static_assert(!VG || TMIX, "row groups are a form of mixed-radix rows");
    constexpr int ngrp = VG ? 2 : 1;
    int vt = tid, grp = 0;
    if constexpr (VG) {
        vt = (itab + tk.tmap_off + (1 << (EB + 8)))[tid & 127];
        grp = __builtin_amdgcn_readfirstlane(tid >> 7);
    }
    const int vlane = vt & 63, vwave = vt >> 6;

    const uint32_t xF = bk.xF + (uint32_t)tid * VEC;
    const T *psi = psi_arena + tk.psi_off;
    T *bel = bel_arena + (MODE == 1 ? tk.bel_off : 0);   // distribute always stores (virtual cliques: scratch)
#ifdef JT_EXPERIMENT              // (timing experiments, wrong results: JTP_FLOW_DEBUG 16 = unit tasks run four rows only, 32 = one partial copy staged)
    const int total = (UNIT && (fl.dbg & 16)) ? U : tk.total;
#else
    const int total = tk.total;                       // loop iterations of this workgroup (>= U)
#endif
    const int dbg = tk.debug;
#ifdef JT_STAMPS
    double *stamp_out = msg_arena + tk.dbg_off + (int64_t)(fl.blk_base + bindex) * JT_NSTAMP;
#endif
    JT_STAMP_AT(0, FLOW ? t_entry : __builtin_amdgcn_s_memrealtime());
    // What staging needs from the task record, read HERE: the element loads below are issued by inline assembly that the
    // compiler treats as a barrier for memory operations - left where they are used, these scalar loads would be one more
    // dependent round trip between the workgroup's start and its message loads.
    constexpr int NI = NIN > 0 ? NIN : 1, NO = NOUT > 0 ? NOUT : 1;
    int64_t sm_off[NI], sm_ps[NI];
    int sm_npart[NI], sm_nfree[NI], sm_lds[NI];
    bool sm_same[NI];
    uint32_t sm_fp[NI][4], sm_hiv[NI];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const JtMsg &m = tk.msg[k];
        const uint32_t *fpw = reinterpret_cast<const uint32_t *>(m.free_pos);
        const uint32_t fp[4] = {fpw[0], fpw[1], fpw[2], fpw[3]};
        sm_off[k] = m.off + bk.gbase[k] + (m.fixed ? fl.fix_shift : 0);
        sm_ps[k] = m.pstride;
#ifdef JT_EXPERIMENT
        sm_npart[k] = (UNIT && (fl.dbg & 32)) ? 1 : m.npart;
#else
        sm_npart[k] = m.npart;
#endif
        sm_nfree[k] = m.nfree;
        sm_lds[k] = m.lds_off;
        sm_same[k] = m.same_launch != 0;
        sm_fp[k][0] = fp[0], sm_fp[k][1] = fp[1], sm_fp[k][2] = fp[2], sm_fp[k][3] = fp[3];
        sm_hiv[k] = jt_sub_hi(fp, m.nfree, lane);
    }
    // ... and what the flush needs.  Collect pass: kept in registers (read after the loop these are one more dependent round
    // trip on the hand-over to the parent: config 2 collect 2.52 -> 2.40 ms).  Distribute pass: the registers they would occupy
    // through the loop cost more (spills) than they save, so lane 0 parks the few words in LDS (flow_ctl) and the flush reads
    // them back from there - round 2 read the task record again, a dependent trip to memory of 1.0-1.6 us in front of the
    // stores of every hand-over.
    constexpr bool EARLY_OUT = EARLY_FLUSH;      // (the chain build of the distribute pass has the registers too)
    int64_t so_at[NO];
    int so_nfree[NO];
    uint32_t so_glo[NO], so_hiv[NO], so_fp[NO][4];
    uint32_t *park = flow_ctl != nullptr ? flow_ctl + 28 : nullptr;       // [NOUT][8]: at (2 words), nfree, free_pos (4 words)
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        const JtMsg &m = tk.msg[JT_MAX_IN + j];
        const uint32_t *fpw = reinterpret_cast<const uint32_t *>(m.free_pos);
        const uint32_t fp[4] = {fpw[0], fpw[1], fpw[2], fpw[3]};
        const int64_t at = m.off + (int64_t)bk.pnum[j] * m.pstride + bk.gbase[JT_MAX_IN + j];
        if constexpr (EARLY_OUT) {
            so_at[j] = at;
            so_nfree[j] = m.nfree;
            so_fp[j][0] = fp[0], so_fp[j][1] = fp[1], so_fp[j][2] = fp[2], so_fp[j][3] = fp[3];
        } else if (park != nullptr && tid == 0) {
            park[8 * j + 0] = (uint32_t)at, park[8 * j + 1] = (uint32_t)((uint64_t)at >> 32), park[8 * j + 2] = (uint32_t)m.nfree;
            park[8 * j + 3] = fp[0], park[8 * j + 4] = fp[1], park[8 * j + 5] = fp[2], park[8 * j + 6] = fp[3];
        }
    }
    // outgoing message j's epilogue follows every 2^run_j iterations (JtTask::out_run)
    int rmask[NOUT > 0 ? NOUT : 1];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) rmask[j] = (1 << ((tk.out_run >> (8 * j)) & 0xffu)) - 1;
    // hard evidence of this evidence set on this clique (jtp_set_evidence): table entries whose index
    // contradicts it count as zero (read here, before the first store: see the scalar-cache note below)
    uint32_t ev_mask = 0, ev_val = 0;
    if (fl.ev != nullptr) {
        ev_mask = fl.ev[2 * tk.pnode];
        ev_val = fl.ev[2 * tk.pnode + 1];
    }
    uint32_t loop_pos[JT_MAX_ITER_LOG2];                // logical index bit of loop-counter bit t (beyond the loop: bit 31, never set)
#pragma unroll
    for (int t = 0; t < JT_MAX_ITER_LOG2; ++t) loop_pos[t] = t < tk.nA + tk.nR ? tk.loop_pos[t] : 31u;
    T *junk_row = bel_arena + ((size_t)1 << (EB + 8)) + (uint32_t)tid * VEC;
    const bool wr_bel = MODE == 1 && tk.bel_off >= 0;      // (unit tasks: a belief is stored by read-out tasks only)

    // ---- element loads run U iterations ahead of their use.  Iteration i's offsets are row i of
    //      the task's iteration table (host built, held in registers below): the loops do no index
    //      arithmetic beyond one v_readlane per column.
    const int *gtab = itab + tk.itab_off;
    // the first U loads use offsets stored in the task record, so they leave immediately
    // (addresses come from the workgroup record alone: one dependent load after launch)
    // Element vectors travel global -> LDS by LDS-DMA into a wave-private ring of U slots of 1 KiB
    // and are read back with ds_read_b128: no register is a load destination, so the U loads stay
    // in flight across loop iterations whatever the register allocator does; waits are counted by
    // hand (vmcnt counts loads and stores in issue order on CDNA4).
    const T *psi0 = psi_arena + bk.psi_x0 + (uint32_t)tid * VEC;
    // rows that do not exist (JT_NO_ROW: a digit beyond a variable's cardinality, a padding bit) read the arena's
    // zero row; their belief stores go to the row behind it (jtp_internal.h).  The whole chunk may be such.
    const T *zero_row = psi_arena + (uint32_t)tid * VEC;
    const bool chunk_ok = !(bk.flags & JT_BLOCK_INVALID);
    // (uniform) default cache policy for the first table rows, see JtTask::keep_rows.  KEEP: only the kernel that runs both phases
    // in one launch looks at the flag (jt_propagate_flow: the plans with a top that both passes visit close together in time) -
    // the uniform branch around the two forms of the load cost config 2's chain kernels 1.5 % for nothing
    const bool keep_rows = KEEP && (bk.flags & JT_BLOCK_KEEP_ROWS) != 0;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem) + (uint32_t)wave * (U * 1024);
    const char *ring = smem + wave * (U * 1024) + lane * 16;
    // The workgroup's iteration table (<= 64 rows of JT_NCOL ints, host built) lives in registers,
    // row r in lane r; a step reads "row i, column c" with v_readlane: no memory latency on the
    // critical path of a step except the message entries themselves.
    int trow[JT_NCOL];
    // (Vector-memory operations return in issue order, so the staging loads issued behind the first table rows only come
    //  back after them.  Issuing the first round of message loads AHEAD of the rows was tried in round 3: config 4 +-0,
    //  config 3 - whose staging loads are a quarter of its traffic - 12.5 -> 13.1 ms, A/B on one box: the rows go first.)
    // TMIX (plans with a mixed-radix thread part, HostPlan::tmix): the elements of a thread's logical index tid * VEC + e lie
    // at tmap[...] inside a row (-1: no such entry).  Rows are then gathered element by element into registers - ordinary
    // loads the compiler counts itself - instead of the 16-byte LDS-DMA ring; everything downstream (message look-ups,
    // butterflies, epilogues) works on the logical index and is unchanged.
    // (Entries that do not exist load the row's first element - every lane then executes every load, no per-element branch -
    //  and count as zero when the row is consumed; `row0` is a register copy of JtBlock::xF: read through `bk` inside the
    //  loop it was a scalar load and a wait in front of every element load, the belief stores could alias it.)
    int tpo[VEC];
    T tbuf[TMIX ? UT : 1][VEC];
    const uint32_t row0 = bk.xF;
    auto gather_row = [&](const int slot, const uint32_t xrow, const bool ok) {
        if constexpr (TMIX) {
            const T *row = ok ? psi + (row0 + xrow) : psi_arena;             // (uniform; psi_arena: the zero row)
#pragma unroll
            for (int e = 0; e < VEC; ++e) tbuf[slot][e] = row[tpo[e] >= 0 ? tpo[e] : 0];
        }
    };
    auto issue_tables = [&]() {
        if constexpr (UNIT) {
            const int *tm = itab + tk.tmap_off + tid * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) tpo[e] = tm[e];
        } else if constexpr (TMIX) {
            const int *tm = itab + tk.tmap_off + vt * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) tpo[e] = tm[e];
#pragma unroll
            for (int u = 0; u < UT; ++u) {
                const uint32_t x = bk.first_x[u * ngrp + grp];          // (rows 0..7: UT * ngrp <= 8)
                gather_row(u, x, x != JT_NO_ROW);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
                jt_dma16(bk.first_x[u] == JT_NO_ROW ? zero_row : psi0 + bk.first_x[u], __builtin_amdgcn_readfirstlane(ring_lds + u * 1024), keep_rows);
        }
        const int r = lane < total ? lane : total - 1;
        const int4 a = *reinterpret_cast<const int4 *>(gtab + r * JT_NCOL);
        const int4 b = *reinterpret_cast<const int4 *>(gtab + r * JT_NCOL + 4);
        trow[0] = a.x; trow[1] = a.y; trow[2] = a.z; trow[3] = a.w;
        trow[4] = b.x; trow[5] = b.y; trow[6] = b.z; trow[7] = b.w;
        JT_STAMP(1);
    };
    issue_tables();
    if constexpr (EARLY_OUT) {
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            so_glo[j] = jt_sub_lo(so_fp[j], so_nfree[j], tid);
            so_hiv[j] = jt_sub_hi(so_fp[j], so_nfree[j], lane);
        }
    }

    // ---- stage incoming sub-boxes (summing partial copies), zero outgoing sub-boxes -----
    // (the first element loads leave behind the first round of message loads).  Sub-boxes smaller than the workgroup
    // split their partial copies over 256/n thread groups; the loads of ALL such messages are
    // issued together (one round trip to L2 for the whole staging), group sums are combined
    // through LDS in group order (deterministic).  Larger sub-boxes: one thread per entry.
    double *scratch = reinterpret_cast<double *>(smem + tk.itab_lds - JT_STAGE_SCRATCH * NIN);
    const double *msg_cur = msg_arena + fl.cur_off;
    uint64_t wait_t0 = 0;
    for (int attempt = 0;; ++attempt) {
        JT_STAMP_AT(13, attempt + 1);
        // (only in plans made of latency-bound levels, JtTask::settle: chains gain 6 %; next to streaming
        //  levels the extra loads of waiting workgroups cost 1-2 % - both measured)
        const int settle_attempt = tk.settle ? attempt : 0;
        const double *unready = nullptr;          // (FLOW) an entry this thread found not written yet
        // (TMIX: a chunk that does not exist - JT_BLOCK_INVALID - stages nothing, waits for nobody and runs no row: it writes
        //  its all-zero sub-boxes, which the consumers sum and whose entries of the other arena half it re-arms, and ends)
        if (!TMIX || chunk_ok) {
            const double *src[NIN > 0 ? NIN : 1];
            int idx_t[NIN > 0 ? NIN : 1], gp0[NIN > 0 ? NIN : 1], gp1[NIN > 0 ? NIN : 1];
            int64_t ps[NIN > 0 ? NIN : 1];
            bool grouped[NIN > 0 ? NIN : 1], thr_mem[NIN > 0 ? NIN : 1];
            double psum[NIN > 0 ? NIN : 1];
            int maxper = 0;
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                const int nfree = sm_nfree[k];
                const uint32_t fp[4] = {sm_fp[k][0], sm_fp[k][1], sm_fp[k][2], sm_fp[k][3]};
                src[k] = msg_cur + sm_off[k];
                ps[k] = sm_ps[k];
                idx_t[k] = 0;
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (b < nfree) idx_t[k] += ((tid >> b) & 1) << JT_FPOS(fp, b);
                thr_mem[k] = sm_same[k];
                grouped[k] = nfree < 8 && sm_npart[k] > 1;
                psum[k] = 0.0;
                gp0[k] = gp1[k] = 0;
                if (grouped[k]) {
                    const int groups = JT_THREADS >> nfree;   // >= 2
                    const int per = (sm_npart[k] + groups - 1) / groups;
                    gp0[k] = (tid >> nfree) * per;
                    gp1[k] = (gp0[k] + per < sm_npart[k]) ? gp0[k] + per : sm_npart[k];
                    maxper = per > maxper ? per : maxper;
                }
            }
            // Sub-boxes of a workgroup's size and more: one thread per entry and round of 256 entries.  Where entry `it` * 256 + tid
            // of message k's sub-box lies in the message (copy pc):
            // (round 6: the part of the bits above the thread's own eight comes from a 32-row table, row j in lane j - jt_sub_hi, as in
            //  the flush; the scalar deposit loop this replaces ran for every entry of every message: ~20 scalar instructions each)
#ifndef JT_STAGE_SCALAR_HI
            auto entry_at = [&](auto k_tag, int it, int pc) {
                constexpr int k = decltype(k_tag)::value;
                const int idx = idx_t[k] + __builtin_amdgcn_readlane((int)sm_hiv[k], it);
                return src[k] + ((int64_t)pc * ps[k] + idx);
            };
#else
            auto entry_at = [&](auto k_tag, int it, int pc) {
                constexpr int k = decltype(k_tag)::value;
                int idx = idx_t[k];
#pragma unroll
                for (int b = 8; b < JT_MAX_FREE; ++b)
                    if (b < sm_nfree[k]) idx += ((it >> (b - 8)) & 1) << JT_FPOS(sm_fp[k], b);
                return src[k] + ((int64_t)pc * ps[k] + idx);
            };
#endif
            // ... single-copy messages first, ALL of them in lock step: the loads of a round - up to eight entries per thread and
            // message (four when there are three messages or more: registers) - leave together, so a task with two or three
            // such messages pays one round trip to memory where it paid one per message (round 5: the unit tasks of config 3 are
            // two 1024-entry sub-boxes and a few cheap rows each - staging was a third of a workgroup's life).
            {
                constexpr int SU = NIN >= 3 ? 4 : 8;
                int nmax = 0;
#pragma unroll
                for (int k = 0; k < NIN; ++k)
                    if (!grouped[k] && sm_npart[k] == 1) nmax = (1 << sm_nfree[k]) > nmax ? (1 << sm_nfree[k]) : nmax;
                for (int it0 = 0; it0 * JT_THREADS < nmax; it0 += SU) {
                    double c[NIN > 0 ? NIN : 1][SU];
                    jt_static_for<NIN>([&](auto k_tag) {
                        constexpr int k = decltype(k_tag)::value;
                        const int n = (!grouped[k] && sm_npart[k] == 1) ? 1 << sm_nfree[k] : 0;
#pragma unroll
                        for (int u = 0; u < SU; ++u)
                            c[k][u] = (it0 + u) * JT_THREADS + tid < n ? jt_msg_load<FLOW>(entry_at(k_tag, it0 + u, 0), thr_mem[k]) : 0.0;
                    });
                    jt_static_for<NIN>([&](auto k_tag) {
                        constexpr int k = decltype(k_tag)::value;
                        const int n = (!grouped[k] && sm_npart[k] == 1) ? 1 << sm_nfree[k] : 0;
                        double *sub = reinterpret_cast<double *>(smem + sm_lds[k]);
#pragma unroll
                        for (int u = 0; u < SU; ++u) {
                            if ((it0 + u) * JT_THREADS + tid >= n) continue;
                            if (FLOW) c[k][u] = jt_msg_settle<FLOW>(entry_at(k_tag, it0 + u, 0), c[k][u], thr_mem[k], settle_attempt);
                            if (FLOW && jt_unwritten(c[k][u])) unready = entry_at(k_tag, it0 + u, 0);
                            sub[(it0 + u) * JT_THREADS + tid] = 0.0 + c[k][u];
                        }
                    });
                }
            }
            // ... then the messages of several copies, one after the other: 2^plog copies of 8 >> plog entries in flight per thread.
            // Every entry's copies are summed in ascending order from 0.0.
            jt_static_for<NIN>([&](auto k_tag) {
                constexpr int k = decltype(k_tag)::value;
                if (grouped[k] || sm_npart[k] == 1) return;
                const int nfree = sm_nfree[k];
                {
                    double *sub = reinterpret_cast<double *>(smem + sm_lds[k]);
                    const int n = 1 << nfree;
                    const int npart = sm_npart[k];
                    {
                        const int plog = npart >= 8 ? 3 : (npart >= 4 ? 2 : 1);
                        const int pmask = (1 << plog) - 1, E = 8 >> plog;
                        for (int it0 = 0; it0 * JT_THREADS < n; it0 += E) {
                            double sum[4] = {0.0, 0.0, 0.0, 0.0};
                            for (int p0 = 0; p0 < npart; p0 += 1 << plog) {
                                double c[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const int e = u >> plog, pc = p0 + (u & pmask);
                                    const bool ok = (it0 + e) * JT_THREADS + tid < n && pc < npart;
                                    c[u] = ok ? jt_msg_load<FLOW>(entry_at(k_tag, it0 + e, pc), thr_mem[k]) : 0.0;
                                }
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const int e = u >> plog, pc = p0 + (u & pmask);
                                    const bool ok = (it0 + e) * JT_THREADS + tid < n && pc < npart;
                                    if (FLOW && ok) c[u] = jt_msg_settle<FLOW>(entry_at(k_tag, it0 + e, pc), c[u], thr_mem[k], settle_attempt);
                                    if (FLOW && jt_unwritten(c[u])) unready = entry_at(k_tag, it0 + e, pc);
                                }
                                // (plog is uniform: the entry a value belongs to is picked with compile-time indices)
                                if (plog == 1) {
#pragma unroll
                                    for (int u = 0; u < 8; ++u) sum[u >> 1] += c[u];
                                } else if (plog == 2) {
#pragma unroll
                                    for (int u = 0; u < 8; ++u) sum[u >> 2] += c[u];
                                } else {
#pragma unroll
                                    for (int u = 0; u < 8; ++u) sum[0] += c[u];
                                }
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (e < E && (it0 + e) * JT_THREADS + tid < n) sub[(it0 + e) * JT_THREADS + tid] = sum[e];
                        }
                    }
                }
            });
            // grouped messages: every thread sums its range of copies of its entry, all messages at once
            constexpr int GC = NIN >= 3 ? 4 : 8;          // copies in flight per message (register budget)
            for (int p = 0; p < maxper; p += GC) {
                double c[NIN > 0 ? NIN : 1][GC];
#pragma unroll
                for (int k = 0; k < NIN; ++k)
#pragma unroll
                    for (int u = 0; u < GC; ++u)
                        c[k][u] = (grouped[k] && gp0[k] + p + u < gp1[k]) ? jt_msg_load<FLOW>(src[k] + ((int64_t)(gp0[k] + p + u) * ps[k] + idx_t[k]), thr_mem[k]) : 0.0;
#pragma unroll
                for (int k = 0; k < NIN; ++k)
#pragma unroll
                    for (int u = 0; u < GC; ++u) {
                        if (FLOW && grouped[k] && gp0[k] + p + u < gp1[k])
                            c[k][u] = jt_msg_settle<FLOW>(src[k] + ((int64_t)(gp0[k] + p + u) * ps[k] + idx_t[k]), c[k][u], thr_mem[k], settle_attempt);
                        psum[k] += c[k][u];
                        if (FLOW && jt_unwritten(c[k][u])) unready = src[k] + ((int64_t)(gp0[k] + p + u) * ps[k] + idx_t[k]);
                    }
            }
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (grouped[k]) scratch[k * JT_THREADS + tid] = psum[k];
        }
        if (attempt == 0) {
#pragma unroll
            for (int k = 0; k < NOUT; ++k) {
                const JtMsg &m = tk.msg[JT_MAX_IN + k];
                double *sub = reinterpret_cast<double *>(smem + m.lds_off);
                const int n = 1 << m.nfree;
                for (int s = tid; s < n; s += JT_THREADS) sub[s] = 0.0;
            }
            JT_STAMP(2);
        }
        if constexpr (!FLOW || NIN == 0) {
            __syncthreads();
            break;
        } else {
            // Some entry was not ready: ONE lane of the workgroup waits on one such entry (polling with
            // back-off: thousands of workgroups may be waiting at the top of a tree, and their polls
            // share the memory system with the workgroups they wait for), then everybody loads again.
            // The poller gives up after 2 s (100 MHz clock) or when another workgroup did, so that the
            // grid always drains; the host then reports the propagate as failed.
            if (fl.dbg & 4) unready = nullptr;
            if (fl.dbg & 8) unready = msg_cur;                           // fault injection: wait for ever
            uint32_t *slot = flow_ctl + 4 + (attempt & 1) * 12;          // [0..3] wave flags, [4..11] wave candidates
            {
                const uint64_t have = __ballot(unready != nullptr);
                if (have != 0 && lane == (int)__builtin_ctzll(have)) {
                    slot[4 + 2 * wave] = (uint32_t)(uintptr_t)unready;
                    slot[5 + 2 * wave] = (uint32_t)((uintptr_t)unready >> 32);
                }
                if (lane == 0) slot[wave] = have != 0 ? 1u : 0u;
            }
            __syncthreads();
            const uint32_t w3 = slot[3], w2 = slot[2], w1 = slot[1], w0 = slot[0];
            if ((w0 | w1 | w2 | w3) == 0) break;
            if (tid == 0) {
                const int cw = w3 ? 3 : (w2 ? 2 : (w1 ? 1 : 0));          // later waves hold later partial copies
                const double *entry = reinterpret_cast<const double *>((uintptr_t)slot[4 + 2 * cw] | ((uintptr_t)slot[5 + 2 * cw] << 32));
                if (wait_t0 == 0) wait_t0 = __builtin_amdgcn_s_memrealtime();
                uint32_t give_up = 0;
                unsigned spins = 0;
                const uint64_t limit = (fl.dbg & 8) ? 2000000ull : 200000000ull;
                while (jt_unwritten(jt_msg_load<true>(entry)) || (fl.dbg & 8)) {
                    // one load in flight per workgroup is no traffic to speak of: poll back to back while the
                    // wait is young (a short wait is a producer about to finish), then every ~0.5 us, ~1 us
                    if (spins >= 64) __builtin_amdgcn_s_sleep(32);
                    else if (spins >= 16) __builtin_amdgcn_s_sleep(16);
                    if ((++spins & 15u) == 0) {
                        if (__hip_atomic_load(fl.sync + JT_SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) give_up = 1;
                        else if (__builtin_amdgcn_s_memrealtime() - wait_t0 > limit) {
                            __hip_atomic_store(fl.sync + JT_SYNC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(fl.host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            give_up = 1;
                        }
                        if (give_up) break;
                    }
                }
                flow_ctl[1] = give_up;
                JT_STAMP(14);                                         // (the last poll's end: what follows is re-staging)
            }
            __syncthreads();
            if (flow_ctl[1] != 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // LDS-DMA loads must land before the wave ends
                return;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const int nfree = sm_nfree[k];
        if (nfree < 8 && sm_npart[k] > 1 && tid < (1 << nfree)) {
            const int groups = JT_THREADS >> nfree;
            double sum = 0.0;
            for (int g = 0; g < groups; ++g) sum += scratch[k * JT_THREADS + (g << nfree) + tid];
            reinterpret_cast<double *>(smem + sm_lds[k])[tid] = sum;
        }
    }
    __syncthreads();

    JT_STAMP(3);
    // ---- per-thread constants ---------------------------------------------------------------
    int thr[NMSG > 0 ? NMSG : 1];
    const double *in_sub[NIN > 0 ? NIN : 1];
    double *out_sub[NOUT > 0 ? NOUT : 1];
    int in_ew0[NIN > 0 ? NIN : 1], in_ew1[NIN > 0 ? NIN : 1], in_edep[NIN > 0 ? NIN : 1];
    // outgoing-message constants are read here, before the first store of the kernel: later
    // loads of the task record could not use the scalar cache and would drain vmcnt
    int o_rede[NOUT > 0 ? NOUT : 1], o_redl[NOUT > 0 ? NOUT : 1], o_redw[NOUT > 0 ? NOUT : 1];
    int o_ew0[NOUT > 0 ? NOUT : 1], o_ew1[NOUT > 0 ? NOUT : 1];
#pragma unroll
    for (int k = 0; k < NMSG; ++k) {
        const JtMsg &m = tk.msg[k < NIN ? k : JT_MAX_IN + (k - NIN)];
        int t = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) t += ((vlane >> b) & 1) * m.t_w[b];
#pragma unroll
        for (int b = 0; b < 2; ++b) t += ((vwave >> b) & 1) * m.t_w[6 + b];
        thr[k] = t;
        if (k < NIN) {
            in_sub[k < NIN ? k : 0] = reinterpret_cast<const double *>(smem + m.lds_off);
            in_ew0[k < NIN ? k : 0] = m.e_w[0];
            in_ew1[k < NIN ? k : 0] = m.e_w[1];
            in_edep[k < NIN ? k : 0] = m.e_dep;
        } else {
            const int j = k >= NIN ? k - NIN : 0;
            out_sub[j] = reinterpret_cast<double *>(smem + m.lds_off);
            o_rede[j] = m.red_e;
            o_redl[j] = m.red_lane;
            o_redw[j] = m.red_wave;
            o_ew0[j] = m.e_w[0];
            o_ew1[j] = m.e_w[1];
        }
    }

    // (unit tasks: a row is nothing but message look-ups, so every instruction of one counts - the byte address of element e's
    //  entry of message k inside the sub-box is formed HERE, a row adds its offset: one vector add per LDS read)
    uint32_t ua[UNIT ? NI : 1][VEC];
    if constexpr (UNIT) {
#pragma unroll
        for (int k = 0; k < NIN; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                ua[k][e] = (uint32_t)tk.msg[k].lds_off + 8u * (uint32_t)(thr[k] + ((e & 1) ? in_ew0[k] : 0) + ((e & 2) ? in_ew1[k] : 0));
    }

    double acc[NOUT > 0 ? NOUT : 1][VEC];
#pragma unroll
    for (int j = 0; j < NOUT; ++j)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[j][e] = 0.0;

    // fold this thread's sums for outgoing message j (one run of iterations) into its sub-box
    auto epilogue = [&](auto j_tag, const int oo_j, const bool mine) {
        {
            constexpr int j = decltype(j_tag)::value;
            const int red_e = o_rede[j], red_lane = o_redl[j], red_wave = o_redw[j];
            constexpr bool compact = VG;
            if (compact && !mine) {
                // (compact mixed-radix rows: the other group's sums are due, this wave's run goes on - its sums stay as they are;
                //  the four barriers of the turns below)
                for (int ph = 0; ph < 4; ++ph) __syncthreads();
                return;
            }
            if constexpr ((UNIT || TMIX) && MODE == 0) {
                // (which entries of the thread part exist does not change from row to row: applied to the sums of a run, not to
                //  every product; mixed-radix rows: an entry that does not exist was gathered from the row's first element)
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (tpo[e] < 0) acc[j][e] = 0.0;
            }
            if constexpr (VEC == 4) {
                if (red_e & 1) {
                    acc[j][0] += acc[j][1];
                    acc[j][2] += acc[j][3];
                }
                if (red_e & 2) {
                    acc[j][0] += acc[j][2];
                    acc[j][1] += acc[j][3];
                }
            } else {
                if (red_e & 1) acc[j][0] += acc[j][1];
            }
            if constexpr (compact) {
                // compact mixed-radix rows: the threads that share an output slot are wherever the clique's list put them - every
                // thread adds its own sums to its slot, the LDS serialises the lanes of one wave, the waves take turns (a fixed order)
                const int slot_c = oo_j + thr[NIN + j];
                for (int ph = 0; ph < 4; ++ph) {
                    if (wave == ph) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            if ((e & red_e) == 0) {
                                const int eo = ((e & 1) ? o_ew0[j] : 0) + ((e & 2) ? o_ew1[j] : 0);
                                __hip_atomic_fetch_add(&out_sub[j][slot_c + eo], acc[j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                    __syncthreads();
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[j][e] = 0.0;
                return;
            }
            // (all elements take part, also those folded away above: no branch per element - their sums are not stored)
            if (!(dbg & 4)) jt_lane_sums<VEC>(acc[j], red_lane);
            const bool rep = (lane & red_lane) == 0;
            const int slot = oo_j + thr[NIN + j];
            const int nph = (dbg & 8) ? 1 : 1 << __builtin_popcount((unsigned)red_wave);
            // waves that share slots (wave bits not in the message) take turns, in wave order
            int myph = 0;
            if (red_wave == 1) myph = wave & 1;
            else if (red_wave == 2) myph = wave >> 1;
            else if (red_wave == 3) myph = wave;
            for (int ph = 0; ph < nph; ++ph) {
                if (rep && myph == ph) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        if ((e & red_e) == 0) {
                            const int eo = ((e & 1) ? o_ew0[j] : 0) + ((e & 2) ? o_ew1[j] : 0);
                            // one slot, one lane per phase: an LDS add without return (ds_add_f64) is the same
                            // sum in the same order as read-add-write, minus the wait for the read
                            __hip_atomic_fetch_add(&out_sub[j][slot + eo], acc[j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
                if (nph > 1) __syncthreads();
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[j][e] = 0.0;
        }
    };

    // make the table rows "arrived" for the compiler before the loop (they came by vector loads;
    // otherwise every use inside the loop would be guarded by a vmcnt(0) wait)
#pragma unroll
    for (int c = 0; c < JT_NCOL; ++c) asm volatile("" : "+v"(trow[c]));

    // one iteration: consume `slot`, refill it for iteration i + U, multiply, accumulate
    // SLOT = i mod U (static), YOUNGER = vector-memory operations issued after the DMA of iteration i
    // that may still be outstanding when iteration i is consumed
    // (compact mixed-radix rows: a step serves row i * ngrp + grp of the iteration table - nit steps in all)
    const int nit = TMIX ? (total + ngrp - 1) / ngrp : total;
    auto step = [&](auto slot_tag, auto younger_tag, const int i) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int YOUNGER = decltype(younger_tag)::value;
        double p[VEC];
        if constexpr (UNIT) {
            // (filled in below, once the row is known to exist)
        } else if constexpr (TMIX) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) p[e] = (MODE == 0 || tpo[e] >= 0) ? (double)tbuf[SLOT][e] : 0.0;
            const int inext = (i + UT < nit) ? i + UT : nit - 1;
            const int rnext = inext * ngrp + grp;                       // (uniform per wave)
            const uint32_t xnext = (uint32_t)__builtin_amdgcn_readlane(trow[0], rnext < total ? rnext : total - 1);
            gather_row(SLOT, xnext, xnext != JT_NO_ROW && chunk_ok && i + UT < nit && rnext < total);
        } else {
        jt_wait_vmcnt<YOUNGER>();
        const VT v = *reinterpret_cast<const VT *>(ring + SLOT * 1024);
        p[0] = (double)v.x;
        p[1] = (double)v.y;
        if constexpr (VEC == 4) {
            p[2] = (double)v.z;
            p[3] = (double)v.w;
        }
        {   // refill the slot for iteration i + U (the ds_read above has returned: `v` was converted).
            // Past the end the last row is loaded again (cache resident) so that the count of
            // operations in flight stays the same in every step.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int inext = (i + U < total) ? i + U : total - 1;
            const uint32_t xnext = (uint32_t)__builtin_amdgcn_readlane(trow[0], inext);
            // (always non-temporal here: JtTask::keep_rows is honoured by the FIRST loads only - a choice of cache policy in
            //  this loop, a uniform branch around two forms of the load, cost config 2 3.5 % whichever way it went; the workgroups
            //  of the levels nearest the root have four rows - all of them first loads)
            jt_dma16((xnext == JT_NO_ROW || !chunk_ok) ? zero_row : psi + (xF + xnext),
                     __builtin_amdgcn_readfirstlane(ring_lds + SLOT * 1024));
        }
        }
        // (compact mixed-radix rows: the second group's last step may have no row - an odd number of rows)
        const int lraw = TMIX ? i * ngrp + grp : i;
        const bool row_there = !TMIX || lraw < total;          // (uniform per wave)
        const int li = row_there ? lraw : total - 1;
        const uint32_t xoff = (uint32_t)__builtin_amdgcn_readlane(trow[0], li);
        const bool row_ok = xoff != JT_NO_ROW && chunk_ok && row_there;   // (uniform)
        if constexpr (TMIX && !UNIT) {
            if (!row_there) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) p[e] = 0.0;
            }
        }
        // TMIX: the table holds existing rows only; the row's place in the full loop nest and the ends of the outgoing
        // messages' runs come with it (jtp_plan.cpp, plan_loops)
        const uint32_t rowinfo = TMIX ? (uint32_t)__builtin_amdgcn_readlane(trow[1 + JT_MAX_IN], li) : 0u;
        const uint32_t inest = TMIX ? (rowinfo >> 16) & 63u : (uint32_t)i;
        if constexpr (UNIT && MODE == 0) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) p[e] = 1.0;          // (entries that do not exist: see the epilogue; rows: below)
        } else if constexpr (UNIT) {
            const double one = row_ok ? 1.0 : 0.0;             // (uniform)
#pragma unroll
            for (int e = 0; e < VEC; ++e) p[e] = tpo[e] >= 0 ? one : 0.0;
        }
        if (ev_mask != 0) {                                // (uniform: no vector instruction is spent without evidence)
            // LOGICAL index of this thread's first element (evidence masks are over index bits, element offsets are
            // physical): chunk bits + the row's loop bits + thread part
            uint32_t x0 = bk.lxF + (uint32_t)vt * VEC;
#pragma unroll
            for (int t = 0; t < JT_MAX_ITER_LOG2; ++t) x0 += ((inest >> t) & 1u) << loop_pos[t];
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (((x0 + e) & ev_mask) != ev_val) p[e] = 0.0;
        }
        int ioff[4], ooff[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) ioff[k] = k < NIN ? __builtin_amdgcn_readlane(trow[1 + k], li) : 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) ooff[j] = j < NOUT ? __builtin_amdgcn_readlane(trow[1 + JT_MAX_IN + j], li) : 0;
        if constexpr (TMIX && NOUT > 0) ooff[0] &= 0xffff;
        double in[NIN > 0 ? NIN : 1][VEC];
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            if constexpr (UNIT) {
                const uint32_t rb = (uint32_t)ioff[k] << 3;        // (scalar)
                if (in_edep[k]) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) in[k][e] = *reinterpret_cast<const double *>(smem + (ua[k][e] + rb));
                } else {
                    const double t = *reinterpret_cast<const double *>(smem + (ua[k][0] + rb));
#pragma unroll
                    for (int e = 0; e < VEC; ++e) in[k][e] = t;
                }
                continue;
            }
            const int base = ioff[k] + thr[k];
            if (in_edep[k]) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int eo = ((e & 1) ? in_ew0[k] : 0) + ((e & 2) ? in_ew1[k] : 0);
                    in[k][e] = in_sub[k][base + eo];
                }
            } else {
                const double t = in_sub[k][base];
#pragma unroll
                for (int e = 0; e < VEC; ++e) in[k][e] = t;
            }
        }
        if constexpr (MODE == 0 && UNIT) {
            // a row that does not exist (uniform) adds nothing; without evidence an element is the product of its message
            // entries and nothing else (p = 1 is folded away: (1 * a) * b and a * b are the same double)
            if (row_ok) {
                if (ev_mask == 0) {                            // (uniform)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        double w = NIN > 0 ? in[0][e] : 1.0;
#pragma unroll
                        for (int k = 1; k < NIN; ++k) w *= in[k][e];
#pragma unroll
                        for (int j = 0; j < NOUT; ++j) acc[j][e] += w;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        double w = p[e];
#pragma unroll
                        for (int k = 0; k < NIN; ++k) w *= in[k][e];
#pragma unroll
                        for (int j = 0; j < NOUT; ++j) acc[j][e] += w;
                    }
                }
            }
        } else if constexpr (MODE == 0) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                double w = p[e];
#pragma unroll
                for (int k = 0; k < NIN; ++k) w *= in[k][e];
                // (several outputs - read-out tasks only, jt_marginals: every one the sum over its own complement)
#pragma unroll
                for (int j = 0; j < NOUT; ++j) acc[j][e] += w;
            }
        } else {
            double b[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                double pre = p[e];
#pragma unroll
                for (int k = 0; k < NPAR; ++k) pre *= in[k][e];
                // all-but-one products over the children: prefix * suffix
                double suf[NOUT + 1];
                suf[NOUT] = 1.0;
#pragma unroll
                for (int j = NOUT - 1; j >= 0; --j) suf[j] = suf[j + 1] * in[NPAR + j][e];
                double pref = pre;
#pragma unroll
                for (int j = 0; j < NOUT; ++j) {
                    acc[j][e] += pref * suf[j + 1];
                    pref *= in[NPAR + j][e];
                }
                b[e] = pref;
            }
            if constexpr (BEL) if (!UNIT || wr_bel) {
                // beliefs are written once and not read again by this propagate: a streaming store
                // keeps them from sitting dirty in the last-level cache, where the next collect's
                // reads would have to push them out (measured: collect 0.25 -> 0.22 ms)
                typedef T ext_t __attribute__((ext_vector_type(VEC)));
                ext_t ov;
                ov[0] = (T)b[0];
                ov[1] = (T)b[1];
                if constexpr (VEC == 4) {
                    ov[2] = (T)b[2];
                    ov[3] = (T)b[3];
                }
                if constexpr (TMIX) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (row_ok && tpo[e] >= 0) __builtin_nontemporal_store(ov[e], bel + (row0 + xoff + (uint32_t)tpo[e]));
                } else
                __builtin_nontemporal_store(ov, reinterpret_cast<ext_t *>(row_ok ? bel + (xF + xoff) : junk_row));
            }
        }
        if constexpr (NOUT > 0) {
            if (!(dbg & 1)) {
                constexpr bool compact = VG;
                if constexpr (compact) {
                    // Two rows per step: a group's sums of a run are complete when the run ends at its own row or at the next one (the
                    // other group's); the epilogue has barriers, so EVERY wave goes through it when either group's are - the first
                    // group's rows are 2 i, 2 i + 2, ..., the second's 2 i + 1, ... (run-end flags beyond the last row: none)
                    auto endbit = [&](const int r, const int j) {
                        return r < total ? ((uint32_t)__builtin_amdgcn_readlane(trow[1 + JT_MAX_IN], r) >> (24 + j)) & 1u : 0u;
                    };
                    auto both = [&](auto j_tag) {
                        constexpr int j = decltype(j_tag)::value;
                        const uint32_t e0 = endbit(2 * i, j), e1 = endbit(2 * i + 1, j), e2 = endbit(2 * i + 2, j);
                        if (e0 | e1 | e2) epilogue(j_tag, ooff[j], grp == 0 ? (e0 | e1) != 0 : (row_there && (e1 | e2) != 0));
                    };
                    both(std::integral_constant<int, 0>{});
                    if constexpr (NOUT > 1) both(std::integral_constant<int, 1>{});
                    if constexpr (NOUT > 2) both(std::integral_constant<int, 2>{});
                } else {
                auto run_ends = [&](const int j) { return TMIX ? ((rowinfo >> (24 + j)) & 1u) != 0 : (i & rmask[j]) == rmask[j]; };
                if (run_ends(0)) epilogue(std::integral_constant<int, 0>{}, ooff[0], true);
                if constexpr (NOUT > 1)
                    if (run_ends(1)) epilogue(std::integral_constant<int, 1>{}, ooff[1], true);
                if constexpr (NOUT > 2)
                    if (run_ends(2)) epilogue(std::integral_constant<int, 2>{}, ooff[2], true);
                }
            }
        }
    };

    JT_STAMP(4);
    // Younger operations when iteration i is consumed: the U-1 later element loads, plus (distribute)
    // one belief store per step already executed since that load was issued: U of them in steady
    // state, k in step k of the first group (its loads were issued in the prologue, before any store).
    constexpr int ST = (MODE == 1) ? 1 : 0;
    using std::integral_constant;
    if constexpr (TMIX) {
        // any number of rows from 1 to 64 (the rows that exist); none at all for a chunk that does not
        if (chunk_ok)
            for (int i0 = 0; i0 < nit; i0 += UT) {
                step(integral_constant<int, 0>{}, integral_constant<int, 0>{}, i0);
                if (i0 + 1 < nit) step(integral_constant<int, 1>{}, integral_constant<int, 0>{}, i0 + 1);
                if (i0 + 2 < nit) step(integral_constant<int, 2>{}, integral_constant<int, 0>{}, i0 + 2);
                if (i0 + 3 < nit) step(integral_constant<int, 3>{}, integral_constant<int, 0>{}, i0 + 3);
                if constexpr (UT == 8) {
                    if (i0 + 4 < nit) step(integral_constant<int, 4 % UT>{}, integral_constant<int, 0>{}, i0 + 4);
                    if (i0 + 5 < nit) step(integral_constant<int, 5 % UT>{}, integral_constant<int, 0>{}, i0 + 5);
                    if (i0 + 6 < nit) step(integral_constant<int, 6 % UT>{}, integral_constant<int, 0>{}, i0 + 6);
                    if (i0 + 7 < nit) step(integral_constant<int, 7 % UT>{}, integral_constant<int, 0>{}, i0 + 7);
                }
            }
    } else {
    {
        step(integral_constant<int, 0>{}, integral_constant<int, U - 1 + 0 * ST>{}, 0);
        JT_STAMP(5);
        step(integral_constant<int, 1>{}, integral_constant<int, U - 1 + 1 * ST>{}, 1);
        JT_STAMP(6);
        if (total > 2) {              // (a workgroup of two iterations - searched splits on latency-bound levels: the
                                      //  other slots hold repeats of its last row)
            step(integral_constant<int, 2>{}, integral_constant<int, U - 1 + 2 * ST>{}, 2);
            JT_STAMP(7);
            step(integral_constant<int, 3>{}, integral_constant<int, U - 1 + 3 * ST>{}, 3);
            JT_STAMP(8);
        }
        if constexpr (U == 8) {
            if (total > 4) {          // (a workgroup of four iterations: the other four slots hold repeats of its last row)
                step(integral_constant<int, 4 % U>{}, integral_constant<int, U - 1 + 4 * ST>{}, 4);
                step(integral_constant<int, 5 % U>{}, integral_constant<int, U - 1 + 5 * ST>{}, 5);
                step(integral_constant<int, 6 % U>{}, integral_constant<int, U - 1 + 6 * ST>{}, 6);
                step(integral_constant<int, 7 % U>{}, integral_constant<int, U - 1 + 7 * ST>{}, 7);
            }
        }
    }
    for (int i0 = U; i0 < total; i0 += U) {
        step(integral_constant<int, 0>{}, integral_constant<int, U - 1 + U * ST>{}, i0);
        step(integral_constant<int, 1>{}, integral_constant<int, U - 1 + U * ST>{}, i0 + 1);
        step(integral_constant<int, 2>{}, integral_constant<int, U - 1 + U * ST>{}, i0 + 2);
        step(integral_constant<int, 3>{}, integral_constant<int, U - 1 + U * ST>{}, i0 + 3);
        if constexpr (U == 8) {
            step(integral_constant<int, 4 % U>{}, integral_constant<int, U - 1 + U * ST>{}, i0 + 4);
            step(integral_constant<int, 5 % U>{}, integral_constant<int, U - 1 + U * ST>{}, i0 + 5);
            step(integral_constant<int, 6 % U>{}, integral_constant<int, U - 1 + U * ST>{}, i0 + 6);
            step(integral_constant<int, 7 % U>{}, integral_constant<int, U - 1 + U * ST>{}, i0 + 7);
        }
    }
    }
    JT_STAMP(9);

    // ---- flush outgoing sub-boxes as this chunk's partial copy ----------------------------------
    if constexpr (NOUT > 0) {
        if (dbg & 1) {
            if (acc[0][0] == 12345.678) msg_arena[0] = acc[0][0];     // keep the sums alive
            return;
        }
        __syncthreads();
        JT_STAMP(10);
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            if constexpr (!EARLY_OUT) {
                uint32_t fp[4];
                if (park != nullptr) {                     // parked in LDS at the start (dataflow kernels); uniform: back into scalars
                    auto word = [&](int i) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)park[8 * j + i]); };
                    so_at[j] = (int64_t)((uint64_t)word(0) | ((uint64_t)word(1) << 32));
                    so_nfree[j] = (int)word(2);
                    fp[0] = word(3), fp[1] = word(4), fp[2] = word(5), fp[3] = word(6);
                } else {
                    const JtMsg &m = tk.msg[JT_MAX_IN + j];
                    const uint32_t *fpw = reinterpret_cast<const uint32_t *>(m.free_pos);
                    so_at[j] = m.off + (int64_t)bk.pnum[j] * m.pstride + bk.gbase[JT_MAX_IN + j];
                    so_nfree[j] = m.nfree;
                    fp[0] = fpw[0], fp[1] = fpw[1], fp[2] = fpw[2], fp[3] = fpw[3];
                }
                so_glo[j] = jt_sub_lo(fp, so_nfree[j], tid);
                so_hiv[j] = jt_sub_hi(fp, so_nfree[j], lane);
            }
            const int64_t at = so_at[j];
            double *dst = msg_arena + fl.cur_off + at + fl.out_shift;
            double *oth = msg_arena + fl.oth_off + at;          // the half the next propagate will use
            const bool mark = fl.oth_off >= 0;
            const int n = 1 << so_nfree[j];
            for (int s = tid, it = 0; s < n; s += JT_THREADS, ++it) {
                const uint32_t idx = so_glo[j] + (uint32_t)__builtin_amdgcn_readlane((int)so_hiv[j], it);
                jt_msg_store<FLOW>(dst + idx, out_sub[j][s]);
                if (mark) oth[idx] = __longlong_as_double((long long)JT_UNWRITTEN);
            }
        }
    }
    JT_STAMP(11);
#ifdef JT_STAMPS
    if (dbg & 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        JT_STAMP(12);
    }
#endif
}

// Reduce task: entry i of the sum = the sum of the partial copies of entry i, JT_REDUCE_ENTRIES entries per
// workgroup.  Wave w sums copies [w * per, (w + 1) * per) of its lane's entry with ALL its loads in flight at once
// (up to 16 per lane; a message of 64 copies used to cost eight dependent round trips - 40 us on a loaded chip, the
// whole critical path of the top of a tree), the four partial sums are then added in wave order: a fixed order,
// bit-reproducible.  In a dataflow launch the copies may still be in the making: a wave loads until none of its
// entries is the "unwritten" marker (one poller per wave; the producers are the workgroups just ahead in the list).
// MULTI (multi-set plans; their messages have few copies): the four waves take two evidence sets each instead of a
// quarter of the copies - set w, then set w + 4 - and every wave stores its own sums.
template <bool FLOW, bool MULTI = false>
__device__ __forceinline__ void jt_reduce(const JtTask &tk, const JtBlock &bk, double *__restrict__ msg_arena,
                                          const JtFlow &fl, const int sidv = 0) {
    __shared__ double red_part[4][JT_REDUCE_ENTRIES];
    __shared__ uint32_t red_abort;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = (int64_t)1 << tk.nbits;
    const int64_t i = (int64_t)bk.xF + lane;
    const JtMsg &src = tk.msg[0];
    const bool active = i < n;
    const int npart = src.npart;
    if constexpr (MULTI) {
        for (int set = wave; set < JT_MSETS; set += 4) {
            double *arena = msg_arena + (int64_t)__builtin_amdgcn_readlane(sidv, set) * fl.set_stride;      // (MULTI: jt_mpass, `sidv`)
            const double *copies = arena + fl.cur_off + src.off + (active ? i : 0);
            double sum = 0.0;
            uint64_t t0 = 0;
            for (int p0 = 0; p0 < npart; p0 += 16) {
                double c[16];
                for (;;) {
                    const double *unready = nullptr;
#pragma unroll
                    for (int u = 0; u < 16; ++u) c[u] = (active && p0 + u < npart) ? jt_msg_load<FLOW>(copies + (int64_t)(p0 + u) * src.pstride) : 0.0;
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (FLOW && jt_unwritten(c[u])) unready = copies + (int64_t)(p0 + u) * src.pstride;
                    if (!FLOW || (fl.dbg & 4)) break;
                    const uint64_t have = __ballot(unready != nullptr);
                    if (have == 0) break;
                    bool give_up = false;
                    if (lane == (int)__builtin_ctzll(have)) {
                        if (t0 == 0) t0 = __builtin_amdgcn_s_memrealtime();
                        unsigned spins = 0;
                        while (jt_unwritten(jt_msg_load<true>(unready))) {
                            if (spins < 4) __builtin_amdgcn_s_sleep(8);
                            else __builtin_amdgcn_s_sleep(32);
                            if ((++spins & 15u) == 0) {
                                if (__hip_atomic_load(fl.sync + JT_SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) give_up = true;
                                else if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
                                    __hip_atomic_store(fl.sync + JT_SYNC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    __hip_atomic_store(fl.host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    give_up = true;
                                }
                                if (give_up) break;
                            }
                        }
                    }
                    if (__any(give_up)) return;              // (no barrier below on this path: every wave finds the flag itself)
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) sum += c[u];
            }
            if (active) {
                const int64_t at = tk.msg[JT_MAX_IN].off + i;
                jt_msg_store<FLOW>(arena + fl.cur_off + at, sum);
                if (fl.oth_off >= 0) arena[fl.oth_off + at] = __longlong_as_double((long long)JT_UNWRITTEN);
            }
        }
        return;
    }
    const int per = (npart + 3) >> 2;
    const int p_lo = wave * per, p_hi = (p_lo + per < npart) ? p_lo + per : npart;
    const double *copies = msg_arena + fl.cur_off + src.off + (active ? i : 0);
    const int64_t ps = src.pstride;
    double sum = 0.0;
    uint64_t t0 = 0;
    bool gave_up = false;
    if (threadIdx.x == 0) red_abort = 0;
    for (int p0 = p_lo; p0 < p_hi && !gave_up; p0 += 16) {
        double c[16];
        for (;;) {
            const double *unready = nullptr;
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u] = (active && p0 + u < p_hi) ? jt_msg_load<FLOW>(copies + (int64_t)(p0 + u) * ps) : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (FLOW && jt_unwritten(c[u])) unready = copies + (int64_t)(p0 + u) * ps;
            if (!FLOW || (fl.dbg & 4)) break;
            const uint64_t have = __ballot(unready != nullptr);
            if (have == 0) break;
            bool give_up = false;
            if (lane == (int)__builtin_ctzll(have)) {
                if (t0 == 0) t0 = __builtin_amdgcn_s_memrealtime();
                unsigned spins = 0;
                while (jt_unwritten(jt_msg_load<true>(unready))) {
                    if (spins < 4) __builtin_amdgcn_s_sleep(8);
                    else __builtin_amdgcn_s_sleep(32);
                    if ((++spins & 15u) == 0) {
                        if (__hip_atomic_load(fl.sync + JT_SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) give_up = true;
                        else if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
                            __hip_atomic_store(fl.sync + JT_SYNC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(fl.host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            give_up = true;
                        }
                        if (give_up) break;
                    }
                }
            }
            if (__any(give_up)) {
                gave_up = true;
                break;
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) sum += c[u];
    }
    __syncthreads();                               // (red_abort = 0 is visible)
    if (gave_up && lane == 0) red_abort = 1;
    red_part[wave][lane] = sum;
    __syncthreads();
    if (red_abort != 0) return;                    // the grid drains; the host runs the propagate again per level
    if (wave == 0 && active) {
        const double total = ((red_part[0][lane] + red_part[1][lane]) + red_part[2][lane]) + red_part[3][lane];
        const int64_t at = tk.msg[JT_MAX_IN].off + i;
        jt_msg_store<FLOW>(msg_arena + fl.cur_off + at, total);
        if (fl.oth_off >= 0) msg_arena[fl.oth_off + at] = __longlong_as_double((long long)JT_UNWRITTEN);
    }
}

template <typename T>
__global__ __launch_bounds__(JT_THREADS) void jt_reduce_level(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                              const int *__restrict__ itab, const T *__restrict__ psi,
                                                              T *__restrict__ bel, double *__restrict__ msg, JtFlow fl) {
    const JtBlock &bk = blk[blockIdx.x];
    jt_reduce<false>(tasks[bk.task], bk, msg, fl);
}

// ------------------------------------------------------------------------------------------
// The lean unit pass (round 6; JtLean, jtp_internal.h).  A unit task of ONE outgoing message whose incoming tables have one copy each,
// run for an evidence set that observes nothing:
//     out[S] = sum_{C \ S} prod_k in_k[S_k]          (computation.py:79-88 for a clique the reference keeps as length-1 axes,
//                                                     junctiontree.py:52-61; every downward message of such a clique likewise)
// Same sub-boxes, same LDS offsets, same iteration table, same marker protocol and the same epilogue as jt_pass<..., UNIT> - what is
// gone is the interpretation: no bit-deposit loops over free_pos[] (the host stored the weights), no branch per message and row on
// e_dep (the tables that depend on a thread's element bits come first and their number NE is a template parameter), no table row,
// no evidence, no belief, no ring, one read of a 704-byte record instead of scattered reads of a 2.3 KB one.  Where no incoming
// table depends on the element bits (NE = 0) the VEC elements of a thread share ONE product and one accumulator.
// Product order: in[0] * in[1] * ... in the RECORD's order (element-dependent tables first) - a fixed order, the same in every
// launch mode; the generic pass multiplies in JtTask order, so the two agree to rounding, not bit for bit.
// NOUT > 1 (read-out tasks only, jt_unit_single: up to three marginals of psi x every incoming table of a unit clique in one pass): every
// output accumulates the same product and folds after its own runs of rows; outputs 1 and 2 are described by the JtLeanMore record
// behind the lean record.
// COPIES (read-out tasks only): an incoming message may have several partial copies (JtLeanMsg::w_hi[5]), summed in copy order.
template <typename T, int NIN, int NE, bool FLOW, int NOUT = 1, bool COPIES = false>
__device__ __forceinline__ void jt_unit_lean(const JtLean &ln, const JtBlock &bk, const int *__restrict__ itab,
                                             double *__restrict__ msg_arena, const JtFlow &fl, uint32_t *flow_ctl) {
    constexpr int VEC = 16 / sizeof(T);               // elements of a thread per row (the plan's thread part: 256 threads x VEC)
    constexpr int NI = NIN > 0 ? NIN : 1;
    constexpr int NACC = NE > 0 ? VEC : 1;
    const JtLeanMore &lx = *reinterpret_cast<const JtLeanMore *>(&ln + 1);       // (read only where NOUT > 1)
    auto outm = [&](auto j_tag) -> const JtLeanMsg & {
        constexpr int j = decltype(j_tag)::value;
        if constexpr (j == 0) return ln.out;
        else return lx.out[j - 1];
    };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = ln.total;

    // ---- everything that does not depend on a message: the iteration table (row r in lane r), the thread's validity
    int trow[JT_NCOL];
    {
        const int *gtab = itab + ln.itab_off;
        const int r = lane < total ? lane : total - 1;
        const int4 a = *reinterpret_cast<const int4 *>(gtab + r * JT_NCOL);
        const int4 b = *reinterpret_cast<const int4 *>(gtab + r * JT_NCOL + 4);
        trow[0] = a.x; trow[1] = a.y; trow[2] = a.z; trow[3] = a.w;
        trow[4] = b.x; trow[5] = b.y; trow[6] = b.z; trow[7] = b.w;
    }
    uint32_t dead = 0;                                // bit e: element e of this thread names no entry of the clique
    if (ln.some_invalid) {
        const int *tm = itab + ln.tmap_off + tid * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) dead |= tm[e] < 0 ? 1u << e : 0u;
    }
    const bool chunk_ok = !(bk.flags & JT_BLOCK_INVALID);

    // a sub-box entry's place in its message: weights . bits of (round, thread)
    auto lo_of = [&](const JtLeanMsg &m) {
        uint32_t g = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) g += (((uint32_t)tid >> b) & 1u) * (uint32_t)m.w_lo[b];
        return g;
    };
    auto hi_of = [&](const JtLeanMsg &m) {            // (lane j: round j)
        uint32_t g = 0;
#pragma unroll
        for (int b = 0; b < JT_MAX_FREE - 8; ++b) g += (((uint32_t)lane >> b) & 1u) * (uint32_t)m.w_hi[b];
        return g;
    };

    // ---- stage the incoming sub-boxes, all of them in lock step: four rounds of 256 entries of every message in flight together
    const double *msg_cur = msg_arena + fl.cur_off;
    const double *src[NI];
    uint32_t lo[NI], hiv[NI];
    int nent[NI];
    bool thr_mem[NI];
    int rounds = 0;
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const JtLeanMsg &m = ln.in[k];
        const int gb = m.src == 0 ? bk.gbase[0] : (m.src == 1 ? bk.gbase[1] : (m.src == 2 ? bk.gbase[2] : bk.gbase[3]));
        src[k] = msg_cur + (m.off + gb + ((m.flags & 2) ? fl.fix_shift : 0));
        lo[k] = lo_of(m);
        hiv[k] = hi_of(m);
        nent[k] = 1 << m.nfree;
        thr_mem[k] = (m.flags & 1) != 0;
        const int r = (nent[k] + JT_THREADS - 1) >> 8;
        rounds = r > rounds ? r : rounds;
    }
    uint32_t out_lo[NOUT], out_hiv[NOUT];
    jt_static_for<NOUT>([&](auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        out_lo[j] = lo_of(outm(j_tag));
        out_hiv[j] = hi_of(outm(j_tag));
    });
    uint64_t wait_t0 = 0;
    for (int attempt = 0;; ++attempt) {
        const int settle_attempt = ln.settle ? attempt : 0;
        const double *unready = nullptr;
        for (int it0 = 0; it0 < rounds; it0 += 4) {
            double c[NI][4];
            const double *at[NI][4];
#pragma unroll
            for (int k = 0; k < NIN; ++k)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    at[k][u] = src[k] + (lo[k] + (uint32_t)__builtin_amdgcn_readlane((int)hiv[k], it0 + u));
                    c[k][u] = (it0 + u) * JT_THREADS + tid < nent[k] ? jt_msg_load<FLOW>(at[k][u], thr_mem[k]) : 0.0;
                }
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                double *sub = reinterpret_cast<double *>(smem + ln.in[k].lds_off);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if ((it0 + u) * JT_THREADS + tid >= nent[k]) continue;
                    if (FLOW) c[k][u] = jt_msg_settle<FLOW>(at[k][u], c[k][u], thr_mem[k], settle_attempt);
                    if (FLOW && jt_unwritten(c[k][u])) unready = at[k][u];
                    double sum = 0.0 + c[k][u];
                    if constexpr (COPIES) {
                        const int ncopy = ln.in[k].w_hi[5];
                        const int64_t cstride = ln.in[k].w_hi[6];
                        for (int p0 = 1; p0 < ncopy; p0 += 4) {
                            double d[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) d[q] = p0 + q < ncopy ? jt_msg_load<FLOW>(at[k][u] + (int64_t)(p0 + q) * cstride, thr_mem[k]) : 0.0;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                if (p0 + q >= ncopy) continue;
                                if constexpr (FLOW) {         // (folded marginals: the copies are in the making like the first one)
                                    const double *ap = at[k][u] + (int64_t)(p0 + q) * cstride;
                                    d[q] = jt_msg_settle<FLOW>(ap, d[q], thr_mem[k], settle_attempt);
                                    if (jt_unwritten(d[q])) unready = ap;
                                }
                                sum += d[q];
                            }
                        }
                    }
                    sub[(it0 + u) * JT_THREADS + tid] = sum;
                }
            }
        }
        if (attempt == 0) {
            jt_static_for<NOUT>([&](auto j_tag) {
                double *sub = reinterpret_cast<double *>(smem + outm(j_tag).lds_off);
                const int n = 1 << outm(j_tag).nfree;
                for (int s = tid; s < n; s += JT_THREADS) sub[s] = 0.0;
            });
        }
        if constexpr (!FLOW || NIN == 0) {
            __syncthreads();
            break;
        } else {
            // (the wait of jt_pass: one lane polls one entry that was not ready, with back-off; it gives up after 2 s or when another
            //  workgroup did, so that the grid always drains)
            if (fl.dbg & 4) unready = nullptr;
            if (fl.dbg & 8) unready = msg_cur;
            uint32_t *slot = flow_ctl + 4 + (attempt & 1) * 12;
            {
                const uint64_t have = __ballot(unready != nullptr);
                if (have != 0 && lane == (int)__builtin_ctzll(have)) {
                    slot[4 + 2 * wave] = (uint32_t)(uintptr_t)unready;
                    slot[5 + 2 * wave] = (uint32_t)((uintptr_t)unready >> 32);
                }
                if (lane == 0) slot[wave] = have != 0 ? 1u : 0u;
            }
            __syncthreads();
            const uint32_t w3 = slot[3], w2 = slot[2], w1 = slot[1], w0 = slot[0];
            if ((w0 | w1 | w2 | w3) == 0) break;
            if (tid == 0) {
                const int cw = w3 ? 3 : (w2 ? 2 : (w1 ? 1 : 0));
                const double *entry = reinterpret_cast<const double *>((uintptr_t)slot[4 + 2 * cw] | ((uintptr_t)slot[5 + 2 * cw] << 32));
                if (wait_t0 == 0) wait_t0 = __builtin_amdgcn_s_memrealtime();
                uint32_t give_up = 0;
                unsigned spins = 0;
                const uint64_t limit = (fl.dbg & 8) ? 2000000ull : 200000000ull;
                while (jt_unwritten(jt_msg_load<true>(entry)) || (fl.dbg & 8)) {
                    if (spins >= 64) __builtin_amdgcn_s_sleep(32);
                    else if (spins >= 16) __builtin_amdgcn_s_sleep(16);
                    if ((++spins & 15u) == 0) {
                        if (__hip_atomic_load(fl.sync + JT_SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) give_up = 1;
                        else if (__builtin_amdgcn_s_memrealtime() - wait_t0 > limit) {
                            __hip_atomic_store(fl.sync + JT_SYNC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(fl.host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            give_up = 1;
                        }
                        if (give_up) break;
                    }
                }
                flow_ctl[1] = give_up;
            }
            __syncthreads();
            if (flow_ctl[1] != 0) return;
        }
    }

    // ---- per-thread constants: the byte address inside LDS of this thread's entry of every incoming sub-box (a row adds its offset)
    auto slot_of = [&](const JtLeanMsg &m) {
        int t = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) t += ((lane >> b) & 1) * m.t_w[b];
#pragma unroll
        for (int b = 0; b < 2; ++b) t += ((wave >> b) & 1) * m.t_w[6 + b];
        return t;
    };
    uint32_t ua[NI][NE > 0 ? VEC : 1];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const JtLeanMsg &m = ln.in[k];
        const int t = slot_of(m);
        ua[k][0] = (uint32_t)m.lds_off + 8u * (uint32_t)t;
        if constexpr (NE > 0) {
            if (k < NE) {
#pragma unroll
                for (int e = 1; e < VEC; ++e) ua[k][e] = ua[k][0] + 8u * (uint32_t)(((e & 1) ? m.e_w[0] : 0) + ((e & 2) ? m.e_w[1] : 0));
            }
        }
    }
    double *out_sub[NOUT];
    int thr_out[NOUT], o_ew0[NOUT], o_ew1[NOUT], red_e[NOUT], red_lane[NOUT], red_wave[NOUT], rmask[NOUT], nph[NOUT], myph[NOUT];
    bool rep[NOUT];
    jt_static_for<NOUT>([&](auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        const JtLeanMsg &m = outm(j_tag);
        out_sub[j] = reinterpret_cast<double *>(smem + m.lds_off);
        thr_out[j] = slot_of(m);
        o_ew0[j] = m.e_w[0], o_ew1[j] = m.e_w[1];
        red_e[j] = j == 0 ? ln.red_e : lx.red_e[j > 0 ? j - 1 : 0];
        red_lane[j] = j == 0 ? ln.red_lane : lx.red_lane[j > 0 ? j - 1 : 0];
        red_wave[j] = j == 0 ? ln.red_wave : lx.red_wave[j > 0 ? j - 1 : 0];
        rmask[j] = j == 0 ? ln.rmask : lx.rmask[j > 0 ? j - 1 : 0];
        rep[j] = (lane & red_lane[j]) == 0;
        nph[j] = 1 << __builtin_popcount((unsigned)red_wave[j]);
        myph[j] = 0;
        if (red_wave[j] == 1) myph[j] = wave & 1;
        else if (red_wave[j] == 2) myph[j] = wave >> 1;
        else if (red_wave[j] == 3) myph[j] = wave;
    });

    double acc[NOUT][NACC];
#pragma unroll
    for (int j = 0; j < NOUT; ++j)
#pragma unroll
        for (int e = 0; e < NACC; ++e) acc[j][e] = 0.0;
    // (the record orders the tables its own way; a table's column of the iteration table is its place in JtTask::msg - picked once)
    int tcol[NI];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const int sk = ln.in[k].src;
        tcol[k] = sk == 0 ? trow[1] : (sk == 1 ? trow[2] : (sk == 2 ? trow[3] : trow[4]));
    }

    // fold this thread's sums of one run of rows into outgoing sub-box j (jt_pass: same sums, same order)
    auto epilogue = [&](auto j_tag, const int oo) {
        constexpr int j = decltype(j_tag)::value;
        double a[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = ((dead >> e) & 1u) ? 0.0 : acc[j][NE > 0 ? e : 0];
        if constexpr (VEC == 4) {
            if (red_e[j] & 1) {
                a[0] += a[1];
                a[2] += a[3];
            }
            if (red_e[j] & 2) {
                a[0] += a[2];
                a[1] += a[3];
            }
        } else {
            if (red_e[j] & 1) a[0] += a[1];
        }
        jt_lane_sums<VEC>(a, red_lane[j]);
        const int slot = oo + thr_out[j];
        for (int ph = 0; ph < nph[j]; ++ph) {
            if (rep[j] && myph[j] == ph) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    if ((e & red_e[j]) == 0) {
                        const int eo = ((e & 1) ? o_ew0[j] : 0) + ((e & 2) ? o_ew1[j] : 0);
                        __hip_atomic_fetch_add(&out_sub[j][slot + eo], a[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            if (nph[j] > 1) __syncthreads();
        }
#pragma unroll
        for (int e = 0; e < NACC; ++e) acc[j][e] = 0.0;
    };

#pragma unroll
    for (int c = 0; c < JT_NCOL; ++c) asm volatile("" : "+v"(trow[c]));
#pragma unroll
    for (int k = 0; k < NIN; ++k) asm volatile("" : "+v"(tcol[k]));

    // ---- the rows: nothing but look-ups and products.  The entries of row i + 1 are asked for before those of row i are used (one
    //      LDS round trip per row otherwise, and nothing to do in it).  CHECK: some row of the loop nest does not exist (a digit
    //      beyond a cardinality: JT_NO_ROW) - such a row adds nothing; plans of power-of-two cardinalities never look.
    struct Entries {
        double e[NE > 0 ? NE : 1][VEC], c[NIN - NE > 0 ? NIN - NE : 1];
    };
    auto fetch = [&](const int i, Entries &x) {
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const uint32_t rb = (uint32_t)__builtin_amdgcn_readlane(tcol[k], i) << 3;
            if constexpr (NE > 0) {
                if (k < NE) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) x.e[k < NE ? k : 0][e] = *reinterpret_cast<const double *>(smem + (ua[k][e] + rb));
                    continue;
                }
            }
            x.c[k - NE >= 0 ? k - NE : 0] = *reinterpret_cast<const double *>(smem + (ua[k][0] + rb));
        }
    };
    auto use = [&](const Entries &x) {
        if constexpr (NIN == 0) {
#pragma unroll
            for (int j = 0; j < NOUT; ++j) acc[j][0] += 1.0;
        } else if constexpr (NE == 0) {
            double w = x.c[0];
#pragma unroll
            for (int k = 1; k < NIN; ++k) w *= x.c[k];
#pragma unroll
            for (int j = 0; j < NOUT; ++j) acc[j][0] += w;
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                double w = x.e[0][e];
#pragma unroll
                for (int k = 1; k < NE; ++k) w *= x.e[k][e];
#pragma unroll
                for (int k = NE; k < NIN; ++k) w *= x.c[k - NE];
#pragma unroll
                for (int j = 0; j < NOUT; ++j) acc[j][e] += w;
            }
        }
    };
    auto rows = [&](auto check_tag) {
        constexpr bool CHECK = decltype(check_tag)::value;
        Entries cur, nxt;
        fetch(0, cur);
        for (int i = 0; i < total; ++i) {
            fetch(i + 1 < total ? i + 1 : i, nxt);
            if (!CHECK || (uint32_t)__builtin_amdgcn_readlane(trow[0], i) != JT_NO_ROW) use(cur);
            jt_static_for<NOUT>([&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
                if ((i & rmask[j]) == rmask[j]) epilogue(j_tag, __builtin_amdgcn_readlane(trow[1 + JT_MAX_IN + j], i));
            });
            cur = nxt;
        }
    };
    if (chunk_ok) {                                       // (a chunk whose own digits do not exist writes its zeros and nothing else)
        if (ln.some_norow) rows(std::integral_constant<bool, true>{});
        else rows(std::integral_constant<bool, false>{});
    }

    // ---- flush the outgoing sub-boxes as this chunk's partial copies; the same entries of the other arena half become "unwritten"
    __syncthreads();
    jt_static_for<NOUT>([&](auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        const JtLeanMsg &m = outm(j_tag);
        const int64_t at = m.off + (int64_t)bk.pnum[j] * (j == 0 ? ln.out_pstride : lx.out_pstride[j > 0 ? j - 1 : 0]) + bk.gbase[JT_MAX_IN + j];
        double *dst = msg_arena + fl.cur_off + at + fl.out_shift;
        double *oth = msg_arena + fl.oth_off + at;
        const bool mark = fl.oth_off >= 0;
        const int n = 1 << m.nfree;
        for (int s = tid, it = 0; s < n; s += JT_THREADS, ++it) {
            const uint32_t idx = out_lo[j] + (uint32_t)__builtin_amdgcn_readlane((int)out_hiv[j], it);
            jt_msg_store<FLOW>(dst + idx, out_sub[j][s]);
            if (mark) oth[idx] = __longlong_as_double((long long)JT_UNWRITTEN);
        }
    });
}

// (which of the ten row loops: incoming tables x those among them that depend on the element bits)
template <typename T, bool FLOW>
__device__ __forceinline__ void jt_unit_lean_dispatch(const JtLean &ln, const JtBlock &bk, const int *__restrict__ itab,
                                                      double *__restrict__ msg, const JtFlow &fl, uint32_t *flow_ctl) {
#define JT_LEAN(NIN, NE) jt_unit_lean<T, NIN, NE, FLOW>(ln, bk, itab, msg, fl, flow_ctl); break
    switch (ln.n_in * 8 + ln.n_e) {
        case 0: JT_LEAN(0, 0);
        case 8: JT_LEAN(1, 0);
        case 9: JT_LEAN(1, 1);
        case 16: JT_LEAN(2, 0);
        case 17: JT_LEAN(2, 1);
        case 18: JT_LEAN(2, 2);
        case 24: JT_LEAN(3, 0);
        case 25: JT_LEAN(3, 1);
        case 26: JT_LEAN(3, 2);
        default: JT_LEAN(3, 3);                           // (jtp_make_lean: at most three incoming tables)
    }
#undef JT_LEAN
}

// Read-out tasks of unit cliques (jtp_get_marginals -> jt_single; no launch waits on anything): up to four incoming tables - the parent's
// message, the static table and two children, all of them - and one to three marginals per pass.
// FLOW: the same tasks FOLDED into a propagate's dataflow launch (JtTask::fold, jtp_plan.cpp fold_marginals): their inputs are messages of
// this launch, waited for like any other.
template <typename T, bool FLOW = false>
__device__ __forceinline__ void jt_unit_lean_readout(const JtLean &ln, const JtBlock &bk, const int *__restrict__ itab,
                                                     double *__restrict__ msg, const JtFlow &fl, uint32_t *flow_ctl = nullptr) {
#define JT_LEAN_OUT(NIN, NE)                                                                          \
    if (ln.n_out == 1) jt_unit_lean<T, NIN, NE, FLOW, 1, true>(ln, bk, itab, msg, fl, flow_ctl);            \
    else if (ln.n_out == 2) jt_unit_lean<T, NIN, NE, FLOW, 2, true>(ln, bk, itab, msg, fl, flow_ctl);       \
    else jt_unit_lean<T, NIN, NE, FLOW, 3, true>(ln, bk, itab, msg, fl, flow_ctl);                          \
    break
    switch (ln.n_in * 8 + ln.n_e) {
        case 0: JT_LEAN_OUT(0, 0);
        case 8: JT_LEAN_OUT(1, 0);
        case 9: JT_LEAN_OUT(1, 1);
        case 16: JT_LEAN_OUT(2, 0);
        case 17: JT_LEAN_OUT(2, 1);
        case 18: JT_LEAN_OUT(2, 2);
        case 24: JT_LEAN_OUT(3, 0);
        case 25: JT_LEAN_OUT(3, 1);
        case 26: JT_LEAN_OUT(3, 2);
        case 27: JT_LEAN_OUT(3, 3);
        case 32: JT_LEAN_OUT(4, 0);
        case 33: JT_LEAN_OUT(4, 1);
        case 34: JT_LEAN_OUT(4, 2);
        case 35: JT_LEAN_OUT(4, 3);
        default: JT_LEAN_OUT(4, 4);
    }
#undef JT_LEAN_OUT
}

// A dataflow workgroup whose record says "lean" (JT_BLOCK_LEAN) goes from its workgroup record straight to the task's lean record:
// one dependent scalar round trip less than through the task record.  (A clique that hosts an observed variable runs the generic pass;
// first_x[5] = the task's planner node, the index of its entry of the evidence table.)
// FOLD: the kernel can run folded marginal tasks (JT_BLOCK_FOLD; jt_propagate_flow_marg only - the other dataflow kernels end such a
// workgroup at once, as every kernel does where the clique hosts an observed variable: the engine then forms those marginals by
// the read-out, jtp_get_marginals).
template <typename T, bool FOLD = false>
__device__ __forceinline__ bool jt_lean_block(const JtBlock &bk, const int *__restrict__ itab, double *__restrict__ msg, const JtFlow &fl,
                                              uint32_t *flow_ctl) {
    if (!(bk.flags & JT_BLOCK_LEAN)) return false;        // (a folded task always has a lean record: jtp_plan.cpp finish())
    const bool observed = fl.ev != nullptr && fl.ev[2 * bk.first_x[5]] != 0;
    const int64_t at = (int64_t)((uint64_t)bk.first_x[6] | ((uint64_t)bk.first_x[7] << 32));
    if (bk.flags & JT_BLOCK_FOLD) {
        if constexpr (FOLD) {
            if (!observed) jt_unit_lean_readout<T, true>(*reinterpret_cast<const JtLean *>(itab + at), bk, itab, msg, fl, flow_ctl);
        }
        return true;
    }
    if (observed) return false;                          // (this clique hosts an observed variable: the generic pass)
    jt_unit_lean_dispatch<T, true>(*reinterpret_cast<const JtLean *>(itab + at), bk, itab, msg, fl, flow_ctl);
    return true;
}

// Unit tasks (JtTask::unit): their shapes differ from the table-keeping tasks' - the static table is one more incoming one
// (collect: up to three children, or the static table and two; distribute: the parent's message and / or the static table, then
// up to three children) - so they are dispatched here, by every kernel that may meet one.
#define JT_UNIT_PASS(NIN, NOUT, MODE) jt_pass<T, NIN, NOUT, MODE, FLOW, true, TMIX, false, true, false>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry)
#define JT_UNIT_BELIEF(NIN) jt_pass<T, NIN, 0, 1, FLOW, true, TMIX, false, true, true>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry)
template <typename T, bool FLOW, bool TMIX, bool FOLD = false>
__device__ __forceinline__ void jt_unit_collect(const JtTask &tk, const JtBlock &bk, const int *__restrict__ itab, const T *__restrict__ psi,
                                                T *__restrict__ bel, double *__restrict__ msg, const JtFlow &fl, uint32_t bindex,
                                                uint32_t *flow_ctl, uint64_t t_entry) {
    if constexpr (FOLD && !TMIX && !FLOW) {
        // a marginal task folded into the propagate: the lean pass or nothing.  Only the per-level distribute kernel meets one here
        // (FOLD): dataflow workgroups of such tasks are taken by jt_lean_block, and no other launch list holds them.
        if (tk.fold) {
            if (tk.lean_off > 0 && (fl.ev == nullptr || fl.ev[2 * tk.pnode] == 0))
                jt_unit_lean_readout<T, false>(*reinterpret_cast<const JtLean *>(itab + tk.lean_off), bk, itab, msg, fl);
            return;
        }
    }
    if constexpr (!TMIX) {
        // (round 6) a task of one outgoing message and single-copy inputs on a clique that hosts no observed variable of this
        // evidence set (the engine passes no table at all when nothing is observed): the lean pass
        if (tk.lean_off > 0 && (fl.ev == nullptr || fl.ev[2 * tk.pnode] == 0)) {
            jt_unit_lean_dispatch<T, FLOW>(*reinterpret_cast<const JtLean *>(itab + tk.lean_off), bk, itab, msg, fl, flow_ctl);
            return;
        }
    }
    switch (tk.n_in) {
        case 0: JT_UNIT_PASS(0, 1, 0); break;
        case 1: JT_UNIT_PASS(1, 1, 0); break;
        case 2: JT_UNIT_PASS(2, 1, 0); break;
        default: JT_UNIT_PASS(3, 1, 0); break;
    }
}
template <typename T, bool FLOW, bool TMIX>
__device__ __forceinline__ void jt_unit_distribute(const JtTask &tk, const JtBlock &bk, const int *__restrict__ itab, const T *__restrict__ psi,
                                                   T *__restrict__ bel, double *__restrict__ msg, const JtFlow &fl, uint32_t bindex,
                                                   uint32_t *flow_ctl, uint64_t t_entry) {
    switch ((tk.n_in - tk.n_out) * 4 + tk.n_out) {        // (inputs that are not children) x children
        case 0: JT_UNIT_PASS(0, 0, 1); break;
        case 1: JT_UNIT_PASS(1, 1, 1); break;
        case 2: JT_UNIT_PASS(2, 2, 1); break;
        case 3: JT_UNIT_PASS(3, 3, 1); break;
        case 4: JT_UNIT_PASS(1, 0, 1); break;
        case 5: JT_UNIT_PASS(2, 1, 1); break;
        case 6: JT_UNIT_PASS(3, 2, 1); break;
        case 7: JT_UNIT_PASS(4, 3, 1); break;
        case 8: JT_UNIT_PASS(2, 0, 1); break;
        case 9: JT_UNIT_PASS(3, 1, 1); break;
        default: JT_UNIT_PASS(4, 2, 1); break;
    }
}
// read-out tasks of unit cliques (no launch waits on anything): one to three marginals of psi x every incoming table per pass,
// or the belief itself into the scratch arena
template <typename T, bool TMIX>
__device__ __forceinline__ void jt_unit_single(const JtTask &tk, const JtBlock &bk, const int *__restrict__ itab, const T *__restrict__ psi,
                                               T *__restrict__ bel, double *__restrict__ msg, const JtFlow &fl, uint32_t bindex) {
    constexpr bool FLOW = false;
    uint32_t *flow_ctl = nullptr;
    const uint64_t t_entry = 0;
    if (tk.mode == 0) {
        switch (tk.n_out * 8 + tk.n_in) {
            case 8: JT_UNIT_PASS(0, 1, 0); break;
            case 9: JT_UNIT_PASS(1, 1, 0); break;
            case 10: JT_UNIT_PASS(2, 1, 0); break;
            case 11: JT_UNIT_PASS(3, 1, 0); break;
            case 12: JT_UNIT_PASS(4, 1, 0); break;
            case 16: JT_UNIT_PASS(0, 2, 0); break;
            case 17: JT_UNIT_PASS(1, 2, 0); break;
            case 18: JT_UNIT_PASS(2, 2, 0); break;
            case 19: JT_UNIT_PASS(3, 2, 0); break;
            case 20: JT_UNIT_PASS(4, 2, 0); break;
            case 24: JT_UNIT_PASS(0, 3, 0); break;
            case 25: JT_UNIT_PASS(1, 3, 0); break;
            case 26: JT_UNIT_PASS(2, 3, 0); break;
            case 27: JT_UNIT_PASS(3, 3, 0); break;
            default: JT_UNIT_PASS(4, 3, 0); break;
        }
    } else {
        switch (tk.n_in) {            // (the belief itself, into the scratch arena)
            case 0: JT_UNIT_BELIEF(0); break;
            case 1: JT_UNIT_BELIEF(1); break;
            case 2: JT_UNIT_BELIEF(2); break;
            case 3: JT_UNIT_BELIEF(3); break;
            default: JT_UNIT_BELIEF(4); break;
        }
    }
}

// Entry points (these names appear in rocprofv3 traces).  One launch covers every clique of
// one tree level, whatever its number of neighbours: the workgroup dispatches on its task.
template <typename T>
__global__ __launch_bounds__(JT_THREADS, 4) void jt_collect_level(const JtTask *__restrict__ tasks,
                                                               const JtBlock *__restrict__ blk, const int *__restrict__ itab,
                                                               const T *__restrict__ psi, T *__restrict__ bel,
                                                               double *__restrict__ msg, JtFlow fl) {
    const JtBlock &bk = blk[blockIdx.x];
    const JtTask &tk = tasks[bk.task];
    if (tk.unit) {
        jt_unit_collect<T, false, false>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x, nullptr, 0);
        return;
    }
    switch (tk.n_in) {
        case 0: jt_pass<T, 0, 1, 0>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        case 1: jt_pass<T, 1, 1, 0>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        case 2: jt_pass<T, 2, 1, 0>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        default: jt_pass<T, 3, 1, 0>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
    }
}

template <typename T>
__global__ __launch_bounds__(JT_THREADS, 3) void jt_distribute_level(const JtTask *__restrict__ tasks,
                                                                  const JtBlock *__restrict__ blk, const int *__restrict__ itab,
                                                                  const T *__restrict__ psi, T *__restrict__ bel,
                                                                  double *__restrict__ msg, JtFlow fl) {
    const JtBlock &bk = blk[blockIdx.x];
    const JtTask &tk = tasks[bk.task];
    if (tk.unit) {           // (a unit clique's downward messages are marginalisations of their own: mode 0 in the distribute phase)
        if (tk.mode == 0) jt_unit_collect<T, false, false, true>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x, nullptr, 0);
        else jt_unit_distribute<T, false, false>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x, nullptr, 0);
        return;
    }
    switch ((tk.n_in - tk.n_out) * 4 + tk.n_out) {
        case 0: jt_pass<T, 0, 0, 1>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        case 1: jt_pass<T, 1, 1, 1>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        case 2: jt_pass<T, 2, 2, 1>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        case 3: jt_pass<T, 3, 3, 1>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        case 4: jt_pass<T, 1, 0, 1>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        case 5: jt_pass<T, 2, 1, 1>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        case 6: jt_pass<T, 3, 2, 1>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        default: jt_pass<T, 4, 3, 1>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
    }
}

// Dataflow entry points: ONE launch per phase.  Workgroups take the block list in blockIdx order; the
// waits cannot deadlock as long as no workgroup is dispatched before one with a lower index (then the
// lowest unfinished workgroup is always running, and it never waits on an unfinished one).  That is
// how the hardware dispatches; should it ever not be, the wait times out, the host notices and runs
// the propagate again with one launch per level (jtp_engine.hip: check_flow).  With JTP_FLOW_TICKETS
// the list position is drawn from an atomic counter instead, which needs no such assumption but costs
// a memory round trip before anything else can start (~10% on the benchmark tree).
__device__ __forceinline__ uint32_t jt_flow_ticket(const JtFlow &fl, uint32_t *flow_ctl) {
    if (fl.ticket_idx == 0xffffffffu) return blockIdx.x;
    if (threadIdx.x == 0)
        flow_ctl[0] = __hip_atomic_fetch_add(fl.sync + fl.ticket_idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - fl.ticket_base;
    __syncthreads();
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)flow_ctl[0]);
}

#ifndef JT_FLOW_WAVES
#define JT_FLOW_WAVES 4          // (A/B builds: -DJT_FLOW_WAVES=5 / 6 - fewer registers, more workgroups per CU)
#endif
template <typename T>
__global__ __launch_bounds__(JT_THREADS, JT_FLOW_WAVES) void jt_collect_flow(const JtTask *__restrict__ tasks,
                                                              const JtBlock *__restrict__ blk, const int *__restrict__ itab,
                                                              const T *__restrict__ psi, T *__restrict__ bel,
                                                              double *__restrict__ msg, JtFlow fl) {
    __shared__ uint32_t flow_ctl[28 + 8 * JT_MAX_OUT];        // ticket, wait flags and candidates, parked flush records
    const uint64_t t_entry = __builtin_amdgcn_s_memrealtime();
    const uint32_t ticket = jt_flow_ticket(fl, flow_ctl);
    const JtBlock &bk = blk[ticket];
    if (jt_lean_block<T>(bk, itab, msg, fl, flow_ctl)) return;
    const JtTask &tk = tasks[bk.task];
    if (tk.kind != 0) {
        jt_reduce<true>(tk, bk, msg, fl);
        return;
    }
    if (tk.unit) {
        jt_unit_collect<T, true, false>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry);
        return;
    }
    switch (tk.n_in) {
        case 0: jt_pass<T, 0, 1, 0, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 1: jt_pass<T, 1, 1, 0, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 2: jt_pass<T, 2, 1, 0, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        default: jt_pass<T, 3, 1, 0, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
    }
}

template <typename T, bool EARLY = false>
__device__ __forceinline__ void jt_distribute_flow_body(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                        const int *__restrict__ itab, const T *__restrict__ psi, T *__restrict__ bel,
                                                        double *__restrict__ msg, const JtFlow &fl, uint32_t *flow_ctl) {
    const uint64_t t_entry = __builtin_amdgcn_s_memrealtime();
    const uint32_t ticket = jt_flow_ticket(fl, flow_ctl);
    const JtBlock &bk = blk[ticket];
    if (jt_lean_block<T>(bk, itab, msg, fl, flow_ctl)) return;
    const JtTask &tk = tasks[bk.task];
    if (tk.kind != 0) {
        jt_reduce<true>(tk, bk, msg, fl);
        return;
    }
    if (tk.unit) {           // (a unit clique's downward messages are marginalisations of their own: mode 0 in the distribute phase)
        if (tk.mode == 0) jt_unit_collect<T, true, false>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry);
        else jt_unit_distribute<T, true, false>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry);
        return;
    }
    switch ((tk.n_in - tk.n_out) * 4 + tk.n_out) {
        case 0: jt_pass<T, 0, 0, 1, true, EARLY>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 1: jt_pass<T, 1, 1, 1, true, EARLY>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 2: jt_pass<T, 2, 2, 1, true, EARLY>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 3: jt_pass<T, 3, 3, 1, true, EARLY>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 4: jt_pass<T, 1, 0, 1, true, EARLY>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 5: jt_pass<T, 2, 1, 1, true, EARLY>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 6: jt_pass<T, 3, 2, 1, true, EARLY>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        default: jt_pass<T, 4, 3, 1, true, EARLY>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
    }
}

// Both phases in ONE launch (round 3; plans whose phases follow each other without an exchange in between, and whose
// messages are small beside their tables - jtp_plan.cpp: finish()): the block list of the distribute phase simply
// follows that of the collect phase, and the root's distribute workgroups wait for the last upward messages like any
// other consumer.  Saves the second launch's cold start and the gap between the launches (~10 us of 620 on the width-20
// tree); every message is then read through to memory (JtMsg::same_launch), also the upward messages the distribute
// tasks read, because their producers ran in THIS launch.
// FOLD: the build that also runs the marginal tasks folded into the propagate (jt_propagate_flow_marg: plans that have such tasks).  A
// kernel of its own because the plans without them - every hot path of bench.py - pay for the code otherwise: config 3 5.92 -> 6.07 ms
// with one kernel for both (A/B on one box, profiles/r06_ab_fold_hot_path.txt).
template <typename T, bool FOLD>
__device__ __forceinline__ void jt_propagate_flow_body(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                       const int *__restrict__ itab, const T *__restrict__ psi, T *__restrict__ bel,
                                                       double *__restrict__ msg, const JtFlow &fl, uint32_t *flow_ctl) {
    const uint64_t t_entry = __builtin_amdgcn_s_memrealtime();
    const uint32_t ticket = jt_flow_ticket(fl, flow_ctl);
    const JtBlock &bk = blk[ticket];
    if (jt_lean_block<T, FOLD>(bk, itab, msg, fl, flow_ctl)) return;
    const JtTask &tk = tasks[bk.task];
    if (tk.kind != 0) {
        jt_reduce<true>(tk, bk, msg, fl);
        return;
    }
    if (tk.unit) {
        if (tk.mode == 0) jt_unit_collect<T, true, false>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry);
        else jt_unit_distribute<T, true, false>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry);
        return;
    }
    if (tk.mode == 0) {
        switch (tk.n_in) {
            case 0: jt_pass<T, 0, 1, 0, true, true, false, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
            case 1: jt_pass<T, 1, 1, 0, true, true, false, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
            case 2: jt_pass<T, 2, 1, 0, true, true, false, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
            default: jt_pass<T, 3, 1, 0, true, true, false, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        }
        return;
    }
    switch ((tk.n_in - tk.n_out) * 4 + tk.n_out) {
        case 0: jt_pass<T, 0, 0, 1, true, false, false, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 1: jt_pass<T, 1, 1, 1, true, false, false, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 2: jt_pass<T, 2, 2, 1, true, false, false, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 3: jt_pass<T, 3, 3, 1, true, false, false, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 4: jt_pass<T, 1, 0, 1, true, false, false, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 5: jt_pass<T, 2, 1, 1, true, false, false, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        case 6: jt_pass<T, 3, 2, 1, true, false, false, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
        default: jt_pass<T, 4, 3, 1, true, false, false, true>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry); break;
    }
}

template <typename T>
__global__ __launch_bounds__(JT_THREADS, JT_FLOW_WAVES) void jt_propagate_flow(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                                   const int *__restrict__ itab, const T *__restrict__ psi,
                                                                   T *__restrict__ bel, double *__restrict__ msg, JtFlow fl) {
    __shared__ uint32_t flow_ctl[28 + 8 * JT_MAX_OUT];
    jt_propagate_flow_body<T, false>(tasks, blk, itab, psi, bel, msg, fl, flow_ctl);
}

template <typename T>
__global__ __launch_bounds__(JT_THREADS, JT_FLOW_WAVES) void jt_propagate_flow_marg(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                                        const int *__restrict__ itab, const T *__restrict__ psi,
                                                                        T *__restrict__ bel, double *__restrict__ msg, JtFlow fl) {
    __shared__ uint32_t flow_ctl[28 + 8 * JT_MAX_OUT];
    jt_propagate_flow_body<T, true>(tasks, blk, itab, psi, bel, msg, fl, flow_ctl);
}

// Two builds of the distribute pass.  Compiled for four waves per SIMD (128 registers, a few spills) a CU holds four
// workgroups instead of three and a third more table rows are in flight: config 4 0.4505 -> 0.4385 ms (A/B on one box).
// On plans made of latency-bound levels (chains, JtTask::settle) the spills sit on the dependent path - config 2
// 3.56 -> 3.69 ms - so those run the build without (168 registers, three waves).
template <typename T>
__global__ __launch_bounds__(JT_THREADS, JT_FLOW_WAVES) void jt_distribute_flow(const JtTask *__restrict__ tasks,
                                                                    const JtBlock *__restrict__ blk, const int *__restrict__ itab,
                                                                    const T *__restrict__ psi, T *__restrict__ bel,
                                                                    double *__restrict__ msg, JtFlow fl) {
    __shared__ uint32_t flow_ctl[28 + 8 * JT_MAX_OUT];        // ticket, wait flags and candidates, parked flush records
    jt_distribute_flow_body<T>(tasks, blk, itab, psi, bel, msg, fl, flow_ctl);
}

#ifndef JT_CHAIN_WAVES
#define JT_CHAIN_WAVES 3
#endif
template <typename T>
__global__ __launch_bounds__(JT_THREADS, JT_CHAIN_WAVES) void jt_distribute_flow_chain(const JtTask *__restrict__ tasks,
                                                                          const JtBlock *__restrict__ blk, const int *__restrict__ itab,
                                                                          const T *__restrict__ psi, T *__restrict__ bel,
                                                                          double *__restrict__ msg, JtFlow fl) {
    __shared__ uint32_t flow_ctl[28 + 8 * JT_MAX_OUT];        // ticket, wait flags and candidates, parked flush records
    #ifdef JT_CHAIN_NO_EARLY
    jt_distribute_flow_body<T, false>(tasks, blk, itab, psi, bel, msg, fl, flow_ctl);
#else
    jt_distribute_flow_body<T, true>(tasks, blk, itab, psi, bel, msg, fl, flow_ctl);   // (flush records kept in registers)
#endif
}

// ------------------------------------------------------------------------------------------
// Multi-set pass (JTP_MULTISET plans): JT_MSETS evidence sets that share ONE copy of
// the clique tables are served by ONE pass over a table.  Per set s of the group
//     out_s[S] = sum_{C \ S} psi[C] * e_s[C] * prod_k in_{k,s}[S_k]
// (collect, every downward message - planned as its own marginalisation with the parent's message and the
// siblings' upward messages as inputs - and every marginal have this shape; e_s = the set's hard evidence).
// A table row travels HBM -> LDS -> registers once and is multiplied into the G sets' message entries:
// HBM bytes per evidence set fall by G, the arithmetic (two to four f64 operations per element and set)
// becomes the bound.  Each set has its own LDS region of SETB bytes with the sub-boxes of its messages at
// the offsets the single-set planner computed; SETB is a compile-time constant so that the G reads of one
// message entry differ only in the instruction's immediate offset.  The sets' message arenas lie
// fl.set_stride doubles apart, their evidence tables fl.ev_stride words apart.
//
// Evidence: the part of a set's (mask, value) that lies in the ROW bits (chunk + loop bits) is uniform per
// row - a row that contradicts it is skipped for that set; the part in the thread bits (16-byte vector, lane,
// wave) is constant over the loop, so it is applied to the register sums in the epilogue, not per element.
// `sidv` (round 6): lane j < G holds the ARENA SLOT of the j-th evidence set this workgroup serves - entries 8 g .. 8 g + 7 of the
// task's active list (JtFlow::act_ids), or simply 8 g + j; `msg0` is the base of ALL the sets' arenas.
template <typename T, int NIN, int SETB, bool ESUM = false>
__device__ __forceinline__ void jt_mpass(const JtTask &tk, const JtBlock &bk, const int *__restrict__ itab,
                                         const T *__restrict__ psi_arena, double *__restrict__ msg0,
                                         const JtFlow &fl, uint32_t *flow_ctl, uint32_t bindex, const int sidv) {
    constexpr int G = JT_MSETS;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int EB = (VEC == 4) ? 2 : 1;
    constexpr int TBITS = EB + 8;
    constexpr int U = JT_U;
    constexpr bool FLOW = true;
    using VT = typename JtVec<T>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t xF = bk.xF + (uint32_t)tid * VEC;
    const T *psi = psi_arena + tk.psi_off;
    const int total = tk.total;
    const int rmask = (1 << tk.nR) - 1;
    const int64_t sstride = fl.set_stride;
    int64_t soff[G];                                             // (uniform) where the arena of the s-th set of this workgroup starts
#pragma unroll
    for (int s = 0; s < G; ++s) soff[s] = (int64_t)__builtin_amdgcn_readlane(sidv, s) * sstride;
#ifdef JT_STAMPS
    const int dbg = (fl.dbg & 0x80000000u) ? (tk.debug & ~2) : tk.debug;                      // (time stamps of group 0 only)
    double *stamp_out = msg0 + tk.dbg_off + (int64_t)(fl.blk_base + bindex) * JT_NSTAMP;      // (as in jt_pass)
#else
    const int dbg = tk.debug;
#endif
    JT_STAMP(0);

    const int *gtab = itab + tk.itab_off;
    const T *psi0 = psi_arena + bk.psi_x0 + (uint32_t)tid * VEC;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem) + (uint32_t)wave * (U * 1024);
    const char *ring = smem + wave * (U * 1024) + lane * 16;
    const T *zero_row = psi_arena + (uint32_t)tid * VEC;         // what rows that do not exist read (JT_NO_ROW)
    const bool chunk_ok = !(bk.flags & JT_BLOCK_INVALID);
#pragma unroll
    for (int u = 0; u < U; ++u)
        jt_dma16(bk.first_x[u] == JT_NO_ROW ? zero_row : psi0 + bk.first_x[u], __builtin_amdgcn_readfirstlane(ring_lds + u * 1024), fl.n_groups > 1);
    int trow[JT_NCOL];
    {
        const int r = lane < total ? lane : total - 1;
        const int4 a = *reinterpret_cast<const int4 *>(gtab + r * JT_NCOL);
        const int4 b = *reinterpret_cast<const int4 *>(gtab + r * JT_NCOL + 4);
        trow[0] = a.x; trow[1] = a.y; trow[2] = a.z; trow[3] = a.w;
        trow[4] = b.x; trow[5] = b.y; trow[6] = b.z; trow[7] = b.w;
    }

    JT_STAMP(1);
    // evidence of the G sets on this clique, set s in lane s of a register pair (read after the first element
    // loads have left: the table is two dependent loads away; kept in vector registers: sixteen more scalars
    // live across the loop made hipcc spill scalars into it)
    uint32_t ev_m = 0, ev_v = 0;
    if (fl.ev != nullptr && lane < G) {
        ev_m = fl.ev[(size_t)sidv * fl.ev_stride + 2 * tk.pnode];
        ev_v = fl.ev[(size_t)sidv * fl.ev_stride + 2 * tk.pnode + 1];
    }
    // ---- stage the incoming sub-boxes of every set (one thread per entry, partial copies summed in copy
    //      order), zero the outgoing ones; wait for entries still marked unwritten (dataflow launches)
    const double *msg_cur = msg0 + fl.cur_off;
    char *sets = smem + JT_RING_BYTES;                       // region of set s: sets + s * SETB
    const JtMsg &mo = tk.msg[JT_MAX_IN];
    uint64_t wait_t0 = 0;
    for (int attempt = 0;; ++attempt) {
        JT_STAMP_AT(13, attempt + 1);
        const double *unready = nullptr;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const JtMsg &m = tk.msg[k];
            const int nfree = m.nfree;
            const uint32_t *fpw = reinterpret_cast<const uint32_t *>(m.free_pos);
            const uint32_t fp[4] = {fpw[0], fpw[1], fpw[2], fpw[3]};
            const bool thr_mem = m.same_launch != 0;
            const int n = 1 << nfree;
            // (an upward message whose producer did not run for one of the sets - nothing observed below it: that set's entries are the
            //  evidence-free set's, arena slot 0; bit s of `own` = the producer's list holds the s-th set of this workgroup)
            uint32_t own = 0xffu;
            if (fl.skip != nullptr && m.src_task >= 0)
                own = (uint32_t)__ballot(lane < G && fl.skip[(size_t)m.src_task * fl.cap + (uint32_t)sidv] != 0);
            int64_t from[G];
#pragma unroll
            for (int s = 0; s < G; ++s) from[s] = ((own >> s) & 1u) ? soff[s] : (int64_t)0;
            for (int i = tid; i < n; i += JT_THREADS) {
                int idx = 0;
#pragma unroll
                for (int b = 0; b < JT_MAX_FREE; ++b)
                    if (b < nfree) idx += ((i >> b) & 1) << JT_FPOS(fp, b);
                const double *src = msg_cur + m.off + bk.gbase[k] + idx;
                double sum[G];
#pragma unroll
                for (int s = 0; s < G; ++s) sum[s] = 0.0;
                for (int p = 0; p < m.npart; p += 4) {               // the G sets' copies p .. p+3 in flight together
                    double c[G][4];
#pragma unroll
                    for (int s = 0; s < G; ++s)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            c[s][u] = (p + u < m.npart) ? jt_msg_load<FLOW>(src + from[s] + (int64_t)(p + u) * m.pstride, thr_mem) : 0.0;
#pragma unroll
                    for (int s = 0; s < G; ++s)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            sum[s] += c[s][u];
                            if (jt_unwritten(c[s][u])) unready = src + from[s] + (int64_t)(p + u) * m.pstride;
                        }
                }
#pragma unroll
                for (int s = 0; s < G; ++s) reinterpret_cast<double *>(sets + s * SETB + (m.lds_off - JT_RING_BYTES))[i] = sum[s];
            }
        }
        if (attempt == 0) {
            const int n = 1 << mo.nfree;
            for (int i = tid; i < n; i += JT_THREADS)
#pragma unroll
                for (int s = 0; s < G; ++s) reinterpret_cast<double *>(sets + s * SETB + (mo.lds_off - JT_RING_BYTES))[i] = 0.0;
        }
        if constexpr (NIN == 0) {
            __syncthreads();
            break;
        } else {
            if (fl.dbg & 8) unready = msg_cur;                           // fault injection: wait for ever
            uint32_t *slot = flow_ctl + 4 + (attempt & 1) * 12;
            {
                const uint64_t have = __ballot(unready != nullptr);
                if (have != 0 && lane == (int)__builtin_ctzll(have)) {
                    slot[4 + 2 * wave] = (uint32_t)(uintptr_t)unready;
                    slot[5 + 2 * wave] = (uint32_t)((uintptr_t)unready >> 32);
                }
                if (lane == 0) slot[wave] = have != 0 ? 1u : 0u;
            }
            __syncthreads();
            const uint32_t w3 = slot[3], w2 = slot[2], w1 = slot[1], w0 = slot[0];
            if ((w0 | w1 | w2 | w3) == 0) break;
            if (tid == 0) {
                const int cw = w3 ? 3 : (w2 ? 2 : (w1 ? 1 : 0));
                const double *entry = reinterpret_cast<const double *>((uintptr_t)slot[4 + 2 * cw] | ((uintptr_t)slot[5 + 2 * cw] << 32));
                if (wait_t0 == 0) wait_t0 = __builtin_amdgcn_s_memrealtime();
                uint32_t give_up = 0;
                unsigned spins = 0;
                const uint64_t limit = (fl.dbg & 8) ? 2000000ull : 200000000ull;
                while (jt_unwritten(jt_msg_load<true>(entry)) || (fl.dbg & 8)) {
                    if (spins >= 64) __builtin_amdgcn_s_sleep(32);
                    else if (spins >= 16) __builtin_amdgcn_s_sleep(16);
                    if ((++spins & 15u) == 0) {
                        if (__hip_atomic_load(fl.sync + JT_SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) give_up = 1;
                        else if (__builtin_amdgcn_s_memrealtime() - wait_t0 > limit) {
                            __hip_atomic_store(fl.sync + JT_SYNC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(fl.host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            give_up = 1;
                        }
                        if (give_up) break;
                    }
                }
                flow_ctl[1] = give_up;
            }
            __syncthreads();
            if (flow_ctl[1] != 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // LDS-DMA loads must land before the wave ends
                return;
            }
        }
    }

    JT_STAMP(3);
    // ---- per-thread constants (read before the first store of the kernel, see jt_pass) ---------------
    int thr_in[NIN > 0 ? NIN : 1], in_lds[NIN > 0 ? NIN : 1], in_edep[NIN > 0 ? NIN : 1];
    int in_ew0[NIN > 0 ? NIN : 1], in_ew1[NIN > 0 ? NIN : 1];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const JtMsg &m = tk.msg[k];
        int t = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) t += ((lane >> b) & 1) * m.t_w[b];
#pragma unroll
        for (int b = 0; b < 2; ++b) t += ((wave >> b) & 1) * m.t_w[6 + b];
        thr_in[k] = t;
        in_lds[k] = m.lds_off - JT_RING_BYTES;
        in_edep[k] = m.e_dep;
        in_ew0[k] = m.e_w[0];
        in_ew1[k] = m.e_w[1];
    }
    int thr_out = 0;
    {
#pragma unroll
        for (int b = 0; b < 6; ++b) thr_out += ((lane >> b) & 1) * mo.t_w[b];
#pragma unroll
        for (int b = 0; b < 2; ++b) thr_out += ((wave >> b) & 1) * mo.t_w[6 + b];
    }
    const int o_lds = mo.lds_off - JT_RING_BYTES;
    const int o_rede = mo.red_e, o_redl = mo.red_lane, o_redw = mo.red_wave;
    const int o_ew0 = mo.e_w[0], o_ew1 = mo.e_w[1];
    const int o_nfree = mo.nfree;
    const int64_t o_at = mo.off + (int64_t)bk.pnum[0] * mo.pstride + bk.gbase[JT_MAX_IN];
    const uint32_t *ofpw = reinterpret_cast<const uint32_t *>(mo.free_pos);
    const uint32_t ofp[4] = {ofpw[0], ofpw[1], ofpw[2], ofpw[3]};
    // thread part of the evidence: bit (4 s + e) set = element e of this thread agrees with set s
    constexpr uint32_t TMASK = (1u << TBITS) - 1u;
    uint64_t tmatch = 0;
#pragma unroll
    for (int s = 0; s < G; ++s) {
        const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)ev_m, s) & TMASK, v = (uint32_t)__builtin_amdgcn_readlane((int)ev_v, s);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            if (((((uint32_t)tid * VEC + e) ^ v) & m) == 0) tmatch |= 1ull << (4 * s + e);
    }
    // row part (chunk and loop bits): lane s keeps set s's mask and value; one compare + ballot per row
    const uint32_t row_m = ev_m & ~TMASK, row_v = ev_v & ~TMASK;
    const bool has_row_ev = __ballot(row_m != 0) != 0;
    uint32_t loop_pos[JT_MAX_ITER_LOG2];                // logical index bit of loop-counter bit t
#pragma unroll
    for (int t = 0; t < JT_MAX_ITER_LOG2; ++t) loop_pos[t] = t < tk.nA + tk.nR ? tk.loop_pos[t] : 31u;
    // ESUM (JtTask::esum): no message and no evidence involves the element bits - the elements of a vector are
    // summed first and ONE accumulator per evidence set is kept
    constexpr int NACC = ESUM ? 1 : VEC;
    double acc[G][NACC];
#pragma unroll
    for (int s = 0; s < G; ++s)
#pragma unroll
        for (int e = 0; e < NACC; ++e) acc[s][e] = 0.0;

    auto epilogue = [&](const int oo) {
        // thread part of the evidence, then the in-thread and cross-lane sums of every set
#pragma unroll
        for (int s = 0; s < G; ++s) {
#pragma unroll
            for (int e = 0; e < NACC; ++e)
                if (!((tmatch >> (4 * s + e)) & 1ull)) acc[s][e] = 0.0;
            if constexpr (!ESUM) {
                if constexpr (VEC == 4) {
                    if (o_rede & 1) {
                        acc[s][0] += acc[s][1];
                        acc[s][2] += acc[s][3];
                    }
                    if (o_rede & 2) {
                        acc[s][0] += acc[s][2];
                        acc[s][1] += acc[s][3];
                    }
                } else {
                    if (o_rede & 1) acc[s][0] += acc[s][1];
                }
            }
        }
#ifdef JT_MULTI_OLD_SHUFFLES
#pragma nounroll
        for (int b = 0; b < 6; ++b) {
            if ((o_redl >> b) & 1) {
#pragma unroll
                for (int s = 0; s < G; ++s)
#pragma unroll
                    for (int e = 0; e < NACC; ++e)
                        if ((e & o_rede) == 0) acc[s][e] += jt_shfl_xor(acc[s][e], 1 << b);
            }
        }
#else
        // (lane bits 0-3 through DPP row shifts, every set and element together: jt_lane_sums)
#pragma unroll
        for (int s = 0; s < G; ++s) jt_lane_sums<NACC>(acc[s], o_redl);
#endif
        const bool rep = (lane & o_redl) == 0;
        const int slot = oo + thr_out;
        const int nph = 1 << __builtin_popcount((unsigned)o_redw);
        int myph = 0;
        if (o_redw == 1) myph = wave & 1;
        else if (o_redw == 2) myph = wave >> 1;
        else if (o_redw == 3) myph = wave;
        for (int ph = 0; ph < nph; ++ph) {
            if (rep && myph == ph) {
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    double *osub = reinterpret_cast<double *>(sets + s * SETB + o_lds);
#pragma unroll
                    for (int e = 0; e < NACC; ++e) {
                        if ((e & o_rede) == 0) {
                            const int eo = ((e & 1) ? o_ew0 : 0) + ((e & 2) ? o_ew1 : 0);
                            __hip_atomic_fetch_add(&osub[slot + eo], acc[s][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
            if (nph > 1) __syncthreads();
        }
#pragma unroll
        for (int s = 0; s < G; ++s)
#pragma unroll
            for (int e = 0; e < NACC; ++e) acc[s][e] = 0.0;
    };

#pragma unroll
    for (int c = 0; c < JT_NCOL; ++c) asm volatile("" : "+v"(trow[c]));

    // EDEP: some incoming message depends on the element bits of the 16-byte vector (then every message
    // entry is read per element, four reads; otherwise one read per message serves the four elements).
    // The choice is made once per workgroup, outside the loop: inside, a step has no branch at all, so the
    // LDS reads of one evidence set are in flight while the previous set's products are formed.
    auto step = [&](auto slot_tag, auto edep_tag, auto rowev_tag, const int i) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr bool EDEP = decltype(edep_tag)::value;
        constexpr bool ROWEV = decltype(rowev_tag)::value;      // some set of the group observes a variable on the row bits
        constexpr int NV = EDEP ? VEC : 1;
        jt_wait_vmcnt<U - 1>();
        const VT v = *reinterpret_cast<const VT *>(ring + SLOT * 1024);
        double p[VEC];
        p[0] = (double)v.x;
        p[1] = (double)v.y;
        if constexpr (VEC == 4) {
            p[2] = (double)v.z;
            p[3] = (double)v.w;
        }
        {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int inext = (i + U < total) ? i + U : total - 1;
            const uint32_t xnext = (uint32_t)__builtin_amdgcn_readlane(trow[0], inext);
            jt_dma16((xnext == JT_NO_ROW || !chunk_ok) ? zero_row : psi + (xF + xnext),
                     __builtin_amdgcn_readfirstlane(ring_lds + SLOT * 1024), fl.n_groups > 1);
        }
        uint32_t rowok = 0xffffffffu;                       // bit s: the row agrees with set s
        if constexpr (ROWEV) {
            uint32_t xrow = bk.lxF;                         // LOGICAL index of the row (uniform): chunk + loop bits
#pragma unroll
            for (int t = 0; t < JT_MAX_ITER_LOG2; ++t) xrow += (((uint32_t)i >> t) & 1u) << loop_pos[t];
            rowok = (uint32_t)__ballot(((xrow ^ row_v) & row_m) == 0);
        }
        // byte offsets (inside a set's region) of this thread's entries of every incoming message
        int ad[NIN > 0 ? NIN : 1][NV];
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const int base = (__builtin_amdgcn_readlane(trow[1 + k], i) + thr_in[k]) * 8 + in_lds[k];
            ad[k][0] = base;
            if constexpr (EDEP) {
                ad[k][1] = base + in_ew0[k] * 8;
                if constexpr (VEC == 4) {
                    ad[k][2] = base + in_ew1[k] * 8;
                    ad[k][3] = base + (in_ew0[k] + in_ew1[k]) * 8;
                }
            }
        }
        auto load_set = [&](const int s, double (&dst)[NIN > 0 ? NIN : 1][NV]) {
            const char *reg = sets + s * SETB;
#pragma unroll
            for (int k = 0; k < NIN; ++k)
#pragma unroll
                for (int e = 0; e < NV; ++e) dst[k][e] = *reinterpret_cast<const double *>(reg + ad[k][e]);
        };
        // A row that contradicts a set's evidence contributes nothing to that set: its factor rs (1.0 or 0.0,
        // uniform) is the multiplier of the accumulating fma.  Two-stage pipeline over the sets: the entries of
        // set s + 1 are requested before the products of set s are formed.
        double psum = 0.0;
        if constexpr (ESUM) {
            psum = p[0] + p[1];
            if constexpr (VEC == 4) psum += p[2] + p[3];
        }
        if constexpr (!EDEP) {
            // No message depends on the element bits: one entry per message and set, so the entries of ALL eight sets
            // are requested at once (16 LDS reads in flight for two messages; a look-ahead of one set left every
            // set's products waiting a full LDS latency: 0.57 us per row with three waves per SIMD to hide it) and
            // ONE product of a set's entries serves the four elements (NIN multiplications + VEC fused multiply-adds
            // per set and row; with ESUM one).
            double all[G][NIN > 0 ? NIN : 1];
#pragma unroll
            for (int s = 0; s < G; ++s)
#pragma unroll
                for (int k = 0; k < NIN; ++k) all[s][k] = *reinterpret_cast<const double *>(sets + s * SETB + ad[k][0]);
#pragma unroll
            for (int s = 0; s < G; ++s) {
                // (the factor of a row that contradicts the set's evidence is uniform: built in scalar registers it is
                //  one more multiplication per set, not a select as well; without row evidence in the group - decided
                //  once per workgroup - it is not there at all: 24 instead of 56 vector operations per row with two
                //  incoming messages)
                double t = 1.0;
                if constexpr (ROWEV) t = __hiloint2double(__builtin_amdgcn_readfirstlane(((rowok >> s) & 1u) ? 0x3FF00000 : 0), 0);
                if constexpr (NIN > 0 && !ROWEV) {
                    t = all[s][0];
#pragma unroll
                    for (int k = 1; k < NIN; ++k) t *= all[s][k];
                } else {
#pragma unroll
                    for (int k = 0; k < NIN; ++k) t *= all[s][k];
                }
                if constexpr (ESUM) acc[s][0] = __builtin_fma(psum, t, acc[s][0]);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[s][e] = __builtin_fma(p[e], t, acc[s][e]);
                }
            }
        } else {
            double cur[NIN > 0 ? NIN : 1][NV];
#pragma unroll
            for (int s = 0; s < G; ++s) {
                load_set(s, cur);
                if constexpr (ROWEV) {
                    const double rs = __hiloint2double(__builtin_amdgcn_readfirstlane(((rowok >> s) & 1u) ? 0x3FF00000 : 0), 0);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        double w = p[e];
#pragma unroll
                        for (int k = 0; k < NIN; ++k) w *= cur[k][e];
                        acc[s][e] = __builtin_fma(w, rs, acc[s][e]);
                    }
                } else {
                    // (no set of the group observes a variable on the row bits - decided once per workgroup: the last message
                    //  entry is the multiplier of the accumulating fma, NIN operations per element and set instead of NIN + 1)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        if constexpr (NIN == 0) acc[s][e] += p[e];
                        else {
                            double w = p[e];
#pragma unroll
                            for (int k = 0; k + 1 < NIN; ++k) w *= cur[k][e];
                            acc[s][e] = __builtin_fma(w, cur[NIN - 1][e], acc[s][e]);
                        }
                    }
                }
            }
        }
        if ((i & rmask) == rmask) epilogue(__builtin_amdgcn_readlane(trow[1 + JT_MAX_IN], i));
    };

    JT_STAMP(4);
    using std::integral_constant;
    bool any_edep = false;
#pragma unroll
    for (int k = 0; k < NIN; ++k) any_edep = any_edep || in_edep[k] != 0;
    auto loop = [&](auto edep_tag, auto rowev_tag) {
        for (int i0 = 0; i0 < total; i0 += U) {
            step(integral_constant<int, 0>{}, edep_tag, rowev_tag, i0);
            step(integral_constant<int, 1>{}, edep_tag, rowev_tag, i0 + 1);
            step(integral_constant<int, 2>{}, edep_tag, rowev_tag, i0 + 2);
            step(integral_constant<int, 3>{}, edep_tag, rowev_tag, i0 + 3);
        }
    };
    // (ESUM: the planner sets JtTask::esum only where no message has element bits)
    bool plain = true;
    if constexpr (!ESUM) {
        if (any_edep) {
            loop(integral_constant<bool, true>{}, integral_constant<bool, true>{});
            plain = false;
        }
    }
    if (plain) {
        if (has_row_ev) loop(integral_constant<bool, false>{}, integral_constant<bool, true>{});
        else loop(integral_constant<bool, false>{}, integral_constant<bool, false>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the ring's last (repeated) loads land before LDS is given back
    JT_STAMP(9);

    // ---- flush every set's outgoing sub-box as this chunk's partial copy --------------------------------
    __syncthreads();
    {
        const int n = 1 << o_nfree;
        const bool mark = fl.oth_off >= 0;
        for (int i = tid; i < n; i += JT_THREADS) {
            int idx = 0;
#pragma unroll
            for (int b = 0; b < JT_MAX_FREE; ++b)
                if (b < o_nfree) idx += ((i >> b) & 1) << JT_FPOS(ofp, b);
#pragma unroll
            for (int s = 0; s < G; ++s) {
                double *base = msg0 + soff[s] + o_at + idx;
                jt_msg_store<FLOW>(base + fl.cur_off, reinterpret_cast<const double *>(sets + s * SETB + o_lds)[i]);
                if (mark) base[fl.oth_off] = __longlong_as_double((long long)JT_UNWRITTEN);
            }
        }
    }
    JT_STAMP(11);
#ifdef JT_STAMPS
    if (dbg & 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        JT_STAMP(12);
    }
#endif
}

// Evidence-free subtrees (rounds 5-6).  Arena slot 0 of a multi-set plan holds a set that observes nothing.  The upward message of a
// clique below which a set observes nothing IS slot 0's: the set is not on that collect task's active list (JtFlow::act_ids), consumers
// and the read-out take the message from slot 0 (JtFlow::skip / jtp_engine.hip readout_redirect), and the set's own entries of it stay
// "unwritten" in both arena halves - which this pass restores, once, for the (task, set) pairs that LEAVE a list when the evidence changes.
#define JT_FANOUT_RESET 0x20000000
struct JtFanout {
    int64_t off;               // msg-arena offset (doubles) of the entries
    int32_t count;             // doubles
    int32_t flags;             // JT_FANOUT_RESET: the entries of BOTH arena halves of the listed slots become "unwritten"
    uint16_t slot[JT_MSETS];   // the arena slots concerned (0xffff: none)
};
#ifndef JT_INST_TU
// (fl.oth_off: the distance of the second arena half from the first)
__global__ __launch_bounds__(256) void jt_multi_fanout(const JtFanout *__restrict__ list, double *__restrict__ msg, JtFlow fl) {
    const JtFanout f = list[blockIdx.x];
    if (!(f.flags & JT_FANOUT_RESET)) return;
    for (int i = threadIdx.x; i < f.count; i += 256)
#pragma unroll
        for (int s = 0; s < JT_MSETS; ++s) {
            if (f.slot[s] == 0xffffu) continue;
            double *base = msg + (int64_t)f.slot[s] * fl.set_stride + f.off + i;
            base[0] = __longlong_as_double((long long)JT_UNWRITTEN);
            base[fl.oth_off] = __longlong_as_double((long long)JT_UNWRITTEN);
        }
}
#endif

// Multi-set entry point: grid.y = group of JT_MSETS evidence sets; the block list is that of a whole phase
// (dataflow launch) or of one tree level.  Workgroups of one group only ever wait for workgroups of the same
// group earlier in the list.
#ifndef JT_MULTI_WAVES
#define JT_MULTI_WAVES 3
#endif
template <typename T>
__global__ __launch_bounds__(JT_THREADS, JT_MULTI_WAVES) void jt_multi_flow(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                            const int *__restrict__ itab, const T *__restrict__ psi,
                                                            T *__restrict__ bel, double *__restrict__ msg, JtFlow fl) {
    __shared__ uint32_t flow_ctl[28 + 8 * JT_MAX_OUT];        // ticket, wait flags and candidates, parked flush records
    // 1-D grid, the GROUP index inside the record index: workgroups b .. b+7 of group g, then the same eight records for
    // group g+1, ...  The groups of evidence sets stream the SAME table rows; dispatched next to each other - and, with the
    // round-robin placement of consecutive workgroups over the eight XCDs, on the same XCD - the second to last group find
    // the rows the first one fetched in the L2 / Infinity Cache instead of fetching 1 GiB of tables again per group
    // (round 2: grid.y = group, every group a sweep of its own over all tables).  Order inside a group is unchanged
    // (record index ascending with blockIdx), which is all the dataflow waits need.
    const uint32_t octet = blockIdx.x / (8u * fl.n_groups), within = blockIdx.x % (8u * fl.n_groups);
    const uint32_t grp = within / 8u, rec = octet * 8u + (within & 7u);
    if (rec >= fl.n_blocks) return;
    if (grp != 0) fl.dbg |= 0x80000000u;                     // (diagnostic builds: group 0 writes the time stamps)
    fl.sync += (size_t)grp * fl.sync_stride;
    double *msg0 = msg;                                      // (the sets' arenas: slot x set_stride doubles from here)
    const uint32_t ticket = fl.ticket_idx == 0xffffffffu ? rec : jt_flow_ticket(fl, flow_ctl);
    const JtBlock &bk = blk[ticket];
    if (bk.flags & JT_BLOCK_NULL) return;                    // (padding of a level's records to a multiple of eight)
    const JtTask &tk = tasks[bk.task];
    // Which evidence sets: entries 8 g .. 8 g + 7 of the task's active list (a shorter last run repeats its last entry: the same
    // values are then stored twice), or - without lists - arena slots 8 g .. 8 g + 7.  Nothing on the list for this group: done.
    int sidv = (int)(grp * 8u + (threadIdx.x & 7u));
    bool esum_ok = ((tk.esum_groups >> (grp & 63u)) & 1ull) != 0;
    if (fl.act_n != nullptr) {
        const int n_act = fl.act_n[bk.task];
        if ((int)(grp * 8u) >= n_act) return;
        const int j = (int)(grp * 8u + (threadIdx.x & 7u));
        sidv = (int)fl.act_ids[(size_t)bk.task * fl.cap + (uint32_t)(j < n_act ? j : n_act - 1)];
        esum_ok = fl.esum_oct[(size_t)bk.task * fl.n_groups + grp] != 0;
    }
    if (tk.kind != 0) {
        jt_reduce<true, true>(tk, bk, msg0, fl, sidv);
        return;
    }
    if (tk.setb <= JT_SETB_SMALL && (tk.esum & 1) && esum_ok) {
        switch (tk.n_in) {
            case 0: jt_mpass<T, 0, JT_SETB_SMALL, true>(tk, bk, itab, psi, msg0, fl, flow_ctl, ticket, sidv); break;
            case 1: jt_mpass<T, 1, JT_SETB_SMALL, true>(tk, bk, itab, psi, msg0, fl, flow_ctl, ticket, sidv); break;
            case 2: jt_mpass<T, 2, JT_SETB_SMALL, true>(tk, bk, itab, psi, msg0, fl, flow_ctl, ticket, sidv); break;
            default: jt_mpass<T, 3, JT_SETB_SMALL, true>(tk, bk, itab, psi, msg0, fl, flow_ctl, ticket, sidv); break;
        }
    } else if (tk.setb <= JT_SETB_SMALL) {
        switch (tk.n_in) {
            case 0: jt_mpass<T, 0, JT_SETB_SMALL>(tk, bk, itab, psi, msg0, fl, flow_ctl, ticket, sidv); break;
            case 1: jt_mpass<T, 1, JT_SETB_SMALL>(tk, bk, itab, psi, msg0, fl, flow_ctl, ticket, sidv); break;
            case 2: jt_mpass<T, 2, JT_SETB_SMALL>(tk, bk, itab, psi, msg0, fl, flow_ctl, ticket, sidv); break;
            default: jt_mpass<T, 3, JT_SETB_SMALL>(tk, bk, itab, psi, msg0, fl, flow_ctl, ticket, sidv); break;
        }
    } else {
        switch (tk.n_in) {
            case 0: jt_mpass<T, 0, JT_SETB_LARGE>(tk, bk, itab, psi, msg0, fl, flow_ctl, ticket, sidv); break;
            case 1: jt_mpass<T, 1, JT_SETB_LARGE>(tk, bk, itab, psi, msg0, fl, flow_ctl, ticket, sidv); break;
            case 2: jt_mpass<T, 2, JT_SETB_LARGE>(tk, bk, itab, psi, msg0, fl, flow_ctl, ticket, sidv); break;
            default: jt_mpass<T, 3, JT_SETB_LARGE>(tk, bk, itab, psi, msg0, fl, flow_ctl, ticket, sidv); break;
        }
    }
}

// One task list of single-set passes of any shape (read-out of multi-set plans: beliefs and marginals formed on
// demand from the shared table and one set's final messages; up to four incoming messages).
template <typename T>
__global__ __launch_bounds__(JT_THREADS) void jt_single(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                        const int *__restrict__ itab, const T *__restrict__ psi,
                                                        T *__restrict__ bel, double *__restrict__ msg, JtFlow fl) {
    const JtBlock &bk = blk[blockIdx.x];
    const JtTask &tk = tasks[bk.task];
    if (tk.unit) {
        jt_unit_single<T, false>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x);
        return;
    }
    if (tk.mode == 0) {
        switch (tk.n_in) {
            case 0: jt_pass<T, 0, 1, 0>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
            case 1: jt_pass<T, 1, 1, 0>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
            case 2: jt_pass<T, 2, 1, 0>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
            case 3: jt_pass<T, 3, 1, 0>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
            default: jt_pass<T, 4, 1, 0>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        }
    } else {                                   // belief only: psi * every incoming message
        switch (tk.n_in) {
            case 0: jt_pass<T, 0, 0, 1>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
            case 1: jt_pass<T, 1, 0, 1>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
            case 2: jt_pass<T, 2, 0, 1>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
            case 3: jt_pass<T, 3, 0, 1>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
            default: jt_pass<T, 4, 0, 1>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        }
    }
}

// Marginals of unit cliques by the lean pass (round 6): every task of the list has a lean record (jtp_get_marginals makes them) and the
// evidence set observes nothing (the engine sends the list through jt_single otherwise).  A kernel of its own: jt_single holds the
// generic passes of up to four inputs and three outputs, whose registers would halve the occupancy of these loops.
template <typename T>
__global__ __launch_bounds__(JT_THREADS, 4) void jt_lean_single(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                                const int *__restrict__ itab, const T *__restrict__ psi,
                                                                T *__restrict__ bel, double *__restrict__ msg, JtFlow fl) {
    const JtBlock &bk = blk[blockIdx.x];
    jt_unit_lean_readout<T>(*reinterpret_cast<const JtLean *>(itab + tasks[bk.task].lean_off), bk, itab, msg, fl);
}

// Read-out of single-set plans (jtp_get_marginals, CliqueGraph.marginalize junctiontree.py:229-274): a pass over a BELIEF
// table (the `psi` argument) that forms one to JT_MAX_OUT marginals of it at once - the requests of a model's factors on one
// clique share the read of its table.  Bound: HBM read, sizeof(T) per element.
template <typename T>
__global__ __launch_bounds__(JT_THREADS, 4) void jt_marginals(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                              const int *__restrict__ itab, const T *__restrict__ psi,
                                                              T *__restrict__ bel, double *__restrict__ msg, JtFlow fl) {
    const JtBlock &bk = blk[blockIdx.x];
    const JtTask &tk = tasks[bk.task];
    switch (tk.n_out) {
        case 1: jt_pass<T, 0, 1, 0>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        case 2: jt_pass<T, 0, 2, 0>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
        default: jt_pass<T, 0, 3, 0>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x); break;
    }
}

// ------------------------------------------------------------------------------------------
// Kernels of plans with a mixed-radix thread part (HostPlan::tmix: some clique stores the variables of its low index
// bits at their TRUE cardinalities): the same passes with TMIX = true - a thread reaches its elements through the
// clique's thread map (JtTask::tmap_off) - for every launch style the engine uses with such plans: one launch per
// phase (dataflow), one per level, and the read-out task lists.
template <typename T, bool FLOW, bool VG>
__device__ __forceinline__ void jt_collect_mix_v(const JtTask &tk, const JtBlock &bk, const int *__restrict__ itab, const T *__restrict__ psi,
                                               T *__restrict__ bel, double *__restrict__ msg, const JtFlow &fl, uint32_t bindex,
                                               uint32_t *flow_ctl, uint64_t t_entry) {
    if (tk.unit) {
        if constexpr (FLOW) jt_unit_collect<T, true, true>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry);
        else jt_unit_single<T, true>(tk, bk, itab, psi, bel, msg, fl, bindex);      // (level launches and read-out lists)
        return;
    }
    if (tk.n_out > 1) {                        // read-out tasks: several marginals of one belief table per pass
        if (tk.n_out == 2) jt_pass<T, 0, 2, 0, FLOW, true, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry);
        else jt_pass<T, 0, 3, 0, FLOW, true, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry);
        return;
    }
    switch (tk.n_in) {
        case 0: jt_pass<T, 0, 1, 0, FLOW, true, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
        case 1: jt_pass<T, 1, 1, 0, FLOW, true, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
        case 2: jt_pass<T, 2, 1, 0, FLOW, true, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
        case 3: jt_pass<T, 3, 1, 0, FLOW, true, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
        default: jt_pass<T, 4, 1, 0, FLOW, true, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
    }
}
template <typename T, bool FLOW, bool VG>
__device__ __forceinline__ void jt_distribute_mix_v(const JtTask &tk, const JtBlock &bk, const int *__restrict__ itab, const T *__restrict__ psi,
                                                  T *__restrict__ bel, double *__restrict__ msg, const JtFlow &fl, uint32_t bindex,
                                                  uint32_t *flow_ctl, uint64_t t_entry) {
    if (tk.unit) {
        if (tk.mode == 0) jt_collect_mix_v<T, FLOW, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry);      // (a downward message of a unit clique: its own marginalisation)
        else if (tk.n_out == 0 && !FLOW) jt_unit_single<T, true>(tk, bk, itab, psi, bel, msg, fl, bindex);       // (read-out: up to four incoming tables)
        else jt_unit_distribute<T, FLOW, true>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry);
        return;
    }
    if (tk.n_out == 0) {                       // belief only (leaves, and the read-out of multi-neighbour cliques)
        switch (tk.n_in) {
            case 0: jt_pass<T, 0, 0, 1, FLOW, false, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
            case 1: jt_pass<T, 1, 0, 1, FLOW, false, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
            case 2: jt_pass<T, 2, 0, 1, FLOW, false, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
            case 3: jt_pass<T, 3, 0, 1, FLOW, false, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
            default: jt_pass<T, 4, 0, 1, FLOW, false, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
        }
        return;
    }
    switch ((tk.n_in - tk.n_out) * 4 + tk.n_out) {
        case 1: jt_pass<T, 1, 1, 1, FLOW, false, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
        case 2: jt_pass<T, 2, 2, 1, FLOW, false, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
        case 3: jt_pass<T, 3, 3, 1, FLOW, false, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
        case 5: jt_pass<T, 2, 1, 1, FLOW, false, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
        case 6: jt_pass<T, 3, 2, 1, FLOW, false, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
        default: jt_pass<T, 4, 3, 1, FLOW, false, true, false, false, true, VG>(tk, bk, itab, psi, bel, msg, fl, bindex, flow_ctl, t_entry); break;
    }
}

// (VG: the compact form of the rows, two per step - JtTask::vgroups == 2 on every table-keeping task of the plan, HostPlan::tmix_compact -
//  is a kernel of its own: built into one kernel with the one-row form, either lost 10-20 %)
template <typename T, bool VG = false>
__global__ __launch_bounds__(JT_THREADS, 3) void jt_collect_level_mix(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                                      const int *__restrict__ itab, const T *__restrict__ psi,
                                                                      T *__restrict__ bel, double *__restrict__ msg, JtFlow fl) {
    const JtBlock &bk = blk[blockIdx.x];
    jt_collect_mix_v<T, false, VG>(tasks[bk.task], bk, itab, psi, bel, msg, fl, blockIdx.x, nullptr, 0);
}
template <typename T, bool VG = false>
__global__ __launch_bounds__(JT_THREADS, 3) void jt_distribute_level_mix(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                                         const int *__restrict__ itab, const T *__restrict__ psi,
                                                                         T *__restrict__ bel, double *__restrict__ msg, JtFlow fl) {
    const JtBlock &bk = blk[blockIdx.x];
    jt_distribute_mix_v<T, false, VG>(tasks[bk.task], bk, itab, psi, bel, msg, fl, blockIdx.x, nullptr, 0);
}
// (read-out task lists: marginals - mode 0 - and beliefs of multi-neighbour cliques - mode 1)
template <typename T, bool VG = false>
__global__ __launch_bounds__(JT_THREADS, 3) void jt_single_mix(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                               const int *__restrict__ itab, const T *__restrict__ psi,
                                                               T *__restrict__ bel, double *__restrict__ msg, JtFlow fl) {
    const JtBlock &bk = blk[blockIdx.x];
    const JtTask &tk = tasks[bk.task];
    if (tk.mode == 0) jt_collect_mix_v<T, false, VG>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x, nullptr, 0);
    else jt_distribute_mix_v<T, false, VG>(tk, bk, itab, psi, bel, msg, fl, blockIdx.x, nullptr, 0);
}
template <typename T, bool VG = false>
__global__ __launch_bounds__(JT_THREADS, JT_MIX_WAVES) void jt_collect_flow_mix(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                                     const int *__restrict__ itab, const T *__restrict__ psi,
                                                                     T *__restrict__ bel, double *__restrict__ msg, JtFlow fl) {
    __shared__ uint32_t flow_ctl[28 + 8 * JT_MAX_OUT];
    const uint64_t t_entry = __builtin_amdgcn_s_memrealtime();
    const uint32_t ticket = jt_flow_ticket(fl, flow_ctl);
    const JtBlock &bk = blk[ticket];
    const JtTask &tk = tasks[bk.task];
    if (tk.kind != 0) jt_reduce<true>(tk, bk, msg, fl);
    else jt_collect_mix_v<T, true, VG>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry);
}
template <typename T, bool VG = false>
__global__ __launch_bounds__(JT_THREADS, JT_MIX_WAVES) void jt_distribute_flow_mix(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk,
                                                                        const int *__restrict__ itab, const T *__restrict__ psi,
                                                                        T *__restrict__ bel, double *__restrict__ msg, JtFlow fl) {
    __shared__ uint32_t flow_ctl[28 + 8 * JT_MAX_OUT];
    const uint64_t t_entry = __builtin_amdgcn_s_memrealtime();
    const uint32_t ticket = jt_flow_ticket(fl, flow_ctl);
    const JtBlock &bk = blk[ticket];
    const JtTask &tk = tasks[bk.task];
    if (tk.kind != 0) jt_reduce<true>(tk, bk, msg, fl);
    else jt_distribute_mix_v<T, true, VG>(tk, bk, itab, psi, bel, msg, fl, ticket, flow_ctl, t_entry);
}

// Per-shape entry points, used when the plan is built with JTP_SPLIT_VARIANTS (profiling aid:
// one launch per (level, neighbour count), so rocprofv3 attributes time to each shape).
template <typename T, int NCH>
__global__ __launch_bounds__(JT_THREADS) void jt_collect(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk, const int *__restrict__ itab,
                                                         const T *__restrict__ psi, T *__restrict__ bel,
                                                         double *__restrict__ msg, JtFlow fl) {
    const JtBlock &bk = blk[blockIdx.x];
    jt_pass<T, NCH, 1, 0>(tasks[bk.task], bk, itab, psi, bel, msg, fl, blockIdx.x);
}

template <typename T, int HASP, int NCH>
__global__ __launch_bounds__(JT_THREADS) void jt_distribute(const JtTask *__restrict__ tasks, const JtBlock *__restrict__ blk, const int *__restrict__ itab,
                                                            const T *__restrict__ psi, T *__restrict__ bel,
                                                            double *__restrict__ msg, JtFlow fl) {
    const JtBlock &bk = blk[blockIdx.x];
    jt_pass<T, HASP + NCH, NCH, 1>(tasks[bk.task], bk, itab, psi, bel, msg, fl, blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// Layout conversion and synthetic fill (off the hot path).

__device__ __forceinline__ uint64_t jt_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// digit of variable i at device (physical) element index x: a shift where the variable is a bit field
__device__ __forceinline__ int jt_digit(const JtPackDesc &d, int i, uint32_t x) {
    const uint32_t ds = d.dstride[i];
    if (d.row_elems > 0 && i == d.split_var)               // the variable across the thread part's top bit: low digit of the row, high digit above
        return (int)(((x % (uint32_t)d.row_elems) / ds) % (uint32_t)d.dmod[i]) + ((int)((x / d.split_ds2) % (uint32_t)d.split_mod2) << d.split_lb);
    if (d.row_elems > 0 && d.pos[i] < d.low_bits)          // a mixed-radix digit of the row (thread part at true cardinalities)
        return ds > 0 ? (int)(((x % (uint32_t)d.row_elems) / ds) % (uint32_t)d.dmod[i]) : 0;
    if (ds == (1u << d.pos[i]) && d.dmod[i] == (1 << d.nb[i])) return (int)((x >> d.pos[i]) & ((1u << d.nb[i]) - 1u));
    return ds > 0 ? (int)((x / ds) % (uint32_t)d.dmod[i]) : 0;
}

// device index -> host index; returns false for entries that name no table entry (padding inside the thread part)
__device__ __forceinline__ bool jt_dev_to_host(const JtPackDesc &d, uint32_t x, int64_t &hidx) {
    bool valid = true;
    int64_t h = 0, back = 0;
    for (int i = 0; i < d.nvars; ++i) {
        const int digit = jt_digit(d, i, x);
        valid = valid && (digit < d.card[i]);
        h += (int64_t)digit * d.hstride[i];
        if (d.row_elems > 0 && i == d.split_var)
            back += (int64_t)(digit & ((1 << d.split_lb) - 1)) * d.dstride[i] + (int64_t)(digit >> d.split_lb) * d.split_ds2;
        else
        back += (int64_t)digit * d.dstride[i];
    }
    if (back != (int64_t)x) valid = false;              // index bits no variable owns must be clear
    hidx = h;
    return valid;
}

// MODE 0: arena[x] = stage[host index] (pack);  MODE 1: synthetic fill
// MODE 2: 1 where the index names an entry, else 0 (tables of virtual cliques)
template <typename T, typename S, int MODE>
__global__ __launch_bounds__(256) void jt_pack(JtPackDesc d, const S *__restrict__ stage, T *__restrict__ arena,
                                               uint64_t key, double scale) {
    const int64_t n = d.phys_elems;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (int64_t)gridDim.x * blockDim.x) {
        int64_t h;
        const bool valid = jt_dev_to_host(d, (uint32_t)x, h);
        double v = 0.0;
        if (valid) {
            if constexpr (MODE == 0) v = (double)stage[h];
            else if constexpr (MODE == 2) v = 1.0;
            else {
                const uint64_t bits = jt_splitmix64(key + (uint64_t)h);
                v = (0.5 + (double)(bits >> 11) * (1.0 / 9007199254740992.0)) * scale;
            }
        }
        arena[d.dev_off + x] = (T)v;
    }
}

#ifndef JT_INST_TU      // (the kernels that are not templates are defined in ONE translation unit: the engine's)
// JTP_FAKE_COMM: stand-in for a received message
__global__ __launch_bounds__(256) void jt_fill_value(double *__restrict__ dst, int64_t n, double v) {
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (int64_t)gridDim.x * blockDim.x) dst[x] = v;
}
#endif

// Clique potentials = products of factor tables, written in the cliques' device layouts: CliqueGraph.evaluate
// (junctiontree/junctiontree.py:203-226) for a LIST of cliques in ONE launch (jtp_set_potential_products).
// Bound: HBM writes (sizeof(T) per element; the factor tables are small and sit in LDS, or are gathered through L2).
// A workgroup forms JT_EVAL_ROWS consecutive stored rows of one clique.  The element at x = row * row_len + t has
// digit_i(x) = digit_i(t) + digit_i(row * row_len) for every variable i (jt_digit is additive over the two parts: a variable
// lies inside the row, above it, or - one variable at most, JtEvalTask::straddle - has a low part inside and a high part
// above), so every factor's table index is tin[f](t) + rin[f](row): the divisions happen once per thread and once per
// row.  (Round 3's jt_eval_product decoded every element: 9.2 GiB of config-3 tables in 31 ms, 0.3 TB/s.)
template <typename T>
__global__ __launch_bounds__(256) void jt_eval_batch(const JtEvalTask *__restrict__ tasks, const int32_t *__restrict__ blk_start, int ntasks,
                                                     const JtEvalVar *__restrict__ fvars, const char *__restrict__ stage, T *__restrict__ arena) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T ext_t __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *lds_tab = reinterpret_cast<double *>(smem);
    __shared__ int32_t s_rin[JT_EVAL_ROWS][JT_EVAL_MAX_F];
    __shared__ int32_t s_rdig[JT_EVAL_ROWS], s_rok[JT_EVAL_ROWS];
    // (the records are read from memory, not passed as kernel arguments: hipcc (ROCm 7.2) mis-read the 32-bit arrays of a
    //  kernel-argument struct when indexed with a run-time index - dstride[cvar] came back as dstride[0])
    int lo = 0, hi = ntasks;
    const int b = (int)blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (blk_start[mid] <= b) lo = mid;
        else hi = mid;
    }
    const JtEvalTask &tk = tasks[lo];
    const JtPackDesc &c = tk.clique;
    const int tid = (int)threadIdx.x;
    const int row0 = (b - blk_start[lo]) * JT_EVAL_ROWS;
    const int nrows = tk.n_rows - row0 < JT_EVAL_ROWS ? tk.n_rows - row0 : JT_EVAL_ROWS;
    const int L = tk.row_len, nf = tk.nf, sv = tk.straddle;
    const int scard = sv >= 0 ? c.card[sv] : 1;
    // small factor tables -> LDS, as doubles
#pragma unroll
    for (int f = 0; f < JT_EVAL_MAX_F; ++f) {
        if (f >= nf || tk.flds[f] < 0) continue;
        double *dst = lds_tab + tk.flds[f];
        const int n = tk.felems[f];
        if (tk.fis64[f]) {
            const double *src = reinterpret_cast<const double *>(stage) + tk.foff[f];
            for (int i = tid; i < n; i += 256) dst[i] = src[i];
        } else {
            const float *src = reinterpret_cast<const float *>(stage) + tk.foff[f];
            for (int i = tid; i < n; i += 256) dst[i] = (double)src[i];
        }
    }
    // place x (x = t inside the first row, or x = row * row_len) -> is it the part of a table entry, the straddling
    // variable's part of its digit, and every factor's part of its table index
    auto decode = [&](const uint32_t x, const bool high, int &sdig, int (&fidx)[JT_EVAL_MAX_F]) {
        bool ok = true;
        int64_t back = 0;
        sdig = 0;
        for (int i = 0; i < c.nvars; ++i) {
            const int d = jt_digit(c, i, x);
            if (i == sv) sdig = d;
            else ok = ok && d < c.card[i];
            if (c.row_elems > 0 && i == c.split_var)
                back += high ? (int64_t)(d >> c.split_lb) * c.split_ds2 : (int64_t)d * c.dstride[i];
            else
                back += (int64_t)d * c.dstride[i];
        }
        ok = ok && back == (int64_t)x;          // index bits no variable owns must be clear
#pragma unroll
        for (int f = 0; f < JT_EVAL_MAX_F; ++f) {
            int idx = 0;
            if (f < nf && ok) {
                const JtEvalVar *fv = fvars + tk.fv_off[f];
                for (int j = 0; j < tk.fnv[f]; ++j) {
                    const uint32_t ds = fv[j].ds;
                    const uint32_t xr = fv[j].kind ? x % (uint32_t)c.row_elems : x;
                    int digit = ds > 0 ? (int)((xr / ds) % (uint32_t)fv[j].mod) : 0;
                    if (fv[j].kind == 2) digit += (int)((x / c.split_ds2) % (uint32_t)c.split_mod2) << c.split_lb;
                    idx += digit * fv[j].stride;
                }
            }
            fidx[f] = idx;
        }
        return ok;
    };
    int tin[VEC][JT_EVAL_MAX_F], tdig[VEC];
    bool tok[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int t = tid * VEC + e;
        tok[e] = decode((uint32_t)(t < L ? t : 0), false, tdig[e], tin[e]) && t < L;
    }
    if (tid < nrows) {
        int hd, rin[JT_EVAL_MAX_F];
        const bool ok = decode((uint32_t)(row0 + tid) * (uint32_t)L, true, hd, rin);
        s_rok[tid] = ok ? 1 : 0;
        s_rdig[tid] = hd;
#pragma unroll
        for (int f = 0; f < JT_EVAL_MAX_F; ++f) s_rin[tid][f] = rin[f];
    }
    __syncthreads();
    const bool active = tid * VEC < L;
    T *row = arena + c.dev_off + (int64_t)row0 * L + tid * VEC;
    for (int r = 0; r < nrows; ++r, row += L) {
        const bool rok = s_rok[r] != 0;
        const int hd = s_rdig[r];
        bool ok[VEC];
        double v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            ok[e] = tok[e] && rok && tdig[e] + hd < scard;
            v[e] = 1.0;
        }
        if (tk.accumulate && active) {
            const ext_t old = *reinterpret_cast<const ext_t *>(row);
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = (double)old[e];
        }
#pragma unroll
        for (int f = 0; f < JT_EVAL_MAX_F; ++f) {
            if (f >= nf) continue;
            const int ri = s_rin[r][f];
            if (tk.flds[f] >= 0) {
                const double *tab = lds_tab + tk.flds[f];
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] *= tab[ok[e] ? tin[e][f] + ri : 0];
            } else if (tk.fis64[f]) {
                const double *tab = reinterpret_cast<const double *>(stage) + tk.foff[f];
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] *= tab[ok[e] ? tin[e][f] + ri : 0];
            } else {
                const float *tab = reinterpret_cast<const float *>(stage) + tk.foff[f];
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] *= (double)tab[ok[e] ? tin[e][f] + ri : 0];
            }
        }
        if (active) {
            ext_t ov;
#pragma unroll
            for (int e = 0; e < VEC; ++e) ov[e] = ok[e] ? (T)v[e] : (T)0;
            __builtin_nontemporal_store(ov, reinterpret_cast<ext_t *>(row));
        }
    }
}

// host index -> device index
__device__ __forceinline__ uint32_t jt_host_to_dev(const JtPackDesc &d, int64_t h) {
    uint32_t x = 0;
    for (int i = d.nvars - 1; i >= 0; --i) {
        const int c = d.card[i];
        const int digit = (int)(h % c);
        h /= c;
        if (d.row_elems > 0 && i == d.split_var)
            x += (uint32_t)(digit & ((1 << d.split_lb) - 1)) * d.dstride[i] + (uint32_t)(digit >> d.split_lb) * d.split_ds2;
        else
        x += (uint32_t)digit * d.dstride[i];
    }
    return x;
}

template <typename T, typename S>
__global__ __launch_bounds__(256) void jt_unpack(JtPackDesc d, const T *__restrict__ arena, S *__restrict__ stage) {
    for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < d.host_elems;
         h += (int64_t)gridDim.x * blockDim.x)
        stage[h] = (S)arena[d.dev_off + jt_host_to_dev(d, h)];
}

// message(s) -> host order: out[h] = (sum_p up[p]) * (dn ? sum_p dn[p] : 1)
template <typename S>
__global__ __launch_bounds__(256) void jt_msg_unpack(JtPackDesc d, const double *__restrict__ up, int up_npart,
                                                     const double *__restrict__ dn, int dn_npart, int64_t pstride,
                                                     S *__restrict__ stage) {
    for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < d.host_elems;
         h += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t x = jt_host_to_dev(d, h);
        double u = 0.0;
        for (int p = 0; p < up_npart; ++p) u += up[(int64_t)p * pstride + x];
        if (dn) {
            double w = 0.0;
            for (int p = 0; p < dn_npart; ++p) w += dn[(int64_t)p * pstride + x];
            u *= w;
        }
        stage[h] = (S)u;
    }
}

#ifndef JT_INST_TU
// batched marginal read-out: request blockIdx.y, entries strided over blockIdx.x
// (round 6: a factor marginal is a few dozen entries of hundreds of partial copies - one per workgroup of the pass that formed it; a
//  thread per entry added them one after the other, 94 us for config 3's 1831 requests.  A request of at most 128 entries now spreads
//  its copies over 256 / entries thread groups - group g takes copies g, g + G, ... - whose sums are added in group order: a fixed
//  order, the same bits on every call.)
__global__ __launch_bounds__(256) void jt_marg_unpack(const JtMargDesc *__restrict__ descs, const double *__restrict__ scratch_buf,
                                                      double *__restrict__ stage, const double *__restrict__ arena_cur) {
    const JtMargDesc &m = descs[blockIdx.y];
    const double *scratch = m.in_arena ? arena_cur : scratch_buf;       // (a marginal a folded task left in the message arena)
    __shared__ double part[256];
    const int64_t ne = m.d.host_elems;
    if (ne <= 128 && gridDim.x == 1) {
        int w = 1;
        while (w < ne) w <<= 1;                                   // entries rounded up to a power of two
        const int G = 256 / w, g = (int)threadIdx.x / w, h = (int)threadIdx.x % w;
        double u = 0.0;
        if (h < ne) {
            const uint32_t x = jt_host_to_dev(m.d, h);
            for (int p = g; p < m.npart; p += G) u += scratch[m.src_off + (int64_t)p * m.pstride + x];
        }
        part[threadIdx.x] = u;
        __syncthreads();
        if (g == 0 && h < ne) {
            double t = part[h];
            for (int k = 1; k < G; ++k) t += part[k * w + h];
            stage[m.dst_off + h] = t;
        }
        return;
    }
    for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < ne; h += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t x = jt_host_to_dev(m.d, h);
        double u = 0.0;
        for (int p = 0; p < m.npart; ++p) u += scratch[m.src_off + (int64_t)p * m.pstride + x];
        stage[m.dst_off + h] = u;
    }
}
#endif

// ------------------------------------------------------------------------------------------
// Explicit instantiation lists.  The message-passing kernels are compiled in translation units of their own, in parallel
// (jtp_inst_*.hip: one family and storage type each; build.py), every other translation unit sees them as `extern template`:
// one translation unit with everything took 3.5 minutes to compile.  X = `extern` or nothing.
#define JT_KARGS(T) const JtTask *, const JtBlock *, const int *, const T *, T *, double *, JtFlow
#define JT_INST_FLOW(X, T)                                                   \
    X template __global__ void jt_collect_flow<T>(JT_KARGS(T));              \
    X template __global__ void jt_distribute_flow<T>(JT_KARGS(T));           \
    X template __global__ void jt_distribute_flow_chain<T>(JT_KARGS(T));
#define JT_INST_LEVEL(X, T)                                                  \
    X template __global__ void jt_collect_level<T>(JT_KARGS(T));             \
    X template __global__ void jt_distribute_level<T>(JT_KARGS(T));          \
    X template __global__ void jt_reduce_level<T>(JT_KARGS(T));              \
    X template __global__ void jt_single<T>(JT_KARGS(T));                    \
    X template __global__ void jt_lean_single<T>(JT_KARGS(T));               \
    X template __global__ void jt_marginals<T>(JT_KARGS(T));
#define JT_INST_SHAPE(X, T)                                                  \
    X template __global__ void jt_collect<T, 0>(JT_KARGS(T));                \
    X template __global__ void jt_collect<T, 1>(JT_KARGS(T));                \
    X template __global__ void jt_collect<T, 2>(JT_KARGS(T));                \
    X template __global__ void jt_collect<T, 3>(JT_KARGS(T));                \
    X template __global__ void jt_distribute<T, 0, 0>(JT_KARGS(T));          \
    X template __global__ void jt_distribute<T, 0, 1>(JT_KARGS(T));          \
    X template __global__ void jt_distribute<T, 0, 2>(JT_KARGS(T));          \
    X template __global__ void jt_distribute<T, 0, 3>(JT_KARGS(T));          \
    X template __global__ void jt_distribute<T, 1, 0>(JT_KARGS(T));          \
    X template __global__ void jt_distribute<T, 1, 1>(JT_KARGS(T));          \
    X template __global__ void jt_distribute<T, 1, 2>(JT_KARGS(T));          \
    X template __global__ void jt_distribute<T, 1, 3>(JT_KARGS(T));
#define JT_INST_MULTI(X, T) X template __global__ void jt_multi_flow<T>(JT_KARGS(T));
#define JT_INST_BOTH(X, T) X template __global__ void jt_propagate_flow<T>(JT_KARGS(T)); X template __global__ void jt_propagate_flow_marg<T>(JT_KARGS(T));
#define JT_INST_MIX_V(X, T, V)                                                  \
    X template __global__ void jt_collect_level_mix<T, V>(JT_KARGS(T));         \
    X template __global__ void jt_distribute_level_mix<T, V>(JT_KARGS(T));      \
    X template __global__ void jt_single_mix<T, V>(JT_KARGS(T));                \
    X template __global__ void jt_collect_flow_mix<T, V>(JT_KARGS(T));          \
    X template __global__ void jt_distribute_flow_mix<T, V>(JT_KARGS(T));
#define JT_INST_MIX(X, T) JT_INST_MIX_V(X, T, false)
#define JT_INST_MIXC(X, T) JT_INST_MIX_V(X, T, true)
#ifndef JT_INST_TU
JT_INST_FLOW(extern, float)
JT_INST_FLOW(extern, double)
JT_INST_LEVEL(extern, float)
JT_INST_LEVEL(extern, double)
JT_INST_SHAPE(extern, float)
JT_INST_SHAPE(extern, double)
JT_INST_MULTI(extern, float)
JT_INST_MULTI(extern, double)
JT_INST_BOTH(extern, float)
JT_INST_BOTH(extern, double)
JT_INST_MIX(extern, float)
JT_INST_MIX(extern, double)
JT_INST_MIXC(extern, float)
JT_INST_MIXC(extern, double)
#endif
