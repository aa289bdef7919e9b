// Internal structures shared by the host planner (jtp_plan.cpp), the kernels
// (jtp_kernels.hip.h) and the C ABI (jtp_engine.hip).  See DESIGN.md "Data layout in HBM".
//
// Everything on the device is a *bit field*: a variable of cardinality k owns
// ceil(log2 k) consecutive index bits of every table it appears in (tables are zero padded
// when k is not a power of two), so "which separator entry does clique element x touch"
// is a fixed re-arrangement of the bits of x.  A workgroup (256 threads) handles one
// chunk of one clique; the clique's index bits are split into
//     [ F | A and R | T ]        T = low TB bits: thread part (VEC elements x 256 threads)
//   F bits: fixed per workgroup (chunk id)          -> no loop
//   A bits: looped, belong to at least one outgoing message -> outer loop, epilogue each
//   R bits: looped, belong to no outgoing message            -> inner loop, register sums
#pragma once
#include <stdint.h>

#define JT_MAX_IN 4            // incoming messages per task (parent + 3 children)
#define JT_MAX_OUT 3           // outgoing messages per task
#define JT_MAX_MSG (JT_MAX_IN + JT_MAX_OUT)
#define JT_NCOL 8              // iteration-table columns: 0 = element offset, 1..4 = incoming slot
                               // offsets, 5..7 = outgoing slot offsets (iteration i = a * 2^nR + r)
#define JT_MAX_HI 22           // clique bits above the thread part
#define JT_MAX_BITS 31         // max index bits of one clique table
#define JT_MAX_FREE 13         // max log2(entries) of a staged message sub-box
#define JT_THREADS 256
#define JT_REDUCE_ENTRIES 64   // entries of a message summed by one reduce workgroup: its four waves each take a quarter of the
                               // partial copies of those entries (all loads of a wave in flight together), then the four
                               // partial sums are added in wave order
#ifndef JT_RING_BYTES
#define JT_RING_BYTES 16384      // LDS bytes at offset 0: per wave a ring of 4 x 1 KiB element slots (LDS-DMA)
#endif
#define JT_STAGE_SCRATCH 2048    // bytes of LDS for staging sums
#define JT_MIN_ITER_LOG2 2       // a clique table is padded to at least 4 rows; the heuristic splits keep >= 4 loop iterations per workgroup
#ifndef JT_MIN_LOOP_LOG2
#define JT_MIN_LOOP_LOG2 1       // the searched splits (layout policy 4) may go down to 2 iterations (the other ring slots load repeats)
#endif
#define JT_MAX_ITER_LOG2 6       // a workgroup runs at most 64 loop iterations: its offset table lives in
                                 // registers, row r in lane r
#define JT_SYNC_ABORT 0        // dataflow launches, per evidence set: word 0 = abort flag, then one ticket
#define JT_SYNC_HDR 2          // counter per launch segment
// "Not written yet" marker of a message entry (a NaN no arithmetic produces; both 32-bit halves equal so
// that a 32-bit memset fills it).  The message arena has two halves used by alternate propagates; a
// producer writes its entries in this propagate's half and the marker in the other half, so at the
// start of every propagate the half about to be used holds markers only.
#define JT_UNWRITTEN 0x7FF8BEEF7FF8BEEFull
#define JT_MAX_VARS 32         // variables per node
// Physical layout of a clique table (round 2): the thread part (low TB index bits = one 4 KiB row) stays a bit
// field; ABOVE it only the rows that exist are stored - a variable that lies wholly above the thread part counts
// its true cardinality (mixed radix), padding bits store nothing.  Element offsets stay LINEAR in the logical index
// bits (bit k of a variable weighs 2^k x the variable's stride), so every host-built offset table keeps its form;
// rows whose digits do not exist (digit >= cardinality, padding bit set) read the arena's ZERO ROW (offset 0) and
// are marked in the tables with JT_NO_ROW.
#define JT_NO_ROW 0xFFFFFFFFu
#define JT_BLOCK_KEEP_ROWS 2u  // JtBlock::flags: JtTask::keep_rows
#define JT_BLOCK_LEAN 4u       // JtBlock::flags: the task has a JtLean record, at int offset first_x[6] | first_x[7] << 32 of the table buffer (a unit task
                               // loads no rows: first_x is otherwise unused; first_x[5] = JtTask::pnode) - a dataflow workgroup reaches it without
                               // reading the task record
#define JT_BLOCK_FOLD 16u      // JtBlock::flags (with JT_BLOCK_LEAN): a workgroup of a folded marginal task (JtTask::fold): up to four inputs, three outputs
#define JT_BLOCK_NULL 8u       // JtBlock::flags: no workgroup - multi-set plans pad every launch's block list to a multiple of eight records, so
                               // that the eight records a run of workgroups shares (jt_multi_flow) never span two tree levels
#define JT_BLOCK_INVALID 1u    // JtBlock::flags: the chunk's own digits do not exist - every row is the zero row (the
                               // workgroup still writes its - all zero - partial copy and padded message entries)
// Multi-set plans (JTP_MULTISET): evidence sets that share ONE copy of the clique tables are processed
// JT_MSETS at a time by every workgroup - a table row is loaded once and multiplied into the messages of
// each set of the group (kernel jt_multi_*).  Per set a workgroup owns an LDS region of JtTask::setb bytes
// (4 KiB or 16 KiB) holding the set's message sub-boxes.
#ifndef JT_MSETS
#define JT_MSETS 8
#endif
#define JT_SETB_SMALL 4096
#define JT_SETB_LARGE 16384

struct JtMsg {
    int64_t off;               // msg arena offset (doubles) of partial copy 0
    int32_t npart;             // partial copies: summed when read, one written per chunk group
    int32_t pstride;           // doubles between partial copies
    int32_t nfree;             // the workgroup's sub-box has 2^nfree entries
    int32_t lds_off;           // byte offset of the sub-box in dynamic LDS
    int32_t e_w[2];            // sub-box slot weight of clique bits 0..EB-1 (0: bit not in message)
    int32_t t_w[8];            // slot weight of the 6 lane bits then the 2 wave bits
    int32_t red_e;             // outgoing: e bits NOT in the message (summed in-thread)
    int32_t red_lane;          // outgoing: lane bits NOT in the message (summed by shuffles)
    int32_t red_wave;          // outgoing: wave bits NOT in the message (summed through LDS)
    int32_t e_dep;             // incoming: 1 if the message depends on any e bit
    int32_t same_launch;       // incoming: 1 if a dataflow launch may run its producer concurrently (then entries
                               // are read through to memory and checked for the unwritten marker)
    int32_t fixed;             // incoming: 1 = a STATIC table (the potential of a unit clique at the shape its factors cover,
                               // PNode::stat): it lives in the plan's fixed arena, outside the two alternating halves of the message
                               // arena - its address is msg_arena + JtFlow::fix_shift + off - it is never "unwritten" and has one copy
    int32_t src_task;          // incoming, multi-set plans: the collect task that forms this upward message (-1: a downward message).  Where
                               // that task is skipped for a group of evidence sets (JtFlow::skip: nothing observed below it), the
                               // consumer reads the evidence-free group's copy instead of its own group's
    int32_t pad_msg;
    int32_t f_w[JT_MAX_HI];    // weight of F bit j in the message's global index
    int32_t f_p[JT_MAX_HI];    // outgoing: weight of F bit j in the partial-copy number
    uint8_t free_pos[16];      // global-index bit of each sub-box index bit
};

struct JtTask {
    int64_t psi_off;           // element offset in the potential arena (virtual cliques: a 0/1 table)
    int64_t bel_off;           // element offset in the belief arena; < 0: belief not written
    int32_t nbits;             // index bits of the (padded) clique table, >= TB
    int32_t nF, nA, nR;
    int32_t n_in, n_out;
    int32_t lds_bytes;
    int32_t pnode;             // planner node this task belongs to
    int32_t real_bits;         // index bits actually used by variables (<= nbits; rest is padding)
    int32_t debug;             // timing experiments only (JTP_DEBUG): 1 = skip epilogues and flush
    int32_t kind;              // 0: clique pass; 1: reduce task - sum the msg[0].npart partial copies of a
                               // message (2^nbits entries, JT_REDUCE_ENTRIES per workgroup from entry xF) into
                               // msg[JT_MAX_IN].off (multi-set plans: for all JT_MSETS sets of the group)
    int32_t mode;              // clique pass: 0 = marginalise (out = sum psi * ALL incoming: collect, and in multi-set
                               // plans every downward message and marginal), 1 = distribute (belief + all-but-one)
    int32_t setb;              // multi-set plans: bytes of LDS per evidence set (JT_SETB_SMALL / JT_SETB_LARGE), else 0
    int32_t settle;            // dataflow launches: 1 = from the second staging attempt on, a thread re-loads an entry it
                               // finds unwritten itself (plans made of latency-bound levels: chains), see jt_msg_settle
    int32_t esum;              // multi-set plans: bit 0 = the element bits of the 16-byte vector are in NO message of this
                               // task (planner), bit 1 = no evidence set of ANY group observes a variable on them (engine, updated by
                               // jtp_set_evidence; per group: esum_groups).  Where both hold for a group, the four elements are summed
                               // BEFORE they meet the message product - one fused multiply-add per evidence set and row instead of four.
    uint64_t esum_groups;      // multi-set plans: bit (g mod 64) set = no evidence set of group g (nor of any other group sharing the
                               // bit) observes a variable on the element bits of this task's clique: that group sums the four
                               // elements first (round 3 kept ONE flag per task - esum bit 1 - for all groups: with 64 sets of
                               // 16 observations more than half of the tasks lost it for everybody).  Engine, jtp_set_evidence.
    int32_t unit;              // 1: a UNIT clique - no table is stored, every entry that exists counts as 1 (a clique no factor is
                               // assigned to, a virtual clique of the binarisation, or a clique whose factors cover only some of its
                               // variables: their product then enters as one more incoming message, JtMsg::fixed).  The pass loads no
                               // table row: which entries of a row exist comes from the clique's thread map (tmap_off, always set for
                               // such a task), which rows exist from the iteration table (JT_NO_ROW); a belief is stored only when
                               // bel_off >= 0 (read-out tasks: into a scratch arena).  Kernels: jt_pass<..., UNIT = true>.
    int32_t vgroups;           // mixed-radix rows, compact form: 2 = two row groups of 128 threads, the logical thread of each behind the
                               // clique's thread map (tmap_off + 2^TB: 128 ints); 0: every thread is its own logical thread
    int32_t keep_rows;         // 1: the table rows are loaded with the default cache policy instead of non-temporal - this pass
                               // and the next over the same table are close enough in time for the second to find the rows in
                               // the Infinity Cache (the top of a tree: read last by collect, first by distribute; a plan whose
                               // tables fit the cache altogether).  Copied into JtBlock::flags (JT_BLOCK_KEEP_ROWS).
    uint32_t out_run;          // byte j: log2 of the iterations over which outgoing message j's sums stay in registers
                               // (the leading loop-counter bits that message j does not contain: the nR bits no
                               // outgoing message contains, then A bits of other messages only); message j's epilogue
                               // follows every 2^run iterations
    uint32_t f_x[JT_MAX_HI];   // element-offset weight of F bit j (physical)
    uint32_t f_lx[JT_MAX_HI];  // logical index weight of F bit j (1 << bit): evidence masks are over the logical index
    uint8_t loop_pos[8];       // logical index bit of loop-counter bit t (t < nR: R bits, then A bits, those of the fewest
                               // outgoing messages first)
    uint32_t first_x[8];       // element offsets of loop iterations 0..7 (relative to the chunk base; 0 past the end;
                               // JT_NO_ROW: the row does not exist)
    int64_t itab_off;          // offset (ints) of this task's iteration table in the table buffer
    int32_t total;             // loop iterations per workgroup = 2^(nA + nR), 4 .. 64
    int32_t itab_lds;          // byte offset of the iteration table in dynamic LDS
    int64_t dbg_off;           // JTP_DEBUG & 2: msg-arena offset of the time stamps, 16 per workgroup (builds with -DJT_STAMPS)
    int64_t tmap_off;          // plans with a mixed-radix thread part (HostPlan::tmix, kernels *_mix): offset (ints) in the table
                               // buffer of the clique's thread map - 2^TB entries, entry x = element offset inside a row of
                               // logical thread index x, or -1 where x names no table entry; else -1
    int32_t fold;              // 1: a marginal task FOLDED into the propagate (jtp_tree_desc.fold_*): a unit clique's psi x every incoming table
                               // summed onto one to three requested variable sets, scheduled on the level of the clique's downward messages;
                               // runs the lean pass only (jt_unit_lean<..., NOUT>), is skipped where that cannot run (evidence on the clique)
    int32_t pad_fold;
    int64_t lean_off;          // > 0: offset (ints, a multiple of 16) in the table buffer of this task's JtLean record - a unit task of
                               // one outgoing message whose incoming tables have one copy each runs jt_unit_lean (round 6) when its
                               // evidence set observes nothing; 0 (what a zeroed record says): the generic pass
    JtMsg msg[JT_MAX_MSG];     // [0, n_in) incoming; [JT_MAX_IN, JT_MAX_IN + n_out) outgoing
};

// Round 6: what a UNIT task of one outgoing message needs, and nothing else (jt_unit_lean, jtp_kernels.hip.h).  The generic pass
// interprets JtTask - 2.3 KB of record, bit-deposit loops over free_pos[] in scalar code for every staged entry, a branch per
// message and row on e_dep: 1 200 scalar instructions per wave for a workgroup of 13 rows (profiles/r05_counters_c3.txt) - which is
// what bounds a plan made of hundreds of thousands of such workgroups (BASELINE configs[2]).  Here the host has done the deposits:
// a sub-box entry's place in its message is  w_lo . (bits of the thread id) + w_hi . (bits of the round)  with plain weights (0
// beyond nfree), the incoming tables are ORDERED - those that depend on the element bits of a thread's four elements first (`n_e`
// of them: the row loop is compiled per (n_in, n_e) and has no branch) - and the record is 64-byte aligned, read with a few
// s_load_dwordx16.  The record lives in the plan's table buffer (JtTask::lean_off).
struct JtLeanMsg {             // 32 ints
    int64_t off;               // JtMsg::off
    int32_t nfree;
    int32_t lds_off;           // bytes
    int32_t flags;             // 1: same_launch (read through to memory, wait on markers), 2: fixed (a static table)
    int32_t src;               // the message's index in JtTask::msg / JtBlock::gbase
    int32_t e_w[2];            // sub-box slot weights of the element bits
    int32_t w_lo[8];           // place in the MESSAGE of sub-box index bit b, b = 0..7 (as a weight: 1 << free_pos[b]; 0: b >= nfree)
    int32_t w_hi[8];           // ... b = 8..12; read-out tasks (jt_lean_single): [5] = partial copies of an incoming message (summed in copy
                               // order while staging; tasks of a propagate: always one), [6] = doubles between them; [7] spare
    int32_t t_w[8];            // sub-box slot weights of the six lane bits, then the two wave bits
};
struct JtLean {                // 5 * 32 + 16 = 176 ints (+ JtLeanMore)
    JtLeanMsg in[JT_MAX_IN];   // [0, n_e): depend on the element bits; [n_e, n_in): do not
    JtLeanMsg out;
    int32_t n_in, n_e;
    int32_t total;             // rows per workgroup
    int32_t rmask;             // the outgoing message's sums are folded after every row i with (i & rmask) == rmask
    int32_t red_e, red_lane, red_wave;
    int32_t settle;            // JtTask::settle
    int32_t out_pstride;       // doubles between partial copies of the outgoing message
    int32_t some_invalid;      // 1: the clique's thread map has entries that do not exist (it is read, JtLean::tmap_off)
    int64_t tmap_off;          // JtTask::tmap_off
    int64_t itab_off;          // JtTask::itab_off
    int32_t some_norow;        // 1: some row of the workgroups' loop nest does not exist (JT_NO_ROW in the iteration table)
    int32_t n_out;             // 1, or - read-out tasks (jtp_get_marginals: up to JT_MAX_OUT marginals of one unit clique per pass) - 2 or 3:
                               // the further outputs are in the JtLeanMore record right behind this one
};
struct JtLeanMore {            // 2 * 32 + 16 = 80 ints
    JtLeanMsg out[JT_MAX_OUT - 1];
    int32_t rmask[JT_MAX_OUT - 1], red_e[JT_MAX_OUT - 1], red_lane[JT_MAX_OUT - 1], red_wave[JT_MAX_OUT - 1], out_pstride[JT_MAX_OUT - 1];
    int32_t pad[6];
};

// per-launch arguments of the message-passing kernels
struct JtFlow {
    uint32_t *sync;            // dataflow launches: abort flag and segment ticket counters
    uint32_t *host_abort;      // host-visible copy of the abort flag (pinned memory)
    int64_t cur_off;           // offset (doubles) of this propagate's half of the message arena
    int64_t oth_off;           // the other half, to be marked unwritten (< 0: leave it alone)
    uint32_t ticket_idx;       // the segment's ticket counter; 0xffffffff: workgroups run in blockIdx order
    uint32_t ticket_base;      // its value when the launch starts
    uint32_t dbg;              // timing experiments (JTP_FLOW_DEBUG): 4 = no waits (wrong results), 8 = every wait
                               // times out after 20 ms (exercises the fallback to one launch per level)
    uint32_t blk_base;         // index of the launch's first workgroup in the plan's block list (time stamps)
    const uint32_t *ev;        // hard evidence of this evidence set: per planner node (mask, value) over the
                               // clique's index bits, or null: entries with (x & mask) != value count as 0
    // multi-set launches (grid.y = group of JT_MSETS evidence sets): distances between consecutive sets / groups
    int64_t set_stride;        // doubles between the message arenas of consecutive sets
    uint32_t ev_stride;        // uint32 between the evidence tables of consecutive sets
    uint32_t sync_stride;      // uint32 between the sync areas of consecutive groups
    uint32_t n_groups;         // multi-set launches: groups of evidence sets in the launch (the grid is 1-D: jt_multi_flow)
    uint32_t n_blocks;         // ... and workgroup records per group
    int64_t out_shift;         // added to the address of every outgoing entry (doubles): read-out tasks of multi-set
                               // plans read one set's message arena and write into a scratch buffer elsewhere
    // multi-set launches with active lists (below): skip[task * cap + slot] != 0 = arena slot `slot` is on the task's list.  A consumer
    // of an upward message (JtMsg::src_task = the collect task that forms it) takes the entries of every other slot from slot 0, the
    // evidence-free set (null: every set runs every task)
    const uint8_t *skip;
    uint32_t n_tasks;
    uint32_t cap;              // multi-set launches with active lists: arena slots in all (groups x JT_MSETS)
    // Active lists (round 6): WHICH evidence sets a multi-set workgroup serves is per task, not fixed.  Of the collect task of a clique
    // only the sets that observe something below it need the pass (one set in five on the width-20 tree with 16 observations per
    // set; per GROUP of eight sets, round 5's granularity, two in three): act_ids[task * cap + j], j < act_n[task], lists their
    // arena slots - slot 0, the evidence-free set, first - and the workgroups of "group" g of the launch serve entries 8 g .. 8 g + 7
    // of that list (none: they end at once).  skip[task * cap + slot] != 0 = the slot is on the task's list: a consumer takes
    // the upward message of every other slot from slot 0.  Downward tasks list every caller's slot.  Null: the sets of group g
    // are slots 8 g .. 8 g + 7 for every task.
    const uint16_t *act_ids;
    const int32_t *act_n;
    const uint8_t *esum_oct;   // [task * n_groups + g] != 0: no set among entries 8 g .. 8 g + 7 of the task's list observes a variable on
                               // the element bits of its clique (what JtTask::esum_groups says for fixed groups)
    int64_t fix_shift;         // static tables (JtMsg::fixed): offset (doubles) of the plan's fixed arena from the base of THIS
                               // propagate's half of the message arena (the consumer adds it to the message's offset)
};

// one workgroup: which task, and the chunk's decoded bases (so the kernel does no bit decode)
struct JtBlock {               // 96 bytes; everything the first element loads need is in here, so
                               // they leave after ONE dependent load (the task record follows)
    uint32_t task;             // index into the task table
    uint32_t xF;               // element offset of the chunk (F bits deposited)
    int32_t gbase[JT_MAX_MSG]; // per message: global-index base of the chunk's sub-box
    int32_t pnum[JT_MAX_OUT];  // per outgoing message: partial-copy number written by this chunk
    int64_t psi_x0;            // arena element offset of the chunk: task psi_off + xF
    uint32_t first_x[8];       // the task's first_x resolved for this chunk (JT_NO_ROW where the row does not exist)
    uint32_t lxF;              // logical index of the chunk (F bits deposited): evidence masks
    uint32_t flags;            // JT_BLOCK_INVALID
};

// host <-> device layout conversion of one table (pack / unpack / synthetic fill)
struct JtPackDesc {
    int64_t dev_off;           // element offset in the arena
    int32_t nbits;             // device index bits (padded)
    int32_t nvars;
    int64_t host_elems;        // product of host cardinalities
    uint8_t pos[JT_MAX_VARS];  // first device bit of variable i (host axis order) in the LOGICAL index
    uint8_t nb[JT_MAX_VARS];   // bits of variable i
    int32_t card[JT_MAX_VARS]; // cardinality
    int64_t hstride[JT_MAX_VARS]; // host C-order stride in elements (0: broadcast axis)
    // physical layout (tables of cliques; messages and marginals are plain bit fields: dstride = 1 << pos, dmod = 1 << nb)
    uint32_t dstride[JT_MAX_VARS]; // device element stride of variable i's digit (< 2^31; 32 bits: read with a run-time
                                  // index out of the kernel-argument segment, where 64-bit elements were mis-read)
    int32_t dmod[JT_MAX_VARS];    // digits stored along that stride (2^nb where padded, the cardinality where compact)
    int64_t phys_elems;        // elements of the table as stored (zero row not included)
    int32_t low_bits;          // index bits of the thread part (one row = 2^low_bits elements)
    int32_t row_elems;         // > 0: thread part stored at true cardinalities - the digits of variables below low_bits are taken
                               // of (x mod row_elems), x the element index (a row is row_elems long, not 2^low_bits)
    // ... with ONE variable across bit low_bits (PNode::tsplit; -1: none): its digit is low + (high << split_lb), low a
    // radix-2^split_lb digit of the row (dstride / dmod of that variable), high a digit of split_mod2 values along split_ds2
    int32_t split_var;         // index into this record's variables
    int32_t split_lb;
    uint32_t split_ds2;
    int32_t split_mod2;
};

// one request of a batched marginal read-out (jt_marg_unpack): partial copies -> host order
struct JtMargDesc {
    JtPackDesc d;              // layout of the requested variables (dev_off unused)
    int64_t src_off;           // scratch offset (doubles) of partial copy 0
    int64_t pstride;           // doubles between partial copies
    int64_t dst_off;           // offset (doubles) in the staging buffer
    int32_t npart;
    int32_t in_arena;          // 1: the partial copies were left in the MESSAGE arena by a task folded into the propagate (src_off from the
                               // base of the half it used), not in the request list's scratch buffer
};

#define JT_EVAL_MAX_F 8        // factor tables multiplied per pass over a clique (more: further passes that multiply into the table)
// clique potentials of MANY cliques in ONE launch (jt_eval_batch, jtp_set_potential_products): one record per clique (and
// pass of JT_EVAL_MAX_F factors).  A workgroup forms `rows per workgroup` consecutive stored rows of one clique; an element's
// place x = row * row_len + t splits every variable's digit into a part that depends on t only and a part that depends on
// the row only (jt_digit is additive over the two), so the index arithmetic is done once per thread and once per row, not
// once per element (jt_eval_product did it per element: 9.2 GiB of config-3 tables took 31 ms, bound by integer division).
struct JtEvalTask {
    JtPackDesc clique;
    int32_t nf;
    int32_t accumulate;        // 1: multiply into the table already in the arena (factors beyond the first JT_EVAL_MAX_F)
    int32_t row_len;           // elements of one stored row (2^TB, or JtPackDesc::row_elems)
    int32_t n_rows;            // stored rows of the table
    int32_t straddle;          // the clique variable (index into clique.*) whose digit has a part inside the row AND a part
                               // above it (a bit field across bit TB, or JtPackDesc::split_var); -1: none
    int32_t pad;
    int64_t foff[JT_EVAL_MAX_F];        // element offset of factor f's table in the staging buffer (in elements of its own type)
    int32_t fnv[JT_EVAL_MAX_F];
    int32_t fis64[JT_EVAL_MAX_F];
    int32_t flds[JT_EVAL_MAX_F];        // offset (doubles) of the table's copy in LDS, or -1: read from the staging buffer
    int32_t felems[JT_EVAL_MAX_F];
    int32_t fv_off[JT_EVAL_MAX_F];      // factor f's variables: records fv_off[f] .. fv_off[f] + fnv[f] of the launch's JtEvalVar pool
};
// one variable of one factor: where its digit sits in the clique's device index (copied from the clique's record by the host -
// the kernel indexes nothing with an index it has loaded: hipcc (ROCm 7.2) returned the digit of clique.dmod[split_var] for
// clique.dmod[cvar[j]] there) and its stride in the factor's table
struct JtEvalVar {
    uint32_t ds;               // device element stride of the digit
    int32_t mod;               // digits stored along it
    int32_t stride;            // C-order stride in the factor's table (0: broadcast axis)
    int32_t kind;              // 0: taken of x; 1: a mixed-radix digit INSIDE a row (JtPackDesc::row_elems): taken of (x mod row_elems);
                               // 2: the clique's variable across the thread part's top bit (JtPackDesc::split_*)
};
#define JT_EVAL_ROWS 32          // rows per workgroup of jt_eval_batch (128 KiB of table)
#define JT_EVAL_LDS_DOUBLES 6144 // factor tables of one clique kept in LDS (as doubles): 48 KiB

// kernel variant ids: collect with n children; distribute with (has_parent, n children)
enum {
    JT_K_COLLECT0 = 0, JT_K_COLLECT1, JT_K_COLLECT2, JT_K_COLLECT3,
    JT_K_DIST_P0C0, JT_K_DIST_P0C1, JT_K_DIST_P0C2, JT_K_DIST_P0C3,
    JT_K_DIST_P1C0, JT_K_DIST_P1C1, JT_K_DIST_P1C2, JT_K_DIST_P1C3,
    JT_K_COLLECT_LEVEL, JT_K_DISTRIBUTE_LEVEL,      // one launch per tree level
    JT_K_COLLECT_FLOW, JT_K_DISTRIBUTE_FLOW,        // one launch per phase, workgroups wait for their message entries (default)
    JT_K_REDUCE_LEVEL,                              // reduce tasks of one level (per-level launches only)
    JT_K_MULTI_COLLECT, JT_K_MULTI_DISTRIBUTE,      // multi-set plans: JT_MSETS evidence sets per pass over a table
    JT_K_SINGLE,                                    // one task list, any mix of modes and neighbour counts (read-out)
    JT_K_BOTH_FLOW,                                 // collect and distribute in ONE dataflow launch (Segment::phase 2)
    JT_K_MARGINALS,                                 // read-out: up to JT_MAX_OUT marginals of one belief table per pass over it
    JT_K_LEAN_SINGLE,                               // read-out: marginals of unit cliques by the lean pass (round 6, jt_lean_single)
    JT_K_COUNT
};
