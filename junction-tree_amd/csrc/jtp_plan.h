// Host-side plan: what jtp_plan_create compiles a junction tree into.  Pure C++ (no HIP).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/jtprop.h"
#include "jtp_internal.h"

struct PNode {                       // a clique of the (binarised) tree
    int real = -1;                   // clique index in the caller's node list, -1: virtual unit clique
    std::vector<int> vars;           // variable ids, device order (first = lowest bits)
    std::vector<int> pos, nb;        // first bit / number of bits of each variable
    int nbits = 0;                   // padded index bits (>= TB)
    int parent = -1;                 // parent pnode
    int psep = -1;                   // separator to the parent (index into HostPlan::ps)
    std::vector<int> children;       // child pnodes in message order
    int depth = 0;
    int layout = 0;                  // bit-order heuristic used for this clique (jtp_tree_desc.layout_policy)
    int owner = 0;                   // owning rank
    int64_t arena_off = -1;          // element offset in the potential/belief arenas (real, owned)
    // physical layout (jtp_internal.h, JT_NO_ROW): element weight of every logical index bit (0: padding bit), the
    // variables stored at their true cardinality (their bits must stay in ONE of chunk / loop bits in every task),
    // the padding bits above the variables, and the number of elements actually stored
    std::vector<int64_t> bitw;
    std::vector<uint32_t> group_mask;
    std::vector<int> group_pos, group_card;
    uint32_t pad_mask = 0;
    int64_t phys_elems = 0;
    // thread part stored at TRUE cardinalities (round 3, "tmix"): the variables of the low TB index bits as mixed-radix
    // digits - tmap[x] = element offset inside a row of logical thread index x, -1 where x names no entry - and rows of
    // trow elements (the product of their cardinalities, rounded up to the vector width) instead of 2^TB
    bool tmix = false;
    int trow = 0;
    uint32_t tpad_mask = 0;          // index bits below TB that no variable owns
    int tsplit = -1, tsplit_lb = 0;  // (tmix) the variable that straddles bit TB (index into vars; -1: none) and how many of its bits lie
                                     // below TB: those are a radix-2^lb digit of the row, the bits above a digit of ceil(card / 2^lb) values of the
                                     // rows above - taken when that wastes at most a quarter (entries with low + (high << lb) >= card are stored as zeros)
    std::vector<int32_t> tmap;
    // (tmix) compact form: at most 128 logical threads of the thread part own an entry that exists - their list (ascending, padded to
    // 128 with a logical thread that owns none), stored behind tmap in HostPlan::itab; the kernels then serve two rows per step with
    // 128 threads each (JtTask::vgroups = 2).  Empty: every thread is its own logical thread.
    std::vector<int32_t> vmap;
    int64_t tmap_off = -1;           // offset (ints) of tmap in HostPlan::itab
    int collect_task = -1, distribute_task = -1;
    std::vector<int> down_tasks;     // multi-set plans: one task per child (its downward message), child order
    std::vector<int> fold_tasks;     // marginal tasks folded into the propagate (HostPlan::folded), scheduled with the clique's distribute level
    // UNIT cliques (JtTask::unit): no table on the device.  Virtual cliques of the binarisation, cliques that hold no factor,
    // and cliques whose factors cover only `cover` of their variables (jtp_tree_desc.cover_*): the product of those factors is a
    // STATIC table over the covered variables (HostPlan::statics[stat]) that every task of the clique stages like one more
    // incoming message; the reference never materialises the other axes either (junctiontree.py:52-61)
    bool unit = false;
    std::vector<int> cover;          // covered variables (real cliques of plans that were given jtp_tree_desc.cover_*)
    int stat = -1;                   // index into HostPlan::statics, -1: all ones
};

struct PSep {                        // a separator = one message tensor per direction
    int node = -1;                   // node index in the caller's node list, -1: virtual
    std::vector<int> vars, pos, nb;  // device order / bit layout of the message tensor
    int nbits = 0;
    int child = -1, parent = -1;     // pnodes
    int up_npart = 1, dn_npart = 1;  // partial copies written by the producer
    int64_t up_off = -1, dn_off = -1;  // msg arena offsets (doubles); -1: not held by this rank
    // What consumers, exchanges and read-back use: the partial copies themselves, or - when the
    // producer writes many - their sum, formed once by a reduce task right behind the producer
    int64_t up_roff = -1, dn_roff = -1;
    int up_rnpart = 1, dn_rnpart = 1;
    int up_red_task = -1, dn_red_task = -1;
    int dn_task = -1;                // multi-set plans: the task (at the parent clique) that forms the downward message
};

struct Launch {
    int phase = 0;                   // 0 collect, 1 distribute
    int level = 0;
    int variant = 0;                 // JT_K_*
    std::vector<int> tasks;          // indices into HostPlan::tasks
    int64_t blk_off = 0;             // first entry of this launch in HostPlan::blocks
    int nblocks = 0;
    int lds_bytes = 0;
    double alg_bytes = 0;            // algorithmic bytes this launch accounts for
};

struct CommOp {
    int send = 0;                    // 1 send, 0 recv
    int psep = 0;
    int up = 0;                      // 1: upward (collect) message, 0: downward
    int peer = 0;
    int64_t off = 0;                 // msg arena offset (doubles)
    int64_t count = 0;               // doubles (all partial copies)
};

struct Step {
    int kind = 0;                    // 0 launch, 1 comm group
    int first = 0, count = 0;        // launch index, or [first, first+count) in comm
};

// Dataflow schedule: consecutive launches of one phase run as ONE launch whose workgroups take
// their place in the block list from a ticket counter and wait on message completion counters.
struct Segment {
    int phase = 0;                   // 0 collect, 1 distribute, 2 both (the distribute block list behind the collect one)
    int first_launch = 0, n_launch = 0;
    int64_t blk_off = 0;
    int nblocks = 0;
    int lds_bytes = 0;
    int ticket_idx = 0;              // word of the sync buffer
};

// static table of a unit clique: a plain bit field over the covered variables (PNode::cover) in the clique's device order,
// 2^nbits doubles at offset `off` of the plan's FIXED arena (one copy, outside the alternating halves of the message arena)
struct PStatic {
    int pnode = -1;
    std::vector<int> vars, pos, nb;
    int nbits = 0;
    int64_t off = -1;                // doubles, in the fixed arena; -1: the clique is another rank's
};

struct VirtualFill { JtPackDesc d; };   // all-ones table of a virtual clique (1 where the index names an entry, else 0)

struct HostPlan;
// decode chunk number -> workgroup record (element base, message bases, partial numbers)
JtBlock jtp_make_block(const HostPlan &hp, const JtTask &tk, uint32_t task_index, uint32_t chunk);
void jtp_make_lean(const HostPlan &hp, JtTask &tk, std::vector<int32_t> &itab, bool readout = false);

// Development knobs, read from the environment ONCE per plan (jtp_read_knobs, at the top of jtp_build_plan):
// nothing else in the product path calls getenv while planning or propagating.
struct PlanKnobs {
    int debug = 0;                   // JTP_DEBUG: 1 skip epilogues (timing), 2 in-kernel time stamps, ...
    int layout_policy = -1;          // JTP_LAYOUT_POLICY overrides jtp_tree_desc.layout_policy
    int reduce_min = -1;             // JTP_REDUCE_MIN: partial copies from which a reduce task sums them (-1: default)
    double target_blocks_c = 1024.0, target_blocks_d = 1024.0;     // JTP_TARGET_BLOCKS(_D): workgroups per tree level
    bool target_set = false;                                        // ... given in the environment (else twice that in plans of mostly unit cliques)
    int min_block_log2 = 13, max_block_log2 = 16;                   // JTP_MIN/MAX_BLOCK_LOG2
    int multi_min_block_log2 = 16;                                  // JTP_MULTI_MIN_BLOCK_LOG2: ... of multi-set plans
    int max_block_log2_d = 15;                                      // JTP_MAX_BLOCK_LOG2_D: the cap for the distribute pass
    double tiny_level_elems = 2097152.0;                            // JTP_TINY_LEVEL_ELEMS
    int force_level_launches = 0, force_flow = 0, fake_comm = 0;   // JTP_FORCE_LEVEL_LAUNCHES / JTP_FORCE_FLOW / JTP_FAKE_COMM
    unsigned flow_debug = 0;                                        // JTP_FLOW_DEBUG
    int flow_tickets = 0;                                           // JTP_FLOW_TICKETS
    int no_compact = 0;                                             // JTP_NO_COMPACT: every table padded to powers of two
    int search_all = 1;                                             // JTP_SEARCH_ALL=0: the search only where policy 2 would have been chosen (else policy 3 stays)
    int no_search = 0;                                              // JTP_NO_SEARCH: layout policy 2 where the cost-model search (policy 4) would run
    int roctx = 0;                                                  // JTP_ROCTX: roctx ranges around propagates and read-outs
    int longest_first = 1;                                          // JTP_LONGEST_FIRST: within a level, tasks with the longest workgroups go first in the block list
    int lane_low = 2;                                               // JTP_LANE_LOW: a sub-box's index bits that are lane bits come first (0 = message order, 1 = incoming sub-boxes only, 2 = all)
    int top_min_loop = 3;                                           // JTP_TOP_MIN_LOOP: log2 of the fewest rows per workgroup on levels of a few cliques (searched splits)
    int top_loop2 = 2;                                              // JTP_TOP_LOOP2: log2 of the rows per workgroup there
    double top_rows2 = 2048.0;                                      // JTP_TOP_ROWS2: ... and exactly 2^top_loop2 rows (the ring's depth) on levels of at most this many rows in all (0: off)
    double top_share = 0.12;                                        // JTP_TOP_SHARE: ... a clique holding at least this share of its level's elements
    double settle_level_elems = 8388608.0;                          // JTP_SETTLE_LEVEL_ELEMS: tasks on levels of at most this many table elements settle in place
    int no_tmix = 0;                                                // JTP_NO_TMIX: thread parts stay padded bit fields (round-2 layout)
    int no_tsplit = 0;                                              // JTP_NO_TSPLIT: no variable across bit TB in a clique with mixed-radix rows (the first form of round 3)
    double tmix_fill = 0.6;                                         // JTP_TMIX_FILL: mixed-radix rows for cliques whose bit-field thread part would be emptier than this
    int esum_always = 0;                                            // JTP_EXPERIMENT_ESUM_ALWAYS: multi-set tasks keep summing a vector's elements first whatever the evidence (timing experiment, wrong results)
    double keep_rows_mb = 128.0;                                      // JTP_KEEP_ROWS_MB: table rows of the levels nearest the root, up to this many MiB, are loaded with the default cache policy (0: all non-temporal; A/B on one box: config 4 0.6037 -> 0.5990 ms, an 8-rank share of it 198.5 -> 195.4 us)
    int no_ef_share = 0;                                            // JTP_NO_EF_SHARE: multi-set plans compute every set's upward messages (round 4), no evidence-free group to copy from
    int unit_joint_down = 0;                                        // JTP_UNIT_JOINT_DOWN: a unit clique forms its downward messages in ONE mode-1 pass (the first form of round 5) instead of a task each
    int no_vgroups = 0;                                             // JTP_NO_VGROUPS: mixed-radix rows keep one row per step (round 3)
    int keep_invalid = 0;                                           // JTP_KEEP_INVALID: chunks that do not exist stay in the block lists (rounds 2-4)
    int no_unit = 0;                                                // JTP_NO_UNIT: no unit cliques - every clique (virtual ones too) keeps a full table (rounds 1-4)
    int no_fold = 0;                                                // JTP_NO_FOLD: the marginals named at plan creation (jtp_tree_desc.fold_*) are formed by the read-out as before
    int fold = -1;                                                  // JTP_FOLD: -1 where the distribute levels leave the chip's slots idle (fold_marginals), 1 wherever possible, 0 nowhere
    int fold_slots = 1024;                                          // JTP_FOLD_SLOTS: workgroups resident on the chip (256 CUs x 4), the rule's threshold per level
    int no_lean = 0;                                                // JTP_NO_LEAN: unit tasks run the generic pass (round 5), no JtLean records
    double unit_ratio = 4.0;                                        // JTP_UNIT_RATIO: a clique becomes a unit clique when its table is at least this many times its covered part
    int marg_group = JT_MAX_OUT;                                    // JTP_MARG_GROUP: marginals of one belief table formed by one pass over it (1: a pass each, round 3)
    int marg_block_log2 = 0;                                        // JTP_MARG_BLOCK_LOG2: log2 of the elements per workgroup of a marginal pass (0: the 64 rows a workgroup can hold)
    int merge_phases = -1;                                          // JTP_MERGE_PHASES: 1 / 0 = both phases in one dataflow launch / never; -1: where messages are small
};
PlanKnobs jtp_read_knobs();

struct HostPlan {
    PlanKnobs knobs;
    // copy of the description
    int n_vars = 0, n_cliques = 0, n_nodes = 0, dtype = 0, n_ranks = 1, rank = 0, n_batch = 1;
    int device = 0;
    uint32_t flags = 0;
    int lds_budget = 0, block_log2 = 0, layout_policy = 0;
    std::vector<int> card, vbits;
    std::vector<std::vector<int>> node_vars;
    std::vector<int> parent_clique, parent_sep, owner;
    // derived
    int VEC = 4, EB = 2, TB = 10;
    int root = 0;
    std::vector<PNode> pn;
    std::vector<PSep> ps;
    std::vector<int> sep_of_node;    // caller's separator node -> psep (size n_nodes, -1 for cliques)
    std::vector<JtTask> tasks;
    std::vector<std::vector<int>> task_producers;   // per task and incoming message: the task that writes what the consumer reads (-1: another rank)
    std::vector<int> task_variant;
    std::vector<Launch> launches;
    std::vector<JtBlock> blocks;
    std::vector<int32_t> itab;           // iteration tables of all tasks (JtTask::itab_off)
    std::vector<uint32_t> block_chunk;   // chunk number of each block (description/tests)
    // Single-set plans: the chunks whose own digits do not exist (JT_BLOCK_INVALID: tables stored at true cardinalities) have no
    // rows - all they would ever write is the zeros of their partial copies, the same zeros every propagate.  They are NOT in
    // `blocks`: the engine zeroes those copies once per arena half after the arena is (re)initialised (no marker for those
    // entries afterwards: they stay 0.0, "written", for good).  [0]: tasks of mode 0, [1]: mode 1.
    std::vector<JtBlock> init_blocks[2];
    std::vector<uint32_t> init_chunk[2];
    std::vector<VirtualFill> virtual_fills;
    std::vector<PStatic> statics;        // static tables of unit cliques (PNode::stat)
    int64_t fix_doubles = 0;             // size of the fixed arena (doubles)
    std::vector<JtPackDesc> stat_pack;   // per real clique with a static table: host array of the clique (axes of uncovered variables have
                                         // length 1) <-> the static table (dev_off = PStatic::off); nvars = 0 for the others
    // marginals named at plan creation (jtp_tree_desc.fold_*): the request list as jtp_get_marginals keys it, and per request where the
    // propagate leaves it (task < 0: not folded - the clique keeps a table, or the plan's launches cannot take such tasks)
    struct FoldReq { int task = -1, j = 0, npart = 0, out_bits = 0; int64_t off = 0; };
    std::vector<int32_t> fold_key;
    std::vector<int32_t> fold_cliques, fold_var_off, fold_var_ids;
    std::vector<FoldReq> folded;
    bool lean = false;                   // the description named covered variables (jtp_tree_desc.cover_*)
    std::string lean_refused;            // ... and that plan was refused for this reason: this one materialises every table (jtp_plan_create)
    bool has_unit = false;               // some task of this rank's is a unit task
    bool unit_dominated = false;         // most clique elements of the tree belong to unit cliques (decide_units)
    std::vector<std::vector<int>> cover; // per real clique (lean plans)
    int64_t scratch_elems = 0;           // largest table of a unit clique (elements as it WOULD be stored) + the two shared rows: size of the
                                         // scratch arena beliefs of unit cliques are formed in on demand (jtp_get_belief)
    double alg_bytes_full = 0;           // algorithmic bytes with every clique counted at its full shape (SURVEY.md 8d to the letter)
    std::vector<CommOp> comm;
    std::vector<Step> steps;
    std::vector<Segment> segments;
    std::vector<Step> flow_steps;        // as `steps`, launches replaced by segments (kind 0: first = segment)
    int sync_words = 0;                  // uint32 words of the per-evidence-set sync buffer
    std::vector<JtPackDesc> pack;    // per real clique (host order)
    int64_t arena_elems = 0;         // potential arena == belief arena size (elements), zero row included
    double host_table_elems = 0;     // sum of the true sizes of this rank's clique tables (arena_elems / this = padding factor)
    bool compact = true;             // rows above the thread part stored at true cardinalities (JTP_NO_COMPACT clears it)
    bool tmix_compact = false;       // ... and every such clique in the compact form (PNode::vmap, JtTask::vgroups = 2): kernels *_mix<T, true>
    bool tmix = false;               // some clique stores its thread part at true cardinalities: every task then addresses its
                                     // elements through a per-clique map (PNode::tmap) and runs the *_mix kernels
    int64_t msg_doubles = 0;
    int64_t dbg_base = -1;           // JTP_DEBUG & 2: time-stamp region inside the message arena
    int max_lds = 0;
    bool chain_plan = false;     // most cliques sit on levels of a clique or two: a chain of hand-overs (latency bound)
    double alg_bytes = 0;
    // multi-set plans: algorithmic bytes of one pass over the tables (once per GROUP of JT_MSETS evidence sets)
    // and of one set's messages (once per set); alg_bytes = table + msg (one group of one set)
    double alg_table_bytes = 0, alg_msg_bytes = 0;
    bool multiset = false;
    double staging_bytes = 0;        // message bytes all workgroups load while staging (partial copies included)
    double table_bytes = 0;          // clique-table bytes all workgroups stream (reads + belief writes)
    int n_messages = 0;
    std::string json;
};

// Build the plan.  Returns JTP_OK or an error code with `err` set.
int jtp_build_plan(const jtp_tree_desc *desc, HostPlan &hp, std::string &err);

// One-off task: marginalise the table of pnode `p` onto each of the variable lists `out_vars` (one to JT_MAX_OUT of them:
// ONE pass over the table serves them all; message layout: out_vars[j][0] slowest, padded to power-of-two bits).  Output j is
// written as npart[j] partial copies of 2^out_bits[j] doubles at the msg-arena offset the caller puts into
// task.msg[JT_MAX_IN + j].off.  Returns JTP_OK or error.
int jtp_plan_marginal_task(const HostPlan &hp, int pnode, const std::vector<std::vector<int>> &out_vars,
                           JtTask &task, std::vector<int32_t> &itab, std::vector<int> &out_bits, std::vector<int> &npart,
                           std::vector<JtBlock> &blocks, std::string &err, bool with_neighbours = false);

// Multi-set plans keep no belief tables: the belief of clique `pnode` for one evidence set is formed on
// demand as psi * (every incoming message of that set) into the scratch belief arena (mode 1, no outputs).
int jtp_plan_belief_task(const HostPlan &hp, int pnode, JtTask &task, std::vector<int32_t> &itab,
                         std::vector<JtBlock> &blocks, std::string &err);

void jtp_plan_to_json(HostPlan &hp, bool with_tasks);
