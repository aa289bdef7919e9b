/* jtprop.h - C ABI of libjtprop.so, the MI355X (gfx950) sum-product engine that replaces
 * the message-passing hot path of jluttine/junction-tree.
 *
 * Boundary: the reference's per-call seam is
 *     SumProduct(einsum).einsum(...)                 junctiontree/sum_product.py:22-43
 * which is one host round trip per einsum and too fine for a GPU.  The device boundary is
 * therefore `compute_beliefs` granularity - whole junction tree in, all beliefs out:
 *     compute_beliefs(tree, potentials, clique_vars)  junctiontree/computation.py:37-246
 *     JunctionTree.propagate(xs)                      junctiontree/junctiontree.py:297-331
 * Each entry point below names the reference lines it stands in for.
 *
 * Conventions: plain C, no exceptions across the boundary.  Every function returning int
 * returns JTP_OK (0) or a negative JTP_E* code; jtp_last_error() then holds a message
 * (thread local).  Host arrays are C-order in the node's own variable order exactly as the
 * reference passes numpy arrays (README.md:22-41).  A plan is not thread safe; distinct
 * plans are independent.  All device work of a plan runs on the plan's own HIP stream.
 */
#ifndef JTPROP_H
#define JTPROP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JTP_OK 0
#define JTP_EINVAL (-1)     /* bad argument / malformed tree (Python: ValueError)        */
#define JTP_EHIP (-2)       /* HIP runtime error, or no GPU (Python: RuntimeError)      */
#define JTP_ECOMM (-3)      /* RCCL error                                               */
#define JTP_ENOMEM (-4)     /* device or host allocation failed                         */
#define JTP_EUNSUPPORTED (-5) /* structure outside engine limits (e.g. table > 2^31)    */

#define JTP_F32 0           /* clique tables stored as float  (messages are always f64) */
#define JTP_F64 1           /* clique tables stored as double                           */

#define JTP_PLAN_ONLY 1u    /* jtp_tree_desc.flags: plan on the host only, touch no GPU  */
#define JTP_KEEP_ROOT 4u    /* do not re-root the tree at its centre (default: re-root when n_ranks == 1;
                               beliefs do not depend on the root, the number of levels does)     */
#define JTP_SPLIT_VARIANTS 2u /* one launch per (level, neighbour count) instead of per level:
                               profiling aid, attributes device time to each clique shape   */

#define JTP_LEVEL_LAUNCHES 8u /* one launch per tree level instead of the default one per phase, whose
                               workgroups wait for the message entries they need (dataflow)   */

#define JTP_FLOW_TICKETS 16u /* dataflow launches: workgroups draw their place in the block list from an
                               atomic counter instead of relying on in-order workgroup dispatch     */

#define JTP_SHARE_POTENTIALS 32u /* the n_batch evidence sets read ONE set of clique potentials (set through
                               evidence set 0) and differ by their evidence (jtp_set_evidence): BASELINE
                               config 5 without 512 copies of the tables, and with the concurrently running
                               sets finding each other's table reads in the last-level cache             */

#define JTP_MULTISET 64u    /* with JTP_SHARE_POTENTIALS: evidence sets are processed EIGHT at a time by one pass over the
                               shared clique tables (kernel jt_multi_flow): a table row is read from HBM once per
                               group of sets instead of once per set.  No belief tables are kept: clique beliefs and
                               marginals are formed on demand from the shared tables and the set's final messages;
                               jtp_propagate always runs all evidence sets of the plan                        */

#define JTP_NO_COMPACT 128u /* store every clique table padded to powers of two in every variable (round-1 layout).  By
                               default the rows above the thread part are stored at the true cardinalities         */

typedef struct jtp_plan jtp_plan;

/* Structure of one junction tree.  Mirrors the reference's data model
 * (junctiontree/junctiontree.py:141-189, 283-291; README.md:50-77): a node list that is the
 * concatenation of maximal cliques (nodes 0..n_cliques-1) and separators (nodes
 * n_cliques..n_nodes-1), each node a list of variables, plus the tree given here in flat
 * parent form instead of the nested list (the Python layer flattens it). */
typedef struct jtp_tree_desc {
    int32_t struct_size;            /* sizeof(jtp_tree_desc), for ABI versioning             */
    int32_t n_vars;                 /* number of distinct variables                          */
    const int32_t *var_card;        /* [n_vars] cardinality of each variable (>= 1)          */
    int32_t n_cliques;              /* number of maximal cliques (>= 1)                      */
    int32_t n_nodes;                /* n_cliques + number of separators (= 2*n_cliques - 1)  */
    const int32_t *node_var_off;    /* [n_nodes+1] CSR offsets into node_var_ids             */
    const int32_t *node_var_ids;    /* variable ids of each node in its host axis order      */
    const int32_t *parent_clique;   /* [n_cliques] parent clique, -1 for the root            */
    const int32_t *parent_sep;      /* [n_cliques] node index of the separator to the parent */
    int32_t dtype;                  /* JTP_F32 or JTP_F64                                    */
    int32_t device;                 /* HIP device ordinal used by this process               */
    int32_t n_batch;                /* independent evidence sets sharing the structure (>=1) */
    int32_t n_ranks;                /* processes sharing the tree (1 = single GPU)           */
    int32_t rank;                   /* this process                                          */
    const int32_t *clique_owner;    /* [n_cliques] owning rank, or NULL (all rank 0)         */
    uint32_t flags;                 /* JTP_PLAN_ONLY | JTP_SPLIT_VARIANTS | JTP_KEEP_ROOT | JTP_LEVEL_LAUNCHES | ... */
    int32_t lds_budget;             /* bytes of LDS per workgroup the planner may use, 0=default */
    int32_t block_log2;             /* log2 of target elements per workgroup, 0 = automatic  */
    int32_t layout_policy;          /* bit order of the clique tables: 0 = chosen per clique (2 or 3 by the size of
                                       its messages), 1 = host axis order, 2 = variables of the fewest messages
                                       lowest (least message traffic), 3 = message variables in the thread part
                                       (cheapest reduction)                                   */
    /* Which variables of each clique its potential DEPENDS on (NULL: all of them).  The reference leaves every
     * variable of a clique that none of its assigned factors covers as a length-1 axis and never materialises it
     * (junctiontree/junctiontree.py:52-61, evaluate :203-226).  A clique that lists only some of its variables here
     * keeps NO full-size table on the device: its potential is stored at the covered shape and broadcast inside the
     * passes, its belief is formed on demand (jtp_get_belief / jtp_get_marginals), a clique that lists none is all
     * ones and stores nothing.  jtp_set_potential / jtp_set_potential_product(s) of such a clique take arrays whose
     * axes of the other variables have length 1.  (The engine may still materialise a clique whose covered part is
     * most of it; jtp_plan_describe says which.)  Ignored by JTP_MULTISET plans. */
    const int32_t *cover_off;       /* [n_cliques+1] CSR offsets into cover_ids, or NULL          */
    const int32_t *cover_ids;       /* covered variable ids of each clique (a subset of its own)  */
    /* The marginals the caller asks for after every propagate (round 6; 0 / NULL: none named): the list jtp_get_marginals will be
     * called with - per request a clique and the variables to keep, in that order.  JunctionTree.propagate returns factor marginals
     * only (junctiontree/junctiontree.py:264-274, 327-331), so the list is known when the plan is made.  Requests on cliques that keep
     * no table are then formed INSIDE the propagate's launch - tasks of their own on the level of the clique's downward messages,
     * filling slots the dependent levels leave idle - and a jtp_get_marginals call with exactly this list only unpacks them.  A hint:
     * any other list, an evidence set that observes something, or a plan whose launches cannot take such tasks is served as before. */
    int32_t fold_n;
    int32_t fold_pad;
    const int32_t *fold_cliques;    /* [fold_n] clique of each request                            */
    const int32_t *fold_var_off;    /* [fold_n+1] CSR offsets into fold_var_ids                   */
    const int32_t *fold_var_ids;    /* the variables each request keeps                           */
} jtp_tree_desc;

/* Counters of the last jtp_propagate (device time needs jtp_set_profiling(plan, 1)). */
typedef struct jtp_stats {
    int32_t struct_size;
    int32_t n_launches;             /* kernel launches per propagate                         */
    int32_t n_messages;             /* directed tree edges processed = 2*(n_cliques-1)       */
    int32_t n_tasks;
    double  algorithmic_bytes;      /* SURVEY.md 8d definition, this rank's share            */
    double  collect_ms;             /* device time of the collect phase (profiling on)       */
    double  distribute_ms;          /* device time of the distribute phase                   */
    double  kernel_ms[32];          /* device time per kernel variant, mean per propagate    */
    double  kernel_bytes[32];       /* algorithmic bytes processed per kernel variant        */
    int32_t kernel_launches[32];    /* launches per kernel variant                           */
    int32_t flow_fallbacks;         /* dataflow launches that timed out and were re-run per level (expected 0) */
    int32_t launch_mode;            /* of the last jtp_propagate: 0 = one launch per tree level, 1 = dataflow launches in
                                       blockIdx order (the default), 2 = dataflow launches in ticket order            */
    int32_t tickets_used;           /* propagates (counted per evidence set) since plan creation that ran in ticket order */
    int32_t flow_propagates;        /* propagates (per evidence set) since plan creation that ran as dataflow launches  */
    double  device_bytes;           /* device memory the plan allocated at creation (arenas, messages, task tables)     */
    int32_t storage_dtype;          /* JTP_F32 / JTP_F64 the clique tables are stored as: a float32 request is made with float64
                                       tables where the float32 layout cannot be planned (sub-boxes beyond the LDS of a CU) */
    int32_t foreign_seen;           /* propagates since plan creation that found ANOTHER PROCESS with a dataflow propagate in flight
                                       on the device (a shared-memory board, /dev/shm/jtprop_flight_<PCI bus id>) and therefore ran
                                       in ticket order - counted in tickets_used as well                                    */
    double  f64_flops;              /* multi-set plans: float64 operations of one propagate (multiplications and additions of the
                                       element loop, a fused multiply-add counted as two; conversions, index arithmetic, staging and
                                       epilogues not counted): the kernel jt_multi_flow is bound by them, not by HBM; else 0 */
    double  f64_insts;              /* ... and the float64 vector instructions (per lane) that stands for: a multiplication
                                       occupies the pipe as long as a fused multiply-add does                              */
    double  algorithmic_bytes_full; /* algorithmic bytes with EVERY clique counted at its full shape, read in both passes and its
                                       belief written (SURVEY.md 8d to the letter).  `algorithmic_bytes` counts a clique that keeps
                                       no table (jtp_tree_desc.cover_*) as what its potential is: the table over the covered
                                       variables, read once per pass, no belief                                            */
    double  fixed_bytes;            /* bytes of the static tables of such cliques (per evidence set, or shared)            */
    int32_t n_unit_cliques;         /* cliques of the caller's tree that keep no table on the device                       */
    int32_t n_static_tables;        /* ... of which hold factors: their product is a static table over the covered variables */
    int32_t flight_board;           /* 1: this process publishes its in-flight dataflow propagates on the device's shared-memory board
                                       (/dev/shm/jtprop_flight_<PCI bus id>) and sees other processes'; 0: the board could not be opened
                                       (another user's file, no /dev/shm): other PROCESSES on the device are then only noticed by the
                                       2 s time-out; -1: no dataflow propagate has asked for it yet                              */
    int32_t lean_refused;           /* 1: the description named covered variables (cover_*) but the plan that keeps no table for the
                                       uncovered part was refused (JTP_EUNSUPPORTED: a static table's sub-boxes beyond LDS, ...) and
                                       the tree was planned with EVERY table materialised - on lattices many times the device
                                       memory; the reason is the "lean_refused" string of jtp_plan_describe                      */
} jtp_stats;

/* ---- lifetime ------------------------------------------------------------------------- */

/* Compile a tree into a device plan: variable->bit layout of every table, level schedule
 * of collect (computation.py:47-96) and distribute (:140-224), message buffers, kernel task
 * tables, and (n_ranks > 1) the separator exchange schedule.  Replaces the per-call label
 * bookkeeping of sum_product.py:22-43 and the recursion of computation.py:227-243. */
int jtp_plan_create(const jtp_tree_desc *desc, jtp_plan **out);
void jtp_plan_destroy(jtp_plan *plan);

/* JSON description of the plan (layouts, tasks, launches, exchange schedule).  The string is
 * owned by the plan and valid until the plan is destroyed.  Used by tests and DESIGN.md. */
const char *jtp_plan_describe(jtp_plan *plan);

/* ---- data in -------------------------------------------------------------------------- */

/* Upload the potential of clique `node` (0 <= node < n_cliques) for evidence set `batch`.
 * `host` is a C-order array over the node's variables; `shape[i]` is the actual length of
 * axis i: the variable's cardinality, or 1 to broadcast along it (numpy semantics the
 * reference relies on, junctiontree.py:52-61).  `host_dtype` is JTP_F32 or JTP_F64.
 * Stands in for `np.copy(p)` of computation.py:245.  Separator potentials are never
 * uploaded: the reference overwrites them before use.
 * LIFETIME OF `host`: the call enqueues the copy on the plan's stream and does NOT wait for it.  From pageable
 * memory the HIP runtime has staged the caller's bytes when the call returns, so the array may be changed or freed at
 * once.  From PAGE-LOCKED memory (jtp_host_alloc, hipHostMalloc, hipHostRegister) the copy is truly asynchronous:
 * the caller must keep the array alive and unmodified until the next jtp_sync (or any read-out call, which
 * synchronises the evidence set's stream) of this plan.  jtp_set_potential_product copies its tables before it
 * returns and has no such rule. */
int jtp_set_potential(jtp_plan *plan, int32_t batch, int32_t node, const void *host,
                      const int64_t *shape, int32_t host_dtype);

/* One factor table handed to jtp_set_potential_product. */
typedef struct jtp_factor {
    const void *host;               /* C-order table over `var_ids`                           */
    int32_t n_vars;
    int32_t dtype;                  /* JTP_F32 or JTP_F64                                     */
    const int32_t *var_ids;         /* [n_vars] variables (a subset of the clique's)          */
    const int64_t *shape;           /* [n_vars] actual axis lengths: cardinality, or 1        */
} jtp_factor;

/* Clique potential = product of the factor tables assigned to the clique, formed ON THE DEVICE
 * in the clique's layout: CliqueGraph.evaluate for one clique (junctiontree.py:203-226, helper
 * einsum :34-80).  Only the factor tables cross PCIe; variables no factor covers are constant
 * axes (the reference leaves them length 1).  n_factors = 0 gives all ones. */
int jtp_set_potential_product(jtp_plan *plan, int32_t batch, int32_t clique, int32_t n_factors,
                              const jtp_factor *factors);

/* The same for a LIST of cliques - all of CliqueGraph.evaluate (junctiontree.py:203-226: one helper einsum per clique)
 * as ONE host-to-device copy of every factor table and ONE kernel launch over all the listed cliques.  Clique cliques[i]
 * receives the product of factors[factor_off[i] .. factor_off[i+1]); a clique may be listed with no factors (all ones),
 * no clique may be listed twice (JTP_EINVAL).
 * The tables are copied before the call returns.  A caller that knows which factor tables changed since the last call
 * lists only their cliques (junctiontree_amd/junctiontree.py does). */
int jtp_set_potential_products(jtp_plan *plan, int32_t batch, int32_t n_cliques, const int32_t *cliques,
                               const int32_t *factor_off, const jtp_factor *factors);

/* Fill every clique potential on the device with the counter-based synthetic values of
 * junctiontree_amd/synthetic.py: psi[i] = (0.5 + u(seed, node, i)) * scale[node], i the
 * C-order host index.  For benchmarks (no host transfer). */
int jtp_fill_synthetic(jtp_plan *plan, int32_t batch, uint64_t seed, const double *scale);

/* Hard evidence of one evidence set: variable var_ids[i] is observed in state states[i].  Replaces the
 * set's previous evidence (n = 0 clears it).  Equivalent to multiplying a one-hot indicator into one
 * clique that contains the variable (the equivalence tests/test_computation.py:411-459 of the
 * reference demonstrates for apply_evidence, computation.py:11-34), but nothing is rewritten: the kernels
 * skip the table entries that contradict the evidence.  Takes effect at the next jtp_propagate. */
int jtp_set_evidence(jtp_plan *plan, int32_t batch, int32_t n, const int32_t *var_ids, const int32_t *states);

/* ---- compute -------------------------------------------------------------------------- */

/* Collect then distribute for evidence sets [batch_begin, batch_end): the body of
 * compute_beliefs (computation.py:227-243).  Asynchronous; pair with jtp_sync. */
int jtp_propagate(jtp_plan *plan, int32_t batch_begin, int32_t batch_end);
int jtp_sync(jtp_plan *plan);

/* ---- data out ------------------------------------------------------------------------- */

/* Belief of `node` (clique: psi * all incoming messages, computation.py:216-224;
 * separator: up * down, computation.py:210), written to `host` in the node's host axis
 * order at full cardinalities, as `host_dtype`.  Unnormalised, sums to Z. */
int jtp_get_belief(jtp_plan *plan, int32_t batch, int32_t node, void *host, int32_t host_dtype);

/* Marginal of clique `clique`'s belief onto `out_vars` (a subset of its variables, in the
 * requested order), as doubles: CliqueGraph.marginalize for one factor
 * (junctiontree.py:264-274). */
int jtp_get_marginal(jtp_plan *plan, int32_t batch, int32_t clique, const int32_t *out_vars,
                     int32_t n_out, double *host);

/* Many marginals in one go: all of CliqueGraph.marginalize (junctiontree.py:229-274, one
 * einsum per factor) as ONE kernel launch over every requested clique, one conversion launch
 * and one device-to-host copy.  Request i asks clique `cliques[i]` for the variables
 * var_ids[var_off[i] .. var_off[i+1]) (that axis order) and receives them at host + out_off[i]
 * (doubles, C order).  The device tables of a request list are kept with the plan, so asking for
 * the same list again (every propagate of one model does) costs no planning. */
int jtp_get_marginals(jtp_plan *plan, int32_t batch, int32_t n, const int32_t *cliques,
                      const int32_t *var_off, const int32_t *var_ids, const int64_t *out_off,
                      double *host);

/* Z = sum of the root belief (the value the reference computes and drops,
 * computation.py:90-96). */
int jtp_get_z(jtp_plan *plan, int32_t batch, double *z);

/* ---- instrumentation ------------------------------------------------------------------ */

/* Device timing of the next `keep` propagates with hipEvents on the plan's stream (ring; 0
 * switches it off).  per_launch = 0: three events per propagate (start, collect/distribute
 * boundary, end) - cheap enough to stay on in a timed benchmark region; per_launch = 1: an event
 * pair around every launch (about 4 us of idle GPU per event).  jtp_get_stats reports the mean
 * per propagate. */
int jtp_set_profiling(jtp_plan *plan, int32_t keep);
int jtp_set_profiling_granularity(jtp_plan *plan, int32_t per_launch);
/* Time only every `stride`-th propagate (the first one after this call included): an event costs 2-3 us of idle GPU, which a
 * benchmark of 0.6 ms propagates sees (1.3 % with every propagate timed).  Default 1. */
int jtp_set_profiling_stride(jtp_plan *plan, int32_t stride);
/* Device time of a whole REGION of work on the plan's (first) stream: jtp_region_begin records one event, jtp_region_end a
 * second one, waits for it and returns the milliseconds between them.  A benchmark brackets its K timed propagates with the
 * pair and divides by K: no event sits between the propagates, so the figure cannot exceed the wall-clock step. */
int jtp_region_begin(jtp_plan *plan);
int jtp_region_end(jtp_plan *plan, double *ms);
int jtp_get_stats(jtp_plan *plan, jtp_stats *stats);
/* Mean device time (ms) of each of the plan's launches, in schedule order; `n` = capacity of
 * `ms`.  Returns the number of launches (or a negative error).  Needs jtp_set_profiling. */
int jtp_get_launch_ms(jtp_plan *plan, double *ms, int32_t n);
/* Diagnostic: copy `n` doubles at offset `off` of evidence set `batch`'s message arena to the host. */
int jtp_debug_read_msg(jtp_plan *plan, int32_t batch, int64_t off, int64_t n, double *host);
/* Test hooks.  knob "flow_debug": JtFlow::dbg of the following propagates (8 = every dataflow wait times
 * out after 20 ms: exercises the fall-back to one launch per level); "flow": 0 = launch per level from now on. */
int jtp_debug_set(jtp_plan *plan, const char *knob, int64_t value);
/* Name of kernel variant i as it appears in rocprofv3 traces, or NULL past the last one. */
const char *jtp_kernel_name(int32_t variant);

/* ---- multi-GPU: one process per GPU, RCCL point-to-point at subtree cuts --------------- */

/* Rank 0 creates a 128-byte RCCL unique id and hands it to the other ranks out of band. */
int jtp_comm_unique_id(void *id128);
/* Collective: every rank calls it once before creating plans with n_ranks > 1. */
int jtp_comm_init(int32_t rank, int32_t n_ranks, const void *id128, int32_t device);
int jtp_comm_destroy(void);
/* What the communicator itself reports: ncclCommCount, ncclCommUserRank, ncclCommCuDevice (-1 where the loaded library lacks
 * the entry point).  A multi-rank benchmark line quotes it, so that its reader sees RCCL saw N ranks. */
int jtp_comm_info(int32_t *n_ranks, int32_t *rank, int32_t *device);
/* Diagnostic: send `n` doubles from this rank to itself through the communicator (grouped
 * ncclSend + ncclRecv on a private stream) and verify them.  Exercises the RCCL binding on a
 * single GPU, where no second rank can exist. */
int jtp_comm_selftest(int32_t n);

/* ---- misc ----------------------------------------------------------------------------- */

int jtp_device_count(int32_t *count);
/* Free and total memory of a device (hipMemGetInfo): the Python layer budgets its plan cache with it. */
int jtp_device_memory(int32_t device, uint64_t *free_bytes, uint64_t *total_bytes);

/* Page-locked host memory for potentials and results: copies to and from it run at PCIe speed and
 * without an intermediate buffer (a pageable numpy array reads back at ~5 GB/s, a pinned one at
 * ~50).  Nothing in the reference corresponds (it has no device).  An array from here that was handed to
 * jtp_set_potential must stay alive and unmodified until jtp_sync (see the lifetime rule there). */
int jtp_host_alloc(void **ptr, size_t bytes);
int jtp_host_free(void *ptr);
const char *jtp_last_error(void);
const char *jtp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* JTPROP_H */
