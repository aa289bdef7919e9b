// Host planner: junction tree description -> bit layouts, kernel task tables, level
// schedule, message buffers, separator exchange schedule.  Pure C++ (no HIP calls), so it
// also runs in JTP_PLAN_ONLY mode on machines without a GPU.
//
// Replaces, for the whole tree at once, what the reference does per einsum call:
//   label -> axis-number remapping          junctiontree/sum_product.py:22-43
//   recursion order of collect / distribute junctiontree/computation.py:47-96, 140-224
// and removes `remove_message` (computation.py:99-136): every downward message is planned
// as an all-but-one product, never as a division.
#include "jtp_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>

namespace {

int ceil_log2(int k) {
    int b = 0;
    while ((1 << b) < k) ++b;
    return b;
}

int popc(uint32_t x) { return __builtin_popcount(x); }

struct MsgView {
    int psep = -1;
    bool up = true;              // which buffer of the separator
    int8_t dst[32];              // clique bit -> message bit, -1 if the bit is not in the message
    uint32_t mask = 0;           // clique bits that are in the message
    int msg_bits = 0;
};

#define FAIL(code, ...)                                   \
    do {                                                  \
        char _b[512];                                     \
        snprintf(_b, sizeof _b, __VA_ARGS__);             \
        err = _b;                                         \
        return code;                                      \
    } while (0)

// variable -> (pos, nb) lookup inside a node layout
int find_var(const std::vector<int> &vars, int v) {
    for (size_t i = 0; i < vars.size(); ++i)
        if (vars[i] == v) return (int)i;
    return -1;
}

MsgView make_view(const PNode &p, const PSep &s, int psep, bool up) {
    MsgView mv;
    mv.psep = psep;
    mv.up = up;
    mv.msg_bits = s.nbits;
    for (int i = 0; i < 32; ++i) mv.dst[i] = -1;
    for (size_t i = 0; i < s.vars.size(); ++i) {
        int j = find_var(p.vars, s.vars[i]);
        for (int t = 0; t < s.nb[i]; ++t) {
            mv.dst[p.pos[j] + t] = (int8_t)(s.pos[i] + t);
            mv.mask |= 1u << (p.pos[j] + t);
        }
    }
    return mv;
}

// the static table of a unit clique, seen from the clique: one more incoming message (JtMsg::fixed)
MsgView make_view(const PNode &p, const PStatic &s) {
    MsgView mv;
    mv.psep = -1;
    mv.up = true;
    mv.msg_bits = s.nbits;
    for (int i = 0; i < 32; ++i) mv.dst[i] = -1;
    for (size_t i = 0; i < s.vars.size(); ++i) {
        int j = find_var(p.vars, s.vars[i]);
        for (int t = 0; t < s.nb[i]; ++t) {
            mv.dst[p.pos[j] + t] = (int8_t)(s.pos[i] + t);
            mv.mask |= 1u << (p.pos[j] + t);
        }
    }
    return mv;
}

// ---- layout policy 4: a cost model of one task, searched over loop sets (and, in layouts(), over thread parts) -----
// Made for cliques whose messages are not small beside the table (config 3: 2 MiB messages, 8-64 MiB tables), where
// the greedy F/A/R split below ends at 8 iterations per workgroup under 50-70 KiB of sub-boxes and 8-16 partial
// copies.  The model prices a task as workgroups x (start-up + staging + iterations + epilogues + flush) over the
// workgroups the chip holds, floored by its bytes at streaming speed, plus the reduce tasks of its partial copies.
// The constants are from in-kernel time stamps on config 3 and 4 (profiles/r02_stage_times*.txt): they only have to
// rank candidates, not to predict microseconds.
// Loop order of the A bits (loop bits some outgoing message contains): bits of the fewest outgoing messages first, and
// among those the largest class first, so that the messages which do NOT contain the leading bits keep their sums in
// registers over the longest runs (JtTask::out_run).
void order_a_bits(std::vector<int> &Ab, const std::vector<uint32_t> &outs) {
    auto klass = [&](int b) {
        uint32_t k = 0;
        for (size_t j = 0; j < outs.size(); ++j) k |= (outs[j] >> b & 1u) << j;
        return k;
    };
    int size[1 << JT_MAX_OUT] = {0};
    for (int b : Ab) ++size[klass(b)];
    std::stable_sort(Ab.begin(), Ab.end(), [&](int a, int b) {
        const uint32_t ka = klass(a), kb = klass(b);
        if (popc(ka) != popc(kb)) return popc(ka) < popc(kb);
        if (size[ka] != size[kb]) return size[ka] > size[kb];
        if (ka != kb) return ka < kb;
        return a < b;
    });
}

struct CostK {               // constants of the model (JTP_COST_* environment overrides are experiments only)
    // (round 3, tools/c3_sweep.sh again after the kernel changes of the round: stage_fix 5 -> 8 and red_bw 3e6 -> 5e6 take
    //  config 3 from 12.2 to 11.7 ms - fewer, longer collect workgroups - with configs 2, 4, 5 and the rank share within noise;
    //  7 and 9-10 lose it again: the constants choose among a handful of discrete layouts)
    double wg = 1.5, stage_fix = 8.0, stage_bw = 4096.0, iter_c = 0.45, iter_d = 0.55, epi = 0.5, wave = 1.5, lane = 0.1,
           flush_fix = 1.0, flush_bw = 16384.0, bw = 5.0e6, red_fix = 4.0, red_bw = 5.0e6, overlap = 0.5, max_cu = 4, lds_cap = 150.0 * 1024;
    CostK() {
        auto g = [](const char *n, double &v) { if (const char *e = getenv(n)) v = atof(e); };
        g("JTP_COST_WG", wg), g("JTP_COST_STAGE_FIX", stage_fix), g("JTP_COST_STAGE_BW", stage_bw), g("JTP_COST_ITER_C", iter_c);
        g("JTP_COST_ITER_D", iter_d), g("JTP_COST_EPI", epi), g("JTP_COST_WAVE", wave), g("JTP_COST_LANE", lane);
        g("JTP_COST_FLUSH_FIX", flush_fix), g("JTP_COST_FLUSH_BW", flush_bw), g("JTP_COST_BW", bw), g("JTP_COST_RED_FIX", red_fix);
        g("JTP_COST_RED_BW", red_bw), g("JTP_COST_OVERLAP", overlap), g("JTP_COST_MAX_CU", max_cu), g("JTP_COST_LDS_CAP", lds_cap);
    }
};
const CostK &cost_k() {
    static const CostK k;
    return k;
}

struct CostEnv {
    int TB = 10, EB = 2, nbits = 0;
    bool dist = false;           // distribute pass: the table is written as well as read
    bool unit = false;           // unit clique: no table rows are loaded or stored, no element ring in LDS
    int max_iter_log2 = JT_MAX_ITER_LOG2;
    double share = 1.0;          // part of the chip this clique can count on (its share of the level's elements)
    double fill = 1.0;           // rows that exist / rows of the index space (variables stored at their true cardinality)
    long lds_cap = 150 * 1024;
    int red_log2 = 6;            // partial copies from 2^red_log2 on are summed by a reduce task, fewer by the consumers
    bool chain = false;          // latency-bound plan: a consumer that waits for several producers pays a staging attempt each
    int min_loop_log2 = JT_MIN_ITER_LOG2;   // chains: JT_MIN_LOOP_LOG2 (two-iteration workgroups: config 2 6.46 -> 5.75 ms; at the top
                                            // of a tree they cost 3 %: an 8-rank share of config 4 205 -> 212 us)
    std::vector<uint32_t> units; // atomic groups of bits above the thread part (a compact variable stays together)
};

double task_cost_us(const CostEnv &e, const std::vector<uint32_t> &ins, const std::vector<uint32_t> &outs, uint32_t L, long *lds_out = nullptr) {
    const CostK &K = cost_k();
    const uint32_t tmask = (1u << e.TB) - 1u;
    const uint32_t himask = (e.nbits >= 32 ? ~0u : ((1u << e.nbits) - 1u)) & ~tmask;
    const uint32_t F = himask & ~L, cover = tmask | L;
    const int nL = popc(L);
    if (nL < e.min_loop_log2 || nL > e.max_iter_log2 || popc(F) > JT_MAX_HI) return 1e30;
    const double nwg = std::max(1.0, std::ldexp(e.fill, popc(F))), iters = std::ldexp(1.0, nL);
    long lds = e.unit ? 0 : JT_RING_BYTES;
    double stage = 0, flush = 0, red_bytes = 0, epi = 0;
    int n_red = 0;
    for (uint32_t m : ins) {
        const int fb = popc(m & cover);
        if (fb > JT_MAX_FREE) return 1e30;
        lds += (8L << fb) + JT_STAGE_SCRATCH;
        stage += 8.0 * (double)(1L << fb);
    }
    uint32_t allout = 0;
    for (uint32_t o : outs) allout |= o;
    std::vector<int> Ab;
    for (int b = e.TB; b < e.nbits; ++b)
        if ((L & allout) >> b & 1) Ab.push_back(b);
    if (outs.size() > 1) order_a_bits(Ab, outs);
    const int nR = popc(L & ~allout);
    const uint32_t wave_bits = 3u << (e.TB - 2), lane_bits = 63u << e.EB;
    for (uint32_t o : outs) {
        const int fb = popc(o & cover);
        if (fb > JT_MAX_FREE) return 1e30;
        lds += 8L << fb;
        flush += 8.0 * (double)(1L << fb);
        const int np = popc(F & ~o);
        if (np > 6) return 1e30;
        if (np >= e.red_log2) {               // many partial copies: a reduce task sums them once
            red_bytes += (std::ldexp(1.0, np) + 1.0) * 8.0 * std::ldexp(1.0, popc(o));
            ++n_red;
        } else if (np) {                      // fewer: the consumers sum the copies while they stage
            red_bytes += std::ldexp(1.0, np) * 8.0 * std::ldexp(1.0, popc(o));
            if (e.chain) ++n_red;             // (on a chain that wait is on the critical path: priced like the reduce hop)
        }
        // an epilogue folds the register sums into the sub-box: butterflies over summed lane bits, one ordered
        // phase (barriers) per summed wave bit; it follows every run of iterations whose loop bits the message lacks
        int run = nR;
        for (int b : Ab) {
            if (o >> b & 1) break;
            ++run;
        }
        epi += std::ldexp(K.epi * (1.0 + K.wave * popc(~o & wave_bits)) + K.lane * popc(~o & lane_bits), nL - run);
    }
    if (lds > e.lds_cap || lds > (long)K.lds_cap) return 1e30;
    if (lds_out) *lds_out = lds;
    const double epilogues = 1.0;          // (epi already holds every message's epilogues of the whole loop)
    // (start-up, staging and flush are latency chains: record -> addresses -> message loads -> LDS -> barrier cost
    //  6-7 us even for a few KiB)
    const double t_wg = K.wg + (ins.empty() ? 0.0 : K.stage_fix) + stage / K.stage_bw + iters * (e.dist ? K.iter_d : K.iter_c) + epilogues * epi +
                        (outs.empty() ? 0.0 : K.flush_fix) + flush / K.flush_bw;
    // (workgroups a CU holds: LDS, and the kernels' registers - four waves per SIMD)
    const int per_cu = (int)std::min((long)K.max_cu, std::max(1L, 160L * 1024 / (lds + 128)));
    const double conc = std::max(1.0, 256.0 * per_cu * e.share);
    const double t_lat = std::max(t_wg, nwg * t_wg / conc);
    const double bytes = nwg * (e.unit ? 0.0 : iters * 4096.0 * (e.dist ? 2.0 : 1.0)) + nwg * (0.5 * stage + flush);
    // (neither bound hides the other completely: a workgroup's start-up and epilogues issue no loads)
    const double t_bw = bytes / (K.bw * e.share);
    double t = std::max(t_lat, t_bw) + K.overlap * std::min(t_lat, t_bw);
    if (n_red) t += K.red_fix;
    t += red_bytes / (K.red_bw * e.share);
    return t;
}

struct LoopChoice {
    uint32_t L = 0;
    double us = 1e30;
    long lds = 0;
};

// best loop set of one task: every subset of the units with 2..max_iter_log2 bits (`exhaustive`), or units added
// one at a time, cheapest first
LoopChoice search_loops(const CostEnv &e, const std::vector<uint32_t> &ins, const std::vector<uint32_t> &outs, bool exhaustive) {
    LoopChoice best;
    const int n = (int)e.units.size();
    auto consider = [&](uint32_t L) {
        long lds = 0;
        const double us = task_cost_us(e, ins, outs, L, &lds);
        if (us < best.us) best.L = L, best.us = us, best.lds = lds;
        return us;
    };
    if (exhaustive) {
        // depth first over the units, low bits first, pruned by the iteration cap
        std::vector<std::pair<int, uint32_t>> stack;      // (next unit, L)
        stack.push_back({0, 0u});
        while (!stack.empty()) {
            auto [i, L] = stack.back();
            stack.pop_back();
            if (i == n) {
                consider(L);
                continue;
            }
            stack.push_back({i + 1, L});
            if (popc(L | e.units[i]) <= e.max_iter_log2) stack.push_back({i + 1, L | e.units[i]});
        }
    } else {
        uint32_t L = 0;
        for (;;) {
            int pick = -1;
            double pick_us = 1e31;
            for (int i = 0; i < n; ++i) {
                if ((L & e.units[i]) || popc(L | e.units[i]) > e.max_iter_log2) continue;
                const double us = popc(L | e.units[i]) < e.min_loop_log2 ? 1e30 : consider(L | e.units[i]);
                // (below four iterations nothing can be priced: take the unit the fewest messages contain)
                double key = us;
                if (us >= 1e30) {
                    int cnt = 0;
                    for (uint32_t m : ins) cnt += (m & e.units[i]) != 0;
                    for (uint32_t o : outs) cnt += 2 * ((o & e.units[i]) != 0);
                    key = 1e30 + cnt;
                }
                if (key < pick_us) pick_us = key, pick = i;
            }
            if (pick < 0) break;
            L |= e.units[pick];
        }
    }
    return best;
}

// Choose the F / A / R split of the high bits and fill every index table of the task.
// Does the high part (bits >= TB) of logical index `x`, restricted to the bits in `within`, name rows that exist?
// A compact variable whose bits all lie in `within` must have a digit below its cardinality; a padding bit in
// `within` must be clear.  (Variables only partly in `within` cannot occur: their bits stay together.)
bool high_digits_exist(const PNode &p, uint32_t x, uint32_t within) {
    if (x & within & p.pad_mask) return false;
    for (size_t g = 0; g < p.group_mask.size(); ++g) {
        if ((p.group_mask[g] & within) != p.group_mask[g]) continue;
        if ((int)((x & p.group_mask[g]) >> p.group_pos[g]) >= p.group_card[g]) return false;
    }
    return true;
}

int plan_loops(const HostPlan &hp, const PNode &p, JtTask &tk, std::vector<int32_t> &itab, int nbits, int real_bits,
               const std::vector<MsgView> &ins, const std::vector<MsgView> &outs, int block_log2, std::string &err,
               int strict_budget = 0, double share = 1.0) {
    const int TB = hp.TB;
    // the bits of a compact variable (stored at its true cardinality) go to the chunk bits or stay loop bits TOGETHER
    auto unit = [&](int b) {
        for (uint32_t g : p.group_mask)
            if (g >> b & 1) return g;
        return 1u << b;
    };
    // strict_budget (multi-set plans): the sub-boxes of ONE evidence set must fit in that many bytes, whatever
    // it costs in loop iterations (down to 4) - the kernel reserves exactly that much LDS per set
    const int budget = strict_budget > 0 ? strict_budget : (hp.lds_budget > 0 ? hp.lds_budget : 32 * 1024);
    // at most 8 partial copies per outgoing message; small levels (few cliques) may use up to 64
    // so that a lone clique still spreads over >= 128 workgroups
    const int PMAX_LOG2 = block_log2 <= 13 ? 6 : 3;
    const uint32_t himask = nbits >= 32 ? 0 : (((1u << nbits) - 1) & ~((1u << TB) - 1));
    uint32_t allout = 0, everyout = himask;
    for (auto &o : outs) {
        allout |= o.mask;
        everyout &= o.mask;
    }
    if (outs.empty()) everyout = 0;

    auto lds_of = [&](uint32_t F) {
        long total = 0;
        for (auto &m : ins) total += 8L << popc(m.mask & ~F);
        for (auto &m : outs) total += 8L << popc(m.mask & ~F);
        return total;
    };
    auto max_free = [&](uint32_t F) {
        int mx = 0;
        for (auto &m : ins) mx = std::max(mx, popc(m.mask & ~F));
        for (auto &m : outs) mx = std::max(mx, popc(m.mask & ~F));
        return mx;
    };
    auto part_log2 = [&](uint32_t F) {       // worst partial count over the outputs
        int mx = 0;
        for (auto &o : outs) mx = std::max(mx, popc(F & ~o.mask));
        return mx;
    };

    uint32_t F = 0;
    bool searched = false;
    if (p.layout == 4 && strict_budget == 0) {
        // searched split (cost model above): every loop set the iteration cap allows
        CostEnv e;
        e.TB = TB, e.EB = hp.EB, e.nbits = nbits, e.dist = tk.mode == 1, e.share = share, e.unit = tk.unit != 0;
        e.red_log2 = hp.knobs.reduce_min >= 0 ? std::max(0, ceil_log2(std::max(hp.knobs.reduce_min, 1))) : (hp.chain_plan ? 3 : 6);
        e.chain = hp.chain_plan;
        e.min_loop_log2 = hp.chain_plan ? JT_MIN_LOOP_LOG2 : JT_MIN_ITER_LOG2;
        e.max_iter_log2 = std::min(std::max(block_log2 - TB, JT_MIN_ITER_LOG2), JT_MAX_ITER_LOG2);
        if (hp.knobs.top_min_loop > 0 && share >= hp.knobs.top_share && !hp.chain_plan) {
            // levels of at most eight cliques (a clique holding >= 12 % of its level): workgroups of at least 8 rows, not 4,
            // so that the workgroups of two or three such levels are resident at once with their rows in flight - as on a
            // chain - instead of one level filling every slot of the chip and the next paying its start-up behind it.
            // A/B on one box, three times: rank share of config 4 at 8 ranks 196 -> 192.5 us, config 4 on one GPU +-0
            // (16 rows: 0.601 -> 0.631 ms, 216 us; 32 rows: 0.665 ms)
            e.min_loop_log2 = std::max(e.min_loop_log2, std::min(hp.knobs.top_min_loop, nbits - TB));
            e.max_iter_log2 = std::max(e.max_iter_log2, e.min_loop_log2);
            // ... and levels of at most 2048 rows in all (one or two cliques of config 4): exactly FOUR rows, the depth of the
            // element ring - every row of such a workgroup is in flight while it waits for its messages, where rows 5-8 of
            // an eight-row workgroup are only asked for once the loop runs (2.2-3.0 us of every hand-over at the top of a tree,
            // profiles/r03_stage_times_rank0_of_8.txt "more steps"), and one or two such levels still leave room for the next
            // (at most 512 workgroups).  A/B on one box: a rank's share of config 4 at 8 ranks 198.7 -> 189.3 us, config 4
            // 0.5981 -> 0.5948 ms; two rows: 210 us.  Larger levels of few cliques (config 3: 64 MiB tables) keep their long workgroups:
            // held to four rows they took 19.7 instead of 11.7 ms.
            const double lvl_rows = (double)p.phys_elems / std::max(share, 1e-9) / (double)(1 << TB);
            if (hp.knobs.top_rows2 > 0 && lvl_rows <= hp.knobs.top_rows2 && nbits - TB >= hp.knobs.top_loop2) {
                e.min_loop_log2 = std::max(hp.knobs.top_loop2, JT_MIN_LOOP_LOG2);
                e.max_iter_log2 = std::max(hp.knobs.top_loop2, JT_MIN_LOOP_LOG2);
            }
        }
        if (hp.lds_budget > 0) e.lds_cap = (tk.unit ? 0 : JT_RING_BYTES) + hp.lds_budget + JT_STAGE_SCRATCH * (long)ins.size();
        uint32_t seen = 0;
        for (int b = TB; b < nbits; ++b)
            if (!(seen >> b & 1)) e.units.push_back(unit(b)), seen |= unit(b);
        for (size_t g = 0; g < p.group_mask.size(); ++g) e.fill *= (double)p.group_card[g] / (double)(1 << popc(p.group_mask[g]));
        e.fill = std::ldexp(e.fill, -popc(p.pad_mask & himask));
        std::vector<uint32_t> im, om;
        for (auto &m : ins) im.push_back(m.mask);
        for (auto &o : outs) om.push_back(o.mask);
        LoopChoice ch = search_loops(e, im, om, true);
        if (getenv("JTP_PLAN_DEBUG") && nbits >= 24) {
            fprintf(stderr, "pnode %d mode %d nbits %d cap %d share %.3f best L %x us %.1f lds %ld\n", tk.pnode, tk.mode, nbits, e.max_iter_log2, e.share, ch.L, ch.us, ch.lds);
            for (int k = 2; k <= 6; ++k) { CostEnv e2 = e; e2.max_iter_log2 = k; LoopChoice c2 = search_loops(e2, im, om, true); fprintf(stderr, "   cap %d: L %x us %.1f lds %ld\n", k, c2.L, c2.us, c2.lds); }
        }
        if (ch.us < 1e30) F = himask & ~ch.L, searched = true;
    }
    // 1. LDS must fit: fix the high bit that shrinks the staged sub-boxes most.
    bool down_to_four = strict_budget > 0;
    while (!searched && (lds_of(F) > budget || max_free(F) > JT_MAX_FREE)) {
        int best = -1;
        long best_lds = 0;
        int best_part = 0;
        for (int b = TB; b < nbits; ++b) {      // (fitting LDS never goes below 8 iterations: staging a big
                                                //  sub-box for 4 would cost more than it saves; multi-set plans: 4 - and 4
                                                //  for anybody whose sub-boxes do not fit at all otherwise, see below)
            if (F >> b & 1) continue;
            if (nbits - popc(F | unit(b)) < TB + (down_to_four ? 2 : 3)) continue;
            long l = lds_of(F | unit(b));
            int pl = part_log2(F | unit(b));
            if (best < 0 || l < best_lds || (l == best_lds && pl < best_part)) {
                best = b, best_lds = l, best_part = pl;
            }
        }
        if (best < 0 || best_lds >= lds_of(F)) {
            if (strict_budget > 0) FAIL(JTP_EUNSUPPORTED, "message sub-boxes of one evidence set need %ld bytes of LDS (limit %d)", lds_of(F), strict_budget);
            // cannot shrink further: accepted while the whole workgroup - ring, sub-boxes, staging scratch and the kernels'
            // static words - stays inside the CU's 160 KiB (150 KiB of sub-boxes alone, the bound of rounds 1-2, did not:
            // hipFuncSetAttribute refused 170 KiB on a random factor graph, tools/gpu_fuzz_api.py)
            // (4 KiB for the static words: the reduce path's 2 KiB of partial sums and the dataflow control words are in the same kernels)
            if (lds_of(F) + (tk.unit ? 0 : JT_RING_BYTES) + JT_STAGE_SCRATCH * (long)ins.size() + 4096 <= 160 * 1024 && max_free(F) <= JT_MAX_FREE) break;
            // (a marginal onto nearly all variables of a clique of few rows - a factor as wide as its clique, 3^8 entries:
            //  four rows per workgroup before giving up)
            if (!down_to_four) {
                down_to_four = true;
                continue;
            }
            FAIL(JTP_EUNSUPPORTED, "message sub-boxes do not fit in LDS (%ld bytes)", lds_of(F));
        }
        F |= unit(best);
    }
    // 2. Parallelism: split until a workgroup handles at most 2^block_log2 elements, preferring
    //    bits that every outgoing message contains (no partial copies), highest bit first.
    block_log2 = std::max(block_log2, TB + JT_MIN_ITER_LOG2);   // a workgroup always runs >= 4 iterations
    block_log2 = std::min(block_log2, TB + JT_MAX_ITER_LOG2);   // and at most 2^JT_MAX_ITER_LOG2
    while (!searched && nbits - popc(F) > block_log2) {
        int best = -1;
        auto fits = [&](int b) { return nbits - popc(F | unit(b)) >= TB + JT_MIN_ITER_LOG2; };      // >= 4 iterations stay
        for (int b = nbits - 1; b >= TB; --b)
            if (!(F >> b & 1) && (everyout & unit(b)) == unit(b) && fits(b)) {
                best = b;
                break;
            }
        if (best < 0) {
            // otherwise: a bit of SOME outgoing message first (left in the loops it would be an A
            // bit, i.e. an epilogue per iteration, whereas bits of no outgoing message make the free
            // register-summed R loop), then fewest partial copies, then the bit most incoming
            // messages contain (smaller staged sub-boxes), then the highest
            int best_pl = 1 << 30, best_in = -1, best_cls = 9;
            for (int b = nbits - 1; b >= TB; --b) {
                if ((F >> b & 1) || !fits(b)) continue;
                int pl = part_log2(F | unit(b));
                int cls = (allout >> b & 1) ? 0 : 1;
                int nin = 0;
                for (auto &m : ins) nin += (m.mask >> b) & 1;
                if (cls < best_cls || (cls == best_cls && (pl < best_pl || (pl == best_pl && nin > best_in))))
                    best_cls = cls, best_pl = pl, best_in = nin, best = b;
            }
            if (best < 0) break;
            if (best_pl > PMAX_LOG2 && nbits - popc(F) <= TB + JT_MAX_ITER_LOG2) break;
        }
        F |= unit(best);
    }
    if (popc(F) > JT_MAX_HI) FAIL(JTP_EUNSUPPORTED, "too many chunk bits (%d)", popc(F));
    if (nbits - popc(F) > TB + JT_MAX_ITER_LOG2)
        FAIL(JTP_EUNSUPPORTED, "cannot split a table of %d index bits into workgroups of at most 64 rows without splitting a "
                               "variable stored at its true cardinality", nbits);

    std::vector<int> Fb, Ab, Rb;
    for (int b = TB; b < nbits; ++b) {
        if (F >> b & 1) Fb.push_back(b);
        else if (allout >> b & 1) Ab.push_back(b);
        else Rb.push_back(b);
    }
    {
        std::vector<uint32_t> om;
        for (auto &o : outs) om.push_back(o.mask);
        order_a_bits(Ab, om);
    }
    tk.nbits = nbits;
    tk.tmap_off = (hp.tmix || tk.unit) ? p.tmap_off : -1;      // (unit tasks: which entries of a row exist)
    tk.vgroups = (hp.tmix && !tk.unit && !p.vmap.empty()) ? 2 : 0;
    if (hp.tmix_compact && !tk.unit && tk.vgroups != 2) FAIL(JTP_EUNSUPPORTED, "internal: a task of clique %d without its list of logical threads in a compact plan", p.real);
    tk.real_bits = real_bits;
    tk.debug = hp.knobs.debug;
    tk.nF = (int)Fb.size();
    tk.nA = (int)Ab.size();
    tk.nR = (int)Rb.size();
    tk.n_in = (int)ins.size();
    tk.n_out = (int)outs.size();
    if (tk.nA > JT_MAX_HI || tk.nR > JT_MAX_HI) FAIL(JTP_EUNSUPPORTED, "too many loop bits");
    for (int j = 0; j < tk.nF; ++j) {
        tk.f_x[j] = (uint32_t)p.bitw[Fb[j]];
        tk.f_lx[j] = 1u << Fb[j];
    }
    for (int t = 0; t < tk.nR; ++t) tk.loop_pos[t] = (uint8_t)Rb[t];
    for (int t = 0; t < tk.nA; ++t) tk.loop_pos[tk.nR + t] = (uint8_t)Ab[t];
    tk.out_run = 0;
    for (size_t j = 0; j < outs.size(); ++j) {
        int run = tk.nR;
        for (int b : Ab) {
            if (outs[j].mask >> b & 1) break;
            ++run;
        }
        tk.out_run |= (uint32_t)run << (8 * j);
    }
    uint32_t loopmask = 0;
    for (int b : Rb) loopmask |= 1u << b;
    for (int b : Ab) loopmask |= 1u << b;

    int lds = tk.unit ? 0 : JT_RING_BYTES;      // the element ring sits at LDS offset 0 (unit tasks load no rows: no ring)
    // per-message tables
    std::vector<std::vector<int>> slotw;      // [msg][clique bit] -> sub-box slot weight
    auto fill_msg = [&](JtMsg &jm, const MsgView &mv, bool is_out) {
        std::vector<int> sw(32, 0);
        // free message bits = images of clique bits outside F, ascending message bit
        std::vector<std::pair<int, int>> fr;   // (message bit, clique bit)
        for (int b = 0; b < nbits; ++b)
            if ((mv.mask >> b & 1) && !(F >> b & 1)) fr.push_back({mv.dst[b], b});
        std::sort(fr.begin(), fr.end());
        // Sub-box slot order: the index bits that are LANE bits of the clique's thread part come first, the others follow in
        // message order.  The lanes of a half-wave then read one contiguous run of the sub-box (ds_read_b64: 32 lanes x 8 bytes
        // over 64 banks - conflict-free inside 256 bytes), whatever place those variables have in the message; in message order
        // a lane bit of weight >= 32 entries put two lanes on one bank (SQ_LDS_BANK_CONFLICT: a third of all LDS cycles of
        // jt_multi_flow).  64 evidence sets 6.04 -> 5.32 ms, 8 sets 1.02 -> 0.89, config 3 11.78 -> 11.56, configs 2 and 4
        // unchanged (A/B on one box).  Staging and flush follow free_pos[] as before: their global accesses are less contiguous
        // now, which the loop's gain outweighs (slot orders that kept the lowest message bits low measured slower: 5.58 ms).
        // Mixed-radix thread parts have no lane bits: message order.  JTP_LANE_LOW=0: message order, 1: incoming sub-boxes only.
        if (hp.knobs.lane_low > (is_out ? 1 : 0) && !hp.tmix)
            std::stable_partition(fr.begin(), fr.end(), [&](const std::pair<int, int> &x) { return x.second >= hp.EB && x.second < hp.EB + 5; });
        jm.nfree = (int)fr.size();
        jm.src_task = -1;
        for (size_t r = 0; r < fr.size(); ++r) {
            jm.free_pos[r] = (uint8_t)fr[r].first;
            sw[fr[r].second] = 1 << r;
        }
        for (int e = 0; e < 2; ++e) jm.e_w[e] = (e < hp.EB) ? sw[e] : 0;
        for (int t = 0; t < 8; ++t) jm.t_w[t] = sw[hp.EB + t];
        jm.e_dep = 0;
        for (int e = 0; e < hp.EB; ++e) jm.e_dep |= (mv.mask >> e & 1);
        jm.red_e = jm.red_lane = jm.red_wave = 0;
        if (is_out) {
            for (int e = 0; e < hp.EB; ++e) if (!(mv.mask >> e & 1)) jm.red_e |= 1 << e;
            for (int t = 0; t < 6; ++t) if (!(mv.mask >> (hp.EB + t) & 1)) jm.red_lane |= 1 << t;
            for (int t = 0; t < 2; ++t) if (!(mv.mask >> (hp.EB + 6 + t) & 1)) jm.red_wave |= 1 << t;
        }
        int pbit = 0;
        for (int j = 0; j < tk.nF; ++j) {
            int b = Fb[j];
            jm.f_w[j] = (mv.mask >> b & 1) ? (1 << mv.dst[b]) : 0;
            jm.f_p[j] = 0;
            if (is_out && !(mv.mask >> b & 1)) jm.f_p[j] = 1 << pbit++;
        }
        jm.npart = is_out ? (1 << pbit) : 1;      // incoming npart is patched in later
        jm.pstride = 1 << mv.msg_bits;
        jm.lds_off = lds;
        lds += 8 << jm.nfree;
        lds = (lds + 15) & ~15;
        slotw.push_back(sw);
    };
    for (int k = 0; k < tk.n_in; ++k) fill_msg(tk.msg[k], ins[k], false);
    for (int k = tk.n_in; k < JT_MAX_IN; ++k) slotw.push_back(std::vector<int>(32, 0));
    for (int k = 0; k < tk.n_out; ++k) fill_msg(tk.msg[JT_MAX_IN + k], outs[k], true);
    for (int k = tk.n_out; k < JT_MAX_OUT; ++k) slotw.push_back(std::vector<int>(32, 0));

    // iteration table: row i = (a, r), r the fast counter; column 0 = element offset, 1..4 = slot
    // offsets into the incoming sub-boxes, 5..7 = into the outgoing sub-boxes (A bits only)
    tk.total = 1 << (tk.nA + tk.nR);
    itab.assign((size_t)tk.total * JT_NCOL, 0);
    for (int i = 0; i < tk.total; ++i) {
        const int r = i & ((1 << tk.nR) - 1), a = i >> tk.nR;
        int64_t row[JT_NCOL] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t lx = 0;                                    // logical index of the row's loop bits
        for (int t = 0; t < tk.nR; ++t)
            if (r >> t & 1) {
                row[0] += p.bitw[Rb[t]];
                lx |= 1u << Rb[t];
                for (int c = 1; c < JT_NCOL; ++c) row[c] += slotw[c - 1][Rb[t]];
            }
        for (int t = 0; t < tk.nA; ++t)
            if (a >> t & 1) {
                row[0] += p.bitw[Ab[t]];
                lx |= 1u << Ab[t];
                for (int c = 1; c < JT_NCOL; ++c) row[c] += slotw[c - 1][Ab[t]];
            }
        if (!high_digits_exist(p, lx, loopmask)) row[0] = (int64_t)JT_NO_ROW;      // read the zero row instead
        for (int c = 0; c < JT_NCOL; ++c) itab[(size_t)i * JT_NCOL + c] = (int32_t)(uint32_t)row[c];
        if (i < 8) tk.first_x[i] = (uint32_t)row[0];
    }
    if (hp.tmix) {
        // Plans with mixed-radix rows (kernels *_mix) loop over the rows that EXIST only: with cardinality 5 in three bits a
        // loop of two variables is 25 rows, not 64.  Row r of the table is then the r-th existing row, and what the kernels
        // of the other plans derive from the loop counter travels in the upper half of column 1 + JT_MAX_IN: bits 16-21 the
        // row's counter value in the full loop nest (its logical index bits: evidence), bit 24 + j "outgoing message j's
        // run of rows ends here" (JtTask::out_run counts rows of the full nest).
        std::vector<int> live;
        for (int i = 0; i < tk.total; ++i)
            if ((uint32_t)itab[(size_t)i * JT_NCOL] != JT_NO_ROW) live.push_back(i);
        if (live.empty()) FAIL(JTP_EINVAL, "internal: a loop nest without rows");
        std::vector<int32_t> packed(live.size() * JT_NCOL);
        for (size_t r = 0; r < live.size(); ++r) {
            const int i = live[r];
            for (int c = 0; c < JT_NCOL; ++c) packed[r * JT_NCOL + c] = itab[(size_t)i * JT_NCOL + c];
            uint32_t w = (uint32_t)packed[r * JT_NCOL + 1 + JT_MAX_IN];
            if (w >= (1u << 16)) FAIL(JTP_EINVAL, "internal: sub-box offset %u does not fit 16 bits", w);
            w |= (uint32_t)i << 16;
            for (int j = 0; j < tk.n_out; ++j) {
                const int run = (tk.out_run >> (8 * j)) & 0xff;
                if (r + 1 == live.size() || (live[r + 1] >> run) != (i >> run)) w |= 1u << (24 + j);
            }
            packed[r * JT_NCOL + 1 + JT_MAX_IN] = (int32_t)w;
        }
        itab.swap(packed);
        tk.total = (int)live.size();
        for (int i = 0; i < 8; ++i) tk.first_x[i] = i < tk.total ? (uint32_t)itab[(size_t)i * JT_NCOL] : JT_NO_ROW;
    }
    tk.itab_lds = ((lds + 15) & ~15) + JT_STAGE_SCRATCH * tk.n_in;        // sub-boxes, staging scratch per incoming message
    tk.lds_bytes = tk.itab_lds;                           // (the iteration table is register resident)
    if (strict_budget > 0) {
        // element bits in no message of the task: the elements of a 16-byte vector can be summed before they meet
        // the message product (JtTask::esum bit 0; bit 1 - no evidence on those bits - is the engine's)
        bool e_free = !outs.empty();
        for (int k = 0; k < tk.n_in; ++k) e_free = e_free && !tk.msg[k].e_dep;
        for (int k = 0; k < tk.n_out; ++k) e_free = e_free && tk.msg[JT_MAX_IN + k].red_e == (1 << hp.EB) - 1;
        tk.esum = e_free ? 1 : 0;
        if (lds - JT_RING_BYTES > strict_budget) FAIL(JTP_EUNSUPPORTED, "sub-boxes of one evidence set: %d bytes (limit %d)", lds - JT_RING_BYTES, strict_budget);
        tk.setb = strict_budget;
        tk.lds_bytes = JT_RING_BYTES + JT_MSETS * strict_budget;       // ring + one region per evidence set
    }
    return JTP_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------

JtBlock jtp_make_block(const HostPlan &hp, const JtTask &tk, uint32_t task_index, uint32_t chunk) {
    JtBlock b;
    memset(&b, 0, sizeof b);
    b.task = task_index;
    uint32_t fmask = 0;
    for (int j = 0; j < tk.nF; ++j) {
        fmask |= tk.f_lx[j];
        if (!((chunk >> j) & 1u)) continue;
        b.xF += tk.f_x[j];
        b.lxF += tk.f_lx[j];
        for (int k = 0; k < JT_MAX_MSG; ++k) b.gbase[k] += tk.msg[k].f_w[j];
        for (int k = 0; k < JT_MAX_OUT; ++k) b.pnum[k] += tk.msg[JT_MAX_IN + k].f_p[j];
    }
    b.psi_x0 = tk.psi_off + (int64_t)b.xF;
    for (int i = 0; i < 8; ++i) b.first_x[i] = tk.first_x[i];
    // a chunk whose own digits do not exist (a compact variable's digit beyond its cardinality, a padding bit set)
    // has no rows: the workgroup runs on the zero row and writes its all-zero partial copy (jtp_internal.h)
    if (tk.kind == 0 && tk.keep_rows) b.flags |= JT_BLOCK_KEEP_ROWS;
    if (tk.kind == 0 && !high_digits_exist(hp.pn[tk.pnode], b.lxF, fmask)) {
        b.flags |= JT_BLOCK_INVALID;
        b.xF = 0;
        b.psi_x0 = 0;
        for (int i = 0; i < 8; ++i) b.first_x[i] = JT_NO_ROW;
    }
    return b;
}

// The lean record of a unit task (JtLean, jtp_internal.h), appended to `itab` at a 64-byte boundary; JtTask::lean_off says where
// (0: the task runs the generic pass - it keeps a table, has several outputs, stores a belief, belongs to a plan with mixed-radix
// rows or to a multi-set plan, or stages a message of several partial copies).  JTP_NO_LEAN=1: no task gets one.
void jtp_make_lean(const HostPlan &hp, JtTask &tk, std::vector<int32_t> &itab, bool readout) {
    tk.lean_off = 0;
    if (hp.knobs.no_lean || hp.tmix || (hp.multiset && !readout)) return;
    // (tasks of a propagate: one outgoing message, at most three incoming tables - ten row loops in the dataflow kernels; read-out
    //  tasks, whose kernel is off the hot path: up to three marginals of psi x ALL incoming tables of a unit clique)
    const int max_in = readout ? JT_MAX_IN : 3, max_out = readout ? JT_MAX_OUT : 1;
    if (tk.kind != 0 || !tk.unit || tk.mode != 0 || tk.n_out < 1 || tk.n_out > max_out || tk.n_in > max_in || tk.bel_off >= 0 || tk.vgroups) return;
    // (incoming messages of several partial copies: the generic pass splits the copies of a small sub-box over the threads, which the
    //  lock-step staging of the lean pass does not - measured slower inside a propagate; the read-out kernel takes them, copy after copy)
    if (!readout)
        for (int k = 0; k < tk.n_in; ++k)
            if (tk.msg[k].npart != 1) return;
    if ((tk.debug & ~2) != 0) return;                 // (the JTP_DEBUG timing experiments are switches of the generic pass)
    JtLean ln;
    JtLeanMore more;
    memset(&ln, 0, sizeof ln);
    memset(&more, 0, sizeof more);
    auto fill = [&](JtLeanMsg &lm, const JtMsg &m, int src) {
        lm.off = m.off;
        lm.nfree = m.nfree;
        lm.lds_off = m.lds_off;
        lm.flags = (m.same_launch ? 1 : 0) | (m.fixed ? 2 : 0);
        lm.src = src;
        lm.e_w[0] = m.e_w[0], lm.e_w[1] = m.e_w[1];
        for (int b = 0; b < 8; ++b) lm.w_lo[b] = b < m.nfree ? 1 << m.free_pos[b] : 0;
        for (int b = 0; b < 5; ++b) lm.w_hi[b] = 8 + b < m.nfree ? 1 << m.free_pos[8 + b] : 0;
        lm.w_hi[5] = m.npart, lm.w_hi[6] = m.pstride, lm.w_hi[7] = 0;
        for (int t = 0; t < 8; ++t) lm.t_w[t] = m.t_w[t];
    };
    int n = 0;
    for (int pass = 0; pass < 2; ++pass)                      // the tables that depend on the element bits first
        for (int k = 0; k < tk.n_in; ++k)
            if ((tk.msg[k].e_dep != 0) == (pass == 0)) fill(ln.in[n++], tk.msg[k], k);
    for (int k = 0; k < tk.n_in; ++k) ln.n_e += tk.msg[k].e_dep ? 1 : 0;
    for (int j = 0; j < tk.n_out; ++j) {
        const JtMsg &mo = tk.msg[JT_MAX_IN + j];
        const int rmask = (1 << ((tk.out_run >> (8 * j)) & 0xffu)) - 1;
        if (j == 0) {
            fill(ln.out, mo, JT_MAX_IN);
            ln.rmask = rmask;
            ln.red_e = mo.red_e, ln.red_lane = mo.red_lane, ln.red_wave = mo.red_wave;
            ln.out_pstride = mo.pstride;
        } else {
            fill(more.out[j - 1], mo, JT_MAX_IN + j);
            more.rmask[j - 1] = rmask;
            more.red_e[j - 1] = mo.red_e, more.red_lane[j - 1] = mo.red_lane, more.red_wave[j - 1] = mo.red_wave;
            more.out_pstride[j - 1] = mo.pstride;
        }
    }
    ln.n_in = tk.n_in;
    ln.n_out = tk.n_out;
    ln.total = tk.total;
    ln.settle = tk.settle;
    ln.tmap_off = tk.tmap_off;
    ln.itab_off = tk.itab_off;
    if (tk.tmap_off >= 0) {
        const int n_t = 1 << hp.TB;
        for (int x = 0; x < n_t; ++x)
            if (itab[(size_t)tk.tmap_off + x] < 0) ln.some_invalid = 1;
    }
    for (int i = 0; i < tk.total; ++i)
        if ((uint32_t)itab[(size_t)tk.itab_off + (size_t)i * JT_NCOL] == JT_NO_ROW) ln.some_norow = 1;
    while (itab.empty() || itab.size() % 16) itab.push_back(0);
    tk.lean_off = (int64_t)itab.size();
    const int32_t *w = reinterpret_cast<const int32_t *>(&ln);
    itab.insert(itab.end(), w, w + sizeof ln / 4);
    if (tk.n_out > 1) {
        const int32_t *w2 = reinterpret_cast<const int32_t *>(&more);
        itab.insert(itab.end(), w2, w2 + sizeof more / 4);
    }
}

PlanKnobs jtp_read_knobs() {
    PlanKnobs k;
    auto geti = [](const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; };
    auto getd = [](const char *name, double dflt) { const char *v = getenv(name); return v ? atof(v) : dflt; };
    k.debug = geti("JTP_DEBUG", 0);
    k.layout_policy = geti("JTP_LAYOUT_POLICY", -1);
    k.reduce_min = geti("JTP_REDUCE_MIN", -1);
    k.target_blocks_c = getd("JTP_TARGET_BLOCKS", 1024.0);
    k.target_set = getenv("JTP_TARGET_BLOCKS") != nullptr;
    k.target_blocks_d = getd("JTP_TARGET_BLOCKS_D", k.target_blocks_c);
    k.min_block_log2 = geti("JTP_MIN_BLOCK_LOG2", 13);
    k.multi_min_block_log2 = geti("JTP_MULTI_MIN_BLOCK_LOG2", 16);
    k.max_block_log2 = geti("JTP_MAX_BLOCK_LOG2", 16);
    k.max_block_log2_d = geti("JTP_MAX_BLOCK_LOG2_D", std::min(k.max_block_log2, 15));
    k.tiny_level_elems = getd("JTP_TINY_LEVEL_ELEMS", 2097152.0);
    k.force_level_launches = geti("JTP_FORCE_LEVEL_LAUNCHES", 0);
    k.force_flow = geti("JTP_FORCE_FLOW", 0);
    k.fake_comm = geti("JTP_FAKE_COMM", 0);
    k.flow_debug = (unsigned)geti("JTP_FLOW_DEBUG", 0);
    k.flow_tickets = geti("JTP_FLOW_TICKETS", 0);
    k.no_compact = geti("JTP_NO_COMPACT", 0);
    k.no_search = geti("JTP_NO_SEARCH", 0);
    k.search_all = geti("JTP_SEARCH_ALL", 1);
    k.roctx = geti("JTP_ROCTX", 0);
    k.merge_phases = geti("JTP_MERGE_PHASES", -1);
    k.no_tmix = geti("JTP_NO_TMIX", 0);
    k.tmix_fill = getd("JTP_TMIX_FILL", 0.6);
    k.no_tsplit = geti("JTP_NO_TSPLIT", 0);
    k.settle_level_elems = getd("JTP_SETTLE_LEVEL_ELEMS", 8388608.0);
    k.top_min_loop = geti("JTP_TOP_MIN_LOOP", 3);
    k.lane_low = geti("JTP_LANE_LOW", 2);
    k.longest_first = geti("JTP_LONGEST_FIRST", 1);
    k.top_share = getd("JTP_TOP_SHARE", 0.12);
    k.top_rows2 = getd("JTP_TOP_ROWS2", 2048.0);
    k.top_loop2 = geti("JTP_TOP_LOOP2", 2);
    k.keep_rows_mb = getd("JTP_KEEP_ROWS_MB", 128.0);
    k.esum_always = geti("JTP_EXPERIMENT_ESUM_ALWAYS", 0);
    k.marg_group = std::max(1, std::min(geti("JTP_MARG_GROUP", JT_MAX_OUT), JT_MAX_OUT));
    k.marg_block_log2 = geti("JTP_MARG_BLOCK_LOG2", 0);
    k.no_unit = geti("JTP_NO_UNIT", 0);
    k.keep_invalid = geti("JTP_KEEP_INVALID", 0);
    k.no_vgroups = geti("JTP_NO_VGROUPS", 0);
    k.unit_joint_down = geti("JTP_UNIT_JOINT_DOWN", 0);
    k.no_ef_share = geti("JTP_EF_SHARE", 0) ? -1 : geti("JTP_NO_EF_SHARE", 0);      // (-1: the evidence-free group whatever the number of sets)
    k.unit_ratio = getd("JTP_UNIT_RATIO", 4.0);
    k.no_lean = geti("JTP_NO_LEAN", 0);
    k.no_fold = geti("JTP_NO_FOLD", 0);
    k.fold = geti("JTP_FOLD", -1);
    if (k.fold == 0) k.no_fold = 1;
    k.fold_slots = geti("JTP_FOLD_SLOTS", 1024);
    return k;
}

// The planner proper: one method per stage of jtp_build_plan, run in order; what the stages share lives here.
struct PlanBuilder {
    const jtp_tree_desc *d;
    HostPlan &hp;
    std::string &err;
    int N = 0, NP = 0, esize = 4, ALL = 1, maxdepth = 0;
    std::vector<std::vector<double>> lvl_elems[2];       // [phase][owner][level]: elements, to size workgroups
    std::vector<double> task_bytes;                      // algorithmic bytes of every task

    PlanBuilder(const jtp_tree_desc *desc, HostPlan &plan, std::string &e) : d(desc), hp(plan), err(e) {}
    bool mine(int pnode) const { return hp.pn[pnode].owner == hp.rank || hp.pn[pnode].owner == ALL; }
    double host_elems(const std::vector<int> &vars) const {
        double e = 1;
        for (int v : vars) e *= hp.card[v];
        return e;
    }
    int block_log2_for(int phase, int level, int owner, bool tiny_rule = true) const;
    int read_description();      // validate and copy the caller's description
    int link_nodes();            // cliques, separators, reachability, replicated part
    int fold_marginals();        // marginal tasks named at plan creation, behind messages()
    int reroot();                // single rank: root at the tree's centre
    int decide_units();          // which cliques keep no table (all ones, or their factors' product as a static table)
    int binarise();              // at most three children per node (virtual all-ones cliques)
    int depths();
    int layouts();               // bit order of every clique and separator table
    bool searched_order(int c, const std::vector<int> &host, const std::vector<int> &seps, std::vector<int> &order);   // layout policy 4
    bool wants_static(const PNode &p) const { return p.unit && p.real >= 0 && !p.cover.empty(); }
    int arenas();                // table offsets, host<->device conversion records
    int level_work();
    int make_tasks();            // one task per (clique, phase) - multi-set plans: per (clique, child) in distribute
    int messages();              // message arena, reduce tasks, message offsets of every task
    int schedule();              // launches, workgroup records, exchange schedule
    int finish();                // dataflow segments, sync words, time-stamp region
    int run() {
        int (PlanBuilder::*stages[])() = {&PlanBuilder::read_description, &PlanBuilder::link_nodes, &PlanBuilder::reroot,
                                          &PlanBuilder::decide_units, &PlanBuilder::binarise, &PlanBuilder::depths, &PlanBuilder::layouts,
                                          &PlanBuilder::arenas, &PlanBuilder::level_work, &PlanBuilder::make_tasks,
                                          &PlanBuilder::messages, &PlanBuilder::schedule, &PlanBuilder::finish};
        for (auto stage : stages) {
            const int rc = (this->*stage)();
            if (rc != JTP_OK) return rc;
        }
        return JTP_OK;
    }
};

int PlanBuilder::block_log2_for(int phase, int level, int owner, bool tiny_rule) const {
        if (hp.block_log2 > 0) return std::max(hp.block_log2, hp.TB);
        // aim at ~1024 workgroups per tree level (one round of resident workgroups; in a dataflow launch
        // the next level fills the tail), each streaming 16 KiB .. 256 KiB.  Measured on C4: 1024 is
        // 2-3 % faster than 2048 (which was best with one launch per level), 512 and 4096 slower.
        // (plans whose elements are mostly those of unit cliques - no rows to stream, a workgroup is sub-boxes and arithmetic - do
        //  better with twice the workgroups per level: config 3 8.98 -> 8.63 ms, A/B on one box; 4096: 10.9)
        const double target = (phase == 0 ? hp.knobs.target_blocks_c : hp.knobs.target_blocks_d) * (hp.unit_dominated && !hp.knobs.target_set ? 2.0 : 1.0);
        // (the distribute pass - read + write - streams best in workgroups of at most 32 rows: config 4 0.4345 -> 0.4300 ms,
        //  the collect pass in up to 64: 0.2055 against 0.2084 ms; multi-set plans - whose second phase is marginalisations,
        //  not a read + write pass - keep 64: 1.058 against 1.074 ms)
        // (multi-set plans: always the 64 rows a workgroup can hold - a step serves eight evidence sets, so a workgroup's fixed
        //  cost, eight sets' sub-boxes staged through 8-byte loads, weighs more against its loop than in single-set plans:
        //  64 evidence sets 6.73 -> 6.11 ms, env sweep on one box; 512 / 256 / 128 workgroups per level as the target:
        //  6.33 / 6.16 / 6.11;
        //  one group of eight sets alone is too few workgroups for that: 1.03 -> 1.13 ms; 16 sets 1.82 -> 1.80)
        const int lgmin = hp.multiset && hp.n_batch > JT_MSETS ? hp.knobs.multi_min_block_log2 : hp.knobs.min_block_log2;
        const int lgmax = phase == 1 && !hp.multiset ? hp.knobs.max_block_log2_d : hp.knobs.max_block_log2;
        // levels of a clique or two are latency bound: 4 iterations per workgroup, so that every element
        // load is already in flight while the workgroup waits for its messages
        const double tiny = hp.knobs.tiny_level_elems;
        // (searched layouts price a lone workgroup's latency themselves: config 2 6.87 -> 6.49 ms without the rule)
        if (tiny_rule && lvl_elems[phase][owner][level] <= tiny) return hp.TB + JT_MIN_ITER_LOG2;
        double want = lvl_elems[phase][owner][level] / target;
        int lg = lgmin;
        while (lg < lgmax && (double)(1 << (lg + 1)) <= want) ++lg;
        return std::max(lg, hp.TB);
    }

int PlanBuilder::read_description() {
    if (!d) FAIL(JTP_EINVAL, "null description");
    hp.knobs = jtp_read_knobs();
    if (d->struct_size != (int32_t)sizeof(jtp_tree_desc))
        FAIL(JTP_EINVAL, "jtp_tree_desc size mismatch (%d vs %zu)", d->struct_size, sizeof(jtp_tree_desc));
    if (d->n_cliques < 1 || d->n_vars < 0) FAIL(JTP_EINVAL, "empty tree");
    if (d->n_nodes != 2 * d->n_cliques - 1)
        FAIL(JTP_EINVAL, "n_nodes must be 2*n_cliques-1 (got %d for %d cliques)", d->n_nodes, d->n_cliques);
    if (d->dtype != JTP_F32 && d->dtype != JTP_F64) FAIL(JTP_EINVAL, "bad dtype %d", d->dtype);
    if (d->n_batch < 1) FAIL(JTP_EINVAL, "n_batch must be >= 1");
    if (d->n_ranks < 1 || d->rank < 0 || d->rank >= d->n_ranks) FAIL(JTP_EINVAL, "bad rank %d/%d", d->rank, d->n_ranks);

    hp.n_vars = d->n_vars;
    hp.n_cliques = d->n_cliques;
    hp.n_nodes = d->n_nodes;
    hp.dtype = d->dtype;
    hp.n_ranks = d->n_ranks;
    hp.rank = d->rank;
    hp.n_batch = d->n_batch;
    hp.device = d->device;
    hp.flags = d->flags;
    hp.lds_budget = d->lds_budget;
    hp.block_log2 = d->block_log2;
    hp.layout_policy = d->layout_policy;
    if (hp.knobs.layout_policy >= 0) hp.layout_policy = hp.knobs.layout_policy;               // experiments
    hp.compact = !hp.knobs.no_compact && !(d->flags & JTP_NO_COMPACT);
    hp.multiset = (d->flags & JTP_MULTISET) != 0;
    if (hp.multiset) {
        if (d->n_ranks != 1) FAIL(JTP_EUNSUPPORTED, "multi-set plans run on one rank (evidence sets are independent: give every rank its own sets)");
        // Bit order: "epilogue first" (3) measured 2x faster than "traffic first" (2) on the width-20 tree (fewer
        // epilogues per row; its sub-boxes still fit the 4 KiB regions).  jtp_plan_create falls back to 2 - the
        // smallest sub-boxes - when a clique's sub-boxes do not fit one evidence set's LDS region under 3.
        if (hp.layout_policy == 0) hp.layout_policy = 3;
    }
    hp.VEC = d->dtype == JTP_F32 ? 4 : 2;
    hp.EB = d->dtype == JTP_F32 ? 2 : 1;
    hp.TB = hp.EB + 8;
    N = d->n_cliques;
    esize = d->dtype == JTP_F32 ? 4 : 8;

    if ((d->n_vars > 0 && !d->var_card) || !d->node_var_off || !d->parent_clique || !d->parent_sep)
        FAIL(JTP_EINVAL, "null array in the description");
    hp.card.assign(d->var_card, d->var_card + d->n_vars);
    hp.vbits.resize(d->n_vars);
    for (int v = 0; v < d->n_vars; ++v) {
        if (hp.card[v] < 1) FAIL(JTP_EINVAL, "variable %d has cardinality %d", v, hp.card[v]);
        if (hp.card[v] > (1 << 28)) FAIL(JTP_EUNSUPPORTED, "variable %d has cardinality %d (max 2^28)", v, hp.card[v]);
        hp.vbits[v] = ceil_log2(hp.card[v]);
    }
    hp.node_vars.resize(d->n_nodes);
    // (the CSR offsets are the only bound on node_var_ids this ABI has: they must start at 0 and never decrease)
    if (d->node_var_off[0] != 0) FAIL(JTP_EINVAL, "node_var_off[0] must be 0 (got %d)", d->node_var_off[0]);
    if (d->node_var_off[d->n_nodes] > 0 && !d->node_var_ids) FAIL(JTP_EINVAL, "null array in the description");
    for (int n = 0; n < d->n_nodes; ++n) {
        int a = d->node_var_off[n], b = d->node_var_off[n + 1];
        if (a < 0 || b < a) FAIL(JTP_EINVAL, "node_var_off decreases at node %d (%d, %d)", n, a, b);
        if (b - a > JT_MAX_VARS) FAIL(JTP_EUNSUPPORTED, "node %d has %d variables (max %d)", n, b - a, JT_MAX_VARS);
        for (int i = a; i < b; ++i) {
            int v = d->node_var_ids[i];
            if (v < 0 || v >= d->n_vars) FAIL(JTP_EINVAL, "node %d: unknown variable %d", n, v);
            if (find_var(hp.node_vars[n], v) >= 0) FAIL(JTP_EINVAL, "node %d: variable %d listed twice", n, v);
            hp.node_vars[n].push_back(v);
        }
    }
    if (d->fold_n < 0 || (d->fold_n > 0 && (!d->fold_cliques || !d->fold_var_off))) FAIL(JTP_EINVAL, "null array in the description (fold_*)");
    if (d->fold_n > 0) {
        if (d->fold_var_off[0] != 0) FAIL(JTP_EINVAL, "fold_var_off[0] must be 0");
        for (int i = 0; i < d->fold_n; ++i) {
            const int c = d->fold_cliques[i], a = d->fold_var_off[i], b = d->fold_var_off[i + 1];
            if (c < 0 || c >= N) FAIL(JTP_EINVAL, "fold request %d: node %d is not a clique", i, c);
            if (b < a || b - a > JT_MAX_VARS || (b > a && !d->fold_var_ids)) FAIL(JTP_EINVAL, "fold request %d: bad variable list", i);
            for (int k = a; k < b; ++k)
                if (d->fold_var_ids[k] < 0 || d->fold_var_ids[k] >= hp.n_vars) FAIL(JTP_EINVAL, "fold request %d: variable %d out of range", i, d->fold_var_ids[k]);
        }
        hp.fold_cliques.assign(d->fold_cliques, d->fold_cliques + d->fold_n);
        hp.fold_var_off.assign(d->fold_var_off, d->fold_var_off + d->fold_n + 1);
        hp.fold_var_ids.assign(d->fold_var_ids, d->fold_var_ids + d->fold_var_off[d->fold_n]);
        // (the key jtp_get_marginals makes of a request list: n, cliques, offsets, variables)
        hp.fold_key.push_back(d->fold_n);
        hp.fold_key.insert(hp.fold_key.end(), hp.fold_cliques.begin(), hp.fold_cliques.end());
        hp.fold_key.insert(hp.fold_key.end(), hp.fold_var_off.begin(), hp.fold_var_off.end());
        hp.fold_key.insert(hp.fold_key.end(), hp.fold_var_ids.begin(), hp.fold_var_ids.end());
        hp.folded.assign((size_t)d->fold_n, HostPlan::FoldReq());
    }
    hp.lean = d->cover_off != nullptr && !hp.multiset;
    if (d->cover_off) {
        // (validated for every plan that passes them, used by single-set plans)
        if (d->cover_off[0] != 0) FAIL(JTP_EINVAL, "cover_off[0] must be 0 (got %d)", d->cover_off[0]);
        if (d->cover_off[N] > 0 && !d->cover_ids) FAIL(JTP_EINVAL, "null array in the description");
        hp.cover.assign(N, std::vector<int>());
        for (int c = 0; c < N; ++c) {
            const int a = d->cover_off[c], b = d->cover_off[c + 1];
            if (a < 0 || b < a || b - a > JT_MAX_VARS) FAIL(JTP_EINVAL, "cover_off decreases at clique %d (%d, %d)", c, a, b);
            for (int i = a; i < b; ++i) {
                const int v = d->cover_ids[i];
                if (v < 0 || v >= d->n_vars || find_var(hp.node_vars[c], v) < 0) FAIL(JTP_EINVAL, "clique %d: covered variable %d is not one of its variables", c, v);
                if (find_var(hp.cover[c], v) >= 0) FAIL(JTP_EINVAL, "clique %d: covered variable %d listed twice", c, v);
                hp.cover[c].push_back(v);
            }
        }
    }
    hp.parent_clique.assign(d->parent_clique, d->parent_clique + N);
    hp.parent_sep.assign(d->parent_sep, d->parent_sep + N);
    hp.owner.assign(N, 0);
    if (d->clique_owner)
        for (int c = 0; c < N; ++c) {
            hp.owner[c] = d->clique_owner[c];
            // owner == n_ranks: the clique is REPLICATED - every rank holds its table and runs its tasks (the small
            // top of a partitioned tree: its children's upward messages go to every rank, its downward messages are
            // formed where they are consumed, so a propagate needs one exchange instead of two)
            if (hp.owner[c] < 0 || hp.owner[c] > d->n_ranks || (hp.owner[c] == d->n_ranks && d->n_ranks == 1))
                FAIL(JTP_EINVAL, "clique %d: bad owner %d", c, hp.owner[c]);
        }

    return JTP_OK;
}

int PlanBuilder::link_nodes() {
    // ---- nodes and separators -------------------------------------------------------------
    hp.pn.assign(N, PNode());
    hp.sep_of_node.assign(d->n_nodes, -1);
    hp.root = -1;
    for (int c = 0; c < N; ++c) {
        PNode &p = hp.pn[c];
        p.real = c;
        p.owner = hp.owner[c];
        p.parent = hp.parent_clique[c];
        if (p.parent < 0) {
            if (hp.root >= 0) FAIL(JTP_EINVAL, "two roots (%d and %d)", hp.root, c);
            hp.root = c;
        } else if (p.parent >= N || p.parent == c) FAIL(JTP_EINVAL, "clique %d: bad parent %d", c, p.parent);
    }
    if (hp.root < 0) FAIL(JTP_EINVAL, "no root clique");
    for (int c = 0; c < N; ++c) {
        if (c == hp.root) continue;
        int sn = hp.parent_sep[c];
        if (sn < N || sn >= d->n_nodes) FAIL(JTP_EINVAL, "clique %d: separator node %d out of range", c, sn);
        if (hp.sep_of_node[sn] >= 0) FAIL(JTP_EINVAL, "separator node %d used twice", sn);
        for (int v : hp.node_vars[sn]) {
            if (find_var(hp.node_vars[c], v) < 0 || find_var(hp.node_vars[hp.pn[c].parent], v) < 0)
                FAIL(JTP_EINVAL, "separator node %d: variable %d is not in both adjacent cliques", sn, v);
        }
        PSep s;
        s.node = sn;
        s.child = c;
        s.parent = hp.pn[c].parent;
        s.vars = hp.node_vars[sn];
        hp.sep_of_node[sn] = (int)hp.ps.size();
        hp.pn[c].psep = (int)hp.ps.size();
        hp.ps.push_back(s);
        hp.pn[hp.pn[c].parent].children.push_back(c);
    }
    {   // reachability (rejects cycles)
        std::vector<int> q{hp.root};
        std::vector<char> seen(N, 0);
        seen[hp.root] = 1;
        for (size_t i = 0; i < q.size(); ++i)
            for (int k : hp.pn[q[i]].children)
                if (!seen[k]) seen[k] = 1, q.push_back(k);
        if ((int)q.size() != N) FAIL(JTP_EINVAL, "parent pointers do not form a tree");
    }
    ALL = hp.n_ranks;                                              // owner value of replicated cliques
    for (int c = 0; c < N; ++c)
        if (hp.pn[c].owner == ALL && hp.pn[c].parent >= 0 && hp.pn[hp.pn[c].parent].owner != ALL)
            FAIL(JTP_EINVAL, "clique %d is replicated but its parent %d is not (the replicated part must contain the root)", c, hp.pn[c].parent);

    return JTP_OK;
}

int PlanBuilder::reroot() {
    // ---- re-root at the tree's centre (single rank): results do not depend on the root (every
    //      belief is psi times ALL incoming messages), but the number of levels = dependent launches
    //      does: a chain of N cliques needs N/2 levels per phase instead of N.
    if (hp.n_ranks == 1 && !(hp.flags & JTP_KEEP_ROOT) && N > 2) {
        std::vector<std::vector<std::pair<int, int>>> adj(N);      // (neighbour, psep)
        for (int c = 0; c < N; ++c)
            if (c != hp.root) {
                adj[c].push_back({hp.pn[c].parent, hp.pn[c].psep});
                adj[hp.pn[c].parent].push_back({c, hp.pn[c].psep});
            }
        auto bfs = [&](int src, std::vector<int> &dist, std::vector<int> &prev) {
            dist.assign(N, -1);
            prev.assign(N, -1);
            std::vector<int> q{src};
            dist[src] = 0;
            for (size_t i = 0; i < q.size(); ++i)
                for (auto &e : adj[q[i]])
                    if (dist[e.first] < 0) dist[e.first] = dist[q[i]] + 1, prev[e.first] = q[i], q.push_back(e.first);
            return q.back();                                       // a farthest clique
        };
        std::vector<int> d1, d2, p1, p2;
        const int a = bfs(hp.root, d1, p1);
        const int b = bfs(a, d2, p2);                              // a..b is a diameter
        int centre = b;
        for (int steps = d2[b] / 2; steps > 0; --steps) centre = p2[centre];
        if (d2[b] > 0 && centre != hp.root) {
            std::vector<int> order{centre};
            std::vector<int> np(N, -2), nsep(N, -1);
            np[centre] = -1;
            for (size_t i = 0; i < order.size(); ++i)
                for (auto &e : adj[order[i]])
                    if (np[e.first] == -2) np[e.first] = order[i], nsep[e.first] = e.second, order.push_back(e.first);
            for (int c = 0; c < N; ++c) {
                hp.pn[c].parent = np[c];
                hp.pn[c].psep = nsep[c];
                hp.pn[c].children.clear();
            }
            for (int c : order)
                if (np[c] >= 0) {
                    hp.pn[np[c]].children.push_back(c);
                    hp.ps[nsep[c]].child = c;
                    hp.ps[nsep[c]].parent = np[c];
                }
            hp.root = centre;
        }
    }
    return JTP_OK;
}

int PlanBuilder::decide_units() {
    // ---- unit cliques (JtTask::unit): no table, no belief table.  A clique becomes one when the description says its
    //      potential depends on few of its variables: on none (no factor assigned: 365 of the 878 cliques of the config-3
    //      lattice, 98 % of its table bytes), or on a part at most 1 / unit_ratio of the table - the product of its factors
    //      then travels as a static table over the covered variables.  The reference leaves such axes at length 1
    //      (junctiontree.py:52-61); rounds 1-4 of this engine materialised them, and streamed 9 GiB of ones three times per
    //      propagate on that lattice.  A clique covered (nearly) whole keeps its table: streaming it costs less than staging it.
    if (!hp.lean || hp.knobs.no_unit) return JTP_OK;
    double all = 0, unit = 0;
    for (int c = 0; c < N; ++c) {
        PNode &p = hp.pn[c];
        p.cover = hp.cover[c];
        double full = 1, part = 1;
        for (int v : hp.node_vars[c]) full *= hp.card[v];
        for (int v : p.cover) part *= hp.card[v];
        p.unit = p.cover.size() < hp.node_vars[c].size() && part * hp.knobs.unit_ratio <= full;
        all += full;
        unit += p.unit ? full : 0.0;
    }
    hp.unit_dominated = unit > 0.5 * all;
    return JTP_OK;
}

int PlanBuilder::binarise() {
    // ---- binarise: at most 3 children per node, via virtual all-ones cliques ---------------
    // (a unit clique with a static table stages it like one more incoming message: JT_MAX_IN = 4 then leaves room for the
    //  parent's message and TWO children)
    for (int c = 0; c < (int)hp.pn.size(); ++c) {
        const size_t lim = (wants_static(hp.pn[c]) && hp.pn[c].parent >= 0) ? 2 : 3;
        while (hp.pn[c].children.size() > lim) {
            std::vector<int> old = hp.pn[c].children, fresh;
            for (size_t g = 0; g < old.size(); g += 3) {
                size_t ge = std::min(old.size(), g + 3);
                if (ge - g == 1) {
                    fresh.push_back(old[g]);
                    continue;
                }
                PNode v;
                v.real = -1;
                v.unit = !hp.multiset && !hp.knobs.no_unit;     // (all ones: nothing to store)
                v.owner = hp.pn[c].owner;
                v.parent = c;
                for (size_t i = g; i < ge; ++i) {
                    for (int var : hp.ps[hp.pn[old[i]].psep].vars)
                        if (find_var(v.vars, var) < 0) v.vars.push_back(var);
                    v.children.push_back(old[i]);
                }
                if (v.vars.size() > JT_MAX_VARS) FAIL(JTP_EUNSUPPORTED, "virtual clique too wide");
                int vi = (int)hp.pn.size();
                PSep s;
                s.node = -1;
                s.child = vi;
                s.parent = c;
                s.vars = v.vars;
                v.psep = (int)hp.ps.size();
                hp.ps.push_back(s);
                for (size_t i = g; i < ge; ++i) {
                    hp.pn[old[i]].parent = vi;
                    hp.ps[hp.pn[old[i]].psep].parent = vi;
                }
                hp.pn.push_back(v);
                fresh.push_back(vi);
            }
            hp.pn[c].children = fresh;
        }
    }
    NP = (int)hp.pn.size();

    return JTP_OK;
}

int PlanBuilder::depths() {
    // ---- depth ------------------------------------------------------------------------------
    maxdepth = 0;
    {
        std::vector<int> q{hp.root};
        hp.pn[hp.root].depth = 0;
        for (size_t i = 0; i < q.size(); ++i)
            for (int k : hp.pn[q[i]].children) {
                hp.pn[k].depth = hp.pn[q[i]].depth + 1;
                maxdepth = std::max(maxdepth, hp.pn[k].depth);
                q.push_back(k);
            }
    }

    return JTP_OK;
}

// Layout policy 4: the variables of the thread part (and their order) chosen by the cost model of plan_loops' search,
// summed over the clique's collect and distribute tasks.  Candidates: every set of variables that fills the thread
// part (cliques of up to 12 variables), else a hill climb from the "traffic first" order by swapping one variable in
// and one out.  Inside the thread part variables of the fewest messages go lowest (element bits that are summed cost
// nothing, wave bits that are summed cost a barrier phase each), as in policy 2.
bool PlanBuilder::searched_order(int c, const std::vector<int> &host, const std::vector<int> &seps, std::vector<int> &order) {
    const PNode &p = hp.pn[c];
    const int n = (int)host.size(), TB = hp.TB;
    if (n == 0 || n > 31) return false;
    std::vector<int> cnt(n, 0), canon(n), rank_of(n);
    const bool has_static = wants_static(p);
    for (int i = 0; i < n; ++i) {
        for (int sp : seps) cnt[i] += find_var(hp.ps[sp].vars, host[i]) >= 0;
        if (has_static) cnt[i] += find_var(p.cover, host[i]) >= 0;
        canon[i] = i;
    }
    auto waste = [&](int i) { return (double)(1 << hp.vbits[host[i]]) / hp.card[host[i]]; };
    std::stable_sort(canon.begin(), canon.end(), [&](int a, int b) { return cnt[a] != cnt[b] ? cnt[a] < cnt[b] : waste(a) < waste(b); });
    for (int r = 0; r < n; ++r) rank_of[canon[r]] = r;
    int total_bits = 0;
    for (int v : host) total_bits += hp.vbits[v];
    if (total_bits <= TB || total_bits > JT_MAX_BITS) return false;           // one workgroup row: nothing to choose; too large: refused by layouts()
    auto bits_of = [&](uint32_t S) {
        int b = 0;
        for (int i = 0; i < n; ++i)
            if (S >> i & 1) b += hp.vbits[host[i]];
        return b;
    };
    auto valid = [&](uint32_t S) {                // fills the thread part, and would not without its last variable
        if (!S) return false;
        int last = -1;
        for (int r = n - 1; r >= 0 && last < 0; --r)
            if (S >> canon[r] & 1) last = canon[r];
        const int b = bits_of(S);
        return b >= TB && b - hp.vbits[host[last]] < TB;
    };
    struct Eval { double us = 1e30; uint32_t Ld = 0, Lc = 0; };
    std::vector<int> idx;                          // scratch: candidate order as indices into host
    auto order_of = [&](uint32_t S) {
        idx.clear();
        for (int r = 0; r < n; ++r) if (S >> canon[r] & 1) idx.push_back(canon[r]);
        for (int r = 0; r < n; ++r) if (!(S >> canon[r] & 1)) idx.push_back(canon[r]);
    };
    std::vector<int> pos(n);
    auto evaluate = [&](uint32_t S) {
        Eval ev;
        order_of(S);
        int bit = 0;
        for (int i : idx) pos[i] = bit, bit += hp.vbits[host[i]];
        CostEnv e;
        e.TB = TB, e.EB = hp.EB, e.nbits = std::max(bit, TB + JT_MIN_ITER_LOG2);
        e.unit = p.unit;
        e.red_log2 = hp.knobs.reduce_min >= 0 ? std::max(0, ceil_log2(std::max(hp.knobs.reduce_min, 1))) : (hp.chain_plan ? 3 : 6);
        e.chain = hp.chain_plan;
        e.min_loop_log2 = hp.chain_plan ? JT_MIN_LOOP_LOG2 : JT_MIN_ITER_LOG2;
        uint32_t grouped = 0;
        for (int i : idx) {
            const int card = hp.card[host[i]], nb = hp.vbits[host[i]];
            if (pos[i] >= TB && (hp.compact || p.unit) && (card & (card - 1)) != 0) {
                const uint32_t g = ((1u << nb) - 1u) << pos[i];
                e.units.push_back(g), grouped |= g;
                e.fill *= (double)card / (double)(1 << nb);
            }
        }
        for (int b = TB; b < e.nbits; ++b)
            if (!(grouped >> b & 1)) e.units.push_back(1u << b);
        if (hp.compact || p.unit) e.fill = std::ldexp(e.fill, -(e.nbits - std::max(bit, TB)));
        std::sort(e.units.begin(), e.units.end());
        if (hp.lds_budget > 0) e.lds_cap = (p.unit ? 0 : JT_RING_BYTES) + hp.lds_budget + JT_STAGE_SCRATCH * 4L;
        uint32_t stat_mask = 0;
        if (has_static)
            for (int v : p.cover) {
                const int i = find_var(host, v);
                if (i >= 0) stat_mask |= ((1u << hp.vbits[v]) - 1u) << pos[i];
            }
        auto mask_of = [&](int sp) {
            uint32_t m = 0;
            for (int v : hp.ps[sp].vars) {
                const int i = find_var(host, v);
                if (i >= 0) m |= ((1u << hp.vbits[v]) - 1u) << pos[i];
            }
            return m;
        };
        std::vector<uint32_t> kids, none;
        for (int k : p.children) kids.push_back(mask_of(hp.pn[k].psep));
        const double elems = std::ldexp(1.0, e.nbits);
        ev.us = 0;
        if (c != hp.root && p.psep >= 0) {
            e.dist = false;
            e.max_iter_log2 = std::min(std::max(block_log2_for(0, p.depth, p.owner, false) - TB, JT_MIN_ITER_LOG2), JT_MAX_ITER_LOG2);
            e.share = std::min(1.0, elems / std::max(elems, lvl_elems[0][p.owner][p.depth]));
            std::vector<uint32_t> cin = kids;
            if (has_static) cin.push_back(stat_mask);
            LoopChoice ch = search_loops(e, cin, {mask_of(p.psep)}, false);
            ev.us += ch.us, ev.Lc = ch.L;
        }
        {
            e.dist = true;
            e.max_iter_log2 = std::min(std::max(block_log2_for(1, p.depth, p.owner, false) - TB, JT_MIN_ITER_LOG2), JT_MAX_ITER_LOG2);
            e.share = std::min(1.0, elems / std::max(elems, lvl_elems[1][p.owner][p.depth]));
            std::vector<uint32_t> ins;
            if (p.psep >= 0) ins.push_back(mask_of(p.psep));
            if (has_static) ins.push_back(stat_mask);
            ins.insert(ins.end(), kids.begin(), kids.end());
            LoopChoice ch = search_loops(e, ins, kids, false);
            ev.us += ch.us, ev.Ld = ch.L;
        }
        return ev;
    };
    uint32_t bestS = 0;
    Eval best;
    if (n <= 12) {
        for (uint32_t S = 1; S < (1u << n); ++S) {
            if (!valid(S)) continue;
            Eval ev = evaluate(S);
            if (ev.us < best.us) best = ev, bestS = S;
        }
    } else {
        uint32_t S = 0;                              // start: the canonical prefix
        for (int r = 0; r < n && bits_of(S) < TB; ++r) S |= 1u << canon[r];
        best = evaluate(S), bestS = S;
        for (int pass = 0; pass < 4; ++pass) {
            bool better = false;
            for (int i = 0; i < n; ++i) {
                if (!(bestS >> i & 1)) continue;
                for (int j = 0; j < n; ++j) {
                    if (bestS >> j & 1) continue;
                    const uint32_t S2 = (bestS & ~(1u << i)) | (1u << j);
                    if (!valid(S2)) continue;
                    Eval ev = evaluate(S2);
                    if (ev.us < best.us) {
                        best = ev, bestS = S2, better = true;
                        break;                       // i has left the set
                    }
                }
            }
            if (!better) break;
        }
    }
    if (best.us >= 1e30) return false;
    // thread part in canonical order; above it the variables the distribute task loops over first (its rows are
    // then consecutive 4 KiB pieces), then the collect task's, then the chunk bits
    order_of(bestS);
    int bit = 0;
    for (int i : idx) pos[i] = bit, bit += hp.vbits[host[i]];
    auto klass = [&](int i) {
        const uint32_t m = ((1u << hp.vbits[host[i]]) - 1u) << pos[i];
        if (pos[i] < TB) return 0;
        return (m & best.Ld) ? 1 : (m & best.Lc) ? 2 : 3;
    };
    std::vector<int> fin = idx;
    std::stable_sort(fin.begin(), fin.end(), [&](int a, int b) { return klass(a) < klass(b); });
    order.clear();
    for (int i : fin) order.push_back(host[i]);
    return true;
}

int PlanBuilder::layouts() {
    // ---- bit layouts ----------------------------------------------------------------------
    // (level sizes from the padded index spaces, for the searched layouts: level_work() recomputes them from the
    //  physical sizes once the layouts are known)
    for (int ph = 0; ph < 2; ++ph) lvl_elems[ph].assign(hp.n_ranks + 1, std::vector<double>(maxdepth + 1, 0.0));
    for (int c = 0; c < NP; ++c) {
        const PNode &p = hp.pn[c];
        int cb = 0;
        for (int v : (p.real >= 0 ? hp.node_vars[p.real] : p.vars)) cb += hp.vbits[v];
        const double e = std::ldexp(1.0, std::max(cb, hp.TB + JT_MIN_ITER_LOG2));
        if (c != hp.root) lvl_elems[0][p.owner][p.depth] += e;
        lvl_elems[1][p.owner][p.depth] += e;
    }
    {
        int tiny = 0;
        for (int c = 0; c < NP; ++c) tiny += lvl_elems[1][hp.pn[c].owner][hp.pn[c].depth] <= hp.knobs.tiny_level_elems;
        hp.chain_plan = 2 * tiny > NP;
    }
    for (int c = 0; c < NP; ++c) {
        PNode &p = hp.pn[c];
        std::vector<int> host = p.real >= 0 ? hp.node_vars[p.real] : p.vars;
        std::vector<int> seps;
        if (p.psep >= 0) seps.push_back(p.psep);
        for (int k : p.children) seps.push_back(hp.pn[k].psep);
        std::vector<int> order;                       // LSB first
        // Policy 0 chooses per clique between the two heuristics below.  "Epilogue first" (policy 3) suits
        // cliques whose messages are small beside the table (C4: 3 x 8 KiB against 4 MiB); "traffic first"
        // (policy 2) those whose messages are not, and chain-like cliques, whose levels are latency bound
        // and gain from fewer partial copies.  Measured crossover on trees of 64 cliques of 2^20..2^23
        // entries, cardinalities 2..16: message bytes / table bytes ~ 0.1-0.2 for branching cliques of
        // binary variables, 0.03-0.07 with wider ones; chains of any shape tested (cardinality 4..128)
        // were 1.2-1.7x faster traffic first.
        int policy = hp.layout_policy;
        if (policy == 0 && !seps.empty()) {
            double msg_bytes = 0;
            for (int sp : seps) {
                int sb = 0;
                for (int v : hp.ps[sp].vars) sb += hp.vbits[v];
                msg_bytes += 8.0 * (double)((int64_t)1 << sb);
            }
            if (wants_static(p)) {
                int sb = 0;
                for (int v : p.cover) sb += hp.vbits[v];
                msg_bytes += 8.0 * (double)((int64_t)1 << sb);
            }
            int cb = 0;
            for (int v : host) cb += hp.vbits[v];
            const double r = msg_bytes / ((double)((int64_t)1 << std::max(cb, hp.TB + JT_MIN_ITER_LOG2)) * esize);
            // (wide variables move the crossover down: the classes of policy 3 cannot split a variable)
            const double thr = (double)cb / std::max<size_t>(host.size(), 1) >= 2.0 ? 0.04 : 0.12;
            policy = (r >= thr || (p.children.size() <= 1 && r >= 0.01)) ? 2 : 3;
            // where the messages weigh that much: search the thread part with the cost model (multi-set plans keep
            // the heuristic: their sub-boxes have a hard per-set budget that the model does not know)
            if ((policy == 2 || hp.knobs.search_all) && !hp.multiset && !hp.knobs.no_search) policy = 4;
        }
        if (policy == 4 && (seps.empty() || !searched_order(c, host, seps, order))) policy = 2, order.clear();
        p.layout = policy;
        if (policy == 4) {
            // order filled by searched_order
        } else if (policy == 1 || seps.empty()) {
            order.assign(host.rbegin(), host.rend());
        } else if (policy == 2) {
            // Message traffic first (separators nearly as large as the cliques: every message entry is
            // used only a few times): variables in the fewest messages go lowest, so that the elements
            // one workgroup covers (thread part + loops) touch as few distinct entries of each message as
            // possible - a variable absent from a message costs that message's sub-box nothing.
            std::vector<std::pair<int, int>> keyed;           // (messages containing v, position in host order)
            for (size_t i = 0; i < host.size(); ++i) {
                int cnt = 0;
                for (int sp : seps) cnt += find_var(hp.ps[sp].vars, host[i]) >= 0;
                keyed.push_back({cnt, (int)i});
            }
            // (among variables of equally many messages, powers of two lowest: the thread part is the one place
            //  where a cardinality is still padded to a power of two)
            auto waste = [&](int i) { return (double)(1 << hp.vbits[host[i]]) / hp.card[host[i]]; };
            std::stable_sort(keyed.begin(), keyed.end(), [&](const std::pair<int, int> &a, const std::pair<int, int> &b) {
                return a.first != b.first ? a.first < b.first : waste(a.second) < waste(b.second);
            });
            // (moving variables of every message onto the wave bits, to spare the epilogues their barriers,
            //  was tried: the larger sub-boxes cost more than the barriers - config 3 27 -> 37 ms)
            for (auto &kv : keyed) order.push_back(host[kv.second]);
        } else {
            // Classes: priv = in no separator; ponly = only in the parent's; xorc = in some but not
            // all child separators; allc = in every child separator (leaf: in the parent's).
            // Target shape, low to high:  e bits <- priv | lane bits <- xorc | wave bits <- allc |
            // rest of xorc, allc | ponly, priv.  Bits of outgoing messages that sit in the thread
            // part need no cross-lane sum and no outer (A) loop; bits in every outgoing message can
            // be fixed per workgroup (F) without partial copies; everything else up high becomes the
            // register-summed R loop of the distribute pass, which moves twice the bytes of collect.
            int nchild = (int)p.children.size();
            std::vector<int> priv, ponly, part, allc;
            int n_full = 0;
            for (size_t i = 0; i < host.size(); ++i) {
                int v = host[i];
                int in_parent = p.psep >= 0 && find_var(hp.ps[p.psep].vars, v) >= 0;
                int in_child = 0;
                for (int k : p.children) in_child += find_var(hp.ps[hp.pn[k].psep].vars, v) >= 0;
                if (!in_parent && !in_child) priv.push_back(v);
                else if ((nchild > 0 && in_child == nchild) || nchild == 0) {
                    // variables of EVERY message (parent's too) first: they are never summed over in
                    // either pass, so they are the best occupants of the thread part
                    if (in_parent && nchild > 0) allc.insert(allc.begin() + n_full++, v);
                    else allc.push_back(v);
                } else if (in_child == 0) ponly.push_back(v);
                else part.push_back(v);
            }
            // (inside every class, powers of two first: they are the ones taken into the thread part, the one place
            //  where a cardinality is still padded to a power of two)
            for (std::vector<int> *cls : {&priv, &ponly, &part})
                std::stable_sort(cls->begin(), cls->end(), [&](int a, int b) {
                    return (double)(1 << hp.vbits[a]) / hp.card[a] < (double)(1 << hp.vbits[b]) / hp.card[b];
                });
            auto take = [&](std::vector<int> &from, int want_bits) {
                int got = 0;
                while (!from.empty() && got < want_bits) {
                    int v = from.front();
                    from.erase(from.begin());
                    order.push_back(v);
                    got += hp.vbits[v];
                }
                return got;
            };
            auto bits_of = [&](const std::vector<int> &l) {
                int b = 0;
                for (int v : l) b += hp.vbits[v];
                return b;
            };
            // (variables are not split: a wide variable taken for the element bits spills into the
            // lane bits, so without a private variable prefer one that outgoing messages contain)
            int got = take(priv, hp.EB);
            if (got < hp.EB) got += take(part, hp.EB - got);
            if (got < hp.EB) got += take(allc, hp.EB - got);
            if (got < hp.EB) got += take(ponly, hp.EB - got);
            int lane = got > hp.EB ? got - hp.EB : 0;       // bits a wide variable already spilled
            lane += take(part, 6 - std::min(lane, 6));
            // lanes prefer message bits (no shuffle sum) but leave two allc bits for the waves
            while (lane < 6 && !allc.empty() && bits_of(allc) - hp.vbits[allc.front()] >= 2) lane += take(allc, 1);
            if (lane < 6) lane += take(ponly, 6 - lane);
            if (lane < 6) lane += take(priv, 6 - lane);
            if (lane < 6) lane += take(allc, 6 - lane);
            int wave = take(allc, 2);
            if (wave < 2) wave += take(part, 2 - wave);
            if (wave < 2) wave += take(ponly, 2 - wave);
            if (wave < 2) wave += take(priv, 2 - wave);
            // Above the thread part: bits of no outgoing message first (they become the register-summed R
            // loop), bits of every child separator last (they become the chunk bits F), so that a workgroup's
            // loop rows are consecutive 4 KiB pieces of the table wherever the classes allow.  Rows strided
            // by 16-64 KiB stream 10-20 % slower than consecutive ones (tools/dma_bench.hip: 5.0 against
            // 6.2 TB/s read, 4.5 against 5.7 read+write); measured on C4: 1.5 %.
            take(priv, 1 << 20);
            take(ponly, 1 << 20);
            take(part, 1 << 20);
            take(allc, 1 << 20);
        }
        p.vars = order;
        p.pos.clear();
        p.nb.clear();
        // Thread part at true cardinalities (round 3): where a variable of the low TB index bits is not a power of two, those
        // variables become mixed-radix digits of a row of prod(card) elements instead of 2^TB - five variables of cardinality 3
        // in ten bits stored 4.2 x the table (round 2).  Such a clique keeps every variable wholly below or wholly above bit TB
        // (bits in between are padding: tpad_mask), and all tasks of the plan reach their elements through PNode::tmap.
        {
            // (Which cliques: those whose bit-field thread part would be filled to less than PlanKnobs::tmix_fill, 0.6 -
            //  cardinality 3: (3/4)^5 = 0.24, 5: 0.24, 6: 0.42.  Fuller ones keep the bit fields: a bit-field thread part may
            //  hold the low bit of one more variable, so it needs fewer rows - cardinality 7, width 7: fill 0.67, 0.30 ms
            //  against 0.46 ms with mixed-radix rows for 1.7 x the memory; tools/odd_time.py.)
            int b = 0;
            double fill = 1.0;
            for (int v : p.vars) {
                if (b + hp.vbits[v] <= hp.TB) fill *= (double)hp.card[v] / (double)(1 << hp.vbits[v]);
                b += hp.vbits[v];
            }
            // (a unit clique stores nothing: no rows to pack - it keeps the bit-field thread part, whose entries that name
            //  no table entry its thread map marks)
            p.tmix = fill < hp.knobs.tmix_fill && hp.compact && !hp.multiset && !hp.knobs.no_tmix && !p.unit;
        }
        int bit = 0;
        p.tpad_mask = 0;
        p.tsplit = -1, p.tsplit_lb = 0;
        for (int v : p.vars) {
            if (p.tmix && bit < hp.TB && bit + hp.vbits[v] > hp.TB) {
                // A variable across bit TB: its low bits become a radix-2^lb digit of the row and its high bits a digit of the rows
                // above with ceil(card / 2^lb) values - the rows of a bit-field thread part, a third to a half fewer than with the
                // variable moved up whole - where the entries this stores for values >= card (zeros) cost at most a quarter;
                // else the variable moves above bit TB and the bits below it are padding.
                const int lb = hp.TB - bit, card = hp.card[v], hi = (card + (1 << lb) - 1) >> lb;
                if (!hp.knobs.no_tsplit && (double)(hi << lb) <= 1.25 * card) {
                    p.tsplit = (int)p.pos.size(), p.tsplit_lb = lb;
                } else {
                    for (int b = bit; b < hp.TB; ++b) p.tpad_mask |= 1u << b;
                    bit = hp.TB;
                }
            }
            if (p.unit && bit < hp.TB && bit + hp.vbits[v] > hp.TB && (hp.card[v] & (hp.card[v] - 1)) != 0) {
                // A unit clique has no table whose zeros could mark the entries that do not exist: which entries of a ROW exist must
                // depend on the thread alone (PNode::tmap) and which rows exist on the row alone (JT_NO_ROW).  A variable across bit
                // TB whose cardinality is no power of two would tie the two together: it moves above bit TB whole.
                for (int b = bit; b < hp.TB; ++b) p.tpad_mask |= 1u << b;
                bit = hp.TB;
            }
            p.pos.push_back(bit);
            p.nb.push_back(hp.vbits[v]);
            bit += hp.vbits[v];
        }
        if (p.tmix || p.unit)
            for (int b = bit; b < hp.TB; ++b) p.tpad_mask |= 1u << b;
        hp.tmix = hp.tmix || p.tmix;
        if (bit > JT_MAX_BITS) FAIL(JTP_EUNSUPPORTED, "clique %d needs %d index bits (max %d)", p.real, bit, JT_MAX_BITS);
        p.nbits = std::max(bit, hp.TB + JT_MIN_ITER_LOG2);   // >= 4 loop iterations per workgroup
        if (p.nbits - hp.TB > JT_MAX_HI) FAIL(JTP_EUNSUPPORTED, "clique %d too large", p.real);
        // Physical layout (jtp_internal.h, JT_NO_ROW): rows above the thread part.  A variable that starts inside
        // the thread part keeps its bit field (its upper bits double the row stride); a variable wholly above it
        // whose cardinality is not a power of two is stored at its true cardinality - its bits form a group that
        // every task keeps together; index bits above the last variable are padding and store nothing.
        p.bitw.assign(p.nbits, 0);
        p.group_mask.clear();
        p.group_pos.clear();
        p.group_card.clear();
        p.pad_mask = 0;
        for (int b = 0; b < hp.TB && b < p.nbits; ++b) p.bitw[b] = (int64_t)1 << b;
        int64_t mult = (int64_t)1 << hp.TB;
        p.trow = 1 << hp.TB;
        p.tmap.clear();
        if (p.tmix) {
            // row = the thread-part variables as mixed-radix digits, first variable fastest
            std::vector<int64_t> tstride(p.vars.size(), 0);
            int64_t prod = 1;
            for (size_t i = 0; i < p.vars.size(); ++i) {
                if (p.pos[i] + p.nb[i] <= hp.TB) tstride[i] = prod, prod *= hp.card[p.vars[i]];
                else if ((int)i == p.tsplit) tstride[i] = prod, prod <<= p.tsplit_lb;          // the low bits of the variable across TB
            }
            p.trow = (int)((prod + hp.VEC - 1) / hp.VEC * hp.VEC);
            p.tmap.assign((size_t)1 << hp.TB, -1);
            for (uint32_t x = 0; x < (1u << hp.TB); ++x) {
                if (x & p.tpad_mask) continue;
                int64_t off = 0;
                bool ok = true;
                for (size_t i = 0; i < p.vars.size() && ok; ++i) {
                    if ((int)i == p.tsplit) {
                        off += (int64_t)((x >> p.pos[i]) & ((1u << p.tsplit_lb) - 1u)) * tstride[i];       // (every low value has a place)
                        continue;
                    }
                    if (p.pos[i] + p.nb[i] > hp.TB) continue;
                    const int digit = (int)((x >> p.pos[i]) & ((1u << p.nb[i]) - 1u));
                    ok = digit < hp.card[p.vars[i]];
                    off += digit * tstride[i];
                }
                if (ok) p.tmap[x] = (int32_t)off;
            }
            for (int b = 0; b < hp.TB && b < p.nbits; ++b) p.bitw[b] = 0;      // (inside a row: tmap, not bit weights)
            mult = p.trow;
            // compact form (round 5): the logical threads that own an entry, if two waves hold them all
            p.vmap.clear();
            if (!hp.multiset && !hp.knobs.no_vgroups && !p.unit) {
                std::vector<int32_t> owners;
                int spare = -1;
                for (int t = 0; t < JT_THREADS; ++t) {
                    bool any = false;
                    for (int e = 0; e < hp.VEC; ++e) any = any || p.tmap[(size_t)t * hp.VEC + e] >= 0;
                    if (any) owners.push_back(t);
                    else if (spare < 0) spare = t;
                }
                if (owners.size() <= 128 && (owners.size() == 128 || spare >= 0)) {
                    p.vmap = owners;
                    p.vmap.resize(128, spare);
                }
            }
        }
        for (size_t i = 0; i < p.vars.size(); ++i) {
            const int pos = p.pos[i], nb = p.nb[i], card = hp.card[p.vars[i]];
            if (pos + nb <= hp.TB) continue;
            // (a unit clique has no table whose zeros could stand for a digit beyond the cardinality: its rows are always counted at
            //  the true cardinalities, JTP_NO_COMPACT or not)
            const bool whole = pos >= hp.TB && (hp.compact || p.unit) && (card & (card - 1)) != 0;
            // (the variable across TB of a mixed-radix clique: its high bits are a digit of ceil(card / 2^lb) values)
            const int hi = (int)i == p.tsplit ? (card + (1 << p.tsplit_lb) - 1) >> p.tsplit_lb : 0;
            const bool split_group = hi > 0 && (hi & (hi - 1)) != 0;
            for (int k = std::max(0, hp.TB - pos); k < nb; ++k) {
                p.bitw[pos + k] = whole ? mult << k : (split_group ? mult << (k - (hp.TB - pos)) : mult);
                if (!whole && !split_group) mult <<= 1;
            }
            if (whole) {
                p.group_mask.push_back(((1u << nb) - 1u) << pos);
                p.group_pos.push_back(pos);
                p.group_card.push_back(card);
                mult *= card;
            } else if (split_group) {
                p.group_mask.push_back(((1u << (nb - (hp.TB - pos))) - 1u) << hp.TB);
                p.group_pos.push_back(hp.TB);
                p.group_card.push_back(hi);
                mult *= hi;
            }
        }
        for (int b = std::max(bit, hp.TB); b < p.nbits; ++b) {
            if (hp.compact || p.unit) p.pad_mask |= 1u << b; // weight 0, exists only when clear
            else p.bitw[b] = mult, mult <<= 1;
        }
        p.phys_elems = mult;
        if (mult > ((int64_t)1 << 31)) FAIL(JTP_EUNSUPPORTED, "clique %d too large", p.real);
    }
    for (PNode &p : hp.pn)
        if (!p.tmix && (hp.tmix || p.unit)) {
            // bit-field rows: the identity map, so that one kernel family serves every task of a plan with mixed-radix rows; a unit
            // clique's map says which entries of a row EXIST (-1: a thread-part variable's digit beyond its cardinality, an index bit
            // below TB that no variable owns) - the zeros a stored table would hold there
            p.tmap.resize((size_t)1 << hp.TB);
            for (uint32_t x = 0; x < (1u << hp.TB); ++x) {
                bool ok = true;
                if (p.unit) {
                    ok = !(x & p.tpad_mask);
                    for (size_t i = 0; i < p.vars.size() && ok; ++i)
                        if (p.pos[i] + p.nb[i] <= hp.TB) ok = (int)((x >> p.pos[i]) & ((1u << p.nb[i]) - 1u)) < hp.card[p.vars[i]];
                }
                p.tmap[x] = ok ? (int32_t)x : -1;
            }
        }
    for (size_t s = 0; s < hp.ps.size(); ++s) {
        PSep &sp = hp.ps[s];
        const PNode &ch = hp.pn[sp.child];
        std::vector<int> order;
        for (int v : ch.vars)
            if (find_var(sp.vars, v) >= 0) order.push_back(v);
        sp.vars = order;
        int bit = 0;
        sp.pos.clear();
        sp.nb.clear();
        for (int v : sp.vars) {
            sp.pos.push_back(bit);
            sp.nb.push_back(hp.vbits[v]);
            bit += hp.vbits[v];
        }
        sp.nbits = bit;
        if (bit > 28) FAIL(JTP_EUNSUPPORTED, "separator with %d index bits", bit);
    }
    // static tables of unit cliques: the covered variables in the clique's device order, a plain bit field like a message
    for (int c = 0; c < NP; ++c) {
        PNode &p = hp.pn[c];
        if (!wants_static(p)) continue;
        PStatic st;
        st.pnode = c;
        int bit = 0;
        for (int v : p.vars)
            if (find_var(p.cover, v) >= 0) {
                st.vars.push_back(v);
                st.pos.push_back(bit);
                st.nb.push_back(hp.vbits[v]);
                bit += hp.vbits[v];
            }
        st.nbits = bit;
        if (bit > 28) FAIL(JTP_EUNSUPPORTED, "static table with %d index bits", bit);
        p.stat = (int)hp.statics.size();
        hp.statics.push_back(st);
    }

    return JTP_OK;
}

int PlanBuilder::arenas() {
    // ---- arena offsets (this rank's real cliques) --------------------------------------------
    // rows 0 and 1 of the arenas are shared: row 0 stays all zero (what rows that do not exist read), row 1 takes
    // the belief stores of such rows (a belief arena is read again by the marginal tasks: its row 0 must stay zero)
    hp.arena_elems = (int64_t)2 << hp.TB;
    hp.host_table_elems = 0;
    auto pack_of = [&](const PNode &p, const std::vector<int> &host_vars) {
        JtPackDesc pd;
        memset(&pd, 0, sizeof pd);
        pd.dev_off = p.arena_off;
        pd.nbits = p.nbits;
        pd.nvars = (int)host_vars.size();
        pd.phys_elems = p.phys_elems;
        pd.low_bits = hp.TB;
        int64_t stride = 1;
        for (int i = pd.nvars - 1; i >= 0; --i) {
            const int v = host_vars[i];
            const int j = find_var(p.vars, v);
            bool whole = false;
            for (size_t g = 0; g < p.group_pos.size(); ++g) whole = whole || p.group_pos[g] == p.pos[j];
            pd.pos[i] = (uint8_t)p.pos[j];
            pd.nb[i] = (uint8_t)p.nb[j];
            pd.card[i] = hp.card[v];
            pd.hstride[i] = stride;
            pd.dstride[i] = p.nb[j] > 0 ? (uint32_t)p.bitw[p.pos[j]] : 0u;
            pd.dmod[i] = whole ? hp.card[v] : 1 << p.nb[j];
            if (p.tmix && p.pos[j] + p.nb[j] <= hp.TB) {          // a mixed-radix digit of the row
                int64_t ts = 1;
                for (int jj = 0; jj < j; ++jj)
                    if (p.pos[jj] + p.nb[jj] <= hp.TB) ts *= hp.card[p.vars[jj]];
                pd.dstride[i] = p.nb[j] > 0 ? (uint32_t)ts : 0u;
                pd.dmod[i] = hp.card[v];
            }
            stride *= hp.card[v];
        }
        pd.row_elems = p.tmix ? p.trow : 0;
        pd.host_elems = stride;
        pd.split_var = -1;
        if (p.tmix && p.tsplit >= 0) {
            const int j = p.tsplit, card = hp.card[p.vars[j]];
            for (int i = 0; i < pd.nvars; ++i)
                if (host_vars[i] == p.vars[j]) pd.split_var = i;
            if (pd.split_var >= 0) {
                int64_t ts = 1;
                for (int jj = 0; jj < j; ++jj)
                    if (p.pos[jj] + p.nb[jj] <= hp.TB) ts *= hp.card[p.vars[jj]];
                pd.split_lb = p.tsplit_lb;
                pd.dstride[pd.split_var] = (uint32_t)ts;
                pd.dmod[pd.split_var] = 1 << p.tsplit_lb;
                pd.split_ds2 = (uint32_t)p.bitw[hp.TB];
                pd.split_mod2 = (card + (1 << p.tsplit_lb) - 1) >> p.tsplit_lb;
            }
        }
        return pd;
    };
    std::map<std::vector<int32_t>, int64_t> map_at;          // thread maps already in the table buffer (unit cliques: mostly one)
    hp.fix_doubles = 0;
    // compact mixed-radix rows are kernels of their own (*_mix<T, true>): all of the plan's mixed-radix cliques, or none
    {
        bool all = hp.tmix, any = false;
        // (EVERY table-keeping clique of such a plan runs in the *_mix kernels - also those whose thread part stayed a bit field,
        //  and they have no list: one of them and the plan keeps one row per step)
        for (int c = 0; c < NP; ++c)
            if (!hp.pn[c].unit) {
                any = true;
                all = all && !hp.pn[c].vmap.empty();
            }
        hp.tmix_compact = all && any;
        if (!hp.tmix_compact)
            for (int c = 0; c < NP; ++c) hp.pn[c].vmap.clear();
    }
    hp.scratch_elems = 0;
    for (int c = 0; c < NP; ++c) {
        PNode &p = hp.pn[c];
        if (!mine(c)) continue;
        if (p.unit) {
            // no table: the passes make up the ones (jt_pass<..., UNIT>); beliefs on demand, into a scratch arena laid out like
            // the table would be (rows 0 and 1 shared, as in the arenas)
            p.arena_off = (int64_t)2 << hp.TB;
            hp.scratch_elems = std::max(hp.scratch_elems, p.arena_off + p.phys_elems);
            hp.has_unit = true;
            double he = 1;
            for (int v : p.cover) he *= hp.card[v];
            if (p.stat >= 0) hp.host_table_elems += he;
        } else {
            p.arena_off = hp.arena_elems;
            hp.arena_elems += p.phys_elems;
            hp.arena_elems = (hp.arena_elems + 255) & ~(int64_t)255;
            double he = 1;
            for (int v : p.vars) he *= hp.card[v];
            hp.host_table_elems += he;
            if (p.real < 0) hp.virtual_fills.push_back({pack_of(p, p.vars)});     // virtual clique: a resident 0/1 table
        }
        if (hp.tmix || p.unit) {
            auto it = p.unit ? map_at.find(p.tmap) : map_at.end();
            if (it != map_at.end()) p.tmap_off = it->second;
            else {
                p.tmap_off = (int64_t)hp.itab.size();
                hp.itab.insert(hp.itab.end(), p.tmap.begin(), p.tmap.end());
                hp.itab.insert(hp.itab.end(), p.vmap.begin(), p.vmap.end());          // (compact mixed-radix rows: PNode::vmap)
                if (p.unit) map_at[p.tmap] = p.tmap_off;
            }
        }
        if (p.stat >= 0) {
            // (16-byte aligned and at least two doubles: the kernels that fill it store 16-byte vectors)
            PStatic &st = hp.statics[p.stat];
            st.off = hp.fix_doubles;
            hp.fix_doubles += std::max<int64_t>((int64_t)1 << st.nbits, 2);
            hp.fix_doubles = (hp.fix_doubles + 1) & ~(int64_t)1;
        }
    }
    hp.pack.assign(N, JtPackDesc());
    hp.stat_pack.assign(N, JtPackDesc());
    for (int c = 0; c < N; ++c) {
        hp.pack[c] = pack_of(hp.pn[c], hp.node_vars[c]);
        memset(&hp.stat_pack[c], 0, sizeof(JtPackDesc));
        const PNode &p = hp.pn[c];
        if (p.stat < 0) continue;
        // host array of the clique (its uncovered axes have length 1) <-> the static table
        const PStatic &st = hp.statics[p.stat];
        JtPackDesc &pd = hp.stat_pack[c];
        pd.dev_off = st.off;
        pd.nbits = st.nbits;
        pd.nvars = (int)hp.node_vars[c].size();
        int64_t stride = 1;
        for (int i = pd.nvars - 1; i >= 0; --i) {
            const int v = hp.node_vars[c][i];
            const int j = find_var(st.vars, v);
            pd.pos[i] = j >= 0 ? (uint8_t)st.pos[j] : 0;
            pd.nb[i] = j >= 0 ? (uint8_t)st.nb[j] : 0;
            pd.card[i] = j >= 0 ? hp.card[v] : 1;
            pd.hstride[i] = j >= 0 ? stride : 0;
            pd.dstride[i] = j >= 0 && st.nb[j] > 0 ? 1u << st.pos[j] : 0u;
            pd.dmod[i] = j >= 0 ? 1 << st.nb[j] : 1;
            if (j >= 0) stride *= hp.card[v];
        }
        pd.host_elems = stride;
        pd.phys_elems = (int64_t)1 << st.nbits;
        pd.low_bits = st.nbits;
        pd.row_elems = 0;
        pd.split_var = -1;
    }
    return JTP_OK;
}

int PlanBuilder::level_work() {
    // ---- per (phase, level, rank) work, to size workgroups ----------------------------------------
    // (per owning rank: a rank's launches hold only its own cliques, and every rank must size every
    // task the same way because the partial-copy counts of the cut messages follow from it)
    for (int ph = 0; ph < 2; ++ph) lvl_elems[ph].assign(hp.n_ranks + 1, std::vector<double>(maxdepth + 1, 0.0));
    for (int c = 0; c < NP; ++c) {
        double e = (double)hp.pn[c].phys_elems;
        // (a mixed-radix row holds trow of the 2^TB entries a workgroup step covers: workgroups are sized by steps,
        //  as if the rows were full - sized by elements they came out at 8 rows of 500 bytes each, 6 x slower)
        if (hp.pn[c].tmix && hp.pn[c].trow > 0) e *= (double)(1 << hp.TB) / (double)hp.pn[c].trow;
        if (c != hp.root) lvl_elems[0][hp.pn[c].owner][hp.pn[c].depth] += e;
        lvl_elems[1][hp.pn[c].owner][hp.pn[c].depth] += e;
    }

    return JTP_OK;
}

int PlanBuilder::make_tasks() {
    // ---- tasks ------------------------------------------------------------------------------
    task_bytes.clear();
    for (int c = 0; c < NP; ++c) {
        PNode &p = hp.pn[c];
        int nch = (int)p.children.size();
        for (int phase = 0; phase < 2; ++phase) {
            if (phase == 0 && c == hp.root) continue;
            if (hp.multiset && phase == 1) {
                // one marginalisation per child: down_k = sum psi * down_parent * prod_{j != k} up_j (no belief
                // table is written; beliefs and marginals are formed on demand, jtp_plan_belief_task)
                for (int j = 0; j < nch; ++j) {
                    JtTask tk;
                    memset(&tk, 0, sizeof tk);
                    tk.pnode = c;
                    tk.psi_off = p.arena_off;
                    tk.bel_off = -1;
                    tk.mode = 0;
                    std::vector<MsgView> ins, outs;
                    if (p.psep >= 0) ins.push_back(make_view(p, hp.ps[p.psep], p.psep, false));
                    for (int i = 0; i < nch; ++i)
                        if (i != j) ins.push_back(make_view(p, hp.ps[hp.pn[p.children[i]].psep], hp.pn[p.children[i]].psep, true));
                    const int ks = hp.pn[p.children[j]].psep;
                    outs.push_back(make_view(p, hp.ps[ks], ks, false));
                    int real_bits = 0;
                    for (int nb : p.nb) real_bits += nb;
                    std::vector<int32_t> itab;
                    int rc = plan_loops(hp, p, tk, itab, p.nbits, real_bits, ins, outs, block_log2_for(phase, p.depth, p.owner), err, JT_SETB_SMALL);
                    if (rc != JTP_OK) rc = plan_loops(hp, p, tk, itab, p.nbits, real_bits, ins, outs, block_log2_for(phase, p.depth, p.owner), err, JT_SETB_LARGE);
                    if (rc != JTP_OK) return rc;
                    tk.itab_off = (int64_t)hp.itab.size();
                    hp.itab.insert(hp.itab.end(), itab.begin(), itab.end());
                    const int ti = (int)hp.tasks.size();
                    p.down_tasks.push_back(ti);
                    hp.ps[ks].dn_task = ti;
                    hp.ps[ks].dn_npart = tk.msg[JT_MAX_IN].npart;
                    hp.task_variant.push_back(JT_K_MULTI_DISTRIBUTE);
                    double b = 0, mb = 0;
                    if (p.real >= 0) b += host_elems(hp.node_vars[p.real]) * esize;
                    for (auto &m : ins) if (hp.ps[m.psep].node >= 0) mb += host_elems(hp.ps[m.psep].vars) * 8;
                    if (hp.ps[ks].node >= 0) mb += host_elems(hp.ps[ks].vars) * 8 * 2;      // down message + separator belief
                    hp.alg_table_bytes += b;
                    hp.alg_msg_bytes += mb;
                    task_bytes.push_back(b + mb);
                    hp.tasks.push_back(tk);
                }
                continue;
            }
            if (p.unit && phase == 1 && !hp.knobs.unit_joint_down) {
                // A unit clique's distribute pass writes no belief: what is left of it is the downward messages, each a
                // marginalisation of its own (the parent's message, the static table and the SIBLINGS' upward messages in, one
                // message out - mode 0, like the tasks of a multi-set plan) instead of one pass that folds every child's sums on
                // every row: a row of the joint pass costs 3.5 x a row of such a task (163 against 50 vector instructions, the
                // epilogue of the first message on every row), and a leaf has no task at all.  Config 3: distribute 7.4 -> ms below.
                for (int j = 0; j < nch; ++j) {
                    JtTask tk;
                    memset(&tk, 0, sizeof tk);
                    tk.pnode = c;
                    tk.mode = 0;
                    tk.unit = 1;
                    tk.bel_off = -1;
                    std::vector<MsgView> ins, outs;
                    if (p.psep >= 0) ins.push_back(make_view(p, hp.ps[p.psep], p.psep, false));
                    if (p.stat >= 0) ins.push_back(make_view(p, hp.statics[p.stat]));
                    for (int i = 0; i < nch; ++i)
                        if (i != j) ins.push_back(make_view(p, hp.ps[hp.pn[p.children[i]].psep], hp.pn[p.children[i]].psep, true));
                    const int ks = hp.pn[p.children[j]].psep;
                    outs.push_back(make_view(p, hp.ps[ks], ks, false));
                    int real_bits = 0;
                    for (int nb : p.nb) real_bits += nb;
                    std::vector<int32_t> itab;
                    const double share = std::min(1.0, (double)p.phys_elems / std::max(1.0, lvl_elems[phase][p.owner][p.depth]));
                    const int blg = block_log2_for(0, p.depth, p.owner, p.layout != 4);
                    int rc = plan_loops(hp, p, tk, itab, p.nbits, real_bits, ins, outs, hp.block_log2 > 0 ? std::max(hp.block_log2, hp.TB) : blg, err, 0, share);
                    if (rc != JTP_OK) return rc;
                    tk.itab_off = (int64_t)hp.itab.size();
                    hp.itab.insert(hp.itab.end(), itab.begin(), itab.end());
                    const int ti = (int)hp.tasks.size();
                    p.down_tasks.push_back(ti);
                    hp.ps[ks].dn_task = ti;
                    hp.ps[ks].dn_npart = tk.msg[JT_MAX_IN].npart;
                    hp.task_variant.push_back(JT_K_DISTRIBUTE_LEVEL);
                    // (algorithmic bytes of the clique's downward step, counted once: the static table, the parent's message and every
                    //  child's upward message read - booked on the first task - and each child's downward message and separator belief)
                    double b = 0, mb = 0;
                    if (j == 0) {
                        if (p.stat >= 0) b += host_elems(p.cover) * 8;
                        if (p.psep >= 0 && hp.ps[p.psep].node >= 0) mb += host_elems(hp.ps[p.psep].vars) * 8;
                        for (int k : p.children) if (hp.ps[hp.pn[k].psep].node >= 0) mb += host_elems(hp.ps[hp.pn[k].psep].vars) * 8;
                    }
                    if (hp.ps[ks].node >= 0) mb += host_elems(hp.ps[ks].vars) * 8 * 2;
                    double full = 0;
                    if (p.real >= 0 && j == 0) full += host_elems(hp.node_vars[p.real]) * esize * 2;
                    if (mine(c)) hp.alg_bytes_full += full + mb;
                    task_bytes.push_back(b + mb);
                    hp.tasks.push_back(tk);
                }
                continue;
            }
            JtTask tk;
            memset(&tk, 0, sizeof tk);
            tk.pnode = c;
            tk.mode = phase;
            tk.psi_off = mine(c) && !p.unit ? p.arena_off : 0;          // other ranks' tasks are not executed here
            tk.bel_off = phase == 1 && !p.unit ? (mine(c) ? p.arena_off : 0) : -1;   // virtual cliques that keep a table too (scratch)
            tk.unit = p.unit ? 1 : 0;
            std::vector<MsgView> ins, outs;
            // (distribute: the inputs that are not children come first - the parent's message, the clique's static table)
            if (phase == 1 && p.psep >= 0) ins.push_back(make_view(p, hp.ps[p.psep], p.psep, false));
            if (p.stat >= 0) ins.push_back(make_view(p, hp.statics[p.stat]));
            for (int k : p.children) ins.push_back(make_view(p, hp.ps[hp.pn[k].psep], hp.pn[k].psep, true));
            if ((int)ins.size() > JT_MAX_IN) FAIL(JTP_EUNSUPPORTED, "internal: clique %d has %zu incoming tables", p.real, ins.size());
            if (phase == 0) outs.push_back(make_view(p, hp.ps[p.psep], p.psep, true));
            else for (int k : p.children) outs.push_back(make_view(p, hp.ps[hp.pn[k].psep], hp.pn[k].psep, false));
            int real_bits = 0;
            for (int nb : p.nb) real_bits += nb;
            std::vector<int32_t> itab;
            const double steps_scale = p.tmix && p.trow > 0 ? (double)(1 << hp.TB) / (double)p.trow : 1.0;     // (as in level_work)
            const double share = std::min(1.0, (double)p.phys_elems * steps_scale / std::max(1.0, lvl_elems[phase][p.owner][p.depth]));
            // (mixed-radix rows are a quarter of a full row or less, and whole variables - 2 or 3 bits - go in or out of the
            //  loops together: such cliques may always use the 64 rows a workgroup can hold)
            const int blg = p.tmix ? hp.TB + JT_MAX_ITER_LOG2 : block_log2_for(phase, p.depth, p.owner, p.layout != 4);
            int rc = plan_loops(hp, p, tk, itab, p.nbits, real_bits, ins, outs, hp.block_log2 > 0 ? std::max(hp.block_log2, hp.TB) : blg, err,
                                hp.multiset ? JT_SETB_SMALL : 0, share);
            if (rc != JTP_OK && hp.multiset)
                rc = plan_loops(hp, p, tk, itab, p.nbits, real_bits, ins, outs, block_log2_for(phase, p.depth, p.owner), err, JT_SETB_LARGE);
            if (rc != JTP_OK) return rc;
            tk.itab_off = (int64_t)hp.itab.size();
            hp.itab.insert(hp.itab.end(), itab.begin(), itab.end());
            int ti = (int)hp.tasks.size();
            if (phase == 0) {
                p.collect_task = ti;
                hp.ps[p.psep].up_npart = tk.msg[JT_MAX_IN].npart;
                hp.task_variant.push_back(hp.multiset ? JT_K_MULTI_COLLECT : JT_K_COLLECT0 + nch);
            } else {
                p.distribute_task = ti;
                for (int j = 0; j < nch; ++j) hp.ps[hp.pn[p.children[j]].psep].dn_npart = tk.msg[JT_MAX_IN + j].npart;
                hp.task_variant.push_back(JT_K_DIST_P0C0 + 4 * (p.psep >= 0 ? 1 : 0) + nch);
            }
            // algorithmic bytes: clique table read (+ belief written), messages.  A unit clique counts what its
            // potential IS - the static table (doubles), read once per pass, and no belief; `full` counts every clique at its
            // full shape in the storage type, read and belief written (8d to the letter)
            double b = 0, full = 0;
            if (p.real >= 0) full += host_elems(hp.node_vars[p.real]) * esize * (phase == 1 ? 2 : 1);
            if (p.real >= 0 && !p.unit) b += host_elems(hp.node_vars[p.real]) * esize * (phase == 1 ? 2 : 1);
            if (p.stat >= 0) b += host_elems(p.cover) * 8;
            double mb = 0;
            for (auto &m : ins) if (m.psep >= 0 && hp.ps[m.psep].node >= 0) mb += host_elems(hp.ps[m.psep].vars) * 8;
            for (auto &m : outs) if (hp.ps[m.psep].node >= 0) mb += host_elems(hp.ps[m.psep].vars) * 8 * (phase == 1 ? 2 : 1);
            b += mb;
            if (mine(c)) hp.alg_bytes_full += full + mb;
            task_bytes.push_back(b);
            if (hp.multiset) {
                const double tb = p.real >= 0 ? host_elems(hp.node_vars[p.real]) * esize : 0.0;
                hp.alg_table_bytes += tb;
                hp.alg_msg_bytes += b - tb;
            }
            hp.tasks.push_back(tk);
        }
    }

    return JTP_OK;
}

int PlanBuilder::messages() {
    // Settle in place (jt_msg_settle): plans whose cliques mostly sit on latency-bound levels - a clique or two - (chains),
    // and (round 3) the tasks of any plan's NARROW levels - the top of a tree, a rank's share of one - where the hand-over
    // between dependent levels is what the level costs (a rank's share of config 4 at 8 ranks: 201.5 -> 195 us, config 4
    // itself +-0; A/B on one box).  Not on streaming levels: there the re-loads of thousands of waiting workgroups cost more
    // than the round trips they save (round 2).  Tried on top of it and dropped: a two-stage wait - one lane polls an entry
    // the PRODUCER waits for, then every thread spins on its own entries - so that a message is taken one load after it
    // becomes visible: config 2 5.47 -> 5.87 ms, the rank share 195 -> 199 us (the spinning threads of a whole level cost
    // the producers more than the saved round trip).
    for (JtTask &tk : hp.tasks) {
        if (tk.kind != 0) continue;
        const PNode &p = hp.pn[tk.pnode];
        const int phase = ((int)(&tk - hp.tasks.data()) == p.collect_task) ? 0 : 1;
        tk.settle = (hp.chain_plan || lvl_elems[phase][p.owner][p.depth] <= hp.knobs.settle_level_elems) ? 1 : 0;
    }
    // ---- message arena ----------------------------------------------------------------------
    // A message written as many partial copies costs every consuming workgroup (sub-box x copies)
    // loads before it can start, on the critical path of the small levels near the root.  From
    // `red_min` copies on, a reduce task behind the producer sums them once and consumers read the sum.
    // (multi-set plans: 2 and 8 measured within 4 % of each other on the width-20 tree, 8 ahead)
    // Single-set plans (round 2, with eight entry loads in flight per staging thread): only messages of 64 copies get a
    // reduce task - config 3 13.3 -> 12.8 ms, config 4 within noise for any threshold from 8 up.
    // Chains keep 8: there a reduce task between two levels beats every consumer summing eight copies (config 2 6.5 against 6.85 ms).
    const int red_min = hp.knobs.reduce_min >= 0 ? hp.knobs.reduce_min : (hp.multiset || hp.chain_plan ? 8 : 64);
    hp.msg_doubles = 0;
    for (auto &s : hp.ps) {
        if (!mine(s.child) && !mine(s.parent)) continue;
        int64_t n = (int64_t)1 << s.nbits;
        s.up_off = s.up_roff = hp.msg_doubles;
        hp.msg_doubles += n * s.up_npart;
        s.dn_off = s.dn_roff = hp.msg_doubles;
        hp.msg_doubles += n * s.dn_npart;
        s.up_rnpart = s.up_npart;
        s.dn_rnpart = s.dn_npart;
        for (int up = 0; up < 2; ++up) {
            const int npart = up ? s.up_npart : s.dn_npart;
            if (red_min <= 0 || npart < red_min) continue;
            (up ? s.up_roff : s.dn_roff) = hp.msg_doubles;
            (up ? s.up_rnpart : s.dn_rnpart) = 1;
            hp.msg_doubles += n;
            JtTask rt;
            memset(&rt, 0, sizeof rt);
            rt.kind = 1;
            rt.pnode = up ? s.child : s.parent;              // the producer: its rank runs the reduction
            rt.nbits = s.nbits;
            rt.n_in = rt.n_out = 1;
            rt.bel_off = -1;
            rt.msg[0].off = up ? s.up_off : s.dn_off;
            rt.msg[0].npart = npart;
            rt.msg[0].pstride = (int32_t)n;
            rt.msg[0].same_launch = 1;
            rt.msg[JT_MAX_IN].off = up ? s.up_roff : s.dn_roff;
            rt.msg[JT_MAX_IN].npart = 1;
            rt.msg[JT_MAX_IN].pstride = (int32_t)n;
            while (((int64_t)JT_REDUCE_ENTRIES << rt.nF) < n) {
                rt.f_x[rt.nF] = (uint32_t)JT_REDUCE_ENTRIES << rt.nF;
                rt.nF++;
            }
            (up ? s.up_red_task : s.dn_red_task) = (int)hp.tasks.size();
            hp.tasks.push_back(rt);
            hp.task_variant.push_back(JT_K_REDUCE_LEVEL);
            task_bytes.push_back(0.0);
        }
        hp.msg_doubles = (hp.msg_doubles + 1) & ~(int64_t)1;
    }
    // who writes what each task reads (finish() turns it into JtMsg::same_launch once the launches are known)
    hp.task_producers.assign(hp.tasks.size(), std::vector<int>());
    auto up_producer = [&](const PSep &sp) { return sp.up_red_task >= 0 ? sp.up_red_task : hp.pn[sp.child].collect_task; };
    auto dn_producer = [&](const PSep &sp) {
        if (sp.dn_red_task >= 0) return sp.dn_red_task;
        return sp.dn_task >= 0 ? sp.dn_task : hp.pn[sp.parent].distribute_task;        // (multi-set plans and unit cliques: a task per child)
    };
    for (const PSep &sp : hp.ps) {
        if (sp.up_red_task >= 0) hp.task_producers[sp.up_red_task] = {hp.pn[sp.child].collect_task};
        if (sp.dn_red_task >= 0) hp.task_producers[sp.dn_red_task] = {sp.dn_task >= 0 ? sp.dn_task : hp.pn[sp.parent].distribute_task};
    }
    for (size_t t = 0; t < hp.tasks.size(); ++t) {
        JtTask &tk = hp.tasks[t];
        if (tk.kind != 0) continue;
        const PNode &p = hp.pn[tk.pnode];
        {
            std::vector<int> &prod = hp.task_producers[t];
            const bool collect_task = (int)t == p.collect_task;
            size_t skip = p.children.size();                     // multi-set plans: the child a downward-message task serves
            if (!p.down_tasks.empty() && !collect_task)
                for (size_t j = 0; j < p.down_tasks.size(); ++j)
                    if (p.down_tasks[j] == (int)t) skip = j;
            if (!collect_task && p.psep >= 0) prod.push_back(dn_producer(hp.ps[p.psep]));
            if (p.stat >= 0) prod.push_back(-1);                 // the static table: nobody's product
            for (size_t i = 0; i < p.children.size(); ++i)
                if (i != skip) prod.push_back(up_producer(hp.ps[hp.pn[p.children[i]].psep]));
        }
        if (!p.down_tasks.empty() && (int)t != p.collect_task) {          // a downward-message task: which child?
            size_t j = 0;
            while (j < p.down_tasks.size() && p.down_tasks[j] != (int)t) ++j;
            int k = 0;
            if (p.psep >= 0) {
                tk.msg[k].off = hp.ps[p.psep].dn_roff;
                tk.msg[k].npart = hp.ps[p.psep].dn_rnpart;
                tk.msg[k].same_launch = 1;                       // formed by the parent's task in this phase
                ++k;
            }
            if (p.stat >= 0) {
                tk.msg[k].off = std::max<int64_t>(hp.statics[p.stat].off, 0);
                tk.msg[k].npart = 1;
                tk.msg[k].same_launch = 0;
                tk.msg[k].fixed = 1;
                ++k;
            }
            for (size_t i = 0; i < p.children.size(); ++i) {
                if (i == j) continue;
                const PSep &sp = hp.ps[hp.pn[p.children[i]].psep];
                tk.msg[k].off = sp.up_roff;
                tk.msg[k].npart = sp.up_rnpart;
                tk.msg[k].same_launch = 0;                       // finished by the collect launch
                tk.msg[k].src_task = hp.pn[p.children[i]].collect_task;
                ++k;
            }
            tk.msg[JT_MAX_IN].off = hp.ps[hp.pn[p.children[j]].psep].dn_off;
            continue;
        }
        bool collect = (int)t == p.collect_task;
        int k = 0;
        // same_launch: the producer runs in the same dataflow launch as this consumer (same phase, same
        // rank).  Upward messages read during distribute were finished by the collect launch, messages
        // of other ranks arrive by an exchange between launches: those are read with ordinary loads.
        if (!collect && p.psep >= 0) {
            tk.msg[k].off = hp.ps[p.psep].dn_roff;
            tk.msg[k].npart = hp.ps[p.psep].dn_rnpart;
            // (a replicated parent forms the message on this rank, in this phase, with no exchange in between)
            tk.msg[k].same_launch = hp.pn[p.parent].owner == p.owner || hp.pn[p.parent].owner == ALL;
            ++k;
        }
        if (p.stat >= 0) {
            tk.msg[k].off = std::max<int64_t>(hp.statics[p.stat].off, 0);
            tk.msg[k].npart = 1;
            tk.msg[k].same_launch = 0;
            tk.msg[k].fixed = 1;
            ++k;
        }
        for (int ch : p.children) {
            const PSep &s = hp.ps[hp.pn[ch].psep];
            tk.msg[k].off = s.up_roff;
            tk.msg[k].npart = s.up_rnpart;
            tk.msg[k].same_launch = collect && hp.pn[ch].owner == p.owner;
            tk.msg[k].src_task = hp.pn[ch].collect_task;
            ++k;
        }
        if (collect) tk.msg[JT_MAX_IN].off = hp.ps[p.psep].up_off;
        else
            for (size_t j = 0; j < p.children.size(); ++j) tk.msg[JT_MAX_IN + j].off = hp.ps[hp.pn[p.children[j]].psep].dn_off;
    }
    return fold_marginals();
}

// Marginals named at plan creation (jtp_tree_desc.fold_*; round 6).  `JunctionTree.propagate` returns factor marginals only
// (junctiontree/junctiontree.py:264-274, 327-331), and a clique that keeps no table has no belief to take them from: the read-out forms
// psi x (every incoming table) again, per request list, after the propagate - on a tree of such cliques a third of a propagate's work,
// run behind it.  Here the same tasks (jtp_plan_marginal_task: up to three requests of one clique per pass) become tasks OF the
// propagate: on the level of the clique's downward messages - their inputs are the final messages, the parent's produced one level up
// in this launch - where the dependent levels leave slots idle, writing partial copies into a region of the message arena that
// jtp_get_marginals unpacks.  Only the lean pass runs them (jt_unit_lean<..., NOUT>): single-set plans of one rank, no mixed-radix
// rows, not a chain (whose distribute kernel is built without them); the engine falls back to the read-out wherever they did not run.
int PlanBuilder::fold_marginals() {
    if (hp.folded.empty() || hp.multiset || hp.n_ranks != 1 || hp.tmix || hp.chain_plan || hp.knobs.no_fold || hp.knobs.no_lean || (hp.knobs.debug & ~2)) return JTP_OK;
    const int n = (int)hp.folded.size();
    if (hp.knobs.fold < 0) {
        // Where the folded tasks pay (measured, profiles/r06_ab_fold_placement.txt): their workgroups are free where the levels of the
        // distribute phase leave resident slots of the chip idle (the column-sweep tree of config 3: every level under 1 024 workgroups,
        // marginals 3.7 -> 0.24 ms for 1.4 ms more propagate), and cost their own work where the levels fill the chip (the min-fill tree:
        // 6 % of the levels under 1 024; +0.28 ms of propagate for 0.21 ms less read-out, wherever in the launch they are put) - there the
        // read-out's launch, which waits for nobody, does the same work no slower.  So: fold where at least half of the distribute
        // levels are under `fold_slots` workgroups (256 CUs x 4).  JTP_FOLD=1: wherever possible; 0: nowhere.
        std::vector<long> level_blocks(maxdepth + 1, 0);
        for (int c = 0; c < NP; ++c) {
            const PNode &p = hp.pn[c];
            if (!mine(c)) continue;
            if (!p.down_tasks.empty()) {
                for (int t : p.down_tasks) level_blocks[p.depth] += 1L << hp.tasks[t].nF;
            } else if (p.distribute_task >= 0) level_blocks[p.depth] += 1L << hp.tasks[p.distribute_task].nF;
        }
        int levels = 0, idle = 0;
        for (long b : level_blocks)
            if (b > 0) ++levels, idle += b < hp.knobs.fold_slots ? 1 : 0;
        if (2 * idle < levels) return JTP_OK;
    }
    std::vector<std::vector<int>> groups;
    {
        std::map<int, int> open;                             // clique -> its group that still has room
        for (int i = 0; i < n; ++i) {
            const int c = hp.fold_cliques[i];
            if (!hp.pn[c].unit || hp.pn[c].real < 0 || !mine(c)) continue;      // (a clique with a belief table: jt_marginals reads that)
            auto it = open.find(c);
            if (it == open.end() || (int)groups[it->second].size() >= hp.knobs.marg_group) {
                open[c] = (int)groups.size();
                groups.push_back(std::vector<int>());
            }
            groups[open[c]].push_back(i);
        }
    }
    for (const std::vector<int> &grp : groups) {
        const int c = hp.fold_cliques[grp[0]];
        const PNode &p = hp.pn[c];
        std::vector<std::vector<int>> ovs;
        bool ok = true;
        for (int i : grp) {
            std::vector<int> ov(hp.fold_var_ids.begin() + hp.fold_var_off[i], hp.fold_var_ids.begin() + hp.fold_var_off[i + 1]);
            for (size_t a = 0; a < ov.size(); ++a) {
                ok = ok && find_var(p.vars, ov[a]) >= 0;
                for (size_t b = 0; b < a; ++b) ok = ok && ov[a] != ov[b];
            }
            ovs.push_back(ov);
        }
        if (!ok) continue;                                   // (a malformed request: jtp_get_marginals will say so)
        JtTask tk;
        std::vector<int32_t> tab;
        std::vector<int> out_bits, npart;
        std::vector<JtBlock> blk;
        std::string err2;
        if (jtp_plan_marginal_task(hp, c, ovs, tk, tab, out_bits, npart, blk, err2, true) != JTP_OK) continue;
        if (tk.n_in > JT_MAX_IN || tk.n_out > JT_MAX_OUT || tk.vgroups) continue;
        tk.fold = 1;
        tk.debug = hp.knobs.debug;
        tk.itab_off = (int64_t)hp.itab.size();
        if (tk.tmap_off >= 0) tk.tmap_off += tk.itab_off;
        hp.itab.insert(hp.itab.end(), tab.begin(), tab.end());
        const int t = (int)hp.tasks.size();
        for (size_t j = 0; j < grp.size(); ++j) {
            HostPlan::FoldReq &fr = hp.folded[grp[j]];
            fr.task = t, fr.j = (int)j, fr.npart = npart[j], fr.out_bits = out_bits[j], fr.off = hp.msg_doubles;
            tk.msg[JT_MAX_IN + j].off = hp.msg_doubles;
            hp.msg_doubles += ((int64_t)1 << out_bits[j]) * npart[j];
            hp.msg_doubles = (hp.msg_doubles + 1) & ~(int64_t)1;
        }
        // who writes what it reads (finish(): JtMsg::same_launch) - the order of neighbour_inputs
        std::vector<int> prod;
        if (p.psep >= 0) {
            const PSep &sp = hp.ps[p.psep];
            prod.push_back(sp.dn_red_task >= 0 ? sp.dn_red_task : (sp.dn_task >= 0 ? sp.dn_task : hp.pn[sp.parent].distribute_task));
        }
        if (p.stat >= 0) prod.push_back(-1);
        for (int ch : p.children) {
            const PSep &sp = hp.ps[hp.pn[ch].psep];
            prod.push_back(sp.up_red_task >= 0 ? sp.up_red_task : hp.pn[sp.child].collect_task);
        }
        hp.tasks.push_back(tk);
        hp.task_variant.push_back(JT_K_DISTRIBUTE_LEVEL);
        task_bytes.push_back(0.0);
        hp.task_producers.push_back(prod);
        hp.pn[c].fold_tasks.push_back(t);
    }
    return JTP_OK;
}

int PlanBuilder::schedule() {
    // Which passes load their table rows with the default cache policy (JtTask::keep_rows; everything else non-temporal).
    // The levels nearest the root are read LAST by collect and FIRST by distribute: while the tables of levels 0..d (this
    // rank's) stay below knobs.keep_rows_mb, both passes over them keep their rows in the 256 MiB Infinity Cache and the
    // second finds them there instead of in HBM.  A plan whose tables fit altogether keeps every row (a rank's share of
    // config 4 at 8 ranks: 152 MiB).
    if (!hp.multiset && hp.knobs.keep_rows_mb > 0) {
        std::vector<double> level_bytes(maxdepth + 1, 0.0);
        for (int c = 0; c < NP; ++c)
            if (mine(c) && !hp.pn[c].unit) level_bytes[hp.pn[c].depth] += (double)hp.pn[c].phys_elems * esize;
        double cum = 0;
        int keep_depth = -1;
        for (int d = 0; d <= maxdepth; ++d) {
            cum += level_bytes[d];
            if (cum > hp.knobs.keep_rows_mb * 1048576.0) break;
            keep_depth = d;
        }
        for (JtTask &tk : hp.tasks)
            if (tk.kind == 0 && !tk.unit && hp.pn[tk.pnode].depth <= keep_depth) tk.keep_rows = 1;
    }
    // ---- launches, blocks, exchange schedule -----------------------------------------------------
    hp.alg_bytes = 0;
    hp.max_lds = 0;
    std::vector<CommOp> pending;            // comm ops waiting to be grouped before the next launch
    auto flush_comm = [&]() {
        if (pending.empty()) return;
        Step st;
        st.kind = 1;
        st.first = (int)hp.comm.size();
        st.count = (int)pending.size();
        for (auto &op : pending) hp.comm.push_back(op);
        hp.steps.push_back(st);
        pending.clear();
    };
    auto comm_op = [&](int send, int psep, int up, int peer) {
        const PSep &s = hp.ps[psep];
        CommOp op;
        op.send = send;
        op.psep = psep;
        op.up = up;
        op.peer = peer;
        op.off = up ? s.up_roff : s.dn_roff;                 // (the sum, when the producer's rank reduces)
        op.count = ((int64_t)1 << s.nbits) * (up ? s.up_rnpart : s.dn_rnpart);
        pending.push_back(op);
    };
    // Multi-set plans: a launch's block list is padded to a multiple of eight records with records that start no work
    // (JT_BLOCK_NULL).  jt_multi_flow hands runs of eight records to the groups of evidence sets in turn; with active lists (round 6) a
    // workgroup of one group waits for entries another group's workgroup writes, and with every launch - every tree level -
    // starting on a multiple of eight, that producer has the lower blockIdx whatever its group.
    auto pad_launch = [&](const Launch &L) {
        if (!hp.multiset) return;
        while ((hp.blocks.size() - (size_t)L.blk_off) % 8) {
            JtBlock nb;
            memset(&nb, 0, sizeof nb);
            nb.task = L.tasks.empty() ? 0u : (uint32_t)L.tasks[0];
            nb.flags = JT_BLOCK_NULL;
            hp.blocks.push_back(nb);
            hp.block_chunk.push_back(0xffffffffu);
        }
    };
    auto by_level = [&](int level) {
        std::vector<int> v;
        for (int c = 0; c < NP; ++c) if (hp.pn[c].depth == level) v.push_back(c);
        return v;
    };
    auto emit_launches = [&](int phase, int level) {
        std::map<int, std::vector<int>> groups;
        for (int c : by_level(level)) {
            const PNode &p = hp.pn[c];
            if (!mine(c)) continue;
            if (hp.multiset) {
                if (phase == 0 && p.collect_task >= 0) groups[JT_K_MULTI_COLLECT].push_back(p.collect_task);
                if (phase == 1) for (int t : p.down_tasks) groups[JT_K_MULTI_DISTRIBUTE].push_back(t);
                continue;
            }
            if (phase == 1)
                for (int t : p.fold_tasks) groups[JT_K_DISTRIBUTE_LEVEL].push_back(t);      // (marginals folded into the propagate)
            if (phase == 1 && !p.down_tasks.empty()) {       // (a unit clique: a task per downward message)
                for (int t : p.down_tasks) groups[JT_K_DISTRIBUTE_LEVEL].push_back(t);
                continue;
            }
            int t = phase == 0 ? p.collect_task : p.distribute_task;
            if (t < 0) continue;
            int key = hp.task_variant[t];
            // (per-shape launches are a profiling aid of plans whose cliques all keep tables: unit tasks have shapes of their own)
            if (!(hp.flags & JTP_SPLIT_VARIANTS) || hp.has_unit) key = phase == 0 ? JT_K_COLLECT_LEVEL : JT_K_DISTRIBUTE_LEVEL;
            groups[key].push_back(t);
        }
        if (!groups.empty()) flush_comm();
        for (auto &g : groups) {
            Launch L;
            L.phase = phase;
            L.level = level;
            L.variant = g.first;
            L.tasks = g.second;
            if (hp.knobs.longest_first) {
                // Longest workgroups first: a level is over when its LAST workgroup is, and the workgroups of one task all
                // take about as long as each other.  Multi-set plans: the tasks that cannot sum a vector's four elements
                // before the message product (JtTask::esum == 0, 8 % of them on the width-20 tree) run 4-5 x longer per
                // row - sixteen such workgroups, started two thirds into their level, ended 140 us after everybody else.
                auto weight = [&](int t) { return (long)hp.tasks[t].total * (hp.multiset && !hp.tasks[t].esum ? 4 : 1); };
                std::stable_sort(L.tasks.begin(), L.tasks.end(), [&](int a, int b) { return weight(a) > weight(b); });
            }
            L.blk_off = (int64_t)hp.blocks.size();
            for (int t : L.tasks) {
                const JtTask &tk = hp.tasks[t];
                for (uint32_t f = 0; f < (1u << tk.nF); ++f) {
                    const JtBlock b = jtp_make_block(hp, tk, (uint32_t)t, f);
                    if ((b.flags & JT_BLOCK_INVALID) && !hp.knobs.keep_invalid) {
                        // (a chunk that does not exist: zeros, written once per arena - HostPlan::init_blocks)
                        hp.init_blocks[tk.mode ? 1 : 0].push_back(b);
                        hp.init_chunk[tk.mode ? 1 : 0].push_back(f);
                        continue;
                    }
                    hp.blocks.push_back(b);
                    hp.block_chunk.push_back(f);
                }
                L.lds_bytes = std::max(L.lds_bytes, tk.lds_bytes);
                L.alg_bytes += task_bytes[t];
                for (int k = 0; k < tk.n_in; ++k) hp.staging_bytes += (double)(1u << tk.nF) * (8.0 * (1 << tk.msg[k].nfree)) * tk.msg[k].npart;
                if (!tk.unit) hp.table_bytes += (double)hp.pn[tk.pnode].phys_elems * esize * (phase == 1 && !hp.multiset ? 2 : 1);
            }
            pad_launch(L);
            L.nblocks = (int)(hp.blocks.size() - L.blk_off);
            hp.max_lds = std::max(hp.max_lds, L.lds_bytes);
            hp.alg_bytes += L.alg_bytes;
            Step st;
            st.kind = 0;
            st.first = (int)hp.launches.size();
            st.count = 1;
            hp.launches.push_back(L);
            hp.steps.push_back(st);
        }
    };
    // reduce tasks of the messages one level has just produced (its own launch when launching per level)
    auto emit_reduce = [&](int phase, int level) {
        std::vector<int> tasks;
        for (int c : by_level(level)) {
            const PNode &p = hp.pn[c];
            if (!mine(c)) continue;
            if (phase == 0) {
                if (p.psep >= 0 && hp.ps[p.psep].up_red_task >= 0) tasks.push_back(hp.ps[p.psep].up_red_task);
            } else {
                for (int k : p.children)
                    if (hp.ps[hp.pn[k].psep].dn_red_task >= 0) tasks.push_back(hp.ps[hp.pn[k].psep].dn_red_task);
            }
        }
        if (tasks.empty()) return;
        Launch L;
        L.phase = phase;
        L.level = level;
        L.variant = JT_K_REDUCE_LEVEL;
        L.tasks = tasks;
        L.blk_off = (int64_t)hp.blocks.size();
        for (int t : tasks)
            for (uint32_t f = 0; f < (1u << hp.tasks[t].nF); ++f) {
                hp.blocks.push_back(jtp_make_block(hp, hp.tasks[t], (uint32_t)t, f));
                hp.block_chunk.push_back(f);
            }
        pad_launch(L);
        L.nblocks = (int)(hp.blocks.size() - L.blk_off);
        Step st;
        st.kind = 0;
        st.first = (int)hp.launches.size();
        st.count = 1;
        hp.launches.push_back(L);
        hp.steps.push_back(st);
    };
    // Exchange order: ncclSend/ncclRecv (and every transport standing in for them) pair the operations
    // between two ranks in ISSUE order, so both sides of a cut must enumerate the cut edges of one level
    // in the same order whatever the numbering of the cliques: always by the CHILD clique of the edge
    // (ascending), never by the parent's position.
    // A cut edge joins cliques of different owners.  child (rank r) -> replicated parent: r sends the upward message
    // to EVERY other rank, the downward message needs no exchange (each rank's replica forms it; only r uses it).
    auto cut_children = [&](int child_level) {
        std::vector<int> v;                                    // children (ascending) of cut edges at this level
        for (int k : by_level(child_level)) {
            const PNode &ch = hp.pn[k];
            if (ch.parent >= 0 && hp.pn[ch.parent].owner != ch.owner) v.push_back(k);
        }
        return v;
    };
    for (int level = maxdepth; level >= 0; --level) {          // collect
        for (int k : cut_children(level + 1)) {                // receive what this level consumes
            const int po = hp.pn[hp.pn[k].parent].owner;
            if (hp.pn[k].owner != hp.rank && (po == hp.rank || po == ALL)) comm_op(0, hp.pn[k].psep, 1, hp.pn[k].owner);
        }
        if (level >= 1) {
            emit_launches(0, level);
            emit_reduce(0, level);
        }
        for (int c : cut_children(level)) {                    // send what this level produced
            if (hp.pn[c].owner != hp.rank) continue;
            const int po = hp.pn[hp.pn[c].parent].owner;
            if (po == ALL) {
                for (int peer = 0; peer < hp.n_ranks; ++peer)
                    if (peer != hp.rank) comm_op(1, hp.pn[c].psep, 1, peer);
            } else comm_op(1, hp.pn[c].psep, 1, po);
        }
    }
    for (int level = 0; level <= maxdepth; ++level) {          // distribute
        for (int c : cut_children(level)) {
            const int po = hp.pn[hp.pn[c].parent].owner;
            if (hp.pn[c].owner == hp.rank && po != ALL) comm_op(0, hp.pn[c].psep, 0, po);
        }
        emit_launches(1, level);
        emit_reduce(1, level);
        for (int k : cut_children(level + 1)) {
            const int po = hp.pn[hp.pn[k].parent].owner;
            if (po == hp.rank && hp.pn[k].owner != hp.rank) comm_op(1, hp.pn[k].psep, 0, hp.pn[k].owner);
        }
    }
    flush_comm();
    return JTP_OK;
}

int PlanBuilder::finish() {
    // ---- dataflow schedule: runs of launches of one phase become one segment --------------------
    for (const Step &st : hp.steps) {
        if (st.kind == 1) {
            hp.flow_steps.push_back(st);
            continue;
        }
        const Launch &L = hp.launches[st.first];
        const bool extend = !hp.flow_steps.empty() && hp.flow_steps.back().kind == 0 && hp.segments.back().phase == L.phase;
        if (!extend) {
            Segment sg;
            sg.phase = L.phase;
            sg.first_launch = st.first;
            sg.blk_off = L.blk_off;
            sg.ticket_idx = JT_SYNC_HDR + (int)hp.segments.size();
            Step fs;
            fs.kind = 0;
            fs.first = (int)hp.segments.size();
            fs.count = 1;
            hp.segments.push_back(sg);
            hp.flow_steps.push_back(fs);
        }
        Segment &sg = hp.segments.back();
        sg.n_launch++;
        sg.nblocks += L.nblocks;
        sg.lds_bytes = std::max(sg.lds_bytes, L.lds_bytes);
    }
    // Both phases in one launch (jt_propagate_flow): where the distribute segment follows the collect segment directly (no
    // exchange in between) and the messages are small beside the tables - every message of a merged launch is read
    // through to memory, which costs where staging is a large share of the traffic (config 3) and buys nothing on chains.
    {
        const bool merge = hp.knobs.merge_phases == 1 ||
                           (hp.knobs.merge_phases < 0 && !hp.multiset && !hp.chain_plan && !hp.tmix &&
                            // (plans of mostly unit cliques: no tables to speak of - one launch.  Measured the same as two launches once
                            //  the distribute segment ran the two-phase kernel, whose build is the faster one: jtp_engine.hip, get_flow)
                            (hp.staging_bytes * 8.0 <= hp.table_bytes || hp.unit_dominated));
        if (merge && !hp.multiset && !hp.tmix) {
            std::vector<Segment> segs;
            std::vector<Step> fsteps;
            for (const Step &st : hp.flow_steps) {
                if (st.kind == 0 && !fsteps.empty() && fsteps.back().kind == 0 && segs.back().phase == 0 && hp.segments[st.first].phase == 1) {
                    const Segment &b = hp.segments[st.first];
                    Segment &a = segs.back();
                    a.phase = 2;
                    a.n_launch += b.n_launch;
                    a.nblocks += b.nblocks;
                    a.lds_bytes = std::max(a.lds_bytes, b.lds_bytes);
                    continue;
                }
                Step fs = st;
                if (st.kind == 0) {
                    fs.first = (int)segs.size();
                    segs.push_back(hp.segments[st.first]);
                    segs.back().ticket_idx = JT_SYNC_HDR + (int)segs.size() - 1;
                }
                fsteps.push_back(fs);
            }
            hp.segments = segs;
            hp.flow_steps = fsteps;
        }
    }
    // JtMsg::same_launch: the producer of an incoming message runs in the same dataflow launch as its consumer - then the
    // consumer reads the entries through to memory and waits on their "unwritten" markers; messages finished by an earlier
    // launch (or received by an exchange) are read with ordinary loads.
    {
        std::vector<int> seg_of(hp.tasks.size(), -1);
        for (size_t g = 0; g < hp.segments.size(); ++g)
            for (int i = hp.segments[g].first_launch; i < hp.segments[g].first_launch + hp.segments[g].n_launch; ++i)
                for (int t : hp.launches[i].tasks) seg_of[t] = (int)g;
        for (size_t t = 0; t < hp.tasks.size(); ++t) {
            JtTask &tk = hp.tasks[t];
            const std::vector<int> &prod = hp.task_producers[t];
            for (int k = 0; k < tk.n_in && k < (int)prod.size(); ++k)
                tk.msg[k].same_launch = (prod[k] >= 0 && seg_of[t] >= 0 && seg_of[prod[k]] == seg_of[t]) ? 1 : 0;
        }
    }
    hp.sync_words = JT_SYNC_HDR + (int)hp.segments.size();
    if (hp.knobs.debug & 2) {     // time-stamp region, JT_NSTAMP doubles per workgroup (+ one spare set), for -DJT_STAMPS builds
        hp.dbg_base = hp.msg_doubles;
        hp.msg_doubles += ((int64_t)hp.blocks.size() + 1) * 16;
        for (const Launch &L : hp.launches)
            for (int t : L.tasks) hp.tasks[t].dbg_off = hp.dbg_base;
    }
    hp.n_messages = 0;
    for (int c = 0; c < N; ++c)
        if (c != hp.root && (hp.owner[c] == hp.rank || (hp.owner[c] == ALL && hp.rank == 0))) hp.n_messages += 2;
    // Lean records (round 6, JtLean): every field of every task is final here
    for (JtTask &tk : hp.tasks) jtp_make_lean(hp, tk, hp.itab, tk.fold != 0);
    // (a folded marginal task runs as a lean task or not at all: without a record - fold_marginals asks for what jtp_make_lean asks for,
    //  so this does not happen - its requests are the read-out's)
    for (HostPlan::FoldReq &fr : hp.folded)
        if (fr.task >= 0 && hp.tasks[fr.task].lean_off <= 0) fr.task = -1;
    for (JtBlock &b : hp.blocks) {
        const int64_t at = hp.tasks[b.task].lean_off;
        if (hp.tasks[b.task].fold) b.flags |= JT_BLOCK_FOLD;
        if (at > 0) b.flags |= JT_BLOCK_LEAN, b.first_x[5] = (uint32_t)hp.tasks[b.task].pnode, b.first_x[6] = (uint32_t)at, b.first_x[7] = (uint32_t)((uint64_t)at >> 32);
    }
    return JTP_OK;
}

int jtp_build_plan(const jtp_tree_desc *d, HostPlan &hp, std::string &err) {
    return PlanBuilder(d, hp, err).run();
}

// ------------------------------------------------------------------------------------------

// incoming messages of a clique for the read-out tasks: the parent's final downward message, every child's
// final upward message (what consumers read: the reduced sum where a reduce task exists)
// (a unit clique's static table comes with them: src.second = -1 marks it, JtMsg::fixed)
// (`from`: per input, the collect task that forms it - an upward message - or -1: JtMsg::src_task, which tells the engine whose arena a
//  multi-set plan's read-out takes the message from, jtp_engine.hip readout_redirect)
static void neighbour_inputs(const HostPlan &hp, const PNode &p, std::vector<MsgView> &ins, std::vector<std::pair<int64_t, int>> &src,
                             std::vector<int> &from) {
    if (p.psep >= 0) {
        ins.push_back(make_view(p, hp.ps[p.psep], p.psep, false));
        src.push_back({hp.ps[p.psep].dn_roff, hp.ps[p.psep].dn_rnpart});
        from.push_back(-1);
    }
    if (p.stat >= 0) {
        ins.push_back(make_view(p, hp.statics[p.stat]));
        src.push_back({hp.statics[p.stat].off, -1});
        from.push_back(-1);
    }
    for (int k : p.children) {
        const PSep &sp = hp.ps[hp.pn[k].psep];
        ins.push_back(make_view(p, sp, hp.pn[k].psep, true));
        src.push_back({sp.up_roff, sp.up_rnpart});
        from.push_back(hp.pn[k].collect_task);
    }
}

int jtp_plan_marginal_task(const HostPlan &hp, int pnode, const std::vector<std::vector<int>> &out_vars,
                           JtTask &tk, std::vector<int32_t> &itab, std::vector<int> &out_bits, std::vector<int> &npart,
                           std::vector<JtBlock> &blocks, std::string &err, bool with_neighbours) {
    const PNode &p = hp.pn[pnode];
    if (out_vars.empty() || (int)out_vars.size() > JT_MAX_OUT) FAIL(JTP_EINVAL, "internal: %zu marginals in one task", out_vars.size());
    if (with_neighbours && out_vars.size() != 1 && !p.unit) FAIL(JTP_EINVAL, "internal: several marginals in one task of a multi-set plan");
    if (p.unit && !with_neighbours) FAIL(JTP_EINVAL, "internal: a unit clique keeps no belief table to marginalise");
    std::vector<PSep> seps(out_vars.size());
    out_bits.clear();
    for (size_t j = 0; j < out_vars.size(); ++j) {
        PSep &s = seps[j];
        s.vars.assign(out_vars[j].rbegin(), out_vars[j].rend());       // last requested variable = lowest bits
        int bit = 0;
        for (int v : s.vars) {
            if (find_var(p.vars, v) < 0) FAIL(JTP_EINVAL, "variable %d is not in clique %d", v, p.real);
            s.pos.push_back(bit);
            s.nb.push_back(hp.vbits[v]);
            bit += hp.vbits[v];
        }
        s.nbits = bit;
        if (bit > 28) FAIL(JTP_EUNSUPPORTED, "marginal with %d index bits", bit);
        out_bits.push_back(bit);
    }
    memset(&tk, 0, sizeof tk);
    tk.pnode = pnode;
    tk.psi_off = p.unit ? 0 : p.arena_off;
    tk.bel_off = -1;
    tk.unit = p.unit ? 1 : 0;
    tk.mode = 0;                                 // (several outputs: every one of them the sum over its own complement)
    std::vector<MsgView> ins, outs;
    std::vector<std::pair<int64_t, int>> src;
    std::vector<int> from;
    // multi-set plans keep no belief table: the marginal is taken of psi * (every incoming message) directly
    if (with_neighbours) neighbour_inputs(hp, p, ins, src, from);
    if ((int)ins.size() > JT_MAX_IN) FAIL(JTP_EUNSUPPORTED, "clique with %zu neighbours", ins.size());
    for (const PSep &s : seps) outs.push_back(make_view(p, s, -1, true));
    int real_bits = 0;
    for (int nb : p.nb) real_bits += nb;
    // (a pass over a belief table with nothing to stage: the longest workgroups the loop allows, fewest partial copies)
    int rc = plan_loops(hp, p, tk, itab, p.nbits, real_bits, ins, outs, with_neighbours ? 14 : (hp.knobs.marg_block_log2 > 0 ? hp.knobs.marg_block_log2 : hp.TB + JT_MAX_ITER_LOG2), err);
    tk.itab_off = 0;
    if (rc != JTP_OK) return rc;
    if (hp.tmix || p.unit) {                     // the task travels with its own table buffer: the clique's thread map behind its rows
        tk.tmap_off = (int64_t)itab.size();
        itab.insert(itab.end(), p.tmap.begin(), p.tmap.end());
        itab.insert(itab.end(), p.vmap.begin(), p.vmap.end());
    }
    for (size_t k = 0; k < src.size(); ++k) {
        tk.msg[k].off = src[k].first;
        tk.msg[k].npart = src[k].second < 0 ? 1 : src[k].second;
        tk.msg[k].fixed = src[k].second < 0 ? 1 : 0;
        tk.msg[k].same_launch = 0;
        tk.msg[k].src_task = from[k];
    }
    npart.clear();
    for (size_t j = 0; j < out_vars.size(); ++j) npart.push_back(tk.msg[JT_MAX_IN + j].npart);
    blocks.clear();
    for (uint32_t f = 0; f < (1u << tk.nF); ++f) blocks.push_back(jtp_make_block(hp, tk, 0u, f));
    return JTP_OK;
}

int jtp_plan_belief_task(const HostPlan &hp, int pnode, JtTask &tk, std::vector<int32_t> &itab,
                         std::vector<JtBlock> &blocks, std::string &err) {
    const PNode &p = hp.pn[pnode];
    memset(&tk, 0, sizeof tk);
    tk.pnode = pnode;
    tk.psi_off = p.unit ? 0 : p.arena_off;
    tk.bel_off = p.arena_off;                    // (a unit clique: its place in the scratch arena, PlanBuilder::arenas)
    tk.unit = p.unit ? 1 : 0;
    tk.mode = 1;
    std::vector<MsgView> ins, outs;
    std::vector<std::pair<int64_t, int>> src;
    std::vector<int> from;
    neighbour_inputs(hp, p, ins, src, from);
    if ((int)ins.size() > JT_MAX_IN) FAIL(JTP_EUNSUPPORTED, "clique with %zu neighbours", ins.size());
    int real_bits = 0;
    for (int nb : p.nb) real_bits += nb;
    int rc = plan_loops(hp, p, tk, itab, p.nbits, real_bits, ins, outs, 14, err);
    tk.itab_off = 0;
    if (rc != JTP_OK) return rc;
    if (hp.tmix || p.unit) {
        tk.tmap_off = (int64_t)itab.size();
        itab.insert(itab.end(), p.tmap.begin(), p.tmap.end());
        itab.insert(itab.end(), p.vmap.begin(), p.vmap.end());
    }
    for (size_t k = 0; k < src.size(); ++k) {
        tk.msg[k].off = src[k].first;
        tk.msg[k].npart = src[k].second < 0 ? 1 : src[k].second;
        tk.msg[k].fixed = src[k].second < 0 ? 1 : 0;
        tk.msg[k].same_launch = 0;
        tk.msg[k].src_task = from[k];
    }
    blocks.clear();
    for (uint32_t f = 0; f < (1u << tk.nF); ++f) blocks.push_back(jtp_make_block(hp, tk, 0u, f));
    return JTP_OK;
}

// ------------------------------------------------------------------------------------------

namespace {
template <typename It>
void json_list(std::ostringstream &o, It a, It b) {
    o << "[";
    for (It i = a; i != b; ++i) {
        if (i != a) o << ",";
        o << (long long)*i;
    }
    o << "]";
}
template <typename V>
void json_vec(std::ostringstream &o, const V &v) { json_list(o, v.begin(), v.end()); }

void json_msg(std::ostringstream &o, const JtMsg &m, int nF) {
    o << "{\"off\":" << m.off << ",\"npart\":" << m.npart << ",\"pstride\":" << m.pstride
      << ",\"nfree\":" << m.nfree << ",\"lds_off\":" << m.lds_off << ",\"e_w\":";
    json_list(o, m.e_w, m.e_w + 2);
    o << ",\"t_w\":";
    json_list(o, m.t_w, m.t_w + 8);
    o << ",\"red_e\":" << m.red_e << ",\"red_lane\":" << m.red_lane << ",\"red_wave\":" << m.red_wave
      << ",\"e_dep\":" << m.e_dep << ",\"same_launch\":" << m.same_launch << ",\"fixed\":" << m.fixed << ",\"f_w\":";
    json_list(o, m.f_w, m.f_w + nF);
    o << ",\"f_p\":";
    json_list(o, m.f_p, m.f_p + nF);
    o << ",\"free_pos\":";
    json_list(o, m.free_pos, m.free_pos + m.nfree);
    o << "}";
}
}  // namespace

static std::string json_escape(const std::string &s) {
    std::string o;
    for (char c : s) {
        if (c == '"' || c == '\\') o += '\\', o += c;
        else if ((unsigned char)c < 0x20) o += ' ';
        else o += c;
    }
    return o;
}

void jtp_plan_to_json(HostPlan &hp, bool with_tasks) {
    std::ostringstream o;
    o << "{\"version\":1,\"dtype\":" << hp.dtype << ",\"VEC\":" << hp.VEC << ",\"EB\":" << hp.EB << ",\"TB\":" << hp.TB
      << ",\"n_cliques\":" << hp.n_cliques << ",\"n_ranks\":" << hp.n_ranks << ",\"rank\":" << hp.rank
      << ",\"root\":" << hp.root << ",\"arena_elems\":" << hp.arena_elems << ",\"msg_doubles\":" << hp.msg_doubles
      << ",\"dbg_base\":" << hp.dbg_base << ",\"max_lds\":" << hp.max_lds << ",\"alg_bytes\":" << (long long)hp.alg_bytes
      << ",\"staging_bytes\":" << (long long)hp.staging_bytes << ",\"table_bytes\":" << (long long)hp.table_bytes
      << ",\"compact\":" << (hp.compact ? 1 : 0) << ",\"tmix\":" << (hp.tmix ? 1 : 0) << ",\"host_table_elems\":" << (long long)hp.host_table_elems
      << ",\"multiset\":" << (hp.multiset ? 1 : 0) << ",\"alg_table_bytes\":" << (long long)hp.alg_table_bytes
      << ",\"alg_msg_bytes\":" << (long long)hp.alg_msg_bytes
      << ",\"n_messages\":" << hp.n_messages << ",\"n_tasks\":" << hp.tasks.size()
      << ",\"n_blocks\":" << hp.blocks.size() << ",\"tmix_compact\":" << (hp.tmix_compact ? 1 : 0) << ",\"lean\":" << (hp.lean ? 1 : 0) << ",\"has_unit\":" << (hp.has_unit ? 1 : 0)
      << ",\"lean_refused\":\"" << json_escape(hp.lean_refused) << "\",\"fix_doubles\":" << hp.fix_doubles << ",\"scratch_elems\":" << hp.scratch_elems << ",\"alg_bytes_full\":" << (long long)hp.alg_bytes_full;
    o << ",\"statics\":[";
    for (size_t i = 0; i < hp.statics.size(); ++i) {
        const PStatic &st = hp.statics[i];
        if (i) o << ",";
        o << "{\"pnode\":" << st.pnode << ",\"nbits\":" << st.nbits << ",\"off\":" << st.off << ",\"vars\":";
        json_vec(o, st.vars);
        o << ",\"pos\":";
        json_vec(o, st.pos);
        o << ",\"nb\":";
        json_vec(o, st.nb);
        o << "}";
    }
    o << "],\"pnodes\":[";
    for (size_t i = 0; i < hp.pn.size(); ++i) {
        const PNode &p = hp.pn[i];
        if (i) o << ",";
        o << "{\"real\":" << p.real << ",\"parent\":" << p.parent << ",\"psep\":" << p.psep << ",\"depth\":" << p.depth
          << ",\"owner\":" << p.owner << ",\"nbits\":" << p.nbits << ",\"arena_off\":" << p.arena_off
          << ",\"phys_elems\":" << p.phys_elems << ",\"pad_mask\":" << p.pad_mask << ",\"tmix\":" << (p.tmix ? 1 : 0) << ",\"trow\":" << p.trow
          << ",\"tpad_mask\":" << p.tpad_mask << ",\"tsplit\":" << p.tsplit << ",\"tsplit_lb\":" << p.tsplit_lb << ",\"tmap_off\":" << p.tmap_off << ",\"layout\":" << p.layout << ",\"collect_task\":" << p.collect_task << ",\"distribute_task\":" << p.distribute_task
          << ",\"unit\":" << (p.unit ? 1 : 0) << ",\"stat\":" << p.stat << ",\"cover\":";
        json_vec(o, p.cover);
        o << ",\"down_tasks\":";
        json_vec(o, p.down_tasks);
        if (hp.tmix || p.unit) {
            o << ",\"tmap\":";
            json_vec(o, p.tmap);
            o << ",\"vmap\":";
            json_vec(o, p.vmap);
        }
        o << ",\"bitw\":";
        json_vec(o, p.bitw);
        o << ",\"group_mask\":";
        json_vec(o, p.group_mask);
        o << ",\"group_pos\":";
        json_vec(o, p.group_pos);
        o << ",\"group_card\":";
        json_vec(o, p.group_card);
        {
            std::vector<int> cards;
            for (int v : p.vars) cards.push_back(hp.card[v]);
            o << ",\"card\":";
            json_vec(o, cards);
        }
        o << ",\"vars\":";
        json_vec(o, p.vars);
        o << ",\"pos\":";
        json_vec(o, p.pos);
        o << ",\"nb\":";
        json_vec(o, p.nb);
        o << ",\"children\":";
        json_vec(o, p.children);
        o << "}";
    }
    o << "],\"pack\":[";                      // host <-> device conversion records of the real cliques (host variable order)
    for (size_t i = 0; i < hp.pack.size(); ++i) {
        const JtPackDesc &pd = hp.pack[i];
        if (i) o << ",";
        o << "{\"pos\":";
        json_list(o, pd.pos, pd.pos + pd.nvars);
        o << ",\"nb\":";
        json_list(o, pd.nb, pd.nb + pd.nvars);
        o << ",\"card\":";
        json_list(o, pd.card, pd.card + pd.nvars);
        o << ",\"dstride\":";
        json_list(o, pd.dstride, pd.dstride + pd.nvars);
        o << ",\"dmod\":";
        json_list(o, pd.dmod, pd.dmod + pd.nvars);
        o << ",\"split_var\":" << pd.split_var << ",\"split_lb\":" << pd.split_lb << ",\"split_ds2\":" << pd.split_ds2 << ",\"split_mod2\":" << pd.split_mod2;
        o << ",\"phys_elems\":" << pd.phys_elems << "}";
    }
    o << "],\"pseps\":[";
    for (size_t i = 0; i < hp.ps.size(); ++i) {
        const PSep &s = hp.ps[i];
        if (i) o << ",";
        o << "{\"node\":" << s.node << ",\"child\":" << s.child << ",\"parent\":" << s.parent << ",\"nbits\":" << s.nbits
          << ",\"up_npart\":" << s.up_npart << ",\"dn_npart\":" << s.dn_npart << ",\"up_off\":" << s.up_off
          << ",\"dn_off\":" << s.dn_off << ",\"up_roff\":" << s.up_roff << ",\"dn_roff\":" << s.dn_roff
          << ",\"up_rnpart\":" << s.up_rnpart << ",\"dn_rnpart\":" << s.dn_rnpart
          << ",\"up_red_task\":" << s.up_red_task << ",\"dn_red_task\":" << s.dn_red_task << ",\"dn_task\":" << s.dn_task << ",\"vars\":";
        json_vec(o, s.vars);
        o << ",\"pos\":";
        json_vec(o, s.pos);
        o << ",\"nb\":";
        json_vec(o, s.nb);
        o << "}";
    }
    o << "],\"launches\":[";
    for (size_t i = 0; i < hp.launches.size(); ++i) {
        const Launch &L = hp.launches[i];
        if (i) o << ",";
        o << "{\"phase\":" << L.phase << ",\"level\":" << L.level << ",\"variant\":" << L.variant
          << ",\"nblocks\":" << L.nblocks << ",\"blk_off\":" << L.blk_off << ",\"lds_bytes\":" << L.lds_bytes
          << ",\"alg_bytes\":" << (long long)L.alg_bytes << ",\"tasks\":";
        json_vec(o, L.tasks);
        o << "}";
    }
    o << "],\"steps\":[";
    for (size_t i = 0; i < hp.steps.size(); ++i) {
        if (i) o << ",";
        o << "[" << hp.steps[i].kind << "," << hp.steps[i].first << "," << hp.steps[i].count << "]";
    }
    o << "],\"sync_words\":" << hp.sync_words << ",\"segments\":[";
    for (size_t i = 0; i < hp.segments.size(); ++i) {
        const Segment &g = hp.segments[i];
        if (i) o << ",";
        o << "{\"phase\":" << g.phase << ",\"first_launch\":" << g.first_launch << ",\"n_launch\":" << g.n_launch
          << ",\"blk_off\":" << g.blk_off << ",\"nblocks\":" << g.nblocks << ",\"lds_bytes\":" << g.lds_bytes
          << ",\"ticket_idx\":" << g.ticket_idx << "}";
    }
    o << "],\"flow_steps\":[";
    for (size_t i = 0; i < hp.flow_steps.size(); ++i) {
        if (i) o << ",";
        o << "[" << hp.flow_steps[i].kind << "," << hp.flow_steps[i].first << "," << hp.flow_steps[i].count << "]";
    }
    o << "],\"comm\":[";
    for (size_t i = 0; i < hp.comm.size(); ++i) {
        const CommOp &c = hp.comm[i];
        if (i) o << ",";
        o << "{\"send\":" << c.send << ",\"psep\":" << c.psep << ",\"up\":" << c.up << ",\"peer\":" << c.peer
          << ",\"off\":" << c.off << ",\"count\":" << c.count << "}";
    }
    o << "]";
    if (with_tasks) {
        o << ",\"tasks\":[";
        for (size_t t = 0; t < hp.tasks.size(); ++t) {
            const JtTask &tk = hp.tasks[t];
            if (t) o << ",";
            o << "{\"pnode\":" << tk.pnode << ",\"kind\":" << tk.kind << ",\"mode\":" << tk.mode << ",\"unit\":" << tk.unit << ",\"setb\":" << tk.setb << ",\"esum\":" << tk.esum << ",\"variant\":" << hp.task_variant[t] << ",\"psi_off\":" << tk.psi_off
              << ",\"bel_off\":" << tk.bel_off << ",\"nbits\":" << tk.nbits << ",\"real_bits\":" << tk.real_bits << ",\"nF\":" << tk.nF << ",\"nA\":" << tk.nA
              << ",\"nR\":" << tk.nR << ",\"settle\":" << tk.settle << ",\"keep_rows\":" << tk.keep_rows << ",\"tmap_off\":" << tk.tmap_off << ",\"fold\":" << tk.fold << ",\"lean_off\":" << tk.lean_off << ",\"vgroups\":" << tk.vgroups << ",\"out_run\":" << tk.out_run << ",\"n_in\":" << tk.n_in << ",\"n_out\":" << tk.n_out
              << ",\"lds_bytes\":" << tk.lds_bytes << ",\"first_x\":";
            json_list(o, tk.first_x, tk.first_x + 8);
            o << ",\"f_x\":";
            json_list(o, tk.f_x, tk.f_x + tk.nF);
            o << ",\"f_lx\":";
            json_list(o, tk.f_lx, tk.f_lx + tk.nF);
            o << ",\"loop_pos\":";
            json_list(o, tk.loop_pos, tk.loop_pos + tk.nA + tk.nR);
            o << ",\"total\":" << tk.total << ",\"itab_lds\":" << tk.itab_lds << ",\"itab\":[";
            for (int i = 0; i < tk.total; ++i) {
                if (i) o << ",";
                json_list(o, hp.itab.begin() + tk.itab_off + (size_t)i * JT_NCOL, hp.itab.begin() + tk.itab_off + (size_t)(i + 1) * JT_NCOL);
            }
            o << "],\"in\":[";
            for (int k = 0; k < tk.n_in; ++k) {
                if (k) o << ",";
                json_msg(o, tk.msg[k], tk.nF);
            }
            o << "],\"out\":[";
            for (int k = 0; k < tk.n_out; ++k) {
                if (k) o << ",";
                json_msg(o, tk.msg[JT_MAX_IN + k], tk.nF);
            }
            o << "]";
            if (tk.lean_off > 0 && (size_t)tk.lean_off + sizeof(JtLean) / 4 <= hp.itab.size()) {      // (the record as the kernel reads it)
                o << ",\"lean\":";
                json_list(o, hp.itab.begin() + tk.lean_off, hp.itab.begin() + tk.lean_off + sizeof(JtLean) / 4);
            }
            o << "}";
        }
        o << "],\"blocks\":[";
        for (size_t b = 0; b < hp.blocks.size(); ++b) {
            if (b) o << ",";
            const JtBlock &k = hp.blocks[b];
            o << "[" << k.task << "," << hp.block_chunk[b] << "," << k.xF;
            for (int i = 0; i < JT_MAX_MSG; ++i) o << "," << k.gbase[i];
            for (int i = 0; i < JT_MAX_OUT; ++i) o << "," << k.pnum[i];
            o << "," << k.psi_x0;
            for (int i = 0; i < 8; ++i) o << "," << k.first_x[i];
            o << "," << k.lxF << "," << k.flags;
            o << "]";
        }
        o << "],\"init_blocks\":[";
        bool first_init = true;
        for (int m = 0; m < 2; ++m)
            for (size_t b = 0; b < hp.init_blocks[m].size(); ++b) {
                if (!first_init) o << ",";
                first_init = false;
                const JtBlock &k = hp.init_blocks[m][b];
                o << "[" << k.task << "," << hp.init_chunk[m][b] << "," << k.xF;
                for (int i = 0; i < JT_MAX_MSG; ++i) o << "," << k.gbase[i];
                for (int i = 0; i < JT_MAX_OUT; ++i) o << "," << k.pnum[i];
                o << "," << k.psi_x0;
                for (int i = 0; i < 8; ++i) o << "," << k.first_x[i];
                o << "," << k.lxF << "," << k.flags;
                o << "]";
            }
        o << "]";
    }
    o << "}";
    hp.json = o.str();
}
