// Descriptor fuzzer for the host planner (jtp_plan.cpp is pure C++: built here with -fsanitize=address,undefined,
// no HIP).  jtp_build_plan parses caller-supplied descriptors (CSR variable lists, parent pointers, owner arrays):
// for every random descriptor - well formed, or damaged in one of the ways below - it must return JTP_OK or an
// error code with a message, never crash, read out of bounds or overflow.  Valid plans are additionally checked
// for internal consistency (offsets inside the arenas, every block's task in range).
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined \
//       tests/fuzz/fuzz_plan.cpp junction-tree_amd/csrc/jtp_plan.cpp -o /tmp/fuzz_plan && /tmp/fuzz_plan 4000
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <string>
#include <vector>

#include "../../junction-tree_amd/csrc/jtp_plan.h"

struct Desc {
    std::vector<int32_t> card, off, ids, parent, psep, owner, coff, cids;
    bool with_cover = false;
    jtp_tree_desc d;
};

static void build(std::mt19937 &rng, Desc &t, int n_ranks) {
    auto ri = [&](int lo, int hi) { return (int)(rng() % (unsigned)(hi - lo + 1)) + lo; };
    const int N = ri(1, 24);
    const int maxw = ri(1, 6);
    static const int cards[] = {1, 2, 2, 2, 3, 4, 5, 8};
    std::vector<std::vector<int>> cv(N);
    t.card.clear();
    auto fresh = [&]() {
        t.card.push_back(cards[rng() % 8]);
        return (int)t.card.size() - 1;
    };
    t.parent.assign(N, -1);
    for (int i = 0, w = ri(1, maxw); i < w; ++i) cv[0].push_back(fresh());
    for (int c = 1; c < N; ++c) {
        const int p = ri(0, c - 1);
        t.parent[c] = p;
        for (int v : cv[p])
            if ((int)cv[c].size() < maxw - 1 && rng() % 2) cv[c].push_back(v);
        const int nnew = ri(cv[c].empty() ? 1 : 0, maxw - (int)cv[c].size());
        for (int i = 0; i < nnew; ++i) cv[c].push_back(fresh());
    }
    t.off.assign(1, 0);
    t.ids.clear();
    for (int c = 0; c < N; ++c) {
        for (int v : cv[c]) t.ids.push_back(v);
        t.off.push_back((int)t.ids.size());
    }
    t.psep.assign(N, -1);
    for (int c = 1; c < N; ++c) {
        t.psep[c] = N + c - 1;
        for (int v : cv[c])
            for (int u : cv[t.parent[c]])
                if (u == v) t.ids.push_back(v);
        t.off.push_back((int)t.ids.size());
    }
    t.owner.assign(N, 0);
    for (int c = 0; c < N; ++c) t.owner[c] = (int)(rng() % (unsigned)n_ranks);
    // which variables each clique's potential depends on (jtp_tree_desc.cover_*): none, some or all of its own
    t.with_cover = rng() % 2 == 0;
    t.coff.assign(1, 0);
    t.cids.clear();
    for (int c = 0; c < N; ++c) {
        const unsigned mode = rng() % 4;                       // 0: none, 1: all, else: a random subset
        for (int v : cv[c])
            if (mode == 1 || (mode >= 2 && rng() % 2)) t.cids.push_back(v);
        t.coff.push_back((int)t.cids.size());
    }
    memset(&t.d, 0, sizeof t.d);
    t.d.struct_size = sizeof t.d;
    t.d.n_vars = (int)t.card.size();
    t.d.n_cliques = N;
    t.d.n_nodes = 2 * N - 1;
    t.d.dtype = rng() % 2 ? JTP_F32 : JTP_F64;
    t.d.n_batch = ri(1, 3);
    t.d.n_ranks = n_ranks;
    t.d.rank = (int)(rng() % (unsigned)n_ranks);
    t.d.flags = JTP_PLAN_ONLY | (rng() % 4 == 0 ? JTP_KEEP_ROOT : 0u) | (rng() % 5 == 0 ? JTP_SPLIT_VARIANTS : 0u) |
                (n_ranks == 1 && rng() % 3 == 0 ? (JTP_MULTISET | JTP_SHARE_POTENTIALS) : 0u);
    t.d.layout_policy = ri(0, 3);
    t.d.block_log2 = rng() % 3 ? 0 : ri(8, 18);
    t.d.lds_budget = rng() % 3 ? 0 : ri(16, 70000);
}

static void point(Desc &t) {
    t.d.var_card = t.card.data();
    t.d.node_var_off = t.off.data();
    t.d.node_var_ids = t.ids.data();
    t.d.parent_clique = t.parent.data();
    t.d.parent_sep = t.psep.data();
    t.d.clique_owner = t.owner.data();
    t.d.cover_off = t.with_cover ? t.coff.data() : nullptr;
    t.d.cover_ids = t.with_cover ? t.cids.data() : nullptr;
}

static const char *damage(std::mt19937 &rng, Desc &t) {
    const int N = t.d.n_cliques;
    auto pick = [&](int n) { return (int)(rng() % (unsigned)std::max(n, 1)); };
    switch (rng() % 16) {
        case 14: t.with_cover = true; if (!t.cids.empty()) t.cids[pick((int)t.cids.size())] = pick(t.d.n_vars + 2) - 1; return "covered variable not of the clique / unknown / twice";
        case 15: t.with_cover = true; t.coff[pick((int)t.coff.size())] -= 1 + pick(3); return "cover offsets not monotone";
        case 0: t.parent[pick(N)] = pick(N); return "random parent (cycle / two roots / self)";
        case 1: t.parent[pick(N)] = N + 5; return "parent out of range";
        case 2: if (N > 1) t.psep[1 + pick(N - 1)] = pick(2 * N + 3) - 2; return "separator node out of range or reused";
        case 3: if (!t.ids.empty()) t.ids[pick((int)t.ids.size())] = t.d.n_vars + pick(3); return "unknown variable";
        case 4: if (!t.ids.empty()) t.ids[pick((int)t.ids.size())] = -1 - pick(3); return "negative variable";
        case 5: t.off[pick((int)t.off.size())] -= 1 + pick(3); return "CSR offsets not monotone";
        case 6: t.card[pick(t.d.n_vars)] = -pick(3); return "cardinality <= 0";
        case 7: t.card[pick(t.d.n_vars)] = 1 << (12 + pick(19)); return "huge cardinality (table overflow)";
        case 8: t.owner[pick(N)] = t.d.n_ranks + 1 + pick(2); return "owner out of range";
        case 9: t.d.rank = t.d.n_ranks + pick(2); return "rank out of range";
        case 10: t.d.n_nodes += 1 - 2 * (int)(rng() % 2); return "n_nodes != 2N-1";
        case 11: t.d.dtype = 7; return "bad dtype";
        case 12: t.d.n_batch = -pick(3); return "n_batch <= 0";
        default: t.d.struct_size -= 4; return "struct size";
    }
}

static bool consistent(const HostPlan &hp, std::string &why) {
    for (const JtBlock &b : hp.blocks)
        if (b.task >= hp.tasks.size()) return why = "block task out of range", false;
    std::vector<char> runs(hp.tasks.size(), 0);              // tasks this rank executes (the others keep placeholders)
    for (const Launch &L : hp.launches)
        for (int t : L.tasks) {
            if (t < 0 || t >= (int)hp.tasks.size()) return why = "launch task out of range", false;
            runs[t] = 1;
        }
    for (size_t t = 0; t < hp.tasks.size(); ++t) {
        const JtTask &tk = hp.tasks[t];
        if (tk.kind != 0 || !runs[t]) continue;
        if (tk.unit != (hp.pn[tk.pnode].unit ? 1 : 0)) return why = "unit task of a clique that keeps a table (or the reverse)", false;
        if (tk.unit && (tk.psi_off != 0 || tk.bel_off >= 0)) return why = "unit task with a table", false;
        if (!tk.unit && (tk.psi_off < 0 || tk.psi_off + hp.pn[tk.pnode].phys_elems > std::max<int64_t>(hp.arena_elems, 1) + 256)) return why = "psi_off outside the arena", false;
        {   // every row a workgroup of the task touches lies inside the table (or is marked as not existing)
            const int64_t phys = hp.pn[tk.pnode].phys_elems;
            int64_t fmax = 0;
            for (int j = 0; j < tk.nF; ++j) fmax += tk.f_x[j];
            for (int i = 0; i < tk.total; ++i) {
                const uint32_t off = (uint32_t)hp.itab[tk.itab_off + (int64_t)i * JT_NCOL];
                // (a row is 2^TB elements, or - thread part at true cardinalities - PNode::trow)
                if (off != JT_NO_ROW && (int64_t)off + hp.pn[tk.pnode].trow > phys) return why = "row outside the table", false;
            }
            (void)fmax;
        }
        if (tk.itab_off < 0 || tk.itab_off + (int64_t)tk.total * JT_NCOL > (int64_t)hp.itab.size()) return why = "iteration table outside the buffer", false;
        if (hp.tmix || tk.unit) {          // every entry of the clique's thread map: -1 or an offset inside a row
            if (tk.tmap_off < 0 || tk.tmap_off + ((int64_t)1 << hp.TB) > (int64_t)hp.itab.size()) return why = "thread map outside the buffer", false;
            for (int64_t x = 0; x < ((int64_t)1 << hp.TB); ++x) {
                const int32_t po = hp.itab[tk.tmap_off + x];
                if (po < -1 || po >= hp.pn[tk.pnode].trow) return why = "thread map entry outside the row", false;
            }
        } else if (tk.tmap_off != -1) return why = "thread map in a plan without one", false;
        for (int k = 0; k < tk.n_in + tk.n_out; ++k) {
            const JtMsg &m = tk.msg[k < tk.n_in ? k : JT_MAX_IN + (k - tk.n_in)];
            if (m.fixed && (k >= tk.n_in || !tk.unit || m.npart != 1 || m.off < 0 || m.off + m.pstride > hp.fix_doubles)) return why = "static table outside the fixed arena", false;
            if (!m.fixed && (m.off < 0 || m.off + (int64_t)m.npart * m.pstride > hp.msg_doubles)) return why = "message outside the arena", false;
            if (m.nfree < 0 || m.nfree > JT_MAX_FREE) return why = "sub-box too large", false;
        }
    }
    return true;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    std::mt19937 rng(argc > 2 ? (unsigned)atoi(argv[2]) : 12345u);
    int ok = 0, rejected = 0, damaged_ok = 0;
    for (int it = 0; it < iters; ++it) {
        Desc t;
        const int n_ranks = 1 + (int)(rng() % 4);
        build(rng, t, n_ranks);
        const bool hurt = it % 2 == 1;
        const char *what = hurt ? damage(rng, t) : "none";
        point(t);
        if (hurt && rng() % 16 == 0) t.d.clique_owner = nullptr;
        HostPlan hp;
        std::string err;
        const int rc = jtp_build_plan(&t.d, hp, err);
        if (rc == JTP_OK) {
            std::string why;
            if (!consistent(hp, why)) {
                fprintf(stderr, "iteration %d (damage: %s): plan accepted but inconsistent: %s\n", it, what, why.c_str());
                return 1;
            }
            jtp_plan_to_json(hp, true);
            if (hp.json.empty()) return 2;
            ++ok;
            damaged_ok += hurt;
        } else {
            if (err.empty()) {
                fprintf(stderr, "iteration %d: error %d without a message\n", it, rc);
                return 3;
            }
            if (!hurt && rc != JTP_EUNSUPPORTED) {
                fprintf(stderr, "iteration %d: a well-formed descriptor was rejected: %s\n", it, err.c_str());
                return 4;
            }
            ++rejected;
        }
    }
    printf("fuzz_plan: %d descriptors, %d planned (%d of them damaged yet still valid), %d rejected with a message\n", iters, ok, damaged_ok, rejected);
    return 0;
}
