"""CPU emulation of the device task tables, for tests only.

`jtp_plan_describe` exports every index table the HIP kernels consume (csrc/jtp_internal.h:
JtTask / JtMsg).  This module executes those tables with numpy exactly the way
csrc/jtp_kernels.hip.h `jt_pass` does - chunk decode, sub-box staging with partial-copy
summation, incremental A/R loop offsets, per-thread slot offsets, partial-copy flush - so
that the planner (layouts, F/A/R split, message buffers, level schedule) can be checked
against the oracle on machines without a GPU.  It is a checker of the *plan*, not a compute
path: nothing in the product imports it.
"""

import numpy as np

JT_MAX_IN = 4


def _signed(v):
    v = np.asarray(v, dtype=np.int64) & 0xFFFFFFFF
    return np.where(v >= 2 ** 31, v - 2 ** 32, v)


def _scatter(s, free_pos):
    idx = np.zeros_like(s)
    for b, pos in enumerate(free_pos):
        idx += ((s >> b) & 1) << pos
    return idx


class Emulator:
    def __init__(self, desc):
        self.d = desc
        self.VEC, self.EB, self.TB = desc["VEC"], desc["EB"], desc["TB"]
        self.psi = np.zeros(max(desc["arena_elems"], 1), dtype=np.float64)
        self.bel = np.zeros_like(self.psi)
        self.msg = np.full(max(desc["msg_doubles"], 1), np.nan)     # NaN = never written
        self.fix = np.zeros(max(desc.get("fix_doubles", 0), 1))     # static tables of unit cliques (one copy, outside the message arena)
        self.row = 1 << self.TB                                     # elements of one row; rows 0 / 1 of the arenas are shared:
        assert desc["arena_elems"] >= 2 * self.row                  # row 0 all zero (rows that do not exist), row 1 scratch
        for p in desc["pnodes"]:                                    # virtual cliques: resident 0/1 tables
            if p["real"] < 0 and p["arena_off"] >= 0 and not p.get("unit"):
                cards = p["card"]
                n = int(np.prod(cards)) if cards else 1
                digits = np.unravel_index(np.arange(n), cards) if cards else ()
                x = np.zeros(n, dtype=np.int64)
                for pos, dig in zip(p["pos"], digits):
                    x += dig.astype(np.int64) << pos
                self.psi[p["arena_off"] + self._phys(p, x)] = 1.0

    def _phys(self, p, x):
        """physical element offset of logical index x: linear in the index bits (jtp_internal.h) - except, in plans with a
        mixed-radix thread part (`tmix`), inside a row, where the clique's thread map says where logical thread index
        x mod 2^TB lies (callers only ask for entries that exist)"""
        out = np.zeros_like(x)
        for b, w in enumerate(p["bitw"]):
            if b >= self.TB or "tmap" not in p:
                out += ((x >> b) & 1) * w
        if "tmap" in p:
            t = np.asarray(p["tmap"], dtype=np.int64)[x & (self.row - 1)]
            assert np.all(t >= 0), "a table entry whose thread index the map does not know"
            out += t
        return out

    # ---------------------------------------------------------------- layout conversion
    def _dev_index(self, pnode, host_vars, cards):
        """device element index of every host C-order element of a clique"""
        p = self.d["pnodes"][pnode]
        pos = {v: (p["pos"][i], p["nb"][i]) for i, v in enumerate(p["vars"])}
        n = int(np.prod(cards)) if cards else 1
        digits = np.unravel_index(np.arange(n), cards) if cards else ()
        x = np.zeros(n, dtype=np.int64)
        for v, dig in zip(host_vars, digits):
            x += dig.astype(np.int64) << pos[v][0]
        phys = self._phys(p, x)
        assert len(np.unique(phys)) == len(phys) and phys.max(initial=0) < p["phys_elems"]     # a table entry, a place
        return phys

    def _msg_index(self, layout, host_vars, cards):
        """index into a plain bit-field table (a message, a static table) of every C-order element over `host_vars`; variables the
        table does not have are broadcast"""
        pos = {v: layout["pos"][i] for i, v in enumerate(layout["vars"])}
        n = int(np.prod(cards)) if cards else 1
        digits = np.unravel_index(np.arange(n), cards) if cards else ()
        x = np.zeros(n, dtype=np.int64)
        for v, dig in zip(host_vars, digits):
            if v in pos:
                x += dig.astype(np.int64) << pos[v]
        return x

    def set_potential(self, clique, host_vars, cards, array):
        p = self.d["pnodes"][clique]
        if p.get("unit"):
            # no table: the potential may only depend on the covered variables - it is the clique's static table
            arr = np.asarray(array, dtype=np.float64)
            arr = arr.reshape(arr.shape if arr.ndim == len(cards) else [1] * len(cards))
            cov = [v in p["cover"] for v in host_vars]
            assert all(arr.shape[i] == 1 or cov[i] for i in range(len(cards))), "a unit clique's potential depends on an uncovered variable"
            if p["stat"] < 0:
                assert arr.size == 1 and float(arr.reshape(-1)[0]) == 1.0
                return
            st = self.d["statics"][p["stat"]]
            cvars = [v for v, c in zip(host_vars, cov) if c]
            ccards = [k for k, c in zip(cards, cov) if c]
            full = np.broadcast_to(arr.reshape([arr.shape[i] for i in range(len(cards)) if cov[i]]), ccards)
            self.fix[st["off"]:st["off"] + (1 << st["nbits"])] = 0.0
            self.fix[st["off"] + self._msg_index(st, cvars, ccards)] = full.ravel()
            return
        x = self._dev_index(clique, host_vars, cards)
        lo = p["arena_off"]
        self.psi[lo:lo + p["phys_elems"]] = 0.0
        self.psi[lo + x] = np.broadcast_to(np.asarray(array, dtype=np.float64), cards).ravel()

    def belief(self, clique, host_vars, cards):
        p = self.d["pnodes"][clique]
        if p.get("unit"):
            # no belief table is kept: psi (the static table, broadcast) x every incoming message, as the read-out tasks form it
            size = lambda s: 1 << s["nbits"]
            tot = np.ones(int(np.prod(cards)) if cards else 1)
            if p["stat"] >= 0:
                st = self.d["statics"][p["stat"]]
                tot = tot * self.fix[st["off"] + self._msg_index(st, host_vars, cards)]
            if p["psep"] >= 0:
                s = self.d["pseps"][p["psep"]]
                idx = self._msg_index(s, host_vars, cards)
                tot = tot * sum(self.msg[s["dn_roff"] + k * size(s) + idx] for k in range(s["dn_rnpart"]))
            for ch in p["children"]:
                s = self.d["pseps"][self.d["pnodes"][ch]["psep"]]
                idx = self._msg_index(s, host_vars, cards)
                tot = tot * sum(self.msg[s["up_roff"] + k * size(s) + idx] for k in range(s["up_rnpart"]))
            return tot.reshape(cards)
        x = self._dev_index(clique, host_vars, cards)
        return self.bel[p["arena_off"] + x].reshape(cards)

    def folded_marginal(self, task, j, host_vars, cards):
        """output j of a marginal task folded into the propagate (round 6): the sum of its partial copies in the message arena, as
        jt_marg_unpack forms it - a plain bit-field table, the LAST requested variable in the lowest bits"""
        tk = self.d["tasks"][task]
        assert tk["fold"] and j < tk["n_out"]
        m = tk["out"][j]
        p = self.d["pnodes"][tk["pnode"]]
        nb = {v: p["nb"][i] for i, v in enumerate(p["vars"])}
        pos, bit = {}, 0
        for v in reversed(host_vars):
            pos[v] = bit
            bit += nb[v]
        assert m["pstride"] == 1 << bit
        idx = self._msg_index({"vars": list(pos), "pos": list(pos.values())}, host_vars, cards)
        tot = np.zeros(len(idx))
        for q in range(m["npart"]):
            tot = tot + self.msg[m["off"] + q * m["pstride"] + idx]
        return tot.reshape(cards)

    def sep_belief(self, psep, host_vars, cards):
        s = self.d["pseps"][psep]
        pos = {v: s["pos"][i] for i, v in enumerate(s["vars"])}
        n = int(np.prod(cards)) if cards else 1
        digits = np.unravel_index(np.arange(n), cards) if cards else ()
        x = np.zeros(n, dtype=np.int64)
        for v, dig in zip(host_vars, digits):
            x += dig.astype(np.int64) << pos[v]
        size = 1 << s["nbits"]
        up = sum(self.msg[s["up_roff"] + p * size + x] for p in range(s["up_rnpart"]))
        dn = sum(self.msg[s["dn_roff"] + p * size + x] for p in range(s["dn_rnpart"]))
        return (up * dn).reshape(cards)

    # ---------------------------------------------------------------- one workgroup
    def _reduce_block(self, tk, chunk, record, strict):
        """reduce task (csrc/jtp_kernels.hip.h jt_reduce): 256 entries of the sum of the partial copies"""
        assert tk["n_in"] == 1 and tk["n_out"] == 1 and tk["out"][0]["npart"] == 1
        src, dst = tk["in"][0], tk["out"][0]
        n = 1 << tk["nbits"]
        x0 = sum(tk["f_x"][j] for j in range(tk["nF"]) if (chunk >> j) & 1)
        assert x0 == chunk * 64 and record[0] == x0          # JT_REDUCE_ENTRIES per workgroup
        i = x0 + np.arange(64)
        i = i[i < n]
        assert src["pstride"] == n
        tot = np.zeros(len(i))
        for p in range(src["npart"]):
            tot = tot + self.msg[src["off"] + p * src["pstride"] + i]
        if strict:
            assert not np.any(np.isnan(tot)), "a reduce workgroup precedes a producer of its message"
        assert np.all(np.isnan(self.msg[dst["off"] + i])), "a summed entry is written twice"
        self.msg[dst["off"] + i] = tot

    def _block(self, tk, chunk, collect, record=None, strict=False, init=False):
        if tk["kind"] == 1:
            return self._reduce_block(tk, chunk, record, strict)
        VEC, EB = self.VEC, self.EB
        n_in, n_out = tk["n_in"], tk["n_out"]
        ins, outs = tk["in"], tk["out"]
        xF = 0
        gb_in, gb_out, pnum = [0] * n_in, [0] * n_out, [0] * n_out
        for j in range(tk["nF"]):
            if (chunk >> j) & 1:
                xF += tk["f_x"][j]
                for k in range(n_in):
                    gb_in[k] += ins[k]["f_w"][j]
                for k in range(n_out):
                    gb_out[k] += outs[k]["f_w"][j]
                    pnum[k] += outs[k]["f_p"][j]
        NO_ROW = 0xFFFFFFFF
        pn = self.d["pnodes"][tk["pnode"]]
        lxF = sum(tk["f_lx"][j] for j in range(tk["nF"]) if (chunk >> j) & 1)
        fmask = sum(tk["f_lx"])

        def exists(x, within):      # the rows named by the high bits of x (restricted to `within`) exist
            if x & within & pn["pad_mask"]:
                return False
            for g, pos, card in zip(pn["group_mask"], pn["group_pos"], pn["group_card"]):
                assert (g & within) in (0, g), "a variable stored at its true cardinality is split between chunk and loop bits"
                if (g & within) == g and ((x & g) >> pos) >= card:
                    return False
            return True

        chunk_ok = exists(lxF, fmask)
        if record is not None:      # the host-decoded workgroup record must agree with the bit decode
            assert record[20] == lxF and (record[21] & 1) == (0 if chunk_ok else 1)       # (bit 1: JT_BLOCK_KEEP_ROWS, a cache-policy hint)
            if chunk_ok:
                assert record[0] == xF and record[11] == tk["psi_off"] + xF and (not tk.get("unit") or tk["psi_off"] == 0)
                assert bool(record[21] & 16) == bool(tk.get("fold"))       # JT_BLOCK_FOLD (round 6): a marginal task folded into the propagate
                if record[21] & 4:        # JT_BLOCK_LEAN (round 6): a unit task loads no rows - the last three words say where its lean record is
                    assert tk["unit"] and tk["lean_off"] > 0 and tk["lean_off"] % 16 == 0
                    assert list(record[12:17]) == list(tk["first_x"][:5])
                    assert record[17] == tk["pnode"] and (record[18] | (record[19] << 32)) == tk["lean_off"]
                else:
                    assert not tk.get("lean_off") and list(record[12:20]) == list(tk["first_x"])
            else:
                assert record[0] == 0 and record[11] == 0 and all(v == NO_ROW for v in record[12:20])
            assert list(record[1:1 + n_in]) == gb_in and list(record[5:5 + n_out]) == gb_out
            assert list(record[8:8 + n_out]) == pnum
        # staging
        subs = []
        for k, m in enumerate(ins):
            s = np.arange(1 << m["nfree"], dtype=np.int64)
            idx = _scatter(s, m["free_pos"])
            tot = np.zeros(len(s))
            if (self.d.get("tmix") or init) and not chunk_ok:
                # a chunk that does not exist: in mixed-radix plans it stages nothing and waits for nobody (jt_pass: `!TMIX || chunk_ok`);
                # at initialisation (init_blocks) nothing is read at all - its partial copies are zeros whatever comes in
                subs.append(tot)
                continue
            if m.get("fixed"):      # the clique's static table: one copy in the fixed arena, never "unwritten"
                assert m["npart"] == 1 and not m["same_launch"] and tk["unit"]
                subs.append(self.fix[m["off"] + gb_in[k] + idx])
                continue
            for p in range(m["npart"]):
                tot = tot + self.msg[m["off"] + p * m["pstride"] + gb_in[k] + idx]
            if strict:      # dataflow order: every entry read was written by an earlier workgroup
                assert not np.any(np.isnan(tot)), "a workgroup precedes the producer of an entry it reads"
                if not m["same_launch"]:        # read with ordinary loads: must be complete before the launch
                    snap = np.zeros(len(s))
                    for p in range(m["npart"]):
                        snap = snap + self._launch_snapshot[m["off"] + p * m["pstride"] + gb_in[k] + idx]
                    assert not np.any(np.isnan(snap)), "an ordinary load of a message the same launch produces"
            subs.append(tot)
        osubs = [np.zeros(1 << m["nfree"]) for m in outs]

        nA, nR = 1 << tk["nA"], 1 << tk["nR"]
        tid = np.arange(256, dtype=np.int64)
        lane, wave = tid & 63, tid >> 6

        def thread_off(m):
            t = np.zeros(256, dtype=np.int64)
            for b in range(6):
                t += ((lane >> b) & 1) * m["t_w"][b]
            for b in range(2):
                t += ((wave >> b) & 1) * m["t_w"][6 + b]
            return t

        def e_off(m):
            return np.array([((e & 1) and m["e_w"][0]) + ((e & 2) and m["e_w"][1]) for e in range(VEC)],
                            dtype=np.int64)

        for m in outs:      # the kernel relies on: "reduce" bits have weight 0, others do not
            for e in range(EB):
                assert bool(m["red_e"] >> e & 1) == (m["e_w"][e] == 0)
            for b in range(6):
                assert bool(m["red_lane"] >> b & 1) == (m["t_w"][b] == 0)
            for b in range(2):
                assert bool(m["red_wave"] >> b & 1) == (m["t_w"][6 + b] == 0)
        for m in ins:
            assert bool(m["e_dep"]) == any(m["e_w"][e] != 0 for e in range(EB))

        if self.d.get("tmix"):
            # plans with mixed-radix rows: the table holds the rows that exist only, each with its place in the full loop
            # nest (bits 16-21 of column 1 + JT_MAX_IN) and the "run of outgoing message j ends here" flags (bit 24 + j)
            rows = np.asarray(tk["itab"], dtype=np.int64).reshape(-1, 8)
            assert len(rows) == tk["total"] and 1 <= tk["total"] <= nA * nR <= 64
            info = rows[:, 1 + JT_MAX_IN] & 0xFFFFFFFF
            nest = (info >> 16) & 63
            assert np.all(np.diff(nest) > 0) and np.all(info >> (24 + max(n_out, 0)) == 0)
            for i in range(8):
                assert tk["first_x"][i] == ((rows[i, 0] & 0xFFFFFFFF) if i < tk["total"] else NO_ROW)
            for j in range(n_out):
                run = (tk["out_run"] >> (8 * j)) & 0xFF
                ends = (info >> (24 + j)) & 1
                want = np.array([r + 1 == len(nest) or (nest[r + 1] >> run) != (nest[r] >> run) for r in range(len(nest))])
                assert np.array_equal(ends.astype(bool), want), "run ends of outgoing message %d" % j
            rows[:, 1 + JT_MAX_IN] = info & 0xFFFF
            itab = np.zeros((nA * nR, 8), dtype=np.int64)
            itab[:, 0] = NO_ROW
            itab[nest] = rows
            # (offsets of rows that do not exist are not stored: give them their run's, which the checks below compare)
            for j in range(n_out):
                run = (tk["out_run"] >> (8 * j)) & 0xFF
                col = itab[:, 1 + JT_MAX_IN + j].reshape(-1, 1 << run)
                have = (itab[:, 0] != NO_ROW).reshape(-1, 1 << run)
                for g in range(len(col)):
                    if have[g].any():
                        col[g, :] = col[g, have[g]][0]
            itab = itab.reshape(nA, nR, 8)
            xoff = itab[:, :, 0] & 0xFFFFFFFF
        else:
            itab = np.asarray(tk["itab"], dtype=np.int64).reshape(nA, nR, 8)      # row i = a * nR + r
            assert tk["total"] == nA * nR and 2 <= tk["total"] <= 64
            xoff = itab[:, :, 0] & 0xFFFFFFFF
            for i in range(8):
                assert tk["first_x"][i] == (xoff.ravel()[i] if i < tk["total"] else 0)
        # rows that do not exist are marked; the marks must agree with the digits of the row's loop bits
        loopmask = sum(1 << b for b in tk["loop_pos"])
        row_ok = np.ones((nA, nR), dtype=bool)
        for i in range(nA * nR):
            lx = sum(((i >> t) & 1) << b for t, b in enumerate(tk["loop_pos"]))
            ok = exists(lx, loopmask)
            assert ok == (xoff.ravel()[i] != NO_ROW), "row %d: mark and digits disagree" % i
            row_ok[i // nR, i % nR] = ok and chunk_ok
            if ok:
                assert xoff.ravel()[i] == self._phys(pn, np.array([lx], dtype=np.int64))[0]
        xoff = np.where(xoff == NO_ROW, 0, xoff)
        in_off = [_signed(itab[:, :, 1 + k]) for k in range(n_in)]
        out_off = [_signed(itab[:, :, 1 + JT_MAX_IN + j]) for j in range(n_out)]
        for j in range(n_out):      # the kernel reads message j's offsets once per run of 2^run_j iterations (JtTask::out_run)
            run = (tk["out_run"] >> (8 * j)) & 0xFF
            assert tk["nR"] <= run <= tk["nA"] + tk["nR"]
            flat = out_off[j].reshape(-1, 1 << run)
            assert np.all(flat == flat[:, :1])
        # element index of every (a, r, tid, e)
        slot = (tid * VEC)[:, None] + np.arange(VEC)[None, :]                  # logical thread index of (tid, e)
        if "tmap" in pn:        # mixed-radix thread part: the element's place inside the row comes from the clique's map
            tmap = np.asarray(pn["tmap"], dtype=np.int64)[slot]
            assert tk["tmap_off"] == pn["tmap_off"] or tk["tmap_off"] >= 0
        else:
            tmap = slot
        unit = bool(tk.get("unit"))
        assert unit == bool(pn.get("unit")) and (not unit or ("tmap" in pn and tk["tmap_off"] >= 0))
        x = (xF + xoff[:, :, None, None] + np.maximum(tmap, 0)[None, None, :, :]) & 0xFFFFFFFF
        live = np.broadcast_to(row_ok[:, :, None, None], x.shape) & np.broadcast_to((tmap >= 0)[None, None, :, :], x.shape)
        assert x[live].max(initial=0) < pn["phys_elems"]
        assert len(np.unique(x[live])) == x[live].size            # every stored element visited at most once
        if unit:
            # a unit clique: every entry that exists counts as 1 - and the entries that exist are exactly the clique's table
            # (counted over the task's workgroups, checked at the end of the propagate)
            tk["_live"] = tk.get("_live", 0) + int(live.sum())
            p = np.where(live, 1.0, 0.0)
        else:
            p = np.where(live, self.psi[tk["psi_off"] + np.where(live, x, 0)], 0.0)      # rows that do not exist read zeros
        vals = []
        for k, m in enumerate(ins):
            slot = (in_off[k][:, :, None, None]
                    + thread_off(m)[None, None, :, None] + e_off(m)[None, None, None, :])
            assert slot.min() >= 0 and slot.max() < len(subs[k])
            vals.append(subs[k][slot])
        if collect:
            q = p.copy()
            for v in vals:
                q = q * v
            contrib = [q] * n_out      # (several outputs: folded marginal tasks, round 6 - each the sum over its own complement)
        else:
            npar = n_in - n_out
            pre = p.copy()
            for v in vals[:npar]:
                pre = pre * v
            contrib = []
            for j in range(n_out):
                q = pre.copy()
                for i in range(n_out):
                    if i != j:
                        q = q * vals[npar + i]
                contrib.append(q)
            b = pre.copy()
            for v in vals[npar:]:
                b = b * v
            if tk["bel_off"] >= 0:
                assert not unit
                self.bel[tk["bel_off"] + x[live]] = b[live]
        for j, m in enumerate(outs):
            slot = (out_off[j][:, :, None, None]
                    + thread_off(m)[None, None, :, None] + e_off(m)[None, None, None, :])
            assert slot.min() >= 0 and slot.max() < len(osubs[j])
            np.add.at(osubs[j], slot.ravel(), contrib[j].ravel())
        for j, m in enumerate(outs):
            s = np.arange(1 << m["nfree"], dtype=np.int64)
            dst = m["off"] + pnum[j] * m["pstride"] + gb_out[j] + _scatter(s, m["free_pos"])
            assert np.all(np.isnan(self.msg[dst])), "a partial-copy entry is written twice"
            self.msg[dst] = osubs[j]

    # ---------------------------------------------------------------- whole schedule
    def propagate_flow(self, comm=None):
        """Run the dataflow schedule (one launch per phase): workgroups in block-list order, each
        checked to find every message entry it reads already written (NaN = unwritten, as the
        device marks it) - i.e. its producers come earlier in the list, which is what makes the
        device-side waits deadlock free once workgroups draw their list position from a ticket."""
        d = self.d
        self.msg[:] = np.nan
        self._init_blocks()
        tickets = [g["ticket_idx"] for g in d["segments"]]
        assert len(set(tickets)) == len(tickets) and all(2 <= t < d["sync_words"] for t in tickets)
        covered = []
        for kind, first, count in d["flow_steps"]:
            if kind == 1:
                assert comm is not None, "multi-rank plan needs a comm callback"
                comm(d["comm"][first:first + count], self.msg)
                continue
            seg = d["segments"][first]
            launches = d["launches"][seg["first_launch"]:seg["first_launch"] + seg["n_launch"]]
            assert all(L["phase"] == seg["phase"] for L in launches) or (seg["phase"] == 2 and [L["phase"] for L in launches] == sorted(L["phase"] for L in launches))
            assert seg["blk_off"] == launches[0]["blk_off"] and seg["nblocks"] == sum(L["nblocks"] for L in launches)
            assert seg["lds_bytes"] == max(L["lds_bytes"] for L in launches)
            assert all(d["tasks"][t]["kind"] == (1 if L["variant"] == 16 else 0) for L in launches for t in L["tasks"])
            covered += list(range(seg["first_launch"], seg["first_launch"] + seg["n_launch"]))
            self._launch_snapshot = self.msg.copy()
            for blk in d["blocks"][seg["blk_off"]:seg["blk_off"] + seg["nblocks"]]:
                if blk[23] & 8:             # (JT_BLOCK_NULL: multi-set plans pad every launch to a multiple of eight records)
                    assert d.get("multiset")
                    continue
                tk = d["tasks"][blk[0]]
                self._block(tk, blk[1], tk["mode"] == 0, blk[2:], strict=True)
        assert covered == list(range(len(d["launches"])))
        self._check_unit_counts()

    def _init_blocks(self):
        """Single-set plans: the chunks that do not exist are not in the block lists - the engine zeroes their partial copies once
        per arena half when the arena is initialised (jtp_engine.hip zero_padding)."""
        d = self.d
        self._init_seen = {}
        for blk in d.get("init_blocks", []):
            tk = d["tasks"][blk[0]]
            assert tk["kind"] == 0 and (blk[23] & 1) == 1, "only chunks that do not exist may be run at initialisation"
            before = np.isnan(self.msg)
            self._block(tk, blk[1], tk["mode"] == 0, blk[2:], init=True)
            written = before & ~np.isnan(self.msg)
            assert np.all(self.msg[written] == 0.0), "a chunk that does not exist wrote something else than zeros"
            for j, m in enumerate(tk["out"]):        # ... and it wrote its WHOLE partial copy
                sub = np.arange(1 << m["nfree"], dtype=np.int64)
                dst = m["off"] + blk[10 + j] * m["pstride"] + blk[3 + JT_MAX_IN + j] + _scatter(sub, m["free_pos"])
                assert np.all(self.msg[dst] == 0.0)
            self._init_seen.setdefault(blk[0], set()).add(blk[1])

    def _check_unit_counts(self):
        for tk in self.d["tasks"]:
            if tk["kind"] == 0 and "_live" in tk:
                cards = self.d["pnodes"][tk["pnode"]]["card"]
                assert tk.pop("_live") == (int(np.prod(cards)) if cards else 1), "a unit task did not visit exactly its clique's entries"

    def propagate(self, comm=None):
        """Run the plan's step list.  `comm(ops, msg)` executes one exchange group (a list of the
        plan's comm records) on the message arena `msg`; it is required for multi-rank plans."""
        d = self.d
        self.msg[:] = np.nan
        self._init_blocks()
        for kind, first, count in d["steps"]:
            if kind == 1:
                assert comm is not None, "multi-rank plan needs a comm callback"
                comm(d["comm"][first:first + count], self.msg)
                continue
            launch = d["launches"][first]
            blocks = d["blocks"][launch["blk_off"]:launch["blk_off"] + launch["nblocks"]]
            if d.get("multiset"):            # (padded to a multiple of eight records with records that start no work)
                assert launch["blk_off"] % 8 == 0 and launch["nblocks"] % 8 == 0 and sum(1 for b in blocks if b[23] & 8) < 8
                blocks = [b for b in blocks if not (b[23] & 8)]
            else:
                assert not any(b[23] & 8 for b in blocks)
            seen = set()
            for blk in blocks:
                t, chunk = blk[0], blk[1]
                tk = d["tasks"][t]
                assert t in launch["tasks"]
                assert launch["variant"] in (tk["variant"], 12 + launch["phase"])      # per level or per shape
                if d.get("multiset"):       # every task of a multi-set plan is a marginalisation (mode 0)
                    assert tk["kind"] == 1 or (tk["mode"] == 0 and tk["variant"] == 17 + launch["phase"] and tk["setb"] in (4096, 16384))
                else:
                    # (a unit clique's downward messages are marginalisations of their own: mode 0 tasks of the distribute phase)
                    assert tk["kind"] == 1 or (tk["unit"] and tk["mode"] == 0 and (tk["n_out"] == 1 or (tk["fold"] and launch["phase"] == 1))) or ((tk["variant"] < 4) == (launch["phase"] == 0) and tk["mode"] == launch["phase"])
                assert tk["lds_bytes"] <= launch["lds_bytes"]
                seen.add((t, chunk))
                self._block(tk, chunk, tk["mode"] == 0, blk[2:])
            # (every chunk of every task once: in the launch, or - a chunk that does not exist, mixed-radix plans - at initialisation)
            assert all(not ((t, c) in seen) for t in launch["tasks"] for c in self._init_seen.get(t, ()))
            assert len(seen) == len(blocks) == sum((1 << d["tasks"][t]["nF"]) - len(self._init_seen.get(t, ())) for t in launch["tasks"])
        self._check_unit_counts()
