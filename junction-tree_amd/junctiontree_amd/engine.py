"""Host side of the device boundary: a junction tree in the reference's nested-list form
-> a `Plan` on the MI355X (libjtprop.so through ctypes).

The reference walks the nested list recursively on every call
(`junctiontree/computation.py:47-96, 140-224`); here the list is flattened once into parent
arrays (no recursion limit on deep chains), variables are numbered, and the
C side compiles layouts, kernel task tables and the level schedule (`jtp_plan_create`).
"""

import ctypes as C
import json

import numpy as np

from . import _capi

__all__ = ["Plan", "flatten_tree", "plan_for", "cached_plan", "clear_plan_cache", "plan_cache_info", "set_plan_cache_budget"]


def flatten_tree(tree):
    """[clique, (sep, subtree), ...] -> (cliques in pre-order, parent, parent_sep, children).

    The format is the reference's (`README.md:50-66`); tuples and lists are interchangeable
    as in its tests.  Iterative.
    """
    order, parent, parent_sep, children = [], {}, {}, {}
    stack = [(tree, -1, -1)]
    while stack:
        sub, par, sep = stack.pop()
        c = sub[0]
        if c in parent:
            raise ValueError("clique index %r appears twice in the tree" % (c,))
        order.append(c)
        parent[c], parent_sep[c] = par, sep
        children[c] = [(entry[0], entry[1][0]) for entry in sub[1:]]
        for entry in reversed(sub[1:]):
            stack.append((entry[1], c, entry[0]))
    return order, parent, parent_sep, children


def _int_array(values):
    arr = (C.c_int32 * max(len(values), 1))()
    for i, v in enumerate(values):
        arr[i] = int(v)
    return arr


class Plan:
    """A compiled junction tree resident on one GPU (or, with `plan_only`, on the host).

    `node_vars[i]` lists the variable labels of node i of the caller's node list (cliques
    and separators in any numbering, as `compute_beliefs` allows); `sizes` maps a label to
    its cardinality.  `dtype` is the storage type of clique tables ("f32" or "f64");
    messages and accumulation are always float64.
    """

    def __init__(self, tree, node_vars, sizes, dtype="f64", device=0, n_batch=1,
                 n_ranks=1, rank=0, owner=None, plan_only=False, lds_budget=0, block_log2=0,
                 layout_policy=0, split_variants=False, keep_root=False, level_launches=False,
                 flow_tickets=False, share_potentials=False, multiset=False, no_compact=False, root=None, cover=None, fold=None):
        self._lib = _capi.lib()
        self._handle = C.c_void_p()
        order, parent, parent_sep, children = flatten_tree(tree)
        if root is not None and root != order[0]:
            # the same tree hung from another clique (partition.partition_tree: the weighted centroid): the edges on the
            # path root .. old root turn round, every edge keeps its separator
            if root not in parent:
                raise ValueError("root %r is not a clique of the tree" % (root,))
            prev, prev_sep, c = -1, -1, root
            while c != -1:
                nxt, nxt_sep = parent[c], parent_sep[c]
                parent[c], parent_sep[c] = prev, prev_sep
                prev, prev_sep, c = c, nxt_sep, nxt
            order = [root] + [c for c in order if c != root]
        cliques = sorted(order)
        seps = sorted(parent_sep[c] for c in order if parent[c] != -1)
        if set(cliques) & set(seps):
            raise ValueError("a node index is used both as clique and as separator")
        self.cliques, self.seps = cliques, seps
        self.node_ids = cliques + seps                    # ABI node number -> caller's index
        self.abi_of = {n: i for i, n in enumerate(self.node_ids)}
        self.n_cliques = len(cliques)
        self.tree_order = order
        self.parent = parent
        self.root = order[0]

        # More than 32 variables on a node (the C ABI's limit; the reference, through numpy.einsum, takes 52 labels): a
        # table is at most 2^31 entries, so all but 31 of them have cardinality 1 - such variables own no index bit and
        # are kept on the host only: the device sees the nodes without them, arrays lose / regain the length-1 axes by
        # a reshape.  (Only then: trees within the limit are passed on as they are.)
        try:
            wide = any(len(node_vars[n]) > _capi.MAX_VARS for n in self.node_ids)
            self._trivial = {lab for n in self.node_ids for lab in node_vars[n] if int(sizes[lab]) == 1} if wide else set()
        except KeyError as exc:                           # the reference raises KeyError too
            raise KeyError(exc.args[0])
        full_vars = node_vars
        node_vars = {n: [lab for lab in full_vars[n] if lab not in self._trivial] for n in self.node_ids}
        labels = {}
        for n in self.node_ids:
            for lab in node_vars[n]:
                labels.setdefault(lab, len(labels))
        self.var_id = labels
        self.var_labels = list(labels)
        try:
            self.card = [int(sizes[lab]) for lab in self.var_labels]
        except KeyError as exc:
            raise KeyError(exc.args[0])
        self.node_vars = {n: list(full_vars[n]) for n in self.node_ids}
        self.node_shape = {n: tuple(int(sizes[lab]) for lab in full_vars[n]) for n in self.node_ids}
        self.dtype = {"f32": _capi.JTP_F32, "f64": _capi.JTP_F64, np.float32: _capi.JTP_F32,
                      np.float64: _capi.JTP_F64}[dtype]
        self.n_batch = n_batch
        self.rank, self.n_ranks = rank, n_ranks
        self.device = int(device)

        off, ids = [0], []
        for n in self.node_ids:
            ids += [labels[lab] for lab in node_vars[n]]
            off.append(len(ids))
        par = [self.abi_of[parent[c]] if parent[c] != -1 else -1 for c in cliques]
        psep = [self.abi_of[parent_sep[c]] if parent[c] != -1 else -1 for c in cliques]
        self.owner = [0] * len(cliques) if owner is None else [int(owner[c]) for c in cliques]

        d = _capi.TreeDesc()
        d.struct_size = C.sizeof(_capi.TreeDesc)
        d.n_vars = len(self.card)
        self._keep = [_int_array(self.card), _int_array(off), _int_array(ids), _int_array(par),
                      _int_array(psep), _int_array(self.owner)]
        d.var_card, d.node_var_off, d.node_var_ids, d.parent_clique, d.parent_sep, d.clique_owner = \
            [C.cast(a, C.POINTER(C.c_int32)) for a in self._keep]
        # `cover`: which variables of each clique its potential depends on (the union of its factors' variables; the reference
        # leaves the others length-1 axes and never materialises them, junctiontree.py:52-61) - {clique: labels} or a list
        # indexed by the caller's clique index; None: every clique keeps a full table
        self.cover = None
        if cover is not None:
            self.cover = {c: [lab for lab in cover[c] if lab not in self._trivial] for c in cliques}
            coff, cids = [0], []
            for c in cliques:
                unknown = [lab for lab in self.cover[c] if lab not in node_vars[c]]
                if unknown:
                    raise ValueError("clique %r: covered variable %r is not one of its variables" % (c, unknown[0]))
                cids += [labels[lab] for lab in self.cover[c]]
                coff.append(len(cids))
            self._keep += [_int_array(coff), _int_array(cids)]
            d.cover_off, d.cover_ids = [C.cast(a, C.POINTER(C.c_int32)) for a in self._keep[-2:]]
        # `fold`: (clique of every request, labels of every request) - the marginals the caller will ask for after every propagate
        # (`factor_marginals` with the model's factors: junctiontree.py:264-274).  Requests on cliques that keep no table are then formed
        # inside the propagate's launch (jtp_tree_desc.fold_*); the list is built exactly as `_MarginalRequests` builds it.
        if fold is not None and not multiset and n_ranks == 1:
            f_cliques, f_labels = fold
            fv, fo = [], [0]
            for labs in f_labels:
                fv += [labels[lab] for lab in labs if lab not in self._trivial]
                fo.append(len(fv))
            self._keep += [_int_array([self.abi_of[c] for c in f_cliques] + [0]), _int_array(fo), _int_array(fv + [0])]
            d.fold_n = len(f_cliques)
            d.fold_cliques, d.fold_var_off, d.fold_var_ids = [C.cast(a, C.POINTER(C.c_int32)) for a in self._keep[-3:]]
        d.n_cliques = len(cliques)
        d.n_nodes = len(self.node_ids)
        d.dtype = self.dtype
        d.device = device
        d.n_batch = n_batch
        d.n_ranks = n_ranks
        d.rank = rank
        d.flags = ((_capi.JTP_PLAN_ONLY if plan_only else 0) | (_capi.JTP_SPLIT_VARIANTS if split_variants else 0)
                   | (_capi.JTP_KEEP_ROOT if keep_root else 0)
                   | (_capi.JTP_LEVEL_LAUNCHES if level_launches else 0)
                   | (_capi.JTP_FLOW_TICKETS if flow_tickets else 0)
                   | (_capi.JTP_SHARE_POTENTIALS if share_potentials or multiset else 0)
                   | (_capi.JTP_MULTISET if multiset else 0)
                   | (_capi.JTP_NO_COMPACT if no_compact else 0))
        self.multiset = bool(multiset)
        d.lds_budget = lds_budget
        d.block_log2 = block_log2
        d.layout_policy = layout_policy
        _capi.check(self._lib.jtp_plan_create(C.byref(d), C.byref(self._handle)))
        # what the tables are stored as: a float32 request whose layout cannot be planned (sub-boxes beyond the LDS of a CU:
        # cliques of few rows with four or more large separators) is made with float64 tables by jtp_plan_create
        self.requested_dtype = self.dtype
        st = _capi.Stats()
        _capi.check(self._lib.jtp_get_stats(self._handle, C.byref(st)))
        self.dtype = int(st.storage_dtype)
        if st.lean_refused:
            # (jtp_plan_create planned the tree again with every table materialised: say so, and do not pretend to be lean)
            import warnings
            self.cover = None
            warnings.warn("junctiontree_amd: the plan that keeps no table for uncovered variables was refused (%s); every clique "
                          "table is materialised on the device" % self.describe().get("lean_refused", "unsupported"),
                          RuntimeWarning, stacklevel=3)
        if self.dtype != self.requested_dtype:
            import warnings
            warnings.warn("junctiontree_amd: float32 tables asked for, float64 tables made (the float32 layout of this tree does "
                          "not fit the LDS of a CU): twice the device memory", RuntimeWarning, stacklevel=3)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.jtp_plan_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass

    def describe(self):
        return json.loads(self._lib.jtp_plan_describe(self._handle).decode())

    def owns(self, clique):
        """This rank holds the clique: it is the owner, or the clique is replicated (owner == n_ranks)."""
        return self.owner[self.abi_of[clique]] in (self.rank, self.n_ranks) or self.n_ranks == 1

    def _drop_trivial(self, arr, labels):
        """`arr` without the axes of the host-only (one-state) variables; such an axis has length 1, so this is a reshape."""
        for ax, lab in enumerate(labels):
            if lab in self._trivial and arr.shape[ax] != 1:
                raise ValueError("axis %d belongs to variable %r of cardinality 1 but has length %d" % (ax, lab, arr.shape[ax]))
        return arr.reshape([arr.shape[ax] for ax, lab in enumerate(labels) if lab not in self._trivial])

    # ------------------------------------------------------------------ data in
    def set_potential(self, node, array, batch=0):
        """Upload the potential of clique `node` (caller's index).  The array must have one
        axis per variable, each of the variable's cardinality or length 1 (broadcast)."""
        arr = np.asarray(array)
        if arr.dtype not in (np.float32, np.float64):
            arr = arr.astype(np.float64)
        arr = np.ascontiguousarray(arr).reshape(arr.shape)      # ascontiguousarray makes 0-d 1-d
        self._factor_tables = None                              # (what stage_factors believes the device holds is no longer true)
        full = self.node_shape[node]
        if arr.ndim != len(full):
            raise ValueError("potential of node %r has %d axes, its variable list has %d"
                             % (node, arr.ndim, len(full)))
        if self._trivial:
            arr = self._drop_trivial(arr, self.node_vars[node])
        shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
        host_dtype = _capi.JTP_F32 if arr.dtype == np.float32 else _capi.JTP_F64
        _capi.check(self._lib.jtp_set_potential(self._handle, batch, self.abi_of[node],
                                                arr.ctypes.data_as(C.c_void_p), shape, host_dtype))
        # uploads are asynchronous (two staging buffers, no stream synchronisation per call): a page-locked
        # source array must stay alive until the copy has run - hold the last few
        keep = self.__dict__.setdefault("_uploads", [])
        keep.append(arr)
        del keep[:-4]

    def set_potential_product(self, node, arrays, var_lists, batch=0):
        """Potential of clique `node` = product of factor tables, formed on the device in the clique's
        layout (`CliqueGraph.evaluate` for one clique, `junctiontree.py:203-226`): only the factor
        tables are uploaded.  `var_lists[i]` labels the axes of `arrays[i]`; an axis may have length
        1 to broadcast.  No factors gives an all-ones potential."""
        self._factor_tables = None
        keep, recs = [], (_capi.Factor * max(len(arrays), 1))()
        for i, (arr, labels) in enumerate(zip(arrays, var_lists)):
            a = np.asarray(arr)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float64)
            a = np.ascontiguousarray(a).reshape(a.shape)
            if a.ndim != len(labels):
                raise ValueError("factor %d has %d axes but %d variables" % (i, a.ndim, len(labels)))
            if self._trivial:
                a = self._drop_trivial(a, labels)
                labels = [lab for lab in labels if lab not in self._trivial]
            ids = _int_array([self.var_id[lab] for lab in labels])
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            keep += [a, ids, shape]
            recs[i].host = a.ctypes.data_as(C.c_void_p)
            recs[i].n_vars = a.ndim
            recs[i].dtype = _capi.JTP_F32 if a.dtype == np.float32 else _capi.JTP_F64
            recs[i].var_ids = C.cast(ids, C.POINTER(C.c_int32))
            recs[i].shape = C.cast(shape, C.POINTER(C.c_int64))
        _capi.check(self._lib.jtp_set_potential_product(self._handle, batch, self.abi_of[node], len(arrays), recs))

    def stage_factors(self, factor_labels, factor_to_clique, xs, changed=None):
        """`CliqueGraph.evaluate` (`junctiontree.py:203-226`) on the device for a whole factor graph: clique c's potential is
        the product of the tables `xs[f]` with `factor_to_clique[f] == c` (labels `factor_labels[f]`, one axis each; an axis
        may have length 1 to broadcast).  ONE call into the library for all cliques whose tables differ from what this
        plan was last staged with (`jtp_set_potential_products`: one host-to-device copy, one kernel launch); the reference
        recomputes every clique on every call and says so in a FIXME (`junctiontree.py:206-214`).  Returns the number of
        cliques formed (also kept as `staged_cliques`).

        `changed`: None - every table is compared with the values last staged; "all" or the indices of the factors whose
        tables are new - nothing is compared, only the cliques of those factors are formed again, and the caller vouches that
        the factor structure (labels, assignment, shapes, dtypes) is what it was at the last call with these lists."""
        ft = self.__dict__.get("_factor_tables")
        if (changed is not None and ft is not None and ft.src[0] is factor_labels and ft.src[1] is factor_to_clique
                and ft.n_f == len(xs) and ft.prev is not None):
            self.staged_cliques = ft.stage(self, xs, changed)       # (what is looked at is converted there)
            return self.staged_cliques
        arrs = [x if type(x) is np.ndarray else np.asarray(x) for x in xs]
        shapes = [a.shape for a in arrs]
        all_f32 = all(a.dtype == np.float32 for a in arrs)
        # (the structure is compared by VALUE on every call - against list copies, with list.__eq__: a tuple key of 1831 label lists
        #  built anew per call was 0.3 ms of config 3's 7)
        if ft is None or not (ft.all_f32 == all_f32 and ft.shapes == shapes and ft.f2c == list(factor_to_clique) and _same_lists(ft.labels, factor_labels)):
            ft = self._factor_tables = _FactorTables(self, (tuple(map(tuple, factor_labels)), tuple(factor_to_clique), shapes, all_f32))
        ft.src = (factor_labels, factor_to_clique)
        self.staged_cliques = ft.stage(self, arrs)
        return self.staged_cliques

    def factor_marginals(self, factor_labels, factor_to_clique, batch=0, trusted=False):
        """`CliqueGraph.marginalize` (`junctiontree.py:229-274`) on the device: the marginal of clique
        `factor_to_clique[f]`'s belief onto `factor_labels[f]` for every factor, as float64 arrays (views of one buffer).
        `trusted`: the two lists are the objects of the last call and have not been modified (no look at their contents)."""
        req = self.__dict__.get("_marginal_requests")
        if trusted and req is not None and req.src[0] is factor_labels and req.src[1] is factor_to_clique:
            return req.read(self, batch)
        if req is None or not (req.f2c == list(factor_to_clique) and _same_lists(req.labels, factor_labels)):
            req = self._marginal_requests = _MarginalRequests(self, (tuple(map(tuple, factor_labels)), tuple(factor_to_clique)))
        req.src = (factor_labels, factor_to_clique)
        return req.read(self, batch)

    def set_evidence(self, observed, batch=0):
        """Hard evidence of evidence set `batch`: `observed` maps variable label -> observed state; it
        replaces the set's previous evidence ({} clears it) and applies from the next propagate."""
        labels = list(observed)
        if self._trivial:
            for lab in labels:
                if lab in self._trivial and int(observed[lab]) != 0:
                    raise ValueError("variable %r has one state: observed state %r" % (lab, observed[lab]))
            labels = [lab for lab in labels if lab not in self._trivial]
        ids = _int_array([self.var_id[lab] for lab in labels])
        states = _int_array([int(observed[lab]) for lab in labels])
        _capi.check(self._lib.jtp_set_evidence(self._handle, batch, len(labels), C.cast(ids, C.POINTER(C.c_int32)),
                                               C.cast(states, C.POINTER(C.c_int32))))

    def fill_synthetic(self, seed, scales=None, batch=0):
        """Device-side counter-based potentials (see synthetic.synth_values).  `scales` is
        indexed by the caller's clique index."""
        self._factor_tables = None
        sc = None
        if scales is not None:
            sc = (C.c_double * self.n_cliques)(*[float(scales[c]) for c in self.cliques])
        # node keys are the ABI clique numbers; callers that compare with
        # synthetic.synth_values must number cliques 0..N-1 (all recipes do)
        _capi.check(self._lib.jtp_fill_synthetic(self._handle, batch, seed, sc))

    # ------------------------------------------------------------------ compute
    def propagate(self, batch_begin=0, batch_end=None, sync=True):
        end = self.n_batch if batch_end is None else batch_end
        _capi.check(self._lib.jtp_propagate(self._handle, batch_begin, end))
        if sync:
            self.sync()

    def sync(self):
        _capi.check(self._lib.jtp_sync(self._handle))
        self.__dict__.pop("_uploads", None)

    # ------------------------------------------------------------------ data out
    def belief(self, node, batch=0, dtype=np.float64, out=None):
        """Belief of node `node` (clique or separator) in the caller's axis order.  `out`: a C-contiguous
        float32/float64 array of the node's shape to receive it (e.g. from `pinned_empty`)."""
        if out is None:
            out = np.empty(self.node_shape[node], dtype=dtype)
        elif out.shape != tuple(self.node_shape[node]) or not out.flags["C_CONTIGUOUS"] or out.dtype not in (np.float32, np.float64):
            raise ValueError("`out` must be a C-contiguous float32/float64 array of shape %r" % (tuple(self.node_shape[node]),))
        host_dtype = _capi.JTP_F32 if out.dtype == np.float32 else _capi.JTP_F64
        _capi.check(self._lib.jtp_get_belief(self._handle, batch, self.abi_of[node],
                                             out.ctypes.data_as(C.c_void_p), host_dtype))
        return out

    def marginal(self, clique, labels, batch=0):
        """Marginal of the clique belief onto `labels` (in that axis order), float64."""
        shape = tuple(1 if lab in self._trivial else self.card[self.var_id[lab]] for lab in labels)
        labels = [lab for lab in labels if lab not in self._trivial]
        ids = _int_array([self.var_id[lab] for lab in labels])
        out = np.empty(shape, dtype=np.float64)
        _capi.check(self._lib.jtp_get_marginal(self._handle, batch, self.abi_of[clique],
                                               C.cast(ids, C.POINTER(C.c_int32)), len(labels),
                                               out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def marginals(self, requests, batch=0):
        """`requests` = [(clique, labels), ...] -> list of float64 arrays: every marginal from one
        kernel launch and one copy back (`jtp_get_marginals`)."""
        n = len(requests)
        if n == 0:
            return []
        cliques = _int_array([self.abi_of[c] for c, _ in requests])
        var_off, var_ids, out_off, shapes = [0], [], [0], []
        for _, labels in requests:
            ids = [self.var_id[lab] for lab in labels if lab not in self._trivial]
            var_ids += ids
            var_off.append(len(var_ids))
            shape = tuple(1 if lab in self._trivial else self.card[self.var_id[lab]] for lab in labels)
            shapes.append(shape)
            out_off.append(out_off[-1] + int(np.prod(shape, dtype=np.int64)) if shape else out_off[-1] + 1)
        flat = np.empty(out_off[-1], dtype=np.float64)
        offs = (C.c_int64 * (n + 1))(*out_off)
        _capi.check(self._lib.jtp_get_marginals(
            self._handle, batch, n, C.cast(cliques, C.POINTER(C.c_int32)),
            C.cast(_int_array(var_off), C.POINTER(C.c_int32)), C.cast(_int_array(var_ids), C.POINTER(C.c_int32)),
            C.cast(offs, C.POINTER(C.c_int64)), flat.ctypes.data_as(C.POINTER(C.c_double))))
        return [flat[out_off[i]:out_off[i + 1]].reshape(shapes[i]).copy() for i in range(n)]

    def z(self, batch=0):
        val = C.c_double(0.0)
        _capi.check(self._lib.jtp_get_z(self._handle, batch, C.byref(val)))
        return val.value

    # ------------------------------------------------------------------ instrumentation
    def set_profiling(self, keep=1, per_launch=False, stride=1):
        """Time the next `keep` propagates with hipEvents on the plan's stream: three events per
        propagate (phase times), or with `per_launch` a pair around every launch; `stride` > 1 times only
        every stride-th propagate (the events cost 2-3 us of GPU time each)."""
        _capi.check(self._lib.jtp_set_profiling_granularity(self._handle, 1 if per_launch else 0))
        _capi.check(self._lib.jtp_set_profiling_stride(self._handle, int(stride)))
        _capi.check(self._lib.jtp_set_profiling(self._handle, int(keep)))

    def region_begin(self):
        """One hipEvent on the plan's stream now; `region_end` records a second one and returns the device time between
        them in ms (a benchmark's timed steps: no event between the propagates)."""
        _capi.check(self._lib.jtp_region_begin(self._handle))

    def region_end(self):
        ms = C.c_double(0.0)
        _capi.check(self._lib.jtp_region_end(self._handle, C.byref(ms)))
        return ms.value

    def debug_set(self, knob, value):
        """Test hook (`jtp_debug_set`): e.g. ("flow_debug", 8) makes every dataflow wait time out."""
        _capi.check(self._lib.jtp_debug_set(self._handle, knob.encode(), int(value)))

    def launch_ms(self):
        """Mean device time of every launch of the schedule (ms), with its description."""
        d = self.describe()
        n = len(d["launches"])
        buf = (C.c_double * max(n, 1))()
        rc = self._lib.jtp_get_launch_ms(self._handle, buf, n)
        if rc < 0:
            _capi.check(rc)
        return [dict(phase=L["phase"], level=L["level"], variant=L["variant"], nblocks=L["nblocks"],
                     ntasks=len(L["tasks"]), alg_bytes=L["alg_bytes"], ms=buf[i])
                for i, L in enumerate(d["launches"])]

    def stats(self):
        st = _capi.Stats()
        _capi.check(self._lib.jtp_get_stats(self._handle, C.byref(st)))
        tname = "float" if self.dtype == _capi.JTP_F32 else "double"
        kernels = {}
        for v in range(_capi.N_VARIANTS):
            if st.kernel_launches[v]:
                name = self._lib.jtp_kernel_name(v).decode().replace("<T", "<" + tname)
                k = kernels.setdefault(name, {"launches": 0, "ms": 0.0, "bytes": 0.0})      # (two variants may share a kernel)
                k["launches"] += st.kernel_launches[v]
                k["ms"] += st.kernel_ms[v]
                k["bytes"] += st.kernel_bytes[v]
        return {"n_launches": st.n_launches, "n_messages": st.n_messages, "n_tasks": st.n_tasks,
                "algorithmic_bytes": st.algorithmic_bytes, "collect_ms": st.collect_ms,
                "distribute_ms": st.distribute_ms, "kernels": kernels, "flow_fallbacks": st.flow_fallbacks,
                "launch_mode": ("level", "flow", "flow_tickets")[st.launch_mode], "tickets_used": st.tickets_used,
                "flow_propagates": st.flow_propagates, "device_bytes": st.device_bytes, "f64_flops": st.f64_flops, "f64_insts": st.f64_insts,
                "foreign_seen": st.foreign_seen, "algorithmic_bytes_full": st.algorithmic_bytes_full, "fixed_bytes": st.fixed_bytes,
                "n_unit_cliques": st.n_unit_cliques, "n_static_tables": st.n_static_tables, "flight_board": st.flight_board}


_DIGEST_LIMIT = 1 << 20          # bytes: larger factor tables are handed over again on every call rather than compared

# numpy view of an array of jtp_factor records (_capi.Factor): the host pointers of all factors are written in one go
_FACTOR_REC = np.dtype({"names": ["host", "n_vars", "dtype", "var_ids", "shape"], "formats": ["u8", "i4", "i4", "u8", "u8"],
                        "offsets": [0, 8, 12, 16, 24], "itemsize": C.sizeof(_capi.Factor)})


class _FactorTables:
    """The argument tables of `jtp_set_potential_products` for one factor graph on one plan, built once, and the factor
    values the plan was last staged with (small tables: one concatenated copy, compared element by element on the next
    call, so that arrays updated in place are seen; tables above `_DIGEST_LIMIT` always count as changed)."""

    def __init__(self, plan, key):
        labels, f2c, shapes, all_f32 = key
        self.key = key
        # (what later calls are compared with: plain lists, so that `==` against the caller's lists runs in C)
        self.labels, self.f2c, self.shapes, self.all_f32 = [list(l) for l in labels], list(f2c), list(shapes), all_f32
        self.np_dtype = np.float32 if all_f32 else np.float64
        n_f = len(labels)
        if len(f2c) != n_f or len(shapes) != n_f:
            raise ValueError("%d factors, %d clique assignments, %d value arrays" % (n_f, len(f2c), len(shapes)))
        pos_of = {c: i for i, c in enumerate(plan.cliques)}          # caller's clique index -> place in this plan's clique list
        var_ids, var_off, shape_all = [], [0], []
        sizes = np.empty(n_f, dtype=np.int64)
        for f in range(n_f):
            if len(shapes[f]) != len(labels[f]):
                raise ValueError("factor %d has %d axes but %d variables" % (f, len(shapes[f]), len(labels[f])))
            for ax, lab in enumerate(labels[f]):
                if lab in plan._trivial:
                    if shapes[f][ax] != 1:
                        raise ValueError("axis %d of factor %d belongs to variable %r of cardinality 1 but has length %d"
                                         % (ax, f, lab, shapes[f][ax]))
                    continue
                try:
                    var_ids.append(plan.var_id[lab])
                except KeyError as exc:                              # (the reference raises KeyError too)
                    raise KeyError(exc.args[0])
                shape_all.append(shapes[f][ax])
            var_off.append(len(var_ids))
            sizes[f] = int(np.prod(shapes[f], dtype=np.int64)) if len(shapes[f]) else 1
        self._var_ids = np.asarray(var_ids + [0], dtype=np.int32)
        self._shape = np.asarray(shape_all + [0], dtype=np.int64)
        var_off = np.asarray(var_off, dtype=np.int64)
        small = sizes * np.dtype(self.np_dtype).itemsize <= _DIGEST_LIMIT
        self.small_idx = np.flatnonzero(small)
        self.big_idx = np.flatnonzero(~small)
        self.small_off = np.zeros(n_f + 1, dtype=np.int64)           # element offset of every small table in the copy
        np.cumsum(np.where(small, sizes, 0), out=self.small_off[1:])
        self.seg_start = self.small_off[self.small_idx]
        self.f_pos = np.asarray([pos_of[c] for c in f2c], dtype=np.int64)     # factor -> place of its clique
        self.order = np.argsort(self.f_pos, kind="stable")           # factors clique by clique
        n_c = len(plan.cliques)
        self.cliques = np.asarray([plan.abi_of[c] for c in plan.cliques], dtype=np.int32)
        self.factor_off = np.zeros(n_c + 1, dtype=np.int32)
        np.cumsum(np.bincount(self.f_pos, minlength=n_c), out=self.factor_off[1:])
        recs = np.zeros(max(n_f, 1), dtype=_FACTOR_REC)
        o = self.order
        recs["n_vars"][:n_f] = (var_off[1:] - var_off[:-1])[o]
        recs["dtype"][:n_f] = _capi.JTP_F32 if all_f32 else _capi.JTP_F64
        recs["var_ids"][:n_f] = self._var_ids.ctypes.data + 4 * var_off[:-1][o]
        recs["shape"][:n_f] = self._shape.ctypes.data + 8 * var_off[:-1][o]
        self.recs = recs
        self.n_f = n_f
        self.prev = None             # the small tables as last staged
        # every table small, of one shape, at least one axis: `stage` joins them without touching each
        self.all_small_alike = bool(n_f > 0 and len(self.big_idx) == 0 and len(shapes[0]) >= 1 and all(tuple(sh) == tuple(shapes[0]) for sh in shapes))
        self.src = (None, None)      # the caller's list objects this was last used with (Plan.stage_factors)
        self.is_small = small
        self.mine = np.asarray([plan.owns(c) for c in plan.cliques], dtype=bool)

    def stage(self, plan, arrs, named=None):
        """`named`: None = compare every small table with the copy last staged; "all" / factor indices = the caller names the
        tables that changed (`JunctionTree.propagate(xs, changed=...)`): nothing is compared."""
        n_f, item = self.n_f, np.dtype(self.np_dtype).itemsize
        dirty = np.ones(len(self.cliques), dtype=bool)
        if named is not None and self.prev is not None and not isinstance(named, str):
            # the copy last staged is brought up to date in place: only the named tables are looked at
            idx = np.unique(np.asarray(list(named), dtype=np.int64))
            if len(idx) and (idx[0] < 0 or idx[-1] >= n_f):
                raise IndexError("changed: factor index out of range [0, %d)" % n_f)
            flat = self.prev
            for i in idx:
                if self.is_small[i]:
                    a = np.asarray(arrs[i])
                    if a.size != self.small_off[i + 1] - self.small_off[i]:
                        raise ValueError("factor %d changed its shape" % i)
                    flat[self.small_off[i]:self.small_off[i + 1]] = a.reshape(-1)
            dirty[:] = False
            dirty[self.f_pos[idx]] = True
            dirty[self.f_pos[self.big_idx]] = True
        else:
            if named is not None and named != "all" and isinstance(named, str):
                raise ValueError('changed: "all" or an iterable of factor indices')
            if self.all_small_alike:
                # (a pairwise model: 1831 tables of one shape - joined along their first axis, which is the same memory as the
                #  flattened tables one after the other, without a reshape per table)
                flat = np.concatenate(arrs, dtype=self.np_dtype).reshape(-1)
            elif len(self.small_idx):
                flat = np.concatenate([np.asarray(arrs[i]).reshape(-1) for i in self.small_idx], dtype=self.np_dtype)
            else:
                flat = np.empty(0, dtype=self.np_dtype)
            if len(flat) != self.small_off[-1]:
                raise ValueError("the factor tables changed their shapes")
            if self.prev is not None and named is None:
                changed = np.zeros(n_f, dtype=bool)
                changed[self.big_idx] = True
                if len(flat):
                    if len(flat) != len(self.prev):
                        raise ValueError("the factor tables changed their shapes")
                    changed[self.small_idx] = np.logical_or.reduceat(flat != self.prev, self.seg_start)
                dirty[:] = False
                dirty[self.f_pos[changed]] = True
        dirty &= self.mine                                           # (sharded plans: this rank's cliques only)
        todo = np.flatnonzero(dirty)
        if len(todo) == 0:
            self.prev = flat
            return 0
        host = np.zeros(max(n_f, 1), dtype=np.uint64)
        host[self.small_idx] = flat.ctypes.data + item * self.small_off[self.small_idx]
        keep = []
        for i in self.big_idx:
            a = np.asarray(arrs[i])
            a = np.ascontiguousarray(a, dtype=np.float32 if a.dtype == np.float32 else np.float64)
            keep.append(a)
            host[i] = a.ctypes.data
        recs = self.recs
        recs["host"][:n_f] = host[:n_f][self.order]
        for i in self.big_idx:                                        # (their type is their own: float32 stays float32)
            recs["dtype"][np.flatnonzero(self.order == i)] = _capi.JTP_F32 if np.asarray(arrs[i]).dtype == np.float32 else _capi.JTP_F64
        fo = self.factor_off
        if len(todo) == len(self.cliques):
            cl, off, rr = self.cliques, fo, recs
        else:
            # (the same subset call after call - the cliques that hold factors, all of them new: its index tables are kept)
            memo = self.__dict__.get("_subset")
            if memo is None or len(memo[0]) != len(todo) or not np.array_equal(memo[0], todo):
                cl = np.ascontiguousarray(self.cliques[todo])
                counts = (fo[1:] - fo[:-1])[todo]
                off = np.zeros(len(todo) + 1, dtype=np.int32)
                np.cumsum(counts, out=off[1:])
                pick = np.concatenate([np.arange(fo[c], fo[c + 1]) for c in todo]) if off[-1] else np.zeros(0, dtype=np.int64)
                memo = self._subset = (todo.copy(), cl, off, pick)
            _, cl, off, pick = memo
            rr = np.ascontiguousarray(recs[pick]) if off[-1] else recs[:1].copy()
        self.prev = None                                             # (whatever happens below, the device no longer matches it)
        _capi.check(plan._lib.jtp_set_potential_products(plan._handle, 0, len(todo), cl.ctypes.data, off.ctypes.data, rr.ctypes.data))
        self.prev = flat
        return int(len(todo))


class _MarginalRequests:
    """The argument tables of `jtp_get_marginals` for one factor graph on one plan, built once."""

    def __init__(self, plan, key):
        labels, f2c = key
        self.key = key
        self.labels, self.f2c = [list(l) for l in labels], list(f2c)
        self.src = (None, None)      # the caller's list objects this was last used with (Plan.factor_marginals)
        n = len(labels)
        var_ids, var_off, out_off, self.shapes = [], [0], [0], []
        for labs in labels:
            var_ids += [plan.var_id[lab] for lab in labs if lab not in plan._trivial]
            var_off.append(len(var_ids))
            shape = tuple(1 if lab in plan._trivial else plan.card[plan.var_id[lab]] for lab in labs)
            self.shapes.append(shape)
            out_off.append(out_off[-1] + (int(np.prod(shape, dtype=np.int64)) if shape else 1))
        self.n = n
        self.cliques = np.asarray([plan.abi_of[c] for c in f2c] + [0], dtype=np.int32)
        self.var_off = np.asarray(var_off, dtype=np.int32)
        self.var_ids = np.asarray(var_ids + [0], dtype=np.int32)
        self.out_off = np.asarray(out_off, dtype=np.int64)
        self.bounds = out_off
        # (runs of equally shaped results are cut out of the buffer with one reshape: a pairwise model has thousands)
        self.uniform = len(set(self.shapes)) == 1 and n > 0 and len(self.shapes[0]) > 0
        self._args = (self.cliques.ctypes.data_as(C.POINTER(C.c_int32)), self.var_off.ctypes.data_as(C.POINTER(C.c_int32)),
                      self.var_ids.ctypes.data_as(C.POINTER(C.c_int32)), self.out_off.ctypes.data_as(C.POINTER(C.c_int64)))

    def read(self, plan, batch):
        if self.n == 0:
            return []
        flat = np.empty(self.bounds[-1], dtype=np.float64)
        _capi.check(plan._lib.jtp_get_marginals(plan._handle, batch, self.n, *self._args, flat.ctypes.data_as(C.POINTER(C.c_double))))
        if self.uniform:
            return list(flat.reshape((self.n,) + self.shapes[0]))
        b = self.bounds
        return [flat[b[i]:b[i + 1]].reshape(self.shapes[i]) for i in range(self.n)]


def pinned_empty(shape, dtype=np.float64):
    """numpy array in page-locked host memory (jtp_host_alloc): potentials passed from it and beliefs
    read into it (`Plan.belief(..., out=...)`) move at PCIe speed.  Freed with the array."""
    import weakref
    lib = _capi.lib()
    dt = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
    ptr = C.c_void_p()
    _capi.check(lib.jtp_host_alloc(C.byref(ptr), n * dt.itemsize))
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr.value)
    arr = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
    weakref.finalize(buf, lib.jtp_host_free, ptr.value)
    return arr


# ---------------------------------------------------------------------- plan cache
# `compute_beliefs` and `JunctionTree.propagate` are stateless calls in the reference; here the compiled plan behind
# them (device arenas, task tables) is kept for the next call on the same structure.  The cache is bounded by device
# BYTES, least recently used first out (round 2 kept up to 16 plans whatever their size: 2 GiB each for a config-4
# sized tree): by default a quarter of the device's memory, at most 16 plans.

_cache = {}                       # key -> Plan, least recently used first
_cache_stats = {"hits": 0, "misses": 0, "evictions": 0}
_cache_budget = None              # bytes per device; None: a quarter of the device's total memory, asked for at first use
_device_budget = {}               # device -> that quarter
_CACHE_MAX_PLANS = 16


def set_plan_cache_budget(nbytes):
    """Bound the device memory the cached plans may hold together (None: back to the default, a quarter of the
    device).  Plans beyond it are forgotten least recently used first."""
    global _cache_budget
    _cache_budget = None if nbytes is None else int(nbytes)
    _evict(keep=None)


def _budget(device=0):
    """Bytes the cached plans of one device may hold together: what set_plan_cache_budget said, else a quarter of THAT
    device's memory (asked for once per device)."""
    if _cache_budget is not None:
        return _cache_budget
    if device not in _device_budget:
        free, total = C.c_uint64(0), C.c_uint64(0)
        rc = _capi.lib().jtp_device_memory(int(device), C.byref(free), C.byref(total))
        _device_budget[device] = int(total.value // 4) if rc == _capi.JTP_OK and total.value else 1 << 62
    return _device_budget[device]


def _plan_bytes(plan):
    if "_device_bytes" not in plan.__dict__:
        plan._device_bytes = int(plan.stats()["device_bytes"])
    return plan._device_bytes


def _evict(keep):
    """Forget least recently used plans until every device's cached plans fit its budget.  A forgotten plan is destroyed when
    its last holder lets go (JunctionTree.plan() hands these objects out: closing here could pull a plan from under its user)."""
    while _cache:
        per_device = {}
        for p in _cache.values():
            per_device[p.device] = per_device.get(p.device, 0) + _plan_bytes(p)
        over = [dev for dev, b in per_device.items() if b > _budget(dev)]
        if not over and len(_cache) <= _CACHE_MAX_PLANS:
            break
        victim = next((k for k, p in _cache.items() if k is not keep and (not over or p.device in over)), None)
        if victim is None:
            break                     # the plan just asked for is larger than the budget on its own: it stays
        _cache.pop(victim)
        _cache_stats["evictions"] += 1


def plan_cache_info():
    """{"plans", "device_bytes", "budget_bytes", "hits", "misses", "evictions", "widened"} of the plan cache ("widened": cached
    plans asked for with float32 tables and made with float64 ones; budget_bytes: per device, of the first cached plan's)."""
    first = next(iter(_cache.values()), None)
    return dict(_cache_stats, plans=len(_cache), device_bytes=sum(_plan_bytes(p) for p in _cache.values()),
                widened=sum(1 for p in _cache.values() if p.dtype != p.requested_dtype),
                budget_bytes=_budget(first.device) if first is not None else _cache_budget)


def _same_lists(stored, given):
    """`given` (the caller's label lists, or any sequence of sequences) names what `stored` (a list of lists) does.  The common case -
    lists of lists - is one C-level `==`; anything else is converted first."""
    if given == stored:
        return True
    try:
        return [list(x) for x in given] == stored
    except TypeError:
        return False


class _Key:
    """A plan-cache key: the structure tuple with its hash taken once (Python hashes a tuple anew on every dictionary operation -
    for a tree of a thousand cliques 0.03 ms each, three per `cached_plan`)."""
    __slots__ = ("t", "h")

    def __init__(self, t):
        self.t = t
        self.h = hash(t)

    def __hash__(self):
        return self.h

    def __eq__(self, other):
        return self is other or (isinstance(other, _Key) and self.h == other.h and self.t == other.t)


def _freeze(tree):
    order, parent, parent_sep, _ = flatten_tree(tree)
    return tuple((c, parent[c], parent_sep[c]) for c in order)


def cached_plan(key, plan):
    """`plan` if it still is the cache's entry under `key` (then the most recently used one), else None: lets a caller that kept
    (key, weak reference) from `plan_for(..., return_key=True)` skip building the key again - for a tree of a thousand cliques
    that is 1-2 ms of tuple building per call."""
    if plan is None or _cache.get(key) is not plan or not plan._handle:
        return None
    _cache_stats["hits"] += 1
    _cache[key] = _cache.pop(key)         # most recently used last
    return plan


def plan_for(tree, node_vars, sizes, dtype, return_key=False, **kwargs):
    """Return a cached Plan for this structure (plans are expensive relative to tiny trees:
    device allocations and a task-table upload)."""
    order, parent, parent_sep, _ = flatten_tree(tree)
    used = list(order) + [parent_sep[c] for c in order if parent[c] != -1]
    labels = set(lab for n in used for lab in node_vars[n])
    cover = kwargs.get("cover")
    key = _Key((_freeze(tree), tuple((n, tuple(node_vars[n])) for n in used),
                tuple(sorted((repr(k), int(sizes[k])) for k in labels)),
                dtype, tuple(sorted((k, v) for k, v in kwargs.items() if k not in ("cover", "fold"))),
                None if cover is None else tuple(tuple(cover[c]) for c in order),
                # (`fold` - the marginals named ahead - is a hint: a plan made with one list serves every other list by the read-out,
                #  so two models over one tree still share a plan; only WHETHER one was given keys the cache)
                kwargs.get("fold") is not None))
    plan = _cache.pop(key, None)
    if plan is not None:
        _cache_stats["hits"] += 1
        _cache[key] = plan                # most recently used last
        return (plan, key) if return_key else plan
    _cache_stats["misses"] += 1
    try:
        plan = Plan(tree, node_vars, sizes, dtype=dtype, **kwargs)
    except MemoryError:
        # the device is full: let go of every cached plan (those nobody else holds are destroyed now) and try once more
        import gc
        _cache_stats["evictions"] += len(_cache)
        _cache.clear()
        gc.collect()
        plan = Plan(tree, node_vars, sizes, dtype=dtype, **kwargs)
    _cache[key] = plan
    _evict(keep=key)
    return (plan, key) if return_key else plan


def clear_plan_cache():
    for plan in _cache.values():
        plan.close()
    _cache.clear()
