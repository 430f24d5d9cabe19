"""Twin sections of a dual-stream cfg, and the pairing of their commands into two-problem launches.

The reference builds its two backbones as the same layers on two inputs (models.py:288,299-303: sections
[0, second_index) run on the visible image, [second_index, ...) on the LWIR image, and every later stage of the
Fshare cfgs repeats the construction: cfg sections 114-144 / 147-177, 182-200 / 203-221 of
kaist_dyolov4_fshare_global_concat_se3.cfg).  Two sections are TWINS when they have the same definition and their
inputs are twins (or the same section) all the way up to the images: their kernels have identical shapes and are
mutually independent, so one launch can carry both (DykConvDesc.twin and friends, include/dyk_hip.h).

    twin_layers()   cfg level: section index -> its twin (structural hash over the section DAG)
    find_pairs()    command level: commands of twin sections with equal descriptors (every non-pointer field) that
                    the dependency graph leaves unordered
    merge()         dependency graph with every pair contracted into one node, renumbered topologically
"""
import ctypes
import heapq

from . import lib as L

_IMG = ("image",)


def _section_signature(md):
    """what makes two cfg sections the same operator: every key of the definition but the references to other
    sections (those are compared through the inputs)"""
    skip = {"layers", "from", "type"}
    items = []
    for k in sorted(md):
        if k in skip:
            continue
        v = md[k]
        try:
            v = tuple(v.reshape(-1).tolist()) if hasattr(v, "reshape") else (tuple(v) if isinstance(v, (list, tuple)) else v)
        except Exception:
            v = str(v)
        items.append((k, v))
    return (md["type"], tuple(items))


def section_inputs(defs, mods, second):
    """inputs[i] = the sections (or _IMG) section i reads, in operand order"""
    out = []
    for i, md in enumerate(defs):
        t = md["type"]
        if t == "route":
            out.append([int(q) for q in mods[i].layers])
        elif t == "shortcut":
            out.append([i - 1] + [int(q) for q in mods[i].layers])
        elif i == 0 or (second is not None and i == second):
            out.append([_IMG])
        else:
            out.append([i - 1])
    return out


def twin_layers(defs, mods, second):
    """{i: j, j: i} for every pair of sections with the same definition and structurally identical ancestry (a section
    is never its own twin)"""
    if second is None:
        return {}
    inputs = section_inputs(defs, mods, second)
    h, alias = [], set()
    for i, md in enumerate(defs):
        if md["type"] == "route" and len(inputs[i]) == 1:
            h.append(h[inputs[i][0]])         # a single-source [route] IS its source (no commands of its own)
            alias.add(i)
            continue
        key = (_section_signature(md), tuple(_IMG if q is _IMG else h[q] for q in inputs[i]))
        h.append(hash(key))
    groups = {}
    for i, v in enumerate(h):
        if i not in alias:
            groups.setdefault(v, []).append(i)
    twins = {}
    for g in groups.values():
        # two sections per group in the plain case; 2m when a backbone holds m sections of equal definition AND equal
        # input (the two 1x1 convs that open a CSP block): ascending order puts backbone 1 first, so k-th pairs with k-th
        if len(g) % 2 == 0:
            m = len(g) // 2
            for k in range(m):
                twins[g[k]], twins[g[k + m]] = g[k + m], g[k]
    return twins


# ------------------------------------------------------------------------------------------------ command level
def _bytes(d, cls, first, last):
    lo, hi = getattr(cls, first).offset, getattr(cls, last).offset
    return ctypes.string_at(ctypes.addressof(d) + lo, hi - lo)


def _nulls(d, names):
    return tuple(bool(getattr(d, n)) for n in names)


_EW_PTRS = ("a", "b", "out", "p0", "p1", "p2", "p3", "red", "aux", "aux2")
_FIN_PTRS = ("stats", "gamma", "beta", "running_mean", "running_var", "scale", "shift", "save_mean", "save_rstd")


def pair_signature(op, d, plan):
    """hashable summary of everything two commands must share to run as one two-problem launch (the conditions the
    C side checks: dyk_conv_igemm / dyk_conv_wgrad / dyk_fill_ew_pair / dyk_fill_fin_pair), None = op is not pairable"""
    if op == L.OP_CONV:
        if d.flags & L.EPI_BNFWD:
            return None                     # a one-launch conv + BatchNorm waits on its own workgroups: single problem only
        if d.splitk > 1:
            return None                     # split-K across workgroups (private slabs and tile counters): single problem only
        if ((d.tune >> 12) & 0xf) == 7:
            return None                     # persistent pointwise kernels (csrc/conv_pw_kernel.h): single problem only
        if ((d.tune >> 12) & 0xf) == 6:
            return None                     # resident-weight data gradient (csrc/conv_sc.hip): single problem only -- as one half of a
                                            # two-problem launch it would fall back to the generic kernel, whose epilogue rounds differently
        return (op, _bytes(d, L.DykConvDesc, "dtype", "twin"), _nulls(d, ("scale", "shift", "res", "stats", "aux0", "aux1", "add")))
    if op == L.OP_WGRAD:
        if ((d.tune >> 28) & 7) in (2, 3):
            return None                     # row-block 3x3 / pixel-streaming 1x1 kernels (csrc/conv_wgrad_rb.hip, conv_wgrad_ps.hip): single problem only
        return (op, _bytes(d, L.DykWgradDesc, "part_stride", "twin"), bool(d.part))
    if op in (L.OP_BN_ACT_FWD, L.OP_BN_BWD_REDUCE, L.OP_BN_BWD_APPLY, L.OP_AXPBY):
        return (op, _bytes(d, L.DykEwDesc, "dtype", "twin"), _nulls(d, _EW_PTRS))
    if op == L.OP_BN_FINALIZE:
        return (op, _bytes(d, L.DykBnFinalizeDesc, "C", "twin"), _nulls(d, _FIN_PTRS))
    if op == L.OP_BN_FWD_FUSED:
        f = plan._desc_at[d.p[0]]
        a = plan._desc_at[d.p[1]]
        return (op, _bytes(f, L.DykBnFinalizeDesc, "C", "twin"), _nulls(f, _FIN_PTRS),
                _bytes(a, L.DykEwDesc, "dtype", "twin"), _nulls(a, _EW_PTRS))
    return None


def ancestors(deps):
    """anc[i] = bit set of every command that must precede command i (transitive closure of deps)"""
    anc = []
    for i, dl in enumerate(deps):
        m = 0
        for j in dl:
            m |= anc[j] | (1 << j)
        anc.append(m)
    return anc


def worth_pairing(op, d, plan, mode, max_bytes):
    """Measured on MI355X (round 3, C3 at batch 16, every launch timed alone): a two-problem convolution / weight-gradient
    launch takes 1.8-2.1x a single one at EVERY layer (one 4-wave workgroup per CU already saturates the CU's operand
    path), and it chains the two backbones into one dependency chain -- the step got 1.2 ms SLOWER with everything
    paired.  The launch-bound streaming passes do gain: BatchNorm normalise / apply on 5-20 MB tensors 1.1-1.5x for two.
    mode "ew" (default) pairs streaming passes up to `max_bytes` per tensor, "all" everything, "none" nothing."""
    if mode == "all":
        return True
    if mode != "ew" or op in (L.OP_CONV, L.OP_WGRAD):
        return False
    e = plan._desc_at[d.p[1]] if op == L.OP_BN_FWD_FUSED else d
    if op == L.OP_BN_FINALIZE:
        return True
    es = 2 if e.dtype == L.DYK_BF16 else 4
    return float(e.npix) * e.C * es <= max_bytes


def find_pairs(cmds, deps, layer_of, twin_of, plan, mode="ew", max_bytes=48e6):
    """[(i, j)] with i < j: commands of twin sections, k-th against k-th among those of equal signature, that the
    dependency graph does not order"""
    buckets = {}
    for i, (op, d) in enumerate(cmds):
        l = layer_of[i]
        t = twin_of.get(l)
        if t is None:
            continue
        sig = pair_signature(op, d, plan)
        if sig is None or not worth_pairing(op, d, plan, mode, max_bytes):
            continue
        lo, side = (l, 0) if l < t else (t, 1)
        buckets.setdefault((lo, sig), ([], []))[side].append(i)
    anc = ancestors(deps)
    pairs = []
    for (lo, sig), (A, B) in buckets.items():
        if len(A) != len(B):
            continue
        for a, b in zip(A, B):
            i, j = (a, b) if a < b else (b, a)
            if (anc[j] >> i) & 1:
                continue                      # ordered by a dependency chain: not independent
            pairs.append((i, j))
    pairs.sort()
    return pairs


def merge(n, deps, costs, pairs, pair_cost=0.8):
    """Contract every pair into one node.  Returns (members, ndeps, ncosts): members[k] = (i,) or (i, j) in a
    topological numbering of the contracted graph (ties by the smallest original index, so an unpaired list keeps
    its order).  Pairs whose contraction would close a cycle are dropped (each pair alone is acyclic -- its members
    are unordered -- but two pairs can cross: a1 -> b2 and a2 -> b1)."""
    pairs = list(pairs)
    while True:
        node_of = list(range(n))
        for (i, j) in pairs:
            node_of[j] = i                    # the pair's node carries the smaller index
        nd = {}
        for i in range(n):
            k = node_of[i]
            s = nd.setdefault(k, set())
            for j in deps[i]:
                if node_of[j] != k:
                    s.add(node_of[j])
        users = {k: [] for k in nd}
        missing = {}
        for k, s in nd.items():
            missing[k] = len(s)
            for j in s:
                users[j].append(k)
        heap = [k for k, m in missing.items() if m == 0]
        heapq.heapify(heap)
        order = []
        while heap:
            k = heapq.heappop(heap)
            order.append(k)
            for u in users[k]:
                missing[u] -= 1
                if missing[u] == 0:
                    heapq.heappush(heap, u)
        if len(order) == len(nd):
            break
        stuck = {k for k, m in missing.items() if m > 0}
        drop = [p for p in pairs if p[0] in stuck]
        if not drop:
            raise RuntimeError("dependency cycle without a pair in it")
        pairs.remove(drop[0])
    partner = {i: j for (i, j) in pairs}
    new_index = {k: q for q, k in enumerate(order)}
    members, ndeps, ncosts = [], [], []
    for k in order:
        if k in partner:
            members.append((k, partner[k]))
            ncosts.append(pair_cost * (costs[k] + costs[partner[k]]))
        else:
            members.append((k,))
            ncosts.append(costs[k])
        ndeps.append(sorted(new_index[j] for j in nd[k]))
    return members, ndeps, ncosts
