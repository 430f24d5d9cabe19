"""Plan compiler: cfg sections + batch shape + dtype + mode  ->  static device buffers and flat
command lists (forward, backward) of resolved kernel descriptors that libdyk_hip.so executes
natively (dyk_run_commands).

This replaces the reference's eager interpreter loop (models.py:291-305: one Python iteration and
3+ framework launches per cfg section, every step) with a one-time compilation:
  * shapes, channel strides and buffer offsets are resolved once;
  * single-input [route] sections become aliases, multi-input ones write into one concat buffer;
  * BatchNorm statistics are produced by the convolution epilogue, the normalise+activation pass
    carries the residual of a following plain [shortcut];
  * the backward list is generated from the forward one with static knowledge of which gradient
    buffer is written first (store) and which later (accumulate) -- no runtime autograd graph.
"""
import ctypes
import math
import os

import torch

from . import lib as L
from .ops import conv_out_size, dgrad_classes, fwd_taps

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
HEAD_LD = 32            # head conv outputs / their gradients live in 32-channel rows
FWD_SLOTS_CAP = 32    # most replicas of a forward statistics buffer.  32 = every layer's
# finalize rides on its normalise launch (42 launches of 9 us fewer on the forward chain of the target cfg: -0.2 ms in an A/B
# against 256 replicas, round 3; the 10 240 tiles of a 256 x 320 layer then put 320 fp64 atomics on an address -- the backward
# pass has always run such layers with 16 replicas)
FWD_SLOT_WG = 128        # (64 / 256 / 1024: -0.2 / -0.1 / -0.15 ms against 32, round 3; 128: -0.27) conv workgroups per replica of a forward statistics buffer
def _bnfwd_on():
    """conv + BatchNorm forward in one launch on the deep stages (DYK_EPI_BNFWD).  OFF by default (DYK_BNFWD=1 enables): built,
    bit-identical to the two-launch path, and measured NEUTRAL on MI355X (round 3, C3 at batch 16, same box: 31.79 / 31.99 ms
    with it, 31.74 / 31.85 without).  Timed alone, the one launch takes as long as the two it replaces (1x1 256->256 @32x40:
    28.7 us against 15.6 + 11.4; 3x3 512->1024 @16x20: 94 against 68 + 11): the device-wide wait costs three dependent
    agent-scope round trips on this 8-XCD part (arrival atomic, poll, statistics reads that must bypass the XCD's L2) --
    as much as the kernel boundary it removes -- and the residency contract keeps the tuner off the small / K-grouped tiles.
    With release / acquire fences instead of bare atomics it was 73 us (whole-L2 write-back + invalidate per workgroup).
    Never under the two-problem pairing experiments (a launch that waits on its own workgroups is single-problem)."""
    return os.environ.get("DYK_BNFWD", "0") != "0" and os.environ.get("DYK_PAIR", "0") == "0"


BNFWD_MAX_GRID = 256                                             # = dyk_conv_bnfwd_max_grid(): two such launches are resident together
DW_SLOTS = 32                # replicas of a depthwise conv's forward statistics buffer
STAT_SLOTS = 16   # replicas of every per-channel fp64 reduction buffer of the backward (bounds atomic
                                                            # contention; every apply workgroup folds them: 32 -> 16 measured -0.15 ms, 64 +0.4 ms)


def _ru(n, a):
    return (n + a - 1) // a * a


class TRef:
    """A channels-last tensor inside an arena: arena name, byte offset, logical shape, pixel stride."""
    _next = [0]

    def __init__(self, arena, off, B, H, W, C, ld, esize, tid=None):
        self.arena, self.off, self.B, self.H, self.W, self.C, self.ld, self.esize = arena, off, B, H, W, C, ld, esize
        if tid is None:
            tid = TRef._next[0]
            TRef._next[0] += 1
        self.tid = tid

    @property
    def npix(self):
        return self.B * self.H * self.W

    def chan_slice(self, c0, C):
        return TRef(self.arena, self.off + c0 * self.esize, self.B, self.H, self.W, C, self.ld, self.esize)


class Arena:
    def __init__(self, name):
        self.name, self.size, self.tensor = name, 0, None
        self.blocks = []                       # (offset, bytes) of every allocation, ascending (dyk/sched.py)
        self.pads = {}                         # block offset -> trailing pad bytes

    def alloc(self, nbytes, align=256, pad=0):
        """`pad`: zero bytes behind the payload that belong to the block but are never written (K-step over-read room)"""
        off = _ru(self.size, align)
        self.size = off + nbytes + pad
        self.blocks.append((off, nbytes + pad))
        if pad:
            self.pads[off] = pad
        return off

    def materialize(self, device, zero=True):
        n = max(self.size, 256) + 256          # (spare zero bytes: tight-row K steps of the very last pixel read them)
        self.tensor = torch.zeros(n, dtype=torch.uint8, device=device) if zero else \
            torch.empty(n, dtype=torch.uint8, device=device)
        return self.tensor

    def ptr(self, off=0):
        return self.tensor.data_ptr() + off


class Plan:
    """Compiled execution plan.  Build with `compile_plan`."""

    def __init__(self):
        self.fwd, self.bwd = [], []            # lists of (op, desc)
        self.dyn_in = []                       # (desc, 'x'|'y')  patch-gather inputs to patch per call
        self.dyn_dp = []                       # (desc, head index) head-permute-bwd inputs
        self.stem_fuse = []                    # (BatchNorm-backward apply desc, stem weight-gradient desc): fusable per call (engine)
        self.p_out = []                        # per head: torch tensor [B,na,ny,nx,no] fp32
        self.io = None                         # eval: [B, rows, no]
        self.arenas = {}
        self._keep = []
        self._cfwd = self._cbwd = None
        self._cmd_us, self._rw_extra, self._part_extent = {}, {}, {}
        self._wg_groups, self._wg_group_info = {}, []        # grouped weight-gradient launches: lead descriptor -> members
        self.has_bnfwd, self.bnfwd_counters = False, []      # one-launch conv + BatchNorm blocks (DYK_EPI_BNFWD): counter offsets in `ws`

    def bnfwd_error_words(self):
        """device tensor (int32, one element per DYK_EPI_BNFWD launch of the plan): non-zero where the launch gave up waiting for
        its own workgroups (residency contract violated) and skipped the normalise step; None if the plan has no such launch"""
        if not self.bnfwd_counters:
            return None
        words = self.arenas["ws"].tensor.view(torch.int32)
        idx = torch.tensor([off // 4 + 1 for off in self.bnfwd_counters], device=words.device)
        return words[idx]

    def _pack(self, cmds, lanes):
        arr = (L.DykCommand * max(len(cmds), 1))()
        for i, (op, desc) in enumerate(cmds):
            arr[i].op = op
            arr[i].lane = lanes.get(i, 0)
            arr[i].desc = ctypes.addressof(desc)
        return arr

    def finalize(self):
        self._cfwd = self._pack(self.fwd, getattr(self, "fwd_lanes", {}))
        self._cbwd = self._pack(self.bwd, getattr(self, "bwd_lanes", {}))
        self._desc_at = {ctypes.addressof(d): d for d in self._keep if isinstance(d, ctypes.Structure)}
        self._schedules = {}
        self._graphs, self._graph_stats, self._graph_seen = {}, {}, {}

    def schedule(self, which, start, end):
        """dependency schedule of commands [start, end) (dyk/sched.py), built on first use"""
        key = (which, start, end)
        sc = self._schedules.get(key)
        if sc is None:
            from . import sched
            sc = self._schedules[key] = sched.build(self, self.store, which, start, end)
        return sc

    def _dyn_signature(self, which):
        """the per-call pointers inside the descriptors of a list (image batches; for the backward list also the loss
        gradients): a captured graph is only valid for the values it was captured with"""
        sig = []
        for desc, _ in self.dyn_in:
            sig.append((desc.img, desc.in_u8) if isinstance(desc, L.DykStemDesc) else desc.p[0])
        if which == "bwd":
            for desc, _ in self.dyn_dp:
                sig.append(desc.p[0])
        return tuple(sig)

    def _run_graph(self, which, start, end, sc, lp, arr, stream_ptr, failed):
        """launch the hipGraph of this schedule, capturing it first when the per-call pointers are new.  A pass whose
        pointers keep changing (more than a few captures) goes back to direct replay: returns None."""
        lib = L.load()
        key = (which, start, end, self._dyn_signature(which))
        g = self._graphs.get(key)
        if g is None:
            st = self._graph_stats.setdefault((which, start, end), [0, 0])     # [captures, first-seen signatures]
            seen = self._graph_seen.setdefault((which, start, end), {})
            # capture only pointer sets that come back (second sighting): one-off tensors never pay for a capture
            seen[key[3]] = seen.get(key[3], 0) + 1
            if len(seen) > 64:
                seen.clear()
            if seen.get(key[3], 0) < 2 or st[0] >= 8:
                return None
            handle = ctypes.c_void_p()
            sub = ctypes.cast(ctypes.addressof(arr) + start * ctypes.sizeof(L.DykCommand), ctypes.POINTER(L.DykCommand))
            rc = lib.dyk_dag_graph_create(sub, end - start, sc.dep_off, sc.dep_idx, ctypes.byref(handle), ctypes.byref(failed))
            if rc != 0:
                st[0] = 1 << 30                    # capture is not possible here: stay on direct replay
                failed.value = -1
                return None
            st[0] += 1
            g = self._graphs[key] = handle
        return lib.dyk_schedule_graph_launch(g, ctypes.c_void_p(stream_ptr))

    def __del__(self):
        try:
            lib = L.load()
            for g in getattr(self, "_graphs", {}).values():
                lib.dyk_schedule_graph_destroy(g)
        except Exception:
            pass

    def run(self, which, stream_ptr, start=0, end=None):
        arr, n = (self._cfwd, len(self.fwd)) if which == "fwd" else (self._cbwd, len(self.bwd))
        end = n if end is None else end
        if end <= start:
            return
        failed = ctypes.c_int32(-1)
        mode = os.environ.get("DYK_SCHED", "dag")
        if mode == "dag" and os.environ.get("DYK_OVERLAP", "1") != "0":
            sc = self.schedule(which, start, end)
            lp = 0       # (low-priority side streams: 40 ms against 28.5, round 4 -- never)
            rc = None
            # hipGraph replay of the dependency graph: built and tested, OFF by default -- on this stack (HIP runtime of
            # PyTorch-ROCm 7.0) a graph launch of the 435 + 671 child nodes runs the step in 46.3 ms against 36.1 ms for the
            # direct multi-stream replay (batch 1: 19.3 vs 10.7 ms); DYK_GRAPH=1 enables it
            if os.environ.get("DYK_GRAPH", "0") != "0":
                rc = self._run_graph(which, start, end, sc, lp, arr, stream_ptr, failed)
            if rc is None:
                rc = L.load().dyk_run_schedule(arr, sc.array, sc.n, sc.n_streams, lp, ctypes.c_void_p(stream_ptr),
                                               ctypes.byref(failed))
            start = 0
        else:
            sub = ctypes.cast(ctypes.addressof(arr) + start * ctypes.sizeof(L.DykCommand), ctypes.POINTER(L.DykCommand))
            overlap = os.environ.get("DYK_OVERLAP", "1") != "0"
            fn = L.load().dyk_run_commands_overlap if overlap else L.load().dyk_run_commands
            rc = fn(sub, end - start, ctypes.c_void_p(stream_ptr), ctypes.byref(failed))
        if rc != 0 and failed.value >= 0:
            failed.value += start
        if rc != 0:
            cmds = self.fwd if which == "fwd" else self.bwd
            raise L.DykError("plan %s command %d (op %d) failed: %s" % (
                which, failed.value, cmds[failed.value][0] if 0 <= failed.value < len(cmds) else -1,
                L.load().dyk_error_string(rc).decode()))


# ======================================================================================
def compile_plan(model, store, B, H, W, dtype, training, device, dry=False):
    """model: models.YOLO ; store: ParamStore (adopted on `device`).  dry=True resolves shapes, buffers
    and command lists without touching the GPU library (host-logic tests run it on the CPU)."""
    cw = store.alloc_compute(dtype) if dry else store.compute_weights(dtype)
    code = L.DYK_BF16 if dtype == torch.bfloat16 else L.DYK_F32
    es = 2 if dtype == torch.bfloat16 else 4
    plan = Plan()
    act_arena, grad_arena, ws, st_arena = Arena("act"), Arena("grad"), Arena("ws"), Arena("stats")
    plan.arenas = {"act": act_arena, "grad": grad_arena, "ws": ws, "stats": st_arena}
    pending = []        # closures that fill pointers once the arenas exist
    defs = model.module_defs
    mods = model.module_list
    net = model.net_info
    second = net.get("second_index", None)
    v4 = "yolov4" in model.cfg

    # Row length of an activation / gradient tensor.  The MFMA conv reads its K dimension in 32-channel (64-byte) steps;
    # rows used to be padded to that step, which doubles the HBM traffic of EVERY kernel touching a 16-channel tensor
    # (MobileNet stem / first block at 256 x 320: 64 bytes fetched per pixel for 32 used) and adds 25-60 % on 24- / 40- /
    # 72-channel ones.  Tight rows (a multiple of the 16-byte vector): the last K step of a pixel then runs into the
    # NEXT pixel's first channels -- finite values that meet the zero rows of the padded weight matrix (Wc_pad / Wt_pad),
    # so they contribute exactly 0; every arena ends in 256 spare zero bytes for the last pixel of the last tensor.
    tight = True             # rows of ceil(C / 8) * 8 channels (round 2; padding to the 32-channel K step cost the MobileNet cfgs 2 ms)

    def kpad_bytes(ld, esize):
        # A tensor (or a channel slice of a concat buffer) whose channel count is not a multiple of the MFMA conv's
        # 32-channel K step is over-read by up to one K step (64 bytes) behind its LAST pixel: those bytes belong to the
        # block itself (zero from materialisation, never written), so the tail never reaches a neighbouring block -- no
        # ordering against that block's writers is needed and no Inf / NaN of a foreign tensor (an fp32 block read as
        # bf16) can meet the zero weight rows.  Every activation / gradient block gets the pad: which slices of it
        # will be read with a padded K is not known when it is allocated.
        return 64

    def new_act(Bn, Hn, Wn, C, ld=None, esize=None):
        ld = ld or (_ru(C, 8) if tight else _ru(C, 32))
        esize = esize or es
        return TRef("act", act_arena.alloc(Bn * Hn * Wn * ld * esize, pad=kpad_bytes(ld, esize)), Bn, Hn, Wn, C, ld, esize)

    # ---- concat placement: a layer whose output is an input of a multi-source [route] writes straight into its
    # channel slice of the concatenation buffer (no copy at the route).  Decided up front from the channel counts.
    chans = []
    for j, md in enumerate(defs):
        tj = md["type"]
        if tj in ("convolutional", "depthwiseconvolutional"):
            chans.append(md["filters"])
        elif tj == "route":
            chans.append(sum(chans[q] for q in mods[j].layers))
        else:
            chans.append(chans[j - 1] if j else 3)
    concat_slot, route_buf = {}, {}
    for j, md in enumerate(defs):
        if md["type"] == "route" and len(mods[j].layers) > 1:
            c0 = 0
            for q in mods[j].layers:
                head_conv = q + 1 < len(defs) and defs[q + 1]["type"] == "yolo"
                if (q not in concat_slot and chans[q] % 32 == 0 and not head_conv
                        and defs[q]["type"] in ("convolutional", "depthwiseconvolutional", "maxpool", "upsample", "se")
                        and list(mods[j].layers).count(q) == 1):
                    concat_slot[q] = (j, c0, sum(chans[r] for r in mods[j].layers))
                c0 += chans[q]

    nrefs = [0] * len(defs)           # how many sections read the output of section j
    for j, md in enumerate(defs):
        if md["type"] in ("route", "shortcut"):
            for q in mods[j].layers:
                nrefs[q] += 1
        if md["type"] != "route" and j > 0:
            nrefs[j - 1] += 1

    # ---- joint BatchNorm backward over a concatenation (round 5).  A multi-source [route] whose sources are all plain conv +
    # BatchNorm + (one common) activation sections that write their slice of the concat buffer in place and are read by
    # nobody else: the data gradient of the route's only reader produces the gradient of ALL of them in one launch, so it can
    # carry their BatchNorm-backward reduces in its epilogue (DYK_EPI_BNBWD over the concatenated channels) -- if their raw
    # outputs are channel slices of ONE buffer and their scale / shift / mean / rstd vectors sit side by side.  Decided here
    # (layout), confirmed at the route (every source took its slot) and in the backward (sole reader, first writer).
    joint_slot, joint_raw, joint_vecs, joint_of = {}, {}, {}, {}
    if (training and not os.environ.get("DYK_DEBUG_PLAN")
            and os.environ.get("DYK_BNBWD_FUSE", "1") != "0"):
        for j, md in enumerate(defs):
            if md["type"] != "route" or len(mods[j].layers) < 2:
                continue
            srcs_ = list(mods[j].layers)
            ok = all(concat_slot.get(q, (None,))[0] == j and nrefs[q] == 1 and defs[q]["type"] == "convolutional"
                     and defs[q]["batch_normalize"] and defs[q].get("groups", 1) == 1 and q != 0 and q != second
                     and chans[q] % 8 == 0 for q in srcs_)
            if ok and len({defs[q]["activation"] for q in srcs_}) == 1:
                for q in srcs_:
                    joint_slot[q] = concat_slot[q]

    def alloc_out(layer, Bn, Hn, Wn, C):
        """output tensor of cfg section `layer`: a slice of its concat buffer when it has one"""
        slot = concat_slot.get(layer)
        if slot is None:
            return new_act(Bn, Hn, Wn, C)
        j, c0, ctot = slot
        if j not in route_buf:
            route_buf[j] = new_act(Bn, Hn, Wn, ctot)
        buf = route_buf[j]
        assert (buf.H, buf.W) == (Hn, Wn)
        return buf.chan_slice(c0, C)

    def new_ws(nbytes):
        return ws.alloc(nbytes)

    def ptr_of(t):
        return plan.arenas[t.arena].ptr(t.off)

    def later(fn):
        pending.append(fn)

    def ew_desc(a=None, b=None, out=None, C=None, npix=None, act=0, flags=0, alpha=1.0, beta=1.0, Bn=0, Hn=0, Wn=0, k=0):
        d = L.DykEwDesc()
        d.dtype = code
        d.C = C if C is not None else (a.C if a is not None else 0)
        d.npix = npix if npix is not None else (a.npix if a is not None else 0)
        d.lda = a.ld if a is not None else 0
        d.ldb = b.ld if b is not None else 0
        d.ldo = out.ld if out is not None else 0
        d.act, d.flags, d.alpha, d.beta = act, flags, alpha, beta
        d.B, d.H, d.W, d.k = Bn, Hn, Wn, k

        def fill(d=d, a=a, b=b, out=out):
            d.a = ptr_of(a) if a is not None else None
            d.b = ptr_of(b) if b is not None else None
            d.out = ptr_of(out) if out is not None else None
        later(fill)
        plan._keep.append(d)
        return d

    def misc():
        d = L.DykMiscDesc()
        plan._keep.append(d)
        return d

    # ---------------------------------------------------------------- forward
    outs = []                 # per layer: TRef or None
    info = []                 # per layer: dict of what backward needs
    tcons = {}                # TRef tid -> number of sections that read the tensor
    producer_of = {}          # TRef tid of a train-mode conv+BN output -> its record

    def consume(t):
        tcons[t.tid] = tcons.get(t.tid, 0) + 1
    yolo_rows = []

    def defer_to_dw(i, out_layer, out_ref, Bn, Ho, Wo, cout, is_dw, is_direct):
        """OFF by default (DYK_DW_PRE=1 enables it): built for VERDICT r4 #3a, bit-identical (tests), measured SLOWER -- MobileNetV3
        cfg, batch 32, in-call: 18.2-18.3 ms without, 19.1-19.3 ms with (20.8 with the activation switch inside the unrolled
        loops): the depthwise kernels are latency-bound, every VALU op and the extra barrier on their staging path costs more than
        the two streaming passes (at 3-5 TB/s) it removes.
        May section i's conv + BatchNorm + activation leave normalise + activation to its consumer?  Yes when the consumer
        is THE next section, a stride-1 3x3 / 5x5 depthwise conv over the same channels that dyk_dwconv_fwd runs on the
        LDS-tiled kernel, nobody else reads the output (no route / shortcut, no concat placement), bf16 training plan."""
        if (not training or is_dw or is_direct or out_ref is not None or out_layer != i or code != L.DYK_BF16
                or os.environ.get("DYK_DW_PRE", "0") == "0" or os.environ.get("DYK_DEBUG_PLAN") or _bnfwd_on()):
            return False
        if i + 1 >= len(defs) or nrefs[i] != 1 or i in concat_slot or (second is not None and i + 1 == second):
            return False
        nx = defs[i + 1]
        if nx["type"] == "convolutional":
            g = nx.get("groups", 1)
            if not (isinstance(g, int) and g > 1 and g == cout and nx["filters"] == cout and nx.get("stride", 1) == 1
                    and nx["size"] in (3, 5) and nx["pad"] and nx["batch_normalize"]):
                return False
            k2, pad2 = nx["size"], nx["size"] // 2
        elif nx["type"] == "depthwiseconvolutional":
            if nx.get("stride", 1) != 1 or nx.get("size", 3) not in (3, 5):
                return False
            k2, pad2 = nx.get("size", 3), 1
        else:
            return False
        q = L.DykDwDesc()
        q.x = q.y = 1                                    # (shape query only)
        q.dtype, q.B, q.Hi, q.Wi, q.C, q.k, q.stride, q.pad = code, Bn, Ho, Wo, cout, k2, 1, pad2
        q.Ho, q.Wo = conv_out_size(Ho, k2, 1, pad2), conv_out_size(Wo, k2, 1, pad2)
        q.ldx = q.ldy = _ru(cout, 8)
        return L.load().dyk_dwconv_tile_ok(ctypes.byref(q)) == 1

    import functools
    import types
    conv_forward = functools.partial(_emit_conv_forward, types.SimpleNamespace(
        B=B, H=H, W=W, alloc_out=alloc_out, code=code, consume=consume, cw=cw, defer_to_dw=defer_to_dw, defs=defs, es=es,
        ew_desc=ew_desc, joint_raw=joint_raw, joint_slot=joint_slot, joint_vecs=joint_vecs, later=later, misc=misc,
        new_act=new_act, new_ws=new_ws, plan=plan, producer_of=producer_of, ptr_of=ptr_of, st_arena=st_arena, store=store,
        tight=tight, training=training, ws=ws))

    # ---------------------------------------------------------------- forward: _emit_forward below
    stats_memset, fwd_start = _emit_forward(types.SimpleNamespace(
        B=B, alloc_out=alloc_out, concat_slot=concat_slot, consume=consume, conv_forward=conv_forward, defs=defs,
        device=device, es=es, ew_desc=ew_desc, info=info, joint_of=joint_of, joint_raw=joint_raw, joint_vecs=joint_vecs,
        later=later, misc=misc, model=model, mods=mods, new_act=new_act, new_ws=new_ws, outs=outs, plan=plan, ptr_of=ptr_of,
        route_buf=route_buf, second=second, store=store, training=training, v4=v4, ws=ws, yolo_rows=yolo_rows))

    # ---------------------------------------------------------------- backward (training plans): _emit_backward below
    grads = {}
    if training:
        grads = _emit_backward(types.SimpleNamespace(
            B=B, code=code, concat_slot=concat_slot, cw=cw, defs=defs, es=es, ew_desc=ew_desc, grad_arena=grad_arena,
            info=info, joint_of=joint_of, kpad_bytes=kpad_bytes, later=later, misc=misc, mods=mods, new_ws=new_ws, nrefs=nrefs,
            plan=plan, producer_of=producer_of, ptr_of=ptr_of, store=store, tcons=tcons, tight=tight, ws=ws))

    # ---------------------------------------------------------------- the rest is a pipeline of passes over the two lists
    # allocate + emit (above)  ->  materialise  ->  fuse  ->  tune  ->  group  ->  scratch  ->  schedule metadata
    _materialise(plan, device, pending, stats_memset if training else None, st_arena)
    plan.training = training
    plan.store = store
    _post_passes(plan, store, device, dry, training, stats_memset, _backbone_force_layers(defs, mods, second))
    _assign_lanes(plan, defs, mods, second, fwd_start, training)
    _layer_maps(plan, model, defs, mods, second, fwd_start, training)
    plan.finalize()
    plan.info = info
    plan.grads = grads if training else {}
    plan.outs = outs
    plan.shape = (B, H, W)
    plan.dtype = dtype
    plan.training = training
    return plan


# ======================================================================================
# The two emission stages of compile_plan
def _emit_conv_forward(cx, i, x_in, stem_src, wname, bnpre, bnm, bias_name, k, stride, pad, cout, act, groups=1, out_layer=None,
                       out_ref=None, entry=True):
    """emit forward commands of Conv2d [+ BatchNorm2d] [+ activation]; returns (out TRef, info dict).
    wname / bnpre / bias_name are parameter-store names (bnpre None = no BatchNorm)."""
    (B, H, W, alloc_out, code, consume, cw, defer_to_dw, defs, es, ew_desc, joint_raw, joint_slot, joint_vecs, later, misc,
    new_act, new_ws, plan, producer_of, ptr_of, st_arena, store, tight, training, ws) = (cx.B, cx.H, cx.W, cx.alloc_out, cx.code,
    cx.consume, cx.cw, cx.defer_to_dw, cx.defs, cx.es, cx.ew_desc, cx.joint_raw, cx.joint_slot, cx.joint_vecs, cx.later, cx.misc,
    cx.new_act, cx.new_ws, cx.plan, cx.producer_of, cx.ptr_of, cx.st_arena, cx.store, cx.tight, cx.training, cx.ws)
    bn = bnpre is not None
    dw = groups > 1
    rec = {"kind": "conv", "bn": bn, "k": k, "stride": stride, "pad": pad, "cout": cout, "act": act, "i": i,
           "stem": stem_src is not None, "wname": wname, "bnpre": bnpre, "bias_name": bias_name, "dw": dw,
           "entry": entry}                   # entry: the conv reads the section's input (not an intermediate of it)
    if dw:
        if stem_src is not None or groups != x_in.C or cout != x_in.C:
            raise NotImplementedError("grouped convolution that is not depthwise (layer %d)" % i)
        Hi, Wi = x_in.H, x_in.W
        Ho, Wo = conv_out_size(Hi, k, stride, pad), conv_out_size(Wi, k, stride, pad)
        d = L.DykDwDesc()
        plan._keep.append(d)
        d.dtype, d.w = code, store.p_ptr(wname)
        d.B, d.Hi, d.Wi, d.Ho, d.Wo, d.C = B, Hi, Wi, Ho, Wo, cout
        d.k, d.stride, d.pad = k, stride, pad
        d.ldx = x_in.ld
        conv_op = L.OP_DW_FWD
        pre = getattr(x_in, "pre", None)
        if pre is not None:
            # x_in aliases the RAW output of the expansion conv (deferred normalise, below): z is formed on load
            d.pre_act = pre[1]
            later(lambda d=d, pre=pre: setattr(d, "pre", ws.ptr(pre[0])))
    elif (stem_src is not None and k == 3 and pad == 1 and stride in (1, 2) and cout in (16, 32) and bn
          ):
        # Cin=3 stem straight from the image batch (csrc/stem.hip): no float conversion pass, no im2col
        Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
        Hi, Wi = H, W
        d = L.DykStemDesc()
        plan._keep.append(d)
        d.dtype, d.B, d.H, d.W, d.Cout, d.k, d.stride, d.pad, d.Ho, d.Wo = code, B, H, W, cout, k, stride, pad, Ho, Wo
        d.wt = cw["stems_t"][wname].data_ptr()
        plan.dyn_in.append((d, stem_src))
        conv_op = L.OP_STEM_FWD
        rec["stem_direct"] = d
    elif stem_src is not None:
        # Cin=3 stem, general shape: gather k*k*3 patches (zero padded to 32) and run a 1x1 MFMA conv on them
        Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
        patches = new_act(B, Ho, Wo, 32)
        g = misc()
        g.i[0], g.i[1], g.i[2], g.i[3], g.i[4], g.i[5], g.i[6], g.i[7], g.i[8] = B, 3, H, W, k, stride, pad, 32, code
        g.f[0] = 1.0
        later(lambda g=g, patches=patches: g.p.__setitem__(1, ptr_of(patches)))
        plan.dyn_in.append((g, stem_src))
        plan.fwd.append((L.OP_PATCH_GATHER, g))
        x_in = patches
        wfwd_ptr = cw["stems"][wname].data_ptr()
        taps, cin_k, cisy = [(0, 0, 0)], 32, 1
        rec["wgrad"] = dict(x=patches, Cin=k * k * 3, lddw=k * k * 3, taps=[(0, 0, 0)], isy=1, Hi=Ho, Wi=Wo)
        Hi, Wi = Ho, Wo
    else:
        Hi, Wi = x_in.H, x_in.W
        Ho, Wo = conv_out_size(Hi, k, stride, pad), conv_out_size(Wi, k, stride, pad)
        e = store.by_name[wname]
        cin_k = _ru(x_in.C, 32)
        if x_in.ld < cin_k and not tight:
            raise NotImplementedError("conv input rows narrower than the padded K (layer %d)" % i)
        if cin_k != x_in.C:
            wfwd_ptr = cw["Wc_pad"].data_ptr() + cw["fwd_pad_off"][wname] * es
        else:
            wfwd_ptr = cw["Wc"].data_ptr() + e.offset * es
        taps, cisy = fwd_taps(k, pad), stride
        rec["wgrad"] = dict(x=x_in, Cin=x_in.C, lddw=0, taps=taps, isy=stride, Hi=Hi, Wi=Wi)
    rec["x"] = x_in
    if stem_src is None:
        consume(x_in)
    direct = "stem_direct" in rec
    if not dw and not direct:
        d = L.DykConvDesc()
        plan._keep.append(d)
        d.dtype = code
        d.w = wfwd_ptr
        d.B, d.Hi, d.Wi, d.Cin, d.Cout = B, Hi, Wi, cin_k, cout
        d.Hg, d.Wg, d.Ho, d.Wo = Ho, Wo, Ho, Wo
        d.isy = d.isx = cisy
        d.osy = d.osx = 1
        d.ntaps = len(taps)
        for q, (ty, tx, wt) in enumerate(taps):
            d.tdy[q], d.tdx[q], d.twt[q] = ty, tx, wt
        d.ldx = x_in.ld
        conv_op = L.OP_CONV
    if bn:
        if training:
            js = joint_slot.get(out_layer) if (not dw and not direct and stem_src is None and out_ref is None) else None
            if js is not None:
                jj, jc0, jctot = js
                if jj not in joint_raw:
                    joint_raw[jj] = new_act(B, Ho, Wo, jctot)
                    joint_vecs[jj] = new_ws(4 * jctot * 4)
                assert (joint_raw[jj].H, joint_raw[jj].W) == (Ho, Wo)
                y_raw = joint_raw[jj].chan_slice(jc0, cout)
                rec["joint"] = jj
            else:
                y_raw = new_act(B, Ho, Wo, cout)
            z = None if defer_to_dw(i, out_layer, out_ref, B, Ho, Wo, cout, dw, direct) else (
                out_ref if out_ref is not None else alloc_out(out_layer, B, Ho, Wo, cout))
            # replicas of the fp64 statistics accumulators: keep the atomics per address at a few dozen
            tiles = (B * Ho * Wo + 127) // 128
            # (about 32 workgroups per replica), few enough that the consumer folds them cheaply
            slots = min(FWD_SLOTS_CAP, max(4, 1 << max(0, (tiles * max(1, (cout + 127) // 128) // FWD_SLOT_WG - 1).bit_length())))
            if dw:
                slots = DW_SLOTS                 # 32 replicas: the finalize then rides on the normalise pass (measured
                                                 # -0.3 ms per MobileNetV3 step against up to 256 replicas + own launch)
            stats = st_arena.alloc(slots * 2 * cout * 8)   # fp64 replicas; the whole arena is zeroed at the start of a pass
            if js is not None:                   # this layer's columns of the route's scale | shift | mean | rstd rows
                vecs, vs = joint_vecs[jj] + 4 * jc0, 4 * jctot
            else:
                vecs, vs = new_ws(4 * cout * 4), 4 * cout      # scale | shift | mean | rstd, `vs` bytes apart
            rec["vs"] = vs
            d.ldy, d.stats_slots = y_raw.ld, slots
            if not dw and not direct:
                d.act, d.flags = 0, L.EPI_STATS
            # conv + BatchNorm + activation in ONE launch (DYK_EPI_BNFWD) where every workgroup of the conv can be resident at
            # once: the 32 x 40 and 16 x 20 stages (<= dyk_conv_bnfwd_max_grid() workgroups on the 160-pixel tile).  Saves the
            # normalise launch (6-10 us + its 5-7 us gap where one stream runs alone) and the re-read of the raw output.
            fuse_bn = (not dw and not direct and code == L.DYK_BF16 and cout % 8 == 0 and _bnfwd_on()
                       and ((B * Ho * Wo + 159) // 160) * ((cout + 127) // 128) <= BNFWD_MAX_GRID)
            if fuse_bn:
                cnt = new_ws(16)                     # arrivals | error | departures | - : zero at plan creation, re-armed by the
                                                     # launch itself; NOT in the statistics arena (zeroed every pass), so that an
                                                     # error word survives until Plan.bnfwd_error() has looked at it
                plan.bnfwd_counters.append(cnt)
                d.act, d.flags = act, L.EPI_STATS | L.EPI_BNFWD
                d.bn_gamma, d.bn_beta = store.p_ptr(bnpre + "weight"), store.p_ptr(bnpre + "bias")
                d.bn_running_mean, d.bn_running_var = store.r_ptr(bnm, "running_mean"), store.r_ptr(bnm, "running_var")
                d.bn_count, d.bn_momentum, d.bn_eps = B * Ho * Wo, BN_MOMENTUM, BN_EPS
                d.ldy2 = z.ld
                later(lambda d=d, z=z, vecs=vecs, cnt=cnt: (
                    setattr(d, "y2", ptr_of(z)), setattr(d, "scale", ws.ptr(vecs)), setattr(d, "shift", ws.ptr(vecs + vs)),
                    setattr(d, "bn_save_mean", ws.ptr(vecs + 2 * vs)), setattr(d, "bn_save_rstd", ws.ptr(vecs + 3 * vs)),
                    setattr(d, "bn_counter", ws.ptr(cnt))))
                plan.has_bnfwd = True
            if direct:
                later(lambda d=d, y_raw=y_raw, stats=stats: (
                    setattr(d, "y", ptr_of(y_raw)), setattr(d, "stats", st_arena.ptr(stats))))
            else:
                later(lambda d=d, x_in=x_in, y_raw=y_raw, stats=stats: (
                    setattr(d, "x", ptr_of(x_in)), setattr(d, "y", ptr_of(y_raw)), setattr(d, "stats", st_arena.ptr(stats))))
            plan.fwd.append((conv_op, d))
            if fuse_bn:
                rec.update(y_raw=y_raw, z=z, vecs=vecs, conv_desc=d)      # (conv_desc: a plain [shortcut] behind it rides on the epilogue)
                producer_of[z.tid] = rec
                return z, rec
            f = L.DykBnFinalizeDesc()
            plan._keep.append(f)
            f.gamma, f.beta = store.p_ptr(bnpre + "weight"), store.p_ptr(bnpre + "bias")
            f.running_mean, f.running_var = store.r_ptr(bnm, "running_mean"), store.r_ptr(bnm, "running_var")
            f.C, f.count, f.momentum, f.eps, f.slots = cout, B * Ho * Wo, BN_MOMENTUM, BN_EPS, slots
            later(lambda f=f, stats=stats, vecs=vecs, vs=vs: (
                setattr(f, "stats", st_arena.ptr(stats)), setattr(f, "scale", ws.ptr(vecs)),
                setattr(f, "shift", ws.ptr(vecs + vs)), setattr(f, "save_mean", ws.ptr(vecs + 2 * vs)),
                setattr(f, "save_rstd", ws.ptr(vecs + 3 * vs))))
            if defer_to_dw(i, out_layer, out_ref, B, Ho, Wo, cout, dw, direct):
                # Normalise + activation ON LOAD in the consumer (VERDICT r4 #3a): this block's only reader is a stride-1
                # depthwise conv on the LDS-tiled kernel (MobileNet expansion conv -> depthwise, reference models.py:34-62
                # then :41 groups=C): its forward and its weight gradient form z = dtype(act(scale * u + shift)) from the
                # raw output while staging (DykDwDesc.pre) -- the same values this pass would have stored -- so z is never
                # written or read back: two tensor passes over the LARGEST tensors of the net less, per block.  Only
                # the finalize launch remains.  The output TRef aliases the raw tensor and carries the vectors.
                plan.fwd.append((L.OP_BN_FINALIZE, f))
                z = TRef(y_raw.arena, y_raw.off, y_raw.B, y_raw.H, y_raw.W, y_raw.C, y_raw.ld, y_raw.esize)
                z.pre = (vecs, act)
                rec.update(y_raw=y_raw, z=z, vecs=vecs, deferred=True)
                producer_of[z.tid] = rec
                return z, rec
            a = ew_desc(a=y_raw, out=z, act=act)
            later(lambda a=a, vecs=vecs, vs=vs: (setattr(a, "p0", ws.ptr(vecs)), setattr(a, "p1", ws.ptr(vecs + vs))))
            if slots <= 32:
                fm = misc()                       # finalize folded into the normalise + activation launch
                fm.p[0], fm.p[1] = ctypes.addressof(f), ctypes.addressof(a)
                plan.fwd.append((L.OP_BN_FWD_FUSED, fm))
            else:
                plan.fwd.append((L.OP_BN_FINALIZE, f))
                plan.fwd.append((L.OP_BN_ACT_FWD, a))
            rec.update(y_raw=y_raw, z=z, vecs=vecs, bn_act_desc=a)
            producer_of[z.tid] = rec
            return z, rec
        # eval: running statistics folded into an affine (conv epilogue for the MFMA conv, one more
        # streaming pass for the depthwise conv)
        z = out_ref if out_ref is not None else alloc_out(out_layer, B, Ho, Wo, cout)
        vecs = new_ws(2 * cout * 4)
        fo = misc()
        fo.p[0], fo.p[1] = store.p_ptr(bnpre + "weight"), store.p_ptr(bnpre + "bias")
        fo.p[2], fo.p[3] = store.r_ptr(bnm, "running_mean"), store.r_ptr(bnm, "running_var")
        fo.i[0], fo.f[0] = cout, BN_EPS
        later(lambda fo=fo, vecs=vecs: (fo.p.__setitem__(4, ws.ptr(vecs)), fo.p.__setitem__(5, ws.ptr(vecs + 4 * cout))))
        plan.fwd.append((L.OP_BN_FOLD, fo))
        if dw:
            d.ldy = z.ld
            later(lambda d=d, x_in=x_in, z=z: (setattr(d, "x", ptr_of(x_in)), setattr(d, "y", ptr_of(z))))
            plan.fwd.append((conv_op, d))
            a = ew_desc(a=z, out=z, act=act)
            later(lambda a=a, vecs=vecs: (setattr(a, "p0", ws.ptr(vecs)), setattr(a, "p1", ws.ptr(vecs + 4 * cout))))
            plan.fwd.append((L.OP_BN_ACT_FWD, a))
            rec.update(z=z)
            return z, rec
        if direct:
            d.ldy, d.act = z.ld, act
            later(lambda d=d, z=z, vecs=vecs: (
                setattr(d, "y", ptr_of(z)), setattr(d, "scale", ws.ptr(vecs)), setattr(d, "shift", ws.ptr(vecs + 4 * cout))))
            plan.fwd.append((L.OP_STEM_FWD, d))
            rec.update(z=z)
            return z, rec
        d.ldy, d.act, d.flags = z.ld, act, L.EPI_AFFINE
        later(lambda d=d, x_in=x_in, z=z, vecs=vecs: (
            setattr(d, "x", ptr_of(x_in)), setattr(d, "y", ptr_of(z)), setattr(d, "scale", ws.ptr(vecs)),
            setattr(d, "shift", ws.ptr(vecs + 4 * cout))))
        plan.fwd.append((L.OP_CONV, d))
        rec.update(z=z, conv_desc=d)
        return z, rec
    if dw:
        raise NotImplementedError("depthwise conv without batch_normalize (layer %d)" % i)
    # no BN: bias epilogue; detection heads go to fp32 rows of HEAD_LD channels
    is_head = (i + 1 < len(defs) and defs[i + 1]["type"] == "yolo")
    if is_head:
        assert cout <= HEAD_LD
        z = new_act(B, Ho, Wo, cout, ld=HEAD_LD, esize=4)
        d.flags = L.EPI_AFFINE | L.EPI_OUT_F32
    else:
        z = alloc_out(out_layer, B, Ho, Wo, cout)
        d.flags = L.EPI_AFFINE
    d.ldy, d.act = z.ld, act
    d.shift = store.p_ptr(bias_name)
    later(lambda d=d, x_in=x_in, z=z: (setattr(d, "x", ptr_of(x_in)), setattr(d, "y", ptr_of(z))))
    plan.fwd.append((L.OP_CONV, d))
    rec.update(z=z, is_head=is_head)
    if act != 0:
        raise NotImplementedError("activation on a conv without batch_normalize (layer %d)" % i)
    return z, rec

def _emit_forward(cx):
    """The forward command list (compile_plan's first stage): the interpreter loop of reference models.py:291-305 unrolled once,
    section by section, into resolved descriptors; leaves per section what the backward emission needs (`info`) and the output
    tensor (`outs`).  Returns the statistics memset descriptor and the index of the first forward command of every section."""
    (B, alloc_out, concat_slot, consume, conv_forward, defs, device, es, ew_desc, info, joint_of, joint_raw, joint_vecs, later,
    misc, model, mods, new_act, new_ws, outs, plan, ptr_of, route_buf, second, store, training, v4, ws, yolo_rows) = (cx.B,
    cx.alloc_out, cx.concat_slot, cx.consume, cx.conv_forward, cx.defs, cx.device, cx.es, cx.ew_desc, cx.info, cx.joint_of,
    cx.joint_raw, cx.joint_vecs, cx.later, cx.misc, cx.model, cx.mods, cx.new_act, cx.new_ws, cx.outs, cx.plan, cx.ptr_of,
    cx.route_buf, cx.second, cx.store, cx.training, cx.v4, cx.ws, cx.yolo_rows)
    cur = None                # current x
    head_idx = 0
    stats_memset = misc()
    if training:
        plan.fwd.append((L.OP_MEMSET, stats_memset))       # re-arm every BatchNorm statistics replica of the pass
    fwd_start = []            # index of the first forward command of each section
    for i, m in enumerate(defs):
        t = m["type"]
        mod = mods[i]
        rec = {"kind": t, "i": i}
        fwd_start.append(len(plan.fwd))
        if t == "convolutional":
            stem = None
            if i == 0:
                stem = "x"
            elif second is not None and i == second:
                stem = "y"
            pre = "module_list.%d." % i
            k = m["size"]
            if "stride" not in m:
                raise NotImplementedError("anisotropic stride_y/stride_x (layer %d)" % i)
            bn = bool(m["batch_normalize"])
            cur, rec = conv_forward(i, cur, stem, pre + "Conv2d.weight", (pre + "BatchNorm2d.") if bn else None,
                                    mod[1] if bn else None, pre + "Conv2d.bias", k, m["stride"],
                                    k // 2 if m["pad"] else 0, m["filters"], L.ACT_CODES.get(m["activation"], 0),
                                    groups=m.get("groups", 1), out_layer=i)
        elif t == "depthwiseconvolutional":
            # DepthwiseSeparableConv2d (layers.py:218-231): depthwise k x k (padding fixed at 1) + BN + ReLU6,
            # then pointwise 1x1 + BN + ReLU6
            pre = "module_list.%d.conv." % i
            ks = m.get("size", 3)
            if "stride" not in m:
                raise NotImplementedError("anisotropic stride_y/stride_x (layer %d)" % i)
            relu6 = L.ACT_CODES["relu6"]
            mid, rec_dw = conv_forward(i, cur, None, pre + "0.weight", pre + "1.", mod.conv[1], None, ks, m["stride"], 1,
                                       cur.C, relu6, groups=cur.C)
            cur, rec_pw = conv_forward(i, mid, None, pre + "3.weight", pre + "4.", mod.conv[4], None, 1, 1, 0,
                                       m["filters"], relu6, out_layer=i, entry=False)
            rec = {"kind": "dwsep", "i": i, "parts": [rec_dw, rec_pw]}
        elif t == "inception":
            # Inception (layers.py:148-172): 1x1 | 1x1-3x3 | 1x1-3x3-3x3 | maxpool3-1x1, every conv = Conv2d + BN +
            # LeakyReLU(0.1); the last conv of each branch writes its slice of the concatenated output
            x_in = cur
            consume(x_in)                    # (the branch convs count it again: never a fusion candidate)
            pre = "module_list.%d." % i
            leaky = L.ACT_CODES["leaky"]
            specs = [[(m["n1x1"], 1)], [(m["n3x3_reduce"], 1), (m["n3x3"], 3)],
                     [(m["n5x5_reduce"], 1), (m["n5x5"], 3), (m["n5x5"], 3)], [(m["pool_proj"], 1)]]
            ctot = sum(sp[-1][0] for sp in specs)
            cat = alloc_out(i, B, x_in.H, x_in.W, ctot)
            pooled = new_act(B, x_in.H, x_in.W, x_in.C)
            amax = new_ws(x_in.npix * x_in.C) if training else None
            pd = ew_desc(a=x_in, out=pooled, Bn=B, Hn=x_in.H, Wn=x_in.W, k=3)
            if amax is not None:
                later(lambda pd=pd, amax=amax: setattr(pd, "aux", ws.ptr(amax)))
            plan.fwd.append((L.OP_MAXPOOL_FWD, pd))
            branches, c0 = [], 0
            for bi, sp in enumerate(specs):
                h = pooled if bi == 3 else x_in
                recs = []
                for ci, (co, kk) in enumerate(sp):
                    q = pre + "branch%d.%d.conv." % (bi + 1, ci + (1 if bi == 3 else 0))
                    bnm_ = getattr(mod, "branch%d" % (bi + 1))[ci + (1 if bi == 3 else 0)].conv[1]
                    last = ci == len(sp) - 1
                    h, r_ = conv_forward(i, h, None, q + "0.weight", q + "1.", bnm_, None, kk, 1, kk // 2, co, leaky,
                                         out_ref=cat.chan_slice(c0, co) if last else None, entry=(ci == 0))
                    recs.append(r_)
                branches.append((recs, c0, sp[-1][0]))
                c0 += sp[-1][0]
            rec.update(x=x_in, z=cat, branches=branches, pooled=pooled, amax=amax)
            cur = cat
        elif t == "route":
            layers = mod.layers
            if len(layers) == 1:
                cur = outs[layers[0]]
                rec["alias"] = True
            else:
                srcs = [outs[j] for j in layers]
                for s_ in srcs:
                    consume(s_)
                ctot = sum(s.C for s in srcs)
                s0 = srcs[0]
                cat = route_buf.get(i)
                if cat is None:
                    cat = new_act(B, s0.H, s0.W, ctot)
                assert cat.C == ctot
                c0 = 0
                parts = []
                for j, s in zip(layers, srcs):
                    if concat_slot.get(j, (None,))[0] != i:          # not produced in place: copy
                        plan.fwd.append((L.OP_AXPBY, ew_desc(a=s, out=cat.chan_slice(c0, s.C))))
                    parts.append((s, c0))
                    c0 += s.C
                cur = cat
                rec.update(parts=parts, out=cat)
                if i in joint_raw and all(info[j].get("joint") == i for j in layers):
                    c0 = 0
                    jparts = []
                    for j in layers:
                        jparts.append((info[j], c0))
                        c0 += info[j]["cout"]
                    joint_of[cat.tid] = dict(raw=joint_raw[i], vecs=joint_vecs[i], ctot=ctot, parts=jparts, act=info[layers[0]]["act"])
        elif t == "shortcut":
            layers = mod.layers
            x_in = cur
            if len(layers) != 1:
                raise NotImplementedError("[shortcut] with %d sources" % len(layers))
            a = outs[layers[0]]
            # Mismatched channel counts (reference layers.py:78-83): the sum covers the first min(nx, na) channels; the output
            # keeps x's channel count (nx > na: the rest of x passes through -- the unweighted reference writes that case into x
            # IN PLACE, which also changes a routed copy of the previous layer's output; a plan whose previous layer is routed
            # is refused rather than silently diverging).  Channel runs must be whole 16-byte vectors.
            nx, na = x_in.C, a.C
            C = min(nx, na)
            if nx != na:
                vec = 16 // es
                if C % vec or abs(nx - na) % vec:
                    raise NotImplementedError("[shortcut] with channel counts %d / %d that are not whole 16-byte vectors (layer %d)" % (nx, na, i))
                if nx > na and not mod.weight and i > 0 and model.routs[i - 1]:      # (weighted: x * w[0] is a new tensor there)
                    raise NotImplementedError("[shortcut] narrower than its routed input (layer %d): the reference adds in place" % i)
            consume(x_in)
            consume(a)
            prev = info[i - 1] if i > 0 else {}
            fusable = (not mod.weight and nx == na and prev.get("kind") == "conv" and prev.get("bn") and prev.get("z") is x_in
                       and not model.routs[i - 1] and a is not x_in
                       and ("bn_act_desc" in prev or "conv_desc" in prev) and not os.environ.get("DYK_DEBUG_PLAN"))
            if fusable:
                # plain residual add straight after conv+BN+act whose output nobody else reads: the add rides on the
                # normalise+activation pass (training) / the conv epilogue (eval); z of the conv IS the shortcut output
                if "bn_act_desc" in prev:
                    e = prev["bn_act_desc"]
                    e.ldb = a.ld
                    later(lambda e=e, a=a: setattr(e, "b", ptr_of(a)))
                else:
                    cd = prev["conv_desc"]
                    cd.flags |= L.EPI_RESIDUAL
                    cd.ldr = a.ld
                    later(lambda cd=cd, a=a: setattr(cd, "res", ptr_of(a)))
                prev["fused_shortcut"] = True
                rec.update(weighted=False, fused=True, x=x_in, a=a, z=x_in)
                cur = x_in
                outs.append(cur)
                info.append(rec)
                continue
            z = new_act(B, x_in.H, x_in.W, x_in.C)
            xs, asl, zs = (x_in, a, z) if nx == na else (x_in.chan_slice(0, C), a.chan_slice(0, C), z.chan_slice(0, C))
            rest = (x_in.chan_slice(na, nx - na), z.chan_slice(na, nx - na)) if nx > na else None
            if mod.weight:
                weff = new_ws(16)
                wd = misc()
                wd.p[0] = store.p_ptr("module_list.%d.w" % i)
                wd.i[0] = 2
                later(lambda wd=wd, weff=weff: wd.p.__setitem__(1, ws.ptr(weff)))
                plan.fwd.append((L.OP_WFUSE_WEIGHTS, wd))
                e = ew_desc(a=xs, b=asl, out=zs, C=C)
                later(lambda e=e, weff=weff: (setattr(e, "p0", ws.ptr(weff)), setattr(e, "p1", ws.ptr(weff + 4))))
                plan.fwd.append((L.OP_AXPBY, e))
                if rest:
                    e = ew_desc(a=rest[0], out=rest[1], C=nx - na)
                    later(lambda e=e, weff=weff: setattr(e, "p0", ws.ptr(weff)))
                    plan.fwd.append((L.OP_AXPBY, e))
                rec.update(weighted=True, weff=weff)
            else:
                plan.fwd.append((L.OP_AXPBY, ew_desc(a=xs, b=asl, out=zs, C=C)))
                if rest:
                    plan.fwd.append((L.OP_AXPBY, ew_desc(a=rest[0], out=rest[1], C=nx - na)))
                rec.update(weighted=False)
            rec.update(x=x_in, a=a, z=z)
            cur = z
        elif t == "se":
            x_in = cur
            consume(x_in)
            C, Cs = mod.fc1.in_channels, mod.fc1.out_channels
            pooled = new_ws(B * C * 4)
            scale = new_ws(B * C * 4)
            z = alloc_out(i, B, x_in.H, x_in.W, C)
            pd = ew_desc(a=x_in, C=C, Bn=B, Hn=x_in.H, Wn=x_in.W, alpha=1.0 / (x_in.H * x_in.W))
            pparts = new_ws(L.SE_POOL_SPLITS * B * C * 4)     # partial sums of the pixel-split reduction (dyk_se_pool)
            later(lambda pd=pd, pooled=pooled, pparts=pparts: (setattr(pd, "aux", ws.ptr(pooled)), setattr(pd, "aux2", ws.ptr(pparts))))
            plan.fwd.append((L.OP_SE_POOL, pd))
            pre = "module_list.%d." % i
            fd = L.DykSeFcDesc()
            plan._keep.append(fd)
            fd.w1, fd.b1 = store.p_ptr(pre + "fc1.weight"), store.p_ptr(pre + "fc1.bias")
            fd.w2, fd.b2 = store.p_ptr(pre + "fc2.weight"), store.p_ptr(pre + "fc2.bias")
            fd.B, fd.C, fd.Cs = B, C, Cs
            fcws = new_ws(B * (C + 2 * Cs) * 4)               # h | dt1 | t2: parked by the forward call for the backward one
            later(lambda fd=fd, pooled=pooled, scale=scale, fcws=fcws: (
                setattr(fd, "pooled", ws.ptr(pooled)), setattr(fd, "scale", ws.ptr(scale)), setattr(fd, "ws", ws.ptr(fcws))))
            plan.fwd.append((L.OP_SE_FC_FWD, fd))
            sd = ew_desc(a=x_in, out=z, C=C, Bn=B, Hn=x_in.H, Wn=x_in.W)
            later(lambda sd=sd, scale=scale: setattr(sd, "p0", ws.ptr(scale)))
            plan.fwd.append((L.OP_SE_SCALE, sd))
            rec.update(x=x_in, z=z, pooled=pooled, scale=scale, C=C, Cs=Cs, fcws=fcws)
            cur = z
        elif t == "maxpool":
            x_in = cur
            consume(x_in)
            k, stride = m["size"], m["stride"]
            if not (1 <= k <= 15 and 1 <= stride <= 8):
                raise NotImplementedError("[maxpool] size %s stride %s" % (k, stride))
            mp = (k - 1) // 2                                   # nn.MaxPool2d(k, stride, padding=(k-1)//2), models.py:91-94
            Hp, Wp = (x_in.H + 2 * mp - k) // stride + 1, (x_in.W + 2 * mp - k) // stride + 1
            z = alloc_out(i, B, Hp, Wp, x_in.C)
            amax = new_ws(x_in.npix * x_in.C) if training else None
            pd = ew_desc(a=x_in, out=z, Bn=B, Hn=x_in.H, Wn=x_in.W, k=k)
            pd.slots = stride
            if amax is not None:
                later(lambda pd=pd, amax=amax: setattr(pd, "aux", ws.ptr(amax)))
            plan.fwd.append((L.OP_MAXPOOL_FWD, pd))
            rec.update(x=x_in, z=z, amax=amax, k=k, stride=stride)
            cur = z
        elif t == "upsample":
            x_in = cur
            consume(x_in)
            if m["stride"] != 2:
                raise NotImplementedError("[upsample] stride %s" % m["stride"])
            z = alloc_out(i, B, 2 * x_in.H, 2 * x_in.W, x_in.C)
            plan.fwd.append((L.OP_UPSAMPLE_FWD, ew_desc(a=x_in, out=z, Bn=B, Hn=x_in.H, Wn=x_in.W)))
            rec.update(x=x_in, z=z)
            cur = z
        elif t == "yolo":
            y_in = cur                       # head conv output, fp32, ld = HEAD_LD
            consume(y_in)
            na, no = mod.na, mod.no
            ny, nx = y_in.H, y_in.W
            p = torch.empty((B, na, ny, nx, no), dtype=torch.float32, device=device)
            plan.p_out.append(p)
            hd = misc()
            hd.p[1] = p.data_ptr()
            hd.i[0], hd.i[1], hd.i[2], hd.i[3], hd.i[4], hd.i[5] = B, ny, nx, na, no, y_in.ld
            later(lambda hd=hd, y_in=y_in: hd.p.__setitem__(0, ptr_of(y_in)))
            plan.fwd.append((L.OP_HEAD_PERMUTE_FWD, hd))
            rec.update(y=y_in, na=na, no=no, ny=ny, nx=nx, head=head_idx, p=p, stride=mod.stride,
                       anchor_vec=[float(v) for v in mod.anchor_vec.reshape(-1).tolist()])
            yolo_rows.append(na * ny * nx)
            head_idx += 1
        elif t == "dropout":
            # identity at inference (models.py:96-98 builds nn.Dropout); a training plan would silently diverge from
            # the reference, so it is refused like the other unbuilt constructs (no shipped cfg has a [dropout])
            if training and float(m.get("probability", 0.5)) > 0:
                raise NotImplementedError("[dropout] in a training plan (layer %d): only the inference identity is built" % i)
            rec["alias"] = True
        else:
            raise NotImplementedError("cfg section [%s] (layer %d) is not built yet" % (t, i))
        outs.append(cur)
        info.append(rec)

    # eval: decode every head into one [B, rows, no] buffer (models.py:258,315)
    if not training:
        rows_total = sum(yolo_rows)
        no = info[model.yolo_layers[0]]["no"]
        plan.io = torch.empty((B, rows_total, no), dtype=torch.float32, device=device)
        r0 = 0
        for j in model.yolo_layers:
            rec = info[j]
            dd = L.DykDecodeDesc()
            plan._keep.append(dd)
            dd.p, dd.io = rec["p"].data_ptr(), plan.io.data_ptr()
            dd.B, dd.na, dd.ny, dd.nx, dd.no = B, rec["na"], rec["ny"], rec["nx"], rec["no"]
            dd.rows_total, dd.row_offset, dd.v4, dd.stride = rows_total, r0, 1 if v4 else 0, float(rec["stride"])
            for q, v in enumerate(rec["anchor_vec"]):
                dd.anchor_vec[q] = v
            plan.fwd.append((L.OP_YOLO_DECODE, dd))
            r0 += rec["na"] * rec["ny"] * rec["nx"]
    return stats_memset, fwd_start

def _emit_backward(cx):
    """The backward command list of a training plan (compile_plan's second stage): generated statically from the records the
    forward emission left per section (`info`), store-vs-accumulate resolved here, no autograd graph.  `cx` carries the
    emission context: the helpers that allocate tensors / descriptors and defer pointer assignments, the arenas, the cfg."""
    (B, code, concat_slot, cw, defs, es, ew_desc, grad_arena, info, joint_of, kpad_bytes, later, misc, mods, new_ws, nrefs, plan,
    producer_of, ptr_of, store, tcons, tight, ws) = (cx.B, cx.code, cx.concat_slot, cx.cw, cx.defs, cx.es, cx.ew_desc,
    cx.grad_arena, cx.info, cx.joint_of, cx.kpad_bytes, cx.later, cx.misc, cx.mods, cx.new_ws, cx.nrefs, cx.plan, cx.producer_of,
    cx.ptr_of, cx.store, cx.tcons, cx.tight, cx.ws)
    # parameters frozen with requires_grad_(False) (reference train.py:77-82): no weight gradient for them, and no
    # backward at all for the sections below the first one that owns a trainable parameter
    frozen = {e.name for e in store.entries if not e.param.requires_grad}
    first_trainable = min((e.layer for e in store.entries if e.param.requires_grad), default=len(defs))
    grads = {}           # tid -> TRef in the grad arena
    ginit = set()        # tids whose gradient buffer holds a value already
    red_offs = []        # (offset, bytes) fp64 reduction scratch zeroed at the start of backward

    def gref(t, ld=None, C=None):
        if t.tid not in grads:
            ldn = ld or t.ld
            Cn = C or t.C
            g = TRef("grad", grad_arena.alloc(t.npix * ldn * es, pad=kpad_bytes(ldn, es)), t.B, t.H, t.W, Cn, ldn, es, tid=t.tid)
            grads[t.tid] = g
        return grads[t.tid]

    def acc_flag(t):
        """store on first write, accumulate afterwards"""
        if t.tid in ginit:
            return L.EW_ACCUM
        ginit.add(t.tid)
        return 0

    def new_red(nbytes):
        off = new_ws(nbytes)
        red_offs.append((off, nbytes))
        return off

    pending_add = {}     # tid of a skip tensor -> gradient that reaches it over a fused plain [shortcut], not yet added

    def chain_consumer(a, before):
        """the conv whose data gradient can take the [shortcut] gradient of `a` as an addend and reduce the
        BatchNorm backward of a's producer in its epilogue (DYK_EPI_BNBWD | DYK_EPI_ADDEND): `a` is read by that
        conv and the [shortcut] only, the conv comes earlier than layer `before` and covers a in one launch"""
        if os.environ.get("DYK_BNBWD_FUSE", "1") == "0" or os.environ.get("DYK_DEBUG_PLAN"):
            return None
        prod = producer_of.get(a.tid)
        if prod is None or not prod.get("bn"):
            return None
        # readers: the conv, this [shortcut] and, when a is itself the output of a fused [shortcut], that section
        if tcons.get(a.tid, 0) != (3 if prod.get("fused_shortcut") else 2) or a.tid in ginit or a.tid in grads \
                or a.C % (16 // es):
            return None
        readers = [r for r in info[:before] if r.get("kind") == "conv" and r.get("x") is not None
                   and r["x"].tid == a.tid]
        if len(readers) != 1:
            return None
        r = readers[0]
        if r is prod or r.get("dw") or r.get("stem") or r["stride"] != 1 or r["i"] <= prod["i"]:
            return None
        if len(dgrad_classes(r["k"], r["pad"], 1, a.H, a.W)) != 1:
            return None
        return r

    def emit_conv_backward(rec, dy):
        """dy: gradient w.r.t. the conv's raw output (dtype), rows zero padded to a multiple of 32 channels"""
        k, stride, pad, cout, wname = rec["k"], rec["stride"], rec["pad"], rec["cout"], rec["wname"]
        x_in = rec["x"]
        if rec["dw"]:
            wd = L.DykDwDesc()
            plan._keep.append(wd)
            wd.dtype, wd.w, wd.dw = code, store.p_ptr(wname), store.g_ptr(wname)
            wd.B, wd.Hi, wd.Wi, wd.Ho, wd.Wo, wd.C = B, x_in.H, x_in.W, dy.H, dy.W, cout
            wd.k, wd.stride, wd.pad = k, stride, pad
            wd.ldx, wd.ldy = x_in.ld, dy.ld
            later(lambda wd=wd, x=x_in, dy=dy: (setattr(wd, "x", ptr_of(x)), setattr(wd, "y", ptr_of(dy))))
            pre = getattr(x_in, "pre", None)
            if pre is not None:                       # (the input is the producer's RAW output: z is formed on load)
                wd.pre_act = pre[1]
                later(lambda wd=wd, pre=pre: setattr(wd, "pre", ws.ptr(pre[0])))
            if wname not in frozen:
                plan.bwd.append((L.OP_DW_WGRAD, wd))
            if rec["i"] == first_trainable and rec["entry"]:
                return
            gx = gref(x_in)
            gd = L.DykDwDesc()
            plan._keep.append(gd)
            gd.dtype, gd.w = code, store.p_ptr(wname)
            gd.B, gd.Hi, gd.Wi, gd.Ho, gd.Wo, gd.C = B, x_in.H, x_in.W, dy.H, dy.W, cout
            gd.k, gd.stride, gd.pad = k, stride, pad
            gd.ldx, gd.ldy = gx.ld, dy.ld
            first = x_in.tid not in ginit
            gd.flags = acc_flag(x_in)
            later(lambda gd=gd, gx=gx, dy=dy: (setattr(gd, "x", ptr_of(gx)), setattr(gd, "y", ptr_of(dy))))
            # BatchNorm-backward reduce of the layer that produced x_in (the expansion conv of a MobileNet block) folded
            # into this data gradient, under the conditions of the MFMA conv's DYK_EPI_BNBWD (sole reader, first
            # writer of the gradient) -- LDS-tiled kernel only: stride 1, 3x3 / 5x5, bf16
            prod = producer_of.get(x_in.tid)
            if (first and prod is not None and prod is not rec and tcons.get(x_in.tid, 0) == 1
                    and stride == 1 and k in (3, 5) and code == L.DYK_BF16 and "vecs" in prod and prod["vs"] == 4 * x_in.C
                    and not os.environ.get("DYK_DEBUG_PLAN") and os.environ.get("DYK_BNBWD_FUSE", "1") != "0"
                    ):
                prod["red_fused"] = new_red(STAT_SLOTS * 2 * x_in.C * 8)
                prod["keep_dz"] = False
                gd.act, gd.stats_slots, gd.ldr = prod["act"], STAT_SLOTS, prod["y_raw"].ld
                later(lambda gd=gd, prod=prod: (
                    setattr(gd, "res", ptr_of(prod["y_raw"])), setattr(gd, "bn", ws.ptr(prod["vecs"])),
                    setattr(gd, "stats", ws.ptr(prod["red_fused"]))))
            plan.bwd.append((L.OP_DW_DGRAD, gd))
            return
        if "stem_direct" in rec:
            if wname in frozen:
                return
            f = rec["stem_direct"]
            wd = L.DykStemDesc()
            plan._keep.append(wd)
            wd.dtype, wd.B, wd.H, wd.W, wd.Cout, wd.k, wd.stride, wd.pad, wd.Ho, wd.Wo = (
                code, f.B, f.H, f.W, f.Cout, f.k, f.stride, f.pad, f.Ho, f.Wo)
            wd.lddy, wd.dw = dy.ld, store.g_ptr(wname)
            nsegs = f.B * f.Ho * ((f.Wo + 127) // 128)
            spw = max(1, (nsegs + 4095) // 4096)
            planes = (nsegs + spw - 1) // spw                            # == dyk_stem_wgrad_planes
            part = new_ws(planes * f.Cout * 27 * 4)
            later(lambda wd=wd, dy=dy, part=part: (setattr(wd, "dy", ptr_of(dy)), setattr(wd, "part", ws.ptr(part))))
            bap = rec.get("_bn_apply")
            if (bap is not None and bap[5] and code == L.DYK_BF16 and dy.ld == f.Cout and rec["y_raw"].ld == f.Cout
                    ):
                # BatchNorm-backward apply of the stem's own BatchNorm inside this weight gradient (uint8 images: decided per
                # call by the engine, dyk_stem_wgrad_bn_fusable): da and the raw output are read instead of dz, the separate
                # pass over the largest activation of the net is skipped (DYK_EW_SKIP on its descriptor)
                ap, da_ref, red_, vecs_, cout_, _ = bap
                wd.bn_slots = STAT_SLOTS
                wd.bn_dgamma, wd.bn_dbeta = ap.aux, ap.aux2
                later(lambda wd=wd, da_ref=da_ref, yr=rec["y_raw"], red_=red_, vecs_=vecs_: (
                    setattr(wd, "bn_da", ptr_of(da_ref)), setattr(wd, "bn_yraw", ptr_of(yr)),
                    setattr(wd, "bn_vecs", ws.ptr(vecs_)), setattr(wd, "bn_red", ws.ptr(red_))))
                plan.stem_fuse.append((ap, wd))
            plan.dyn_in.append((wd, [k_ for (d_, k_) in plan.dyn_in if d_ is f][0]))
            plan.bwd.append((L.OP_STEM_WGRAD, wd))
            return
        wg = rec["wgrad"]
        wd = L.DykWgradDesc()
        plan._keep.append(wd)
        wd.dtype = code
        wd.dw = store.g_ptr(wname)
        wd.ldx, wd.lddy = wg["x"].ld, dy.ld
        wd.B, wd.Hi, wd.Wi, wd.Cin = B, wg["Hi"], wg["Wi"], wg["Cin"]
        wd.Ho, wd.Wo, wd.Cout = dy.H, dy.W, cout
        wd.isy = wd.isx = wg["isy"]
        wd.ntaps = len(wg["taps"])
        for q, (ty, tx, wt) in enumerate(wg["taps"]):
            wd.tdy[q], wd.tdx[q], wd.twt[q] = ty, tx, wt
        wd.splits, wd.lddw = 0, wg["lddw"]
        later(lambda wd=wd, x=wg["x"], dy=dy: (setattr(wd, "x", ptr_of(x)), setattr(wd, "dy", ptr_of(dy))))
        if wname not in frozen:
            plan.bwd.append((L.OP_WGRAD, wd))
        if rec["stem"] or (rec["i"] == first_trainable and rec["entry"]):
            return                           # nothing trainable upstream: the input gradient is not needed
        gx = gref(x_in)
        first = x_in.tid not in ginit
        e = store.by_name[wname]
        kpad = _ru(cout, 32)
        if dy.ld < kpad and not tight:
            raise NotImplementedError("gradient rows narrower than the padded K (layer %d)" % rec["i"])
        if kpad != cout:
            wt_ptr = cw["Wt_pad"].data_ptr() + cw["bwd_pad_off"][wname] * es
        else:
            wt_ptr = cw["Wt"].data_ptr() + e.offset * es
        # BatchNorm-backward reduce of the producer of x_in folded into this data gradient (DYK_EPI_BNBWD): possible
        # when this launch is the only writer of the gradient (sole reader of the tensor, nothing accumulated yet)
        prod = producer_of.get(x_in.tid)
        addend = pending_add.pop(x_in.tid, None)
        if addend is not None and (not first or addend[1] is not rec or addend[0].ld != gx.ld):
            raise RuntimeError("residual-chain addend of layer %d lost its consumer" % rec["i"])
        fuse = addend is not None or (
            first and prod is not None and prod is not rec and tcons.get(x_in.tid, 0) == 1
            and x_in.C % (16 // es) == 0 and not os.environ.get("DYK_DEBUG_PLAN")
            and os.environ.get("DYK_BNBWD_FUSE", "1") != "0")
        if fuse:
            prod["red_fused"] = new_red(STAT_SLOTS * 2 * x_in.C * 8)
            prod["keep_dz"] = addend is not None
            bw = dict(act=prod["act"], raw=prod["y_raw"], vecs=prod["vecs"], vs=prod["vs"], red=prod["red_fused"])
        # ... or x_in is the output of the LAST [shortcut] of a residual chain, riding on its conv's normalise pass
        # (fused_shortcut): this launch is the only writer of that gradient, which the skip branch still needs as dz itself --
        # the keep-dz (chain) form of the epilogue without an addend (DYK_EPI_ADDEND, add == NULL)
        zero_chain = (not fuse and addend is None and first and prod is not None and prod is not rec
                      and prod.get("fused_shortcut") and prod.get("bn") and "vecs" in prod and tcons.get(x_in.tid, 0) == 2
                      and x_in.C % (16 // es) == 0 and stride == 1 and len(dgrad_classes(k, pad, 1, x_in.H, x_in.W)) == 1
                      and not os.environ.get("DYK_DEBUG_PLAN") and os.environ.get("DYK_BNBWD_FUSE", "1") != "0"
                      )
        if zero_chain:
            fuse = True
            prod["red_fused"] = new_red(STAT_SLOTS * 2 * x_in.C * 8)
            prod["keep_dz"] = True
            bw = dict(act=prod["act"], raw=prod["y_raw"], vecs=prod["vecs"], vs=prod["vs"], red=prod["red_fused"])
        # ... or of ALL the conv + BatchNorm sections concatenated into x_in (joint_of, decided in the forward): one launch,
        # replicas [slots][2][ctot]; every section's apply pass folds its own columns (DykEwDesc.H / W: replica stride and
        # offset of the second sum)
        jr = joint_of.get(x_in.tid) if not fuse else None
        if (jr is not None and first and tcons.get(x_in.tid, 0) == 1 and x_in.C == jr["ctot"] and x_in.C % (16 // es) == 0
                and all(tcons.get(pr_["z"].tid, 0) == 1 and pr_["z"].tid not in grads and pr_["z"].tid not in ginit
                        for pr_, _ in jr["parts"])):
            fuse = True
            jred = new_red(STAT_SLOTS * 2 * jr["ctot"] * 8)
            for pr_, c0_ in jr["parts"]:
                pr_["red_fused"], pr_["red_geom"], pr_["keep_dz"] = jred + 8 * c0_, (2 * jr["ctot"], jr["ctot"]), False
            bw = dict(act=jr["act"], raw=jr["raw"], vecs=jr["vecs"], vs=4 * jr["ctot"], red=jred)
        classes = dgrad_classes(k, pad, stride, x_in.H, x_in.W)
        # the parity classes of a strided conv's data gradient in one launch (DykConvDesc.ncls) when they all have
        # taps, share the launch grid and fit the tap table
        merged = (len(classes) in (2, 4) and all(c[4] for c in classes)
                  and len({(c[2], c[3]) for c in classes}) == 1
                  and sum(len(c[4]) for c in classes) <= L.MAX_TAPS
                  )
        if merged:
            classes = [(0, 0, classes[0][2], classes[0][3], [t for c in classes for t in c[4]], classes)]
        for cls in classes:
            (py, px, Hg, Wg, taps) = cls[:5]
            if not taps and not first:
                continue                      # nothing to accumulate for this parity class
            d = L.DykConvDesc()
            plan._keep.append(d)
            d.dtype = code
            d.w = wt_ptr
            d.B, d.Hi, d.Wi, d.Cin, d.Cout = B, dy.H, dy.W, kpad, x_in.C
            d.Hg, d.Wg, d.Ho, d.Wo = Hg, Wg, x_in.H, x_in.W
            d.isy = d.isx = 1
            d.osy = d.osx = stride
            d.ooy, d.oox = py, px
            d.ntaps = len(taps)
            for q, (ty, tx, wt) in enumerate(taps):
                d.tdy[q], d.tdx[q], d.twt[q] = ty, tx, wt
            if merged:
                d.ncls, q0 = len(cls[5]), 0
                for c, (cpy, cpx, _, _, ctaps) in enumerate(cls[5]):
                    d.cls_first[c], d.cls_ntaps[c], d.cls_ooy[c], d.cls_oox[c] = q0, len(ctaps), cpy, cpx
                    q0 += len(ctaps)
            d.ldx, d.ldy = dy.ld, gx.ld
            d.act, d.flags = 0, (0 if first else L.EPI_ACCUM)
            later(lambda d=d, dy=dy, gx=gx: (setattr(d, "x", ptr_of(dy)), setattr(d, "y", ptr_of(gx))))
            if fuse:
                d.act, d.flags, d.stats_slots, d.ldr = bw["act"], L.EPI_BNBWD, STAT_SLOTS, bw["raw"].ld
                if addend is not None:
                    d.flags |= L.EPI_ADDEND
                    later(lambda d=d, ad=addend[0]: setattr(d, "add", ptr_of(ad)))
                elif zero_chain:
                    d.flags |= L.EPI_ADDEND           # (add stays NULL)
                later(lambda d=d, bw=bw: (
                    setattr(d, "res", ptr_of(bw["raw"])), setattr(d, "scale", ws.ptr(bw["vecs"])),
                    setattr(d, "shift", ws.ptr(bw["vecs"] + bw["vs"])), setattr(d, "aux0", ws.ptr(bw["vecs"] + 2 * bw["vs"])),
                    setattr(d, "aux1", ws.ptr(bw["vecs"] + 3 * bw["vs"])), setattr(d, "stats", ws.ptr(bw["red"]))))
            plan.bwd.append((L.OP_CONV, d))
        ginit.add(x_in.tid)

    def conv_layer_backward(rec):
        """BatchNorm+activation backward (train-mode statistics) followed by the conv gradients"""
        z = rec["z"]
        if z.tid not in ginit:
            return                           # no gradient reaches this layer
        dz = gref(z)
        if rec["bn"]:
            cout, vecs, bnpre = rec["cout"], rec["vecs"], rec["bnpre"]
            fused_red = rec.get("red_fused")          # the producing dgrad already left da and the two sums
            keep_dz = bool(rec.get("keep_dz"))        # ... or, in a residual chain, dz itself (act' still to apply)
            act_bwd = 0 if (fused_red is not None and not keep_dz) else rec["act"]
            if fused_red is not None:
                red = fused_red
            else:
                red = new_red(STAT_SLOTS * 2 * cout * 8)
                r = ew_desc(a=dz, b=rec["y_raw"], act=rec["act"])
                r.slots = STAT_SLOTS
                later(lambda r=r, vecs=vecs, red=red, vs=rec["vs"]: (
                    setattr(r, "p0", ws.ptr(vecs)), setattr(r, "p1", ws.ptr(vecs + vs)),
                    setattr(r, "p2", ws.ptr(vecs + 2 * vs)), setattr(r, "p3", ws.ptr(vecs + 3 * vs)),
                    setattr(r, "red", ws.ptr(red))))
                plan.bwd.append((L.OP_BN_BWD_REDUCE, r))
            dyr = dz
            if os.environ.get("DYK_DEBUG_PLAN") or rec.get("dz_is_addend") or os.environ.get("DYK_KEEP_DZ"):
                # keep dz intact: per-layer gradient dumps / it is still to be added to the skip tensor's gradient /
                # DYK_KEEP_DZ: the SAME commands and fusions as the default plan, only the apply pass writes beside
                # its input instead of over it, so that both sides of every BatchNorm backward stay readable
                # (tests/test_gpu_bwd_bf16.py holds each section of the backward to the oracle on identical inputs)
                dyr = TRef("grad", grad_arena.alloc(dz.npix * dz.ld * es, pad=kpad_bytes(dz.ld, es)), dz.B, dz.H, dz.W, dz.C, dz.ld, es)
            rec["dz_ref"], rec["dy_raw_ref"] = dz, dyr
            ap = ew_desc(a=dz, b=rec["y_raw"], out=dyr, act=act_bwd)        # in place: dz (or da) -> dy_raw
            # the apply pass folds the replicas of the reduction itself and adds them to dgamma / dbeta
            ap.slots = STAT_SLOTS
            ap.aux, ap.aux2 = store.g_ptr(bnpre + "weight"), store.g_ptr(bnpre + "bias")
            if rec.get("red_geom"):           # the replicas are columns of a route's joint reduction (emit_conv_backward)
                ap.H, ap.W = rec["red_geom"]
            later(lambda ap=ap, vecs=vecs, red=red, vs=rec["vs"]: (
                setattr(ap, "p0", ws.ptr(vecs)), setattr(ap, "p1", ws.ptr(vecs + vs)),
                setattr(ap, "p2", ws.ptr(vecs + 2 * vs)), setattr(ap, "p3", ws.ptr(vecs + 3 * vs)),
                setattr(ap, "red", ws.ptr(red))))
            plan.bwd.append((L.OP_BN_BWD_APPLY, ap))
            # (the stem's weight gradient can do this pass on the fly: emit_conv_backward below, DykStemDesc.bn_fused)
            rec["_bn_apply"] = (ap, dz, red, vecs, cout, fused_red is not None and not keep_dz and act_bwd == 0 and dyr is dz)
            dz = dyr
        emit_conv_backward(rec, dz)

    memset_desc = misc()
    plan.bwd.append((L.OP_MEMSET, memset_desc))          # slot 0: clears the fp64 reduction scratch
    plan.bwd_marks = []                                  # (number of commands emitted, layer index) in backward order
    for i in range(len(defs) - 1, -1, -1):
        rec = info[i]
        t = rec["kind"]
        plan.bwd_marks.append((len(plan.bwd), i))
        if rec.get("alias") or i < first_trainable:
            continue
        if t == "yolo":
            y_in = rec["y"]
            gy = gref(y_in, ld=HEAD_LD)          # dtype rows of 32 channels, zero padded
            hd = misc()
            hd.p[2] = store.g_ptr(info[i - 1]["bias_name"])
            hd.i[0], hd.i[1], hd.i[2], hd.i[3], hd.i[4], hd.i[5], hd.i[6] = B, rec["ny"], rec["nx"], rec["na"], rec["no"], HEAD_LD, code
            later(lambda hd=hd, gy=gy: hd.p.__setitem__(1, ptr_of(gy)))
            plan.dyn_dp.append((hd, rec["head"]))
            plan.bwd.append((L.OP_HEAD_PERMUTE_BWD, hd))
            ginit.add(y_in.tid)
        elif t == "conv":
            conv_layer_backward(rec)
        elif t == "dwsep":
            for sub in reversed(rec["parts"]):
                conv_layer_backward(sub)
        elif t == "inception":
            z = rec["z"]
            if z.tid not in ginit:
                continue
            go = gref(z)
            for recs, c0, co in rec["branches"]:           # the branch outputs' gradients are slices of the concat's
                zb = recs[-1]["z"]
                g = go.chan_slice(c0, co)
                g.tid = zb.tid
                grads[zb.tid] = g
                ginit.add(zb.tid)
            for recs, c0, co in reversed(rec["branches"]):
                for sub in reversed(recs):
                    conv_layer_backward(sub)
            pooled, x_in = rec["pooled"], rec["x"]
            if pooled.tid in ginit:
                pd = ew_desc(a=gref(pooled), out=gref(x_in), Bn=B, Hn=x_in.H, Wn=x_in.W, k=3, flags=acc_flag(x_in))
                later(lambda pd=pd, rec=rec: setattr(pd, "aux", ws.ptr(rec["amax"])))
                plan.bwd.append((L.OP_MAXPOOL_BWD, pd))
        elif t == "route":
            out = rec["out"]
            if out.tid not in ginit:
                continue
            go = gref(out)
            for (s, c0), q in zip(rec["parts"], mods[i].layers):
                if concat_slot.get(q, (None,))[0] == i and nrefs[q] == 1 and s.tid not in grads:
                    # produced in place and read by nobody else: its gradient IS the slice of the concat gradient
                    g = go.chan_slice(c0, s.C)
                    g.tid = s.tid
                    grads[s.tid] = g
                    ginit.add(s.tid)
                    continue
                gs = gref(s)
                fl = acc_flag(s)
                plan.bwd.append((L.OP_AXPBY, ew_desc(a=go.chan_slice(c0, s.C), out=gs, C=s.C, flags=fl)))
        elif t == "shortcut":
            z = rec["z"]
            if z.tid not in ginit:
                continue
            dz = gref(z)
            x_in, a = rec["x"], rec["a"]
            if rec.get("fused"):
                # z is the conv's own output: its gradient stays where it is, the skip branch gets a copy --
                # or, in a residual chain, the data gradient of the skip tensor's other reader adds it on the fly
                r = chain_consumer(a, i)
                if r is not None and dz.ld == a.ld:
                    pending_add[a.tid] = (dz, r)
                    info[i - 1]["dz_is_addend"] = True
                    continue
                plan.bwd.append((L.OP_AXPBY, ew_desc(a=dz, out=gref(a), flags=acc_flag(a))))
                continue
            # mismatched channel counts: x gets dz whole; a gets the first min(nx, na) channels of dz, and zeros in its
            # remaining channels when this is the first gradient written into it (nx < na)
            nx, na = x_in.C, a.C
            Cm = min(nx, na)
            dza = dz if nx <= na else dz.chan_slice(0, Cm)

            def a_grad(p0_off=None):
                ga = gref(a)
                first = a.tid not in ginit
                e = ew_desc(a=dza, out=ga if nx >= na else ga.chan_slice(0, Cm), C=Cm, flags=acc_flag(a))
                if p0_off is not None:
                    later(lambda e=e, p0_off=p0_off: setattr(e, "p0", ws.ptr(p0_off)))
                plan.bwd.append((L.OP_AXPBY, e))
                if nx < na and first:
                    tail = ga.chan_slice(nx, na - nx)
                    plan.bwd.append((L.OP_AXPBY, ew_desc(a=tail, out=tail, C=na - nx, alpha=0.0)))     # alpha = 0: `a` is not read

            if rec["weighted"] and nx == na:
                # weighted fusion, equal channel counts: per source ONE pass over dz leaves its fusion-weight dot product AND
                # its scaled gradient copy (dyk_dot with `out`, round 5: two dots + two scaled copies read dz four times)
                red = new_red(16)
                weff = rec["weff"]
                for q, src in enumerate((x_in, a)):
                    dq = ew_desc(a=dz, b=src, out=gref(src), flags=acc_flag(src))
                    later(lambda dq=dq, red=red, weff=weff, q=q: (setattr(dq, "red", ws.ptr(red + 8 * q)), setattr(dq, "p0", ws.ptr(weff + 4 * q))))
                    plan.bwd.append((L.OP_DOT, dq))
                pm = misc()
                pm.p[0], pm.p[2] = store.p_ptr("module_list.%d.w" % i), store.g_ptr("module_list.%d.w" % i)
                pm.i[0] = 2
                later(lambda pm=pm, red=red: pm.p.__setitem__(1, ws.ptr(red)))
                plan.bwd.append((L.OP_WFUSE_BWD_PARAMS, pm))
            elif rec["weighted"]:
                red = new_red(16)
                d0 = ew_desc(a=dz, b=x_in)
                later(lambda d0=d0, red=red: setattr(d0, "red", ws.ptr(red)))
                plan.bwd.append((L.OP_DOT, d0))
                d1 = ew_desc(a=dza, b=a if nx >= na else a.chan_slice(0, Cm), C=Cm)
                later(lambda d1=d1, red=red: setattr(d1, "red", ws.ptr(red + 8)))
                plan.bwd.append((L.OP_DOT, d1))
                pm = misc()
                pm.p[0], pm.p[2] = store.p_ptr("module_list.%d.w" % i), store.g_ptr("module_list.%d.w" % i)
                pm.i[0] = 2
                later(lambda pm=pm, red=red: pm.p.__setitem__(1, ws.ptr(red)))
                plan.bwd.append((L.OP_WFUSE_BWD_PARAMS, pm))
                weff = rec["weff"]
                e = ew_desc(a=dz, out=gref(x_in), flags=acc_flag(x_in))
                later(lambda e=e, weff=weff: setattr(e, "p0", ws.ptr(weff)))
                plan.bwd.append((L.OP_AXPBY, e))
                a_grad(weff + 4)
            else:
                plan.bwd.append((L.OP_AXPBY, ew_desc(a=dz, out=gref(x_in), flags=acc_flag(x_in))))
                a_grad()
        elif t == "se":
            z = rec["z"]
            if z.tid not in ginit:
                continue
            dz = gref(z)
            x_in, C, Cs = rec["x"], rec["C"], rec["Cs"]
            dscale = new_ws(B * C * 4)
            dpooled = new_ws(B * C * 4)
            fcws = rec["fcws"]
            pd = ew_desc(a=dz, b=x_in, C=C, Bn=B, Hn=x_in.H, Wn=x_in.W, alpha=1.0)
            pparts = new_ws(L.SE_POOL_SPLITS * B * C * 4)
            later(lambda pd=pd, dscale=dscale, pparts=pparts: (setattr(pd, "aux", ws.ptr(dscale)), setattr(pd, "aux2", ws.ptr(pparts))))
            plan.bwd.append((L.OP_SE_POOL, pd))
            pre = "module_list.%d." % i
            fd = L.DykSeFcDesc()
            plan._keep.append(fd)
            fd.w1, fd.b1 = store.p_ptr(pre + "fc1.weight"), store.p_ptr(pre + "fc1.bias")
            fd.w2, fd.b2 = store.p_ptr(pre + "fc2.weight"), store.p_ptr(pre + "fc2.bias")
            fd.B, fd.C, fd.Cs = B, C, Cs
            # the data half (dpooled) stays on the chain to dx; the parameter half is a command of its own that the
            # dependency scheduler places like a weight gradient (one call held the chain for three launches: 24 us on
            # each of the 19 squeeze-excitation blocks of the MobileNetV3 cfg).  DYK_SE_SPLIT=0: one command, as before
            split = True
            fg = L.DykSeFcDesc() if split else fd
            if split:
                plan._keep.append(fg)
                fg.w1, fg.b1, fg.w2, fg.b2, fg.B, fg.C, fg.Cs = fd.w1, fd.b1, fd.w2, fd.b2, B, C, Cs
            fg.dw1, fg.db1 = store.g_ptr(pre + "fc1.weight"), store.g_ptr(pre + "fc1.bias")
            fg.dw2, fg.db2 = store.g_ptr(pre + "fc2.weight"), store.g_ptr(pre + "fc2.bias")

            def se_ptrs(fd=fd, fg=fg, rec=rec, dscale=dscale, dpooled=dpooled, fcws=fcws):
                for q in {id(fd): fd, id(fg): fg}.values():
                    q.pooled, q.dscale, q.ws = ws.ptr(rec["pooled"]), ws.ptr(dscale), ws.ptr(fcws)
                fd.dpooled = ws.ptr(dpooled)
            later(se_ptrs)
            plan.bwd.append((L.OP_SE_FC_BWD, fd))
            if split:
                plan.bwd.append((L.OP_SE_FC_BWD, fg))
            gx = gref(x_in)
            # BatchNorm-backward reduce of the conv + BatchNorm layer that produced x_in inside this launch (dyk_se_scale
            # with `red`): this block is the only reader of x_in and writes its gradient first (and last); the gradient
            # stays dz (keep_dz: the apply pass forms act' itself).  22 launches of the MobileNetV3 / 3 of the target cfg
            prod = producer_of.get(x_in.tid)
            fuse = (x_in.tid not in ginit and prod is not None and prod.get("bn") and "vecs" in prod and prod["vs"] == 4 * C
                    and tcons.get(x_in.tid, 0) == 1 and x_in.C % (16 // es) == 0 and prod["y_raw"].C == C
                    and not os.environ.get("DYK_DEBUG_PLAN") and os.environ.get("DYK_BNBWD_FUSE", "1") != "0"
                    )
            sd = ew_desc(a=dz, b=prod["y_raw"] if fuse else None, out=gx, C=C, Bn=B, Hn=x_in.H, Wn=x_in.W,
                         alpha=1.0 / (x_in.H * x_in.W), flags=acc_flag(x_in), act=prod["act"] if fuse else 0)
            later(lambda sd=sd, rec=rec, dpooled=dpooled: (setattr(sd, "p0", ws.ptr(rec["scale"])), setattr(sd, "p1", ws.ptr(dpooled))))
            if fuse:
                prod["red_fused"] = new_red(STAT_SLOTS * 2 * C * 8)
                prod["keep_dz"] = True
                sd.slots = STAT_SLOTS
                later(lambda sd=sd, prod=prod: (setattr(sd, "p2", ws.ptr(prod["vecs"])), setattr(sd, "red", ws.ptr(prod["red_fused"]))))
            plan.bwd.append((L.OP_SE_SCALE, sd))
        elif t == "maxpool":
            z = rec["z"]
            if z.tid not in ginit:
                continue
            dz = gref(z)
            x_in = rec["x"]
            gx = gref(x_in)
            pd = ew_desc(a=dz, out=gx, Bn=B, Hn=x_in.H, Wn=x_in.W, k=rec["k"], flags=acc_flag(x_in))
            pd.slots = rec["stride"]
            later(lambda pd=pd, rec=rec: setattr(pd, "aux", ws.ptr(rec["amax"])))
            plan.bwd.append((L.OP_MAXPOOL_BWD, pd))
        elif t == "upsample":
            z = rec["z"]
            if z.tid not in ginit:
                continue
            dz = gref(z)
            x_in = rec["x"]
            gx = gref(x_in)
            plan.bwd.append((L.OP_UPSAMPLE_BWD, ew_desc(a=dz, out=gx, C=x_in.C, Bn=B, Hn=x_in.H, Wn=x_in.W, flags=acc_flag(x_in))))
    # zero the fp64 reduction scratch before anything accumulates into it
    plan.bwd_marks.append((len(plan.bwd), -1))
    if not red_offs:
        red_offs.append((new_ws(256), 256))
    lo = min(o for o, n in red_offs)
    hi = max(o + n for o, n in red_offs)
    memset_desc.n, memset_desc.i[0] = hi - lo, 0
    later(lambda ms=memset_desc, lo=lo: ms.p.__setitem__(0, ws.ptr(lo)))
    return grads


# ======================================================================================
# Passes over the emitted command lists (compile_plan's second half)
def _materialise(plan, device, pending, stats_memset, st_arena):
    """arenas get their memory, every deferred pointer assignment of the emission runs"""
    for a in plan.arenas.values():
        a.materialize(device)
    if stats_memset is not None:
        stats_memset.p[0] = st_arena.ptr(0)
        stats_memset.n, stats_memset.i[0] = max(st_arena.size, 256), 0
    for fn in pending:
        fn()


def _backbone_force_layers(defs, mods, second):
    """the two backbones of a dual-stream net are enqueued interleaved (dyk_run_commands_overlap): a plane fold must not span the
    boundary between them, or it could run before weight gradients that precede it in the list"""
    force = set()
    if second is not None and 0 < second < len(defs):
        force.add(second - 1)
        for j in range(second, len(defs)):
            if defs[j]["type"] in ("route", "shortcut") and any(q < second for q in mods[j].layers):
                force.add(j - 1)
                break
    return force


def _post_passes(plan, store, device, dry, training, stats_memset, force_layers):
    """fuse (BatchNorm-backward reduces onto the last contributor's data gradient) -> tune (tile / kernel per problem) -> group
    (weight gradients of one geometry into one launch) -> scratch (split-K slabs and counters, weight-gradient planes and their
    fold commands, device tables of the grouped launches)"""
    if training and os.environ.get("DYK_BNBWD_FUSE", "1") != "0" and not os.environ.get("DYK_DEBUG_PLAN"):
        _fuse_late_reduces(plan, store)
    if not dry and os.environ.get("DYK_AUTOTUNE", "1") != "0":
        autotune(plan, _TUNE_CACHE)
    elif training:
        _default_wgrad_tunes(plan)
    if training and os.environ.get("DYK_WGRAD_PARTIALS", "1") != "0":
        _group_wgrads(plan, store)
    plan.sk_ws = plan.sk_cnt = None
    plan.sk_bytes = 0
    if not dry:
        _setup_splitk(plan, device)
        if training and plan.sk_cnt is not None:
            # the pass's statistics memset (head of the forward list, a barrier for the scheduler) also zeroes the tile
            # counters: they re-arm themselves, but only if every split-K launch runs to its end
            stats_memset.p[1], stats_memset.i[1] = plan.sk_cnt.data_ptr(), 4 * plan.sk_cnt.numel()
    plan.part = None
    plan.bwd_cut_ok = None
    if training and not dry and os.environ.get("DYK_WGRAD_PARTIALS", "1") != "0":
        _setup_wgrad_partials(plan, store, device, set(force_layers) | getattr(plan, "_wg_protect_layers", set()))
    if training:
        _finish_wgrad_groups(plan, device)


def _assign_lanes(plan, defs, mods, second, fwd_start, training):
    """branch lanes of the round-1 executor (dyk_run_commands_overlap, DYK_SCHED=lanes): sections [second, F) -- the second
    backbone up to the first section that reads anything of the first one -- are independent of sections [0, second)"""
    plan.fwd_lanes, plan.bwd_lanes = {}, {}
    if second is not None and 0 < second < len(defs):
        F = len(defs)
        for j in range(second, len(defs)):
            if defs[j]["type"] in ("route", "shortcut") and any(q < second for q in mods[j].layers):
                F = j
                break
        if second < F < len(defs):
            lo, hi = fwd_start[second], fwd_start[F]
            if 0 < lo < hi < len(plan.fwd):
                plan.fwd_lanes[1 if training else 0] = 2            # fork at the start (behind the statistics memset)
                for q in range(lo, hi):
                    plan.fwd_lanes[q] = plan.fwd_lanes.get(q, 0) | 1
                plan.fwd_lanes[hi] = plan.fwd_lanes.get(hi, 0) | 4  # join in front of the fusion section
            if training:
                mark = {li: n for n, li in plan.bwd_marks}          # commands emitted before section li's backward
                b0, a0 = mark.get(F - 1), mark.get(second - 1)
                if b0 is not None and a0 is not None and 0 < b0 < a0 < len(plan.bwd):
                    plan.bwd_lanes[b0] = 2                          # fork once the fusion sections' gradients exist
                    for q in range(a0, len(plan.bwd)):
                        plan.bwd_lanes[q] = plan.bwd_lanes.get(q, 0) | 1
    # the side stream finishes last (its launches contend with the chain for CUs): the weight gradients of the last
    # sections differentiated stay on their own streams instead (measured on the per-stream timeline, tools/trace_timeline.py)
    if training:
        wg = [q for q, (op, _) in enumerate(plan.bwd) if op in (L.OP_WGRAD, L.OP_DW_WGRAD)]
        ntail = 0 if plan.part is not None else 20
        for q in wg[len(wg) - ntail:] if ntail > 0 else []:
            plan.bwd_lanes[q] = plan.bwd_lanes.get(q, 0) | 8


def _layer_maps(plan, model, defs, mods, second, fwd_start, training):
    """cfg section of every command (dyk/twins.py pairs the commands of twin sections into two-problem launches)"""
    from . import twins
    plan.twin_layer = twins.twin_layers(defs, mods, second)
    plan.fwd_layer = [-1] * len(plan.fwd)
    for li, c0 in enumerate(fwd_start):
        c1 = fwd_start[li + 1] if li + 1 < len(fwd_start) else (len(plan.fwd) - (0 if training else len(model.yolo_layers)))
        for q in range(c0, c1):
            plan.fwd_layer[q] = li
    plan.bwd_layer = [-1] * len(plan.bwd)
    if training:
        marks = plan.bwd_marks
        for k in range(len(marks) - 1):
            for q in range(marks[k][0], marks[k + 1][0]):
                plan.bwd_layer[q] = marks[k][1]


# ======================================================================================
def _fuse_late_reduces(plan, store):
    """BatchNorm-backward reduce passes whose dz has SEVERAL contributors (CSP splits, routes, weighted fusions): the LAST
    contribution, when it is an accumulating MFMA data gradient covering the whole tensor in one launch, takes the reduce
    into its epilogue in chain mode -- DYK_EPI_BNBWD | DYK_EPI_ADDEND with `add` = its own output: y = acc + y exactly
    as the accumulate epilogue would leave it, and sum(da), sum(da * xhat) of that final dz go to the replicas the apply
    pass folds.  Decided on the RESOLVED command list (who writes the region last is read off the descriptors' access
    sets, dyk/sched.py), independent of how the commands were emitted; the reduce command is dropped."""
    from . import sched
    mem = sched.Memory(plan, store)
    cmds = plan.bwd
    acc = [sched.accesses(op, d, mem, plan) for op, d in cmds]
    removed = []
    for ri, (op, r) in enumerate(cmds):
        if op != L.OP_BN_BWD_REDUCE:
            continue
        es = 2 if r.dtype == L.DYK_BF16 else 4
        target = mem.block(r.a, r.lda * es, r.C * es)
        if target is None:
            continue
        wi = None
        for j in range(ri - 1, -1, -1):
            if acc[j][2]:
                break
            if any(w.overlaps(target) for w in acc[j][1]):
                wi = j
                break
        if wi is None:
            continue
        wop, w = cmds[wi]
        if wop != L.OP_CONV or w.flags != L.EPI_ACCUM or w.ncls > 1 or w.dtype != r.dtype:
            continue
        if w.y != r.a or w.ldy != r.lda or w.Cout != r.C or w.B * w.Ho * w.Wo != r.npix:
            continue
        if w.osy != 1 or w.osx != 1 or w.ooy or w.oox or w.Hg != w.Ho or w.Wg != w.Wo:
            continue
        if w.Cout % (16 // es) or (w.ldy * es) % 16 or (r.ldb * es) % 16 or w.y % 16 or r.b % 16:
            continue
        w.flags = L.EPI_BNBWD | L.EPI_ADDEND
        w.add, w.res, w.ldr = w.y, r.b, r.ldb
        w.scale, w.shift, w.aux0, w.aux1 = r.p0, r.p1, r.p2, r.p3
        w.stats, w.stats_slots, w.act = r.red, (r.slots if r.slots > 0 else 1), r.act
        acc[wi] = sched.accesses(wop, w, mem, plan)
        removed.append(ri)
    if not removed:
        return
    gone = set(removed)
    plan.bwd = [c for q, c in enumerate(cmds) if q not in gone]
    removed.sort()
    import bisect
    plan.bwd_marks = [(cnt - bisect.bisect_left(removed, cnt), layer) for cnt, layer in plan.bwd_marks]
    plan.late_fused = len(removed)


def _setup_splitk(plan, device):
    """scratch of the convolutions the tuner runs with split-K across workgroups (DykConvDesc.splitk > 1): one slab set and
    one run of tile counters PER COMMAND (commands on different streams must not share them; the scheduler need not know these
    blocks: nobody else touches them), carved from one allocation; counters start at zero and re-arm themselves"""
    lib = L.load()
    todo, nbytes, nwords = [], 0, 0
    for op, d in plan.fwd + plan.bwd:
        if op != L.OP_CONV or d.splitk <= 1:
            continue
        nt = ctypes.c_int32(0)
        need = int(lib.dyk_conv_splitk_ws_bytes(ctypes.byref(d), ctypes.byref(nt)))
        if need <= 0:
            d.splitk = 0
            continue
        todo.append((d, nbytes, need, nwords, nt.value))
        nbytes += _ru(need, 256)
        nwords += _ru(nt.value, 64)
    if not todo:
        return
    plan.sk_ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    plan.sk_cnt = torch.zeros(nwords, dtype=torch.int32, device=device)
    plan.sk_bytes = nbytes + 4 * nwords
    for d, off, need, woff, nt in todo:
        d.sk_ws, d.sk_ws_bytes = plan.sk_ws.data_ptr() + off, need
        d.sk_cnt, d.sk_cnt_n = plan.sk_cnt.data_ptr() + 4 * woff, nt


def _default_wgrad_tunes(plan):
    """without the autotuner (DYK_AUTOTUNE=0, dry plans): the kernels the tuner ends up with on the target cfg -- row-block for
    3x3 / pad 1, pixel-streaming for 1x1 / stride 1 (bf16), the per-tap kernel for everything else -- with their default
    configurations, so that untuned plans (the pinned-tile tests, the CPU-side scheduler tests) run the product's kernels"""
    lib = L.load()
    for op, d in plan.bwd:
        if op != L.OP_WGRAD or d.tune:
            continue
        for v, t in ((2, 2 | (1 << 8) | (2 << 28) | WGRAD_EXCLUSIVE), (3, (3 << 28) | 3 | WGRAD_EXCLUSIVE)):
            d.tune = t
            if (v == 2 and os.environ.get("DYK_WGRAD_RB", "1") == "0") or (v == 3 and os.environ.get("DYK_WGRAD_PS", "1") == "0") \
                    or lib.dyk_conv_wgrad_variant(ctypes.byref(d)) != v:
                d.tune = 0
                continue
            break


# members of a grouped weight-gradient launch lie within this many backward commands of each other: a stage or two of one
# backbone (8 residual units ~ 60-100 commands).  Without a window twin layers of the two backbones -- different buffers, no
# conflict -- would share a launch: the first one's gradient would wait for the whole other backbone, and no data-parallel
# bucket could close in between (tests/test_ddp_gloo.py counts the buckets).  Measured in-call (r6_ab_wgrad_group_knobs.log):
# 50 / 100 / 200 commands 26.96 / 26.91 / 26.67 ms; 8 / 16 / 32 members per launch and 128 / 256 / 512 workgroups within +-0.1
WGRAD_GROUP_WINDOW = 200


def _group_wgrads(plan, store):
    """Grouped weight-gradient launches (round 6, DykWgradDesc.group).  The repeated units of a stage have weight gradients of
    ONE geometry that become ready one after the other and are nobody's input before the gradient fold / the optimizer; as
    183 separate launches each pays its launch, prologue and epilogue and needs many K splits (= partial planes) to occupy the
    chip.  Here the weight gradients of equal signature (row-block 3x3 / pixel-streaming 1x1 kernels) are collected into ONE
    launch at the position of the LAST member, as far as the access sets allow: a member may move from position i to j only if
    no command in (i, j] writes what it reads (x, dy: gradient buffers are recycled) or touches what it writes.  The members
    then share the chip: K splits per member = what fills 256 workgroups over the whole group (planes 32 -> 4 on the 64x80
    stage).  Rewrites plan.bwd / plan.bwd_marks; every member's result is what its own launch would give with that split
    count."""
    plan._wg_groups, plan._wg_group_info = {}, []
    maxg = int(os.environ.get("DYK_WGRAD_GROUP", "16"))
    if maxg < 2 or not plan.bwd:
        return
    from . import sched
    lib = L.load()
    cmds = plan.bwd
    n = len(cmds)
    cand = {}
    for i, (op, d) in enumerate(cmds):
        if op != L.OP_WGRAD or d.twin or d.part or d.sk_cnt or not (d.tune & WGRAD_EXCLUSIVE):
            continue
        v = lib.dyk_conv_wgrad_variant(ctypes.byref(d))
        if v not in (2, 3):
            continue
        key = (v, d.dtype, d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.isy, d.ntaps, d.tune, d.splits, d.ldx, d.lddy, d.lddw)
        cand.setdefault(key, []).append(i)
    cand = {k: v for k, v in cand.items() if len(v) >= 2}
    if not cand:
        return
    mem = sched.Memory(plan, store)
    acc = [sched.accesses(op, d, mem, plan) for op, d in cmds]

    def conflict(i, k):
        Ri, Wi, _ = acc[i]
        Rk, Wk, bk = acc[k]
        return bk or any(w.overlaps(r) for w in Wk for r in Ri) or any(a.overlaps(w) for w in Wi for a in Rk + Wk)

    late = {}
    for idxs in cand.values():
        for i in idxs:
            j = i
            for k in range(i + 1, n):
                if conflict(i, k):
                    break
                j = k
            late[i] = j
    layer_at = [-1] * n                       # cfg section whose backward a command belongs to (plan.bwd_marks)
    marks = plan.bwd_marks
    for k in range(len(marks)):
        hi = marks[k + 1][0] if k + 1 < len(marks) else n
        for q in range(marks[k][0], hi):
            layer_at[q] = marks[k][1]
    # positions where the data-parallel exchange closes its default buckets (dyk/ddp.py GEOMETRIC: 50 | 80 | 95 | 99 % of the
    # gradient buffer in backward order; 95 % is also where the one-GPU optimizer step starts early): no launch may hold back a
    # gradient across one of them -- a member before the cut, the launch behind it -- or the bucket could not close there
    total = store.total
    first_off = {}
    for e in store.entries:
        first_off.setdefault(e.layer, e.offset)
    cuts = [total - int(f * total) for f in (0.5, 0.8, 0.95, 0.99)]
    protected = []
    for k in range(1, len(marks)):
        c_end, layer_done = marks[k][0], marks[k - 1][1]
        lo = min((o for l, o in first_off.items() if l >= layer_done), default=total)
        while cuts and lo <= cuts[0]:
            protected.append(c_end)
            cuts.pop(0)
    pos_layer = {cnt: layer for cnt, layer in marks}
    plan._wg_protect_layers = {pos_layer[c] for c in protected}      # (the plane fold is forced there: _setup_wgrad_partials)
    groups = []
    for key, idxs in cand.items():
        cur, lim = [idxs[0]], late[idxs[0]]
        for i in idxs[1:] + [None]:
            if (i is not None and i <= lim and len(cur) < maxg and i - cur[0] <= WGRAD_GROUP_WINDOW
                    and not any(cur[0] < c <= i for c in protected)):
                cur.append(i)
                lim = min(lim, late[i])
                continue
            if len(cur) >= 2:
                groups.append(cur)
            if i is not None:
                cur, lim = [i], late[i]
    if not groups:
        return
    def tiles_of(d):                          # workgroups of one problem per K split
        if lib.dyk_conv_wgrad_variant(ctypes.byref(d)) == 2:
            return -(-d.Cout // 64) * -(-d.Cin // 32)
        cap64 = ((d.tune >> 8) & 0xf) == 1
        bm, bn = (128 if d.Cout > 64 and not cap64 else 64), (128 if d.Cin > 64 and not cap64 else 64)
        return -(-d.Cout // bm) * -(-d.Cin // bn)

    removed = []
    for g in groups:
        members = [cmds[i][1] for i in g]
        lead = members[-1]                    # its command stays where it is: every member's inputs exist by then
        s1 = lib.dyk_conv_wgrad_splits(ctypes.byref(lead))
        if s1 < 1:
            continue
        # K splits per member: as many workgroups over the WHOLE group as ONE member's tuned launch had, at least one per CU.
        # (A model of rounds of 256 workgroups + plane traffic per split, and groups cut to whole rounds -- 5 members of 64 tiles
        # as 4 + 1 -- measured no better in the step and 0.15 ms worse at batch 1: r6_ab_tree_group_shape.log)
        tiles = tiles_of(lead)
        sg = max(1, min(s1, -(-max(tiles * s1, 256) // (tiles * len(g)))))
        # (shorter-lived workgroups -- K splits for 2 / 4 / 8 rounds of 256 -- do not help the chain beside them: 26.74 / 26.85 /
        # 27.37 ms against 26.77, r6_ab_filler_rounds.log)
        for m in members:
            m.splits = sg
        plan._wg_groups[ctypes.addressof(lead)] = members
        plan._wg_group_info.append(dict(lead=lead, lmax=max(layer_at[i] for i in g)))
        t = [plan._cmd_us.get(ctypes.addressof(m)) for m in members]
        if all(x is not None for x in t):
            plan._cmd_us[ctypes.addressof(lead)] = 0.75 * sum(t)
        removed += g[:-1]
    removed.sort()
    gone = set(removed)
    import bisect
    plan.bwd = [c for q, c in enumerate(cmds) if q not in gone]
    plan.bwd_marks = [(cnt - bisect.bisect_left(removed, cnt), layer) for cnt, layer in plan.bwd_marks]
    plan.wgrad_grouped = (len(groups), len(removed) + len(groups))


def _finish_wgrad_groups(plan, device):
    """device tables of the grouped weight-gradient launches (after the planes are handed out), and the data-parallel cut
    positions they forbid: a bucket must not close at a position where a member of a LATER launch still owes the gradient of a
    layer the bucket covers"""
    groups = getattr(plan, "_wg_groups", None)
    if not groups:
        return
    for lead_addr, members in groups.items():
        arr = (L.DykWgradGroupEntry * len(members))()
        for i, m in enumerate(members):
            arr[i].x, arr[i].dy, arr[i].dw, arr[i].part = m.x, m.dy, m.dw, m.part
        tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        lead = members[-1]
        assert ctypes.addressof(lead) == lead_addr
        lead.group, lead.group_n = tab.data_ptr(), len(members)
        plan._keep.append(tab)
    pos = {ctypes.addressof(d): q for q, (op, d) in enumerate(plan.bwd) if op == L.OP_WGRAD}
    marks = plan.bwd_marks
    bad = set()
    for info in plan._wg_group_info:
        P, lmax = pos[ctypes.addressof(info["lead"])], info["lmax"]
        for k in range(1, len(marks)):
            c_end, layer_done = marks[k][0], marks[k - 1][1]
            if c_end <= P and layer_done <= lmax:
                bad.add(c_end)
    if plan.bwd_cut_ok is None:
        plan.bwd_cut_ok = {c for c, _ in marks}
    plan.bwd_cut_ok = set(plan.bwd_cut_ok) - bad


def _setup_wgrad_partials(plan, store, device, force_layers=()):
    """Atomic-free weight gradients: every K split of a weight-gradient launch gets its own plane of a partial buffer
    (dyk_conv_wgrad_splits planes per layer), and a table-driven dyk_grad_reduce folds the planes into the flat gradient
    buffer every ~1/16 of the parameters (at section boundaries, so that the data-parallel exchange can cut there).
    Rewrites plan.bwd / plan.bwd_marks; single-split launches keep their (uncontended, order-free) atomics."""
    lib = L.load()
    G0 = store.G.data_ptr()
    items, total = {}, 0
    for q, (op, d) in enumerate(plan.bwd):
        if op == L.OP_DW_WGRAD:                  # depthwise: one plane [k*k][C] per workgroup row
            splits = lib.dyk_dwconv_wgrad_rows(ctypes.byref(d))
            plane = d.k * d.k * d.C
            if splits < 2 or plane % 4:
                continue
            items[q] = [dict(d=d, splits=splits, plane=plane, part_off=total, g_off=(d.dw - G0) // 4, dw=True)]
            total += splits * plane
            continue
        if op != L.OP_WGRAD:
            continue
        for m in plan._wg_groups.get(ctypes.addressof(d), [d]):          # (a grouped launch: every member has its own planes)
            splits = lib.dyk_conv_wgrad_splits(ctypes.byref(m))
            if splits < 2:
                continue
            plane = m.ntaps * m.Cout * (m.lddw if m.lddw > 0 else m.Cin)   # stems: [Cout][k*k*3] rows of lddw floats
            if plane % 4:
                continue
            items.setdefault(q, []).append(dict(d=m, splits=splits, plane=plane, part_off=total, g_off=(m.dw - G0) // 4))
            total += splits * plane
    if not items:
        return
    plan.part = torch.empty(total, dtype=torch.float32, device=device)
    plan.part_bytes = total * 4
    base = plan.part.data_ptr()
    plan._rw_extra, plan._part_extent = {}, {}
    for it in (it_ for lst in items.values() for it_ in lst):
        d = it["d"]
        if it.get("dw"):
            d.part = base + 4 * it["part_off"]
            plan._part_extent[ctypes.addressof(d)] = 4 * it["splits"] * it["plane"]
        else:
            d.part, d.part_stride, d.splits = base + 4 * it["part_off"], it["plane"], it["splits"]
    target = sum(it["plane"] for lst in items.values() for it in lst) // 16

    def reduce_cmd(entries):
        arr = (L.DykGradReduceEntry * len(entries))()
        chunks = 0
        for i, it in enumerate(entries):
            arr[i].g_off, arr[i].part_off, arr[i].plane = it["g_off"], it["part_off"], it["plane"]
            arr[i].n, arr[i].splits, arr[i].chunk_begin = it["plane"], it["splits"], chunks
            chunks += (it["plane"] + 1023) // 1024
        tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        m = L.DykMiscDesc()
        m.p[0], m.p[1], m.p[2] = G0, base, tab.data_ptr()
        m.i[0], m.i[1] = len(entries), chunks
        plan._keep += [tab, m]
        # read / write footprint for the dependency scheduler: the planes of its entries, their slices of G
        plan._rw_extra[ctypes.addressof(m)] = (
            [(base + 4 * it["part_off"], 4 * it["splits"] * it["plane"]) for it in entries],
            [(G0 + 4 * it["g_off"], 4 * it["plane"]) for it in entries])
        return (L.OP_GRAD_REDUCE, m)

    starts = {}
    for cnt, layer in plan.bwd_marks:
        starts.setdefault(cnt, []).append(layer)
    new, marks, cut_ok, pend, acc = [], [], set(), [], 0
    for idx in range(len(plan.bwd) + 1):
        if idx in starts:
            if pend and (acc >= target or idx == len(plan.bwd) or any(l in force_layers for l in starts[idx])):
                new.append(reduce_cmd(pend))
                pend, acc = [], 0
            if not pend:
                cut_ok.add(len(new))
            for layer in starts[idx]:
                marks.append((len(new), layer))
        if idx < len(plan.bwd):
            new.append(plan.bwd[idx])
            if idx in items:
                pend.extend(items[idx])
                acc += sum(it["plane"] for it in items[idx])
    assert not pend
    plan.bwd, plan.bwd_marks, plan.bwd_cut_ok = new, marks, cut_ok


_TUNE_MS = {}        # problem key -> measured duration (ms) of the chosen configuration
_TUNE_CACHE = {}     # process-wide: the same problem always runs the same tile configuration (bit-reproducible
                     # results across plans / model instances within a process) -- and, through the file below, across processes
_TUNE_FILE = {"path": None, "loaded": False}
# environment switches (analysis knobs) that change a problem's candidate set, its ranking or the kernels behind a choice
_TUNE_KNOBS = ("DYK_CONV_", "DYK_WGRAD_", "DYK_TUNE_", "DYK_BNFWD", "DYK_BNBWD", "DYK_EPI_", "DYK_TIGHT_ROWS", "DYK_RB_",
               "DYK_SPLITK", "DYK_PW_", "DYK_DW_")


def _tune_file_path(device=None):
    """Where the autotuner's choices persist (tile choices fix the summation order of the bf16 path, so two
    processes that tune separately may differ in the last bits).  One JSON file per (library source digest, device name) under
    $DYK_TUNE_CACHE_DIR | $XDG_CACHE_HOME/dyk | ~/.cache/dyk; DYK_TUNE_CACHE=0 turns persistence off, DYK_TUNE_CACHE=<file>
    names the file.  A changed kernel source changes the digest: stale choices are never applied to new kernels."""
    knob = os.environ.get("DYK_TUNE_CACHE", "1")
    if knob == "0":
        return None
    if knob not in ("1", ""):
        return knob
    base = os.environ.get("DYK_TUNE_CACHE_DIR") or os.path.join(
        os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "dyk")
    try:
        sha = L.load().dyk_build_sha().decode()
        name = torch.cuda.get_device_name(device if device is not None else torch.cuda.current_device())
    except Exception:
        return None
    name = "".join(ch if ch.isalnum() else "_" for ch in name)
    # environment switches that change what a key's candidates are / how they are ranked belong to the file's identity
    knobs = "".join("%s=%s;" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith(_TUNE_KNOBS) and k not in (
        "DYK_TUNE_CACHE", "DYK_TUNE_CACHE_DIR", "DYK_TUNE_VERBOSE"))
    import hashlib
    tag = hashlib.sha1(knobs.encode()).hexdigest()[:8] if knobs else "default"
    return os.path.join(base, "tune_%s_%s_%s.json" % (sha, name, tag))


def _tune_cache_load():
    if _TUNE_FILE["loaded"]:
        return
    _TUNE_FILE["loaded"] = True
    path = _TUNE_FILE["path"] = _tune_file_path()
    if not path or not os.path.exists(path):
        return
    import json
    try:
        with open(path) as f:
            data = json.load(f)
    except (OSError, ValueError):
        return
    import ast
    for k, (best, ms) in data.items():
        key = ast.literal_eval(k)
        if key not in _TUNE_CACHE:
            _TUNE_CACHE[key] = tuple(best) if isinstance(best, list) else best
            if ms is not None:
                _TUNE_MS[key] = ms


def _tune_cache_save():
    path = _TUNE_FILE["path"]
    if not path:
        return
    import json
    data = {}
    if os.path.exists(path):                   # merge: another process may have tuned other problems meanwhile (its choices win
        try:                                   # for keys both hold -- the file is the authority once written)
            with open(path) as f:
                data = json.load(f)
        except (OSError, ValueError):
            data = {}
    for key, best in _TUNE_CACHE.items():
        data.setdefault(repr(key), [list(best) if isinstance(best, tuple) else best, _TUNE_MS.get(key)])
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "w") as f:
            json.dump(data, f)
        os.replace(tmp, path)
    except OSError:
        pass


def pw_eligible(d):
    """mirror of conv_pw_eligible (csrc/conv_pw_kernel.h): 1x1 / stride 1 / dense grid, bf16, K <= 256, plain | statistics |
    BatchNorm-backward (| chain) epilogues, weights of one 128-row channel tile <= 64 KB"""
    if d.dtype != L.DYK_BF16 or d.ntaps != 1 or d.tdy[0] or d.tdx[0] or d.ncls > 1 or d.twin or d.splitk > 1:
        return False
    if (d.isy, d.isx, d.osy, d.osx, d.ooy, d.oox) != (1, 1, 1, 1, 0, 0):
        return False
    if not (d.Hg == d.Hi == d.Ho and d.Wg == d.Wi == d.Wo):
        return False
    if d.flags not in (0, L.EPI_STATS, L.EPI_BNBWD, L.EPI_BNBWD | L.EPI_ADDEND):
        return False
    if not (d.flags & L.EPI_BNBWD) and d.act != L.ACT_CODES["linear"]:      # (EPI 0 / 1 never apply an activation)
        return False
    return d.Cin % 32 == 0 and d.Cin <= 256 and d.Cout % 8 == 0 and d.ldx % 8 == 0 and d.ldy % 8 == 0


def _conv_candidates(d):
    """tile configurations of dyk_conv_igemm for one problem: K-step bytes | ring stages << 8 | pixel tile << 12
    (0 = 128, 1 = 80, 2 = 160 pixels, 3 / 4 = halo kernel 4x20 / 8x20; bf16 only) | channel tile << 24 (0 = by Cout, 2 = 64, 1 = 32)
    | 1 << 28 = K-grouped workgroups"""
    es = 2 if d.dtype == L.DYK_BF16 else 4
    bkbs = [64] + ([128] if (d.Cin * es) % 128 == 0 else [])
    tiles = [0, 1, 2] if d.dtype == L.DYK_BF16 else [0]
    if (d.dtype == L.DYK_BF16 and d.ntaps == 9 and d.isy == 1 and d.osy == 1 and d.Hg == d.Hi and d.Wg == d.Wi
            and d.Wi % 20 == 0 and d.Hi % 4 == 0):
        tiles += [3] + ([4] if d.Hi % 8 == 0 else [])       # 3x3 halo kernel, 4x20 / 8x20 pixel patches
    bms = [0] + ([2] if d.Cout > 64 else ([1] if d.Cout > 32 else []))
    out = []
    for bkb in bkbs:
        for pipe in (2, 3, 4, 6):
            for t in tiles:
                for bm in bms:
                    if t in (1, 3, 4) and bm == 1:
                        continue                      # 80-pixel / halo tiles need >= 64 channel rows
                    if t in (3, 4) and pipe != 2:
                        continue                      # the halo kernel has a fixed pipeline
                    out.append(bkb | (pipe << 8) | (t << 12) | (bm << 24))
    if (d.dtype == L.DYK_BF16 and d.ntaps == 9 and d.isy == 1 and d.osy == 1 and d.Hg == d.Hi and d.Wg == d.Wi and d.ncls <= 1
            and d.Cin % 32 == 0 and d.Wi % 20 == 0 and not (d.flags & (L.EPI_OUT_F32 | L.EPI_BNFWD))
            and os.environ.get("DYK_CONV_LT", "1") != "0"):
        # large-tile 3x3 kernels (csrc/conv_lt_kernel.h): 8 waves, 128 x 320 / 256 x 160 / 128 x 160 with two K-groups, halo patch of
        # the fewest rows (the patch width moved no time in tools/lt_probe.py).  The front end falls back to the generic
        # 160-pixel tile where a shape does not fit the map.
        for shape in (1, 2, 3):
            if shape == 2 and d.Cout < 256:
                continue
            out.append((5 << 12) | (shape << 8))
    if (d.dtype == L.DYK_BF16 and d.ntaps == 9 and d.Cout == 32 and d.Cin == 64 and d.flags == L.EPI_BNBWD and d.isy == 1
            and d.Hg % 8 == 0 and d.Wg % 16 == 0 and os.environ.get("DYK_CONV_SC", "1") != "0"):
        # resident-weight 3x3 data gradient into 32-channel tensors (csrc/conv_sc.hip): one patch per tile for every tap and
        # parity class, epilogue from the accumulators; the front end falls back to the generic tile where it does not apply
        out.append(6 << 12)
    if pw_eligible(d) and os.environ.get("DYK_CONV_PW", "1") != "0":
        # persistent resident-weight pointwise kernels (csrc/conv_pw_kernel.h): ring stages 2 | 3 | 4, 64- or (channel tiles <= 64
        # rows) 128-pixel tiles; the front end falls back to the generic 128-pixel tile where a shape does not fit
        for nxs in (2, 3, 4):
            out.append((7 << 12) | (nxs << 8))
            if d.Cout <= 64:
                out.append((7 << 12) | (nxs << 8) | (1 << 24))
            else:
                out.append((7 << 12) | (nxs << 8) | (1 << 25))
    if d.dtype == L.DYK_BF16 and (d.Cin * es) % 128 == 0 and d.Cin >= 128 and os.environ.get("DYK_CONV_KG", "1") != "0":
        # K-grouped workgroups (two 4-wave groups over the two halves of Cin): for tiles that leave a CU one workgroup
        for t in (1, 2):
            for bm in bms:
                if bm != 1:
                    out.append(128 | (2 << 8) | (t << 12) | (bm << 24) | (1 << 28))
    return out


def _conv_split_candidates(d):
    """(tile configuration, S) pairs of dyk_conv_igemm with split-K across workgroups (DykConvDesc.splitk, round 5): for the
    problems whose output has too few tiles to fill 256 CUs with tiles large enough to feed the matrix cores -- the 32x40 /
    16x20 stages (reference models.py:34-62 at strides 16 / 32).  The S slices of a tile fold through private fp32 slabs in
    slice order (bit-reproducible).  Generic 128 / 64-row tiles of 160 / 128 / 80 pixels (2- and 3-stage rings, K-grouped form)
    and, for 3x3 stride-1 problems, the 8-wave large-tile kernels."""
    if os.environ.get("DYK_CONV_SPLITK", "1") == "0":
        return []
    if d.twin or d.ncls > 1 or d.flags & L.EPI_BNFWD:
        return []
    es = 2 if d.dtype == L.DYK_BF16 else 4
    npix = d.B * d.Hg * d.Wg
    max_s = 8                # (12 / 16 slices measured equal to 8 at batch 1, round 5)
    out = []

    def splits(tiles, kchunks):
        # S so that tiles * S is about one to two workgroups per CU, every slice keeping >= 2 K chunks
        return [s for s in (2, 3, 4, 6, 8) if s <= max_s and tiles * s <= 640 and tiles * (s - 1) < 512 and kchunks // s >= 2]
    if (d.Cin * es) % 64:
        return []
    bkbs = [128] if (d.Cin * es) % 128 == 0 else [64]
    tiles_px = [(2, 160), (0, 128), (1, 80)] if d.dtype == L.DYK_BF16 else [(0, 128)]
    bms = [(0, 128 if d.Cout > 64 else (64 if d.Cout > 32 else 32))] + ([(2, 64)] if d.Cout > 64 else [])
    for bkb in bkbs:
        kch = (d.Cin * es) // bkb
        for t, bn in tiles_px:
            for bmc, bm in bms:
                if t == 1 and bm < 64:
                    continue
                tiles = -(-npix // bn) * -(-d.Cout // bm)
                for s in splits(tiles, kch):
                    for pipe in (2, 3):
                        out.append((bkb | (pipe << 8) | (t << 12) | (bmc << 24), s))
                    if d.dtype == L.DYK_BF16 and bkb == 128 and t in (1, 2) and bm >= 64 and kch // s >= 4 and os.environ.get("DYK_CONV_KG", "1") != "0":
                        out.append((128 | (2 << 8) | (t << 12) | (bmc << 24) | (1 << 28), s))
    if (d.dtype == L.DYK_BF16 and d.ntaps == 9 and d.isy == 1 and d.osy == 1 and d.Hg == d.Hi and d.Wg == d.Wi
            and d.Cin % 32 == 0 and d.Wi % 20 == 0 and not (d.flags & L.EPI_OUT_F32) and os.environ.get("DYK_CONV_LT", "1") != "0"):
        for shape, (bm, bn, kg) in ((1, (128, 320, 1)), (2, (256, 160, 1)), (3, (128, 160, 2))):
            if shape == 2 and d.Cout < 256:
                continue
            if npix % bn:
                continue
            tiles = (npix // bn) * -(-d.Cout // bm)
            for s in splits(tiles, (d.Cin // 32) // kg):
                out.append(((5 << 12) | (shape << 8), s))
    return out


WGRAD_EXCLUSIVE = 1 << 20      # the plan's weight gradients have one writer each (store.g_ptr per weight, commands ordered by
                                # their write sets): a single-split row-block launch may read-add-write instead of atomics
_WGRAD_CANDIDATES = [2, 3, 2 | (2 << 8), 2 | (1 << 24), 3 | (1 << 24), 2 | (2 << 8) | (1 << 24), 2 | (1 << 28)]
if os.environ.get("DYK_WGRAD_RB", "1") != "0":
    _WGRAD_CANDIDATES.append(2 | (1 << 8) | (2 << 28) | WGRAD_EXCLUSIVE)
if os.environ.get("DYK_WGRAD_PS", "1") != "0":
    # pixel-streaming 1x1 kernel (round 6, csrc/conv_wgrad_ps.hip): ring stages in the low byte, 64 x 64 tile cap << 8,
    # 64-pixel stages for the 64 x 64 tile << 12; ignored (falls back to the per-tap kernel = candidate 2) where it does not apply
    _WGRAD_PS = 3 << 28
    # Rings of at most 96 KB: a weight gradient is filler work beside the critical chain, and a workgroup that holds 128 KB of
    # LDS keeps every convolution workgroup (45-110 KB) off its CU for its whole life.  Alone, the 4- / 6-stage rings are the
    # faster ones on the deep layers (tools/wgps_probe.py); in the step every launch forced to 2 / 3 / 4 stages measured
    # 26.35 / 26.42 / 26.46 ms against 26.53 with the tuner's free choice (r6_ab_ps_ring_in_step.log)
    _WGRAD_CANDIDATES += [_WGRAD_PS | 2 | WGRAD_EXCLUSIVE, _WGRAD_PS | 3 | WGRAD_EXCLUSIVE,
                          _WGRAD_PS | 4 | (1 << 8) | WGRAD_EXCLUSIVE, _WGRAD_PS | 4 | (1 << 8) | (1 << 12) | WGRAD_EXCLUSIVE]
# LDS ring stages (2 | 3; 4 exists in the kernel, measured never the fastest: DESIGN 9.4) | K-groups per workgroup << 8 | tile cap << 24 (1 = 64 x 64) | 1 << 28 = multi-tap 3x3 kernel
# | 2 << 28 = row-block 3x3 kernel (round 4: 128-pixel block steps, 64 x 32 x 9-tap tiles)


def _time_launch(fn, desc, stream, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L.check(fn(ctypes.byref(desc), stream), "autotune launch")          # warm-up (also loads the code object)
    # (candidates are timed ALONE: timing them beside a side stream of 512 MB copies and / or 4096^3 bf16 products -- "tune under
    # load", rounds 3-4 -- made the step 1.1-2.1 ms slower, profiles/r04_ab_tune_under_load.log; removed in round 6)
    e0.record()
    for _ in range(reps):
        fn(ctypes.byref(desc), stream)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def _refine(cands, times, trial, within=1.10, keep=4, rounds=2, reps=6):
    """Second look at the front runners of a tuning pass.  The first pass times every candidate with three back-to-back
    launches; candidates within a few per cent of each other then win or lose on timer noise, and the step time of one
    build moved by 0.4 ms from run to run on one box (round 3).  The `keep` fastest within `within` of the best are timed
    again, interleaved, `rounds` x `reps` launches each; their times are replaced by the mean of the second look."""
    if len(cands) < 2:
        return times
    best = min(times)
    short = sorted((t, i) for i, t in enumerate(times) if t <= within * best)[:keep]
    if len(short) < 2:
        return times
    acc = {i: 0.0 for _, i in short}
    for _ in range(rounds):
        for _, i in short:
            acc[i] += trial(cands[i], reps)
    out = list(times)
    worst = max(acc.values()) / rounds
    for j, t in enumerate(out):                    # everything outside the short list stays behind it
        if j not in acc:
            out[j] = max(t, worst * 1.0001)
    for i, a in acc.items():
        out[i] = a / rounds
    return out


def autotune(plan, cache=None):
    """Measure, don't guess: for every distinct convolution / weight-gradient problem of the plan, time
    the tile configurations the kernels offer (K-step bytes x LDS ring depth) on the real buffers and
    keep the fastest.  One-off cost at plan compilation (~0.2 s for the 46 problems of the target cfg)."""
    lib = L.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    cache = cache if cache is not None else {}
    if cache is _TUNE_CACHE:
        _tune_cache_load()
    n_known = len(cache)
    store = getattr(plan, "store", None)
    if store is not None and store.wt_ready is not None:
        # data-gradient trials read the transposed packs: a rebuild pending on the optimizer's side stream comes first
        torch.cuda.current_stream().wait_event(store.wt_ready)
    groups = {}
    scratch = [None]                      # partial planes of the weight-gradient trials
    sk_scratch = [None, None]             # split-K slabs / tile counters of the convolution trials
    for (op, d) in plan.fwd + plan.bwd:
        if op == L.OP_CONV:
            key = ("c", d.dtype, d.B, d.Cin, d.Cout, d.Hg, d.Wg, d.ntaps, d.isy, d.osy,
                   d.flags & (L.EPI_STATS | L.EPI_OUT_F32 | L.EPI_BNBWD | L.EPI_BNFWD | L.EPI_ADDEND), d.ncls)
        elif op == L.OP_WGRAD:
            key = ("w", d.dtype, d.B, d.Cin, d.Cout, d.Ho, d.Wo, d.ntaps, d.isy)
        else:
            continue
        groups.setdefault(key, []).append(d)
    for key, descs in groups.items():
        best = cache.get(key)
        if best is None:
            d = descs[0]
            bn_saved = None
            try:
                if key[0] == "c":
                    cands = _conv_candidates(d)
                    fn = lib.dyk_conv_igemm
                    if d.flags & L.EPI_BNFWD:
                        # The trial launches run the REAL descriptor: every one of them would EMA-update the layer's running statistics
                        # (from replica sums that keep accumulating across trials: k x the mean, variance clamped to 0) and rewrite its
                        # saved mean / rstd.  Trials run without those outputs and on freshly zeroed replicas.
                        bn_saved = (d.bn_running_mean, d.bn_running_var, d.bn_save_mean, d.bn_save_rstd)
                        d.bn_running_mean = d.bn_running_var = d.bn_save_mean = d.bn_save_rstd = None
                        # one-launch conv + BatchNorm: generic tiles whose launch fits the residency contract (the front end refuses the others)
                        keep = []
                        for c in cands:
                            if (c >> 28) & 7 or ((c >> 12) & 0xf) in (3, 4):
                                continue
                            d.tune = c
                            if lib.dyk_conv_grid(ctypes.byref(d)) <= BNFWD_MAX_GRID and fn(ctypes.byref(d), stream) == 0:
                                keep.append(c)
                        cands = keep
                        assert cands, "no tile configuration fits the one-launch BatchNorm contract: %s" % (key,)
                        plan.arenas["stats"].tensor.zero_()          # (the filter launches above left sums in the replicas)
                else:
                    cands, fn = _WGRAD_CANDIDATES, lib.dyk_conv_wgrad
                planes_on = key[0] == "w" and plan.training and os.environ.get("DYK_WGRAD_PARTIALS", "1") != "0"
                if planes_on:
                    # Weight gradients are timed the way the step runs them: every K split stores its own partial plane
                    # (_setup_wgrad_partials below; the atomic form penalises exactly the many-split shapes the plane form is
                    # good at) plus the cost of folding that many planes (dyk_grad_reduce, ~2.7 TB/s).  The NUMBER of K splits
                    # is a tuning dimension too: the kernel's own count fills ~3 workgroups per CU, which on the deep layers
                    # (16x20 maps: 5 120 pixels) writes and re-reads several times the operand bytes as planes (512->512 3x3:
                    # 6 planes of 9.4 MB against 10.5 MB of operands) -- half / a quarter of the splits trade idle CUs for
                    # that traffic; one split means no plane at all (single writer, plain accumulation).
                    plane = d.ntaps * d.Cout * (d.lddw if d.lddw > 0 else d.Cin)
                    dev = torch.device("cuda", torch.cuda.current_device())
                    saved_dw = d.dw

                    def room(n):
                        if scratch[0] is None or scratch[0].numel() < n:
                            scratch[0] = None
                            scratch[0] = torch.empty(n, dtype=torch.float32, device=dev)
                        return scratch[0].data_ptr()

                    # Weight of the fold term.  2, not 1: the trial times the weight-gradient launch ALONE, in the step it shares the
                    # chip with three other streams and every plane byte is paid for at the shared rate -- in-call A/B of the whole step
                    # (round 3, C3): weight 0.5 / 1 / 1.5 / 2 / 2.5 / 3 / 4 -> 32.8 / 32.6 / 32.2 / 31.8 / 32.1 / 32.0 / 32.6 ms
                    fold_w = float(os.environ.get("DYK_WGRAD_FOLD_W", "2"))
                    def trial(c, o, reps=3):
                        """time tile configuration c with o K splits (0 = the kernel's own count); None if c does not apply"""
                        d.tune, d.part, d.part_stride, d.splits = c, None, 0, o
                        n = lib.dyk_conv_wgrad_splits(ctypes.byref(d))
                        if n < 1:
                            return None
                        if n >= 2 and plane % 4 == 0:
                            d.part, d.part_stride, d.splits = room(n * plane), plane, n
                            t = _time_launch(fn, d, stream, reps) + fold_w * (n + 1) * plane * 4 / 2.7e9
                        else:
                            d.dw = room(plane)                         # (trial sums must not land in the gradient buffer)
                            t = _time_launch(fn, d, stream, reps)
                            d.dw = saved_dw
                        return t

                    combos, times = [], []
                    for c in cands:
                        d.tune, d.part, d.part_stride, d.splits = c, None, 0, 0
                        if (c >> 28) & 7 and lib.dyk_conv_wgrad_variant(ctypes.byref(d)) != (c >> 28) & 7:
                            continue           # a kernel variant this problem is not eligible for (it would time the fallback again)
                        auto = lib.dyk_conv_wgrad_splits(ctypes.byref(d))
                        if auto < 1:
                            continue
                        opts = {auto}
                        if plane % 4 == 0:
                            opts |= {max(1, auto // 2), max(1, auto // 4), max(1, auto // 8)}
                        for o in sorted(opts, reverse=True):
                            t = trial(c, 0 if o == auto else o)
                            if t is not None:
                                combos.append((c, 0 if o == auto else o))
                                times.append(t)
                    if not combos:                     # no candidate applies to this problem: the kernel's defaults
                        combos, times = [(0, 0)], [float("inf")]
                    else:
                        times = _refine(combos, times, lambda cc, reps: trial(cc[0], cc[1], reps))
                        if os.environ.get("DYK_TUNE_VERBOSE"):
                            print("tune", key, " ".join("%#x/%d:%.1f" % (c[0], c[1], 1e3 * t) for t, c in sorted(zip(times, combos))[:10]), flush=True)
                    d.part, d.part_stride, d.splits, d.dw = None, 0, 0, saved_dw
                    best = combos[times.index(min(times))]
                elif key[0] == "w":
                    # weight gradients without partial planes (DYK_WGRAD_PARTIALS=0: fp32 atomics into the gradient buffer): the
                    # candidates as they are, the kernel's own split count, sums kept out of the gradient buffer
                    saved_dw = d.dw
                    room_t = torch.zeros(d.ntaps * d.Cout * (d.lddw if d.lddw > 0 else d.Cin), dtype=torch.float32,
                                         device=torch.device("cuda", torch.cuda.current_device()))
                    d.dw = room_t.data_ptr()

                    def trial_w(c, reps=3):
                        d.tune = c
                        if (c >> 28) & 7 and lib.dyk_conv_wgrad_variant(ctypes.byref(d)) != (c >> 28) & 7:
                            return float("inf")
                        return _time_launch(fn, d, stream, reps)
                    try:
                        times = _refine(cands, [trial_w(c) for c in cands], trial_w)
                    finally:
                        d.dw = saved_dw
                    best = cands[times.index(min(times))]
                else:
                    sk_saved = (d.sk_ws, d.sk_cnt, d.sk_ws_bytes, d.sk_cnt_n, d.splitk)

                    def trial_c(c, reps=3):
                        if isinstance(c, tuple):
                            # split-K candidate: private scratch for the trial (slabs + zeroed tile counters)
                            d.tune, d.splitk = c
                            nt = ctypes.c_int32(0)
                            need = int(lib.dyk_conv_splitk_ws_bytes(ctypes.byref(d), ctypes.byref(nt)))
                            if need <= 0:
                                return float("inf")
                            dev = torch.device("cuda", torch.cuda.current_device())
                            if sk_scratch[0] is None or sk_scratch[0].numel() < need:
                                sk_scratch[0] = None
                                sk_scratch[0] = torch.empty(need, dtype=torch.uint8, device=dev)
                            if sk_scratch[1] is None or sk_scratch[1].numel() < nt.value:
                                sk_scratch[1] = torch.zeros(max(nt.value, 4096), dtype=torch.int32, device=dev)
                            else:
                                sk_scratch[1].zero_()          # (a trial of another shape must not have left a ticket behind)
                            d.sk_ws, d.sk_ws_bytes = sk_scratch[0].data_ptr(), need
                            d.sk_cnt, d.sk_cnt_n = sk_scratch[1].data_ptr(), nt.value
                            if fn(ctypes.byref(d), stream) != 0:       # (a shape the chosen kernel cannot split: not a candidate)
                                return float("inf")
                        else:
                            d.tune, d.splitk = c, 0
                        if bn_saved is not None:
                            plan.arenas["stats"].tensor.zero_()
                        return _time_launch(fn, d, stream, reps)
                    cands = list(cands) + _conv_split_candidates(d)
                    try:
                        times = _refine(cands, [trial_c(c) for c in cands], trial_c)
                    finally:
                        d.sk_ws, d.sk_cnt, d.sk_ws_bytes, d.sk_cnt_n, d.splitk = sk_saved
                    best = cands[times.index(min(times))]
                    if os.environ.get("DYK_TUNE_VERBOSE"):   # analysis: every candidate's time, fastest first
                        print("tune", key, " ".join(("%#x/%d:%.1f" % (c[0], c[1], 1e3 * t)) if isinstance(c, tuple) else ("%#x:%.1f" % (c, 1e3 * t))
                                                    for t, c in sorted(zip(times, cands), key=lambda tc: tc[0])[:12]), flush=True)
            finally:
                # (also when a trial raises: a live descriptor without its running-statistics outputs would silently stop
                # updating the BatchNorm running statistics -- ADVICE r4)
                if bn_saved is not None:
                    d.bn_running_mean, d.bn_running_var, d.bn_save_mean, d.bn_save_rstd = bn_saved
            cache[key] = best
            if min(times) != float("inf"):
                _TUNE_MS[key] = min(times)
        for d in descs:
            if isinstance(best, tuple) and key[0] == "c":
                d.tune, d.splitk = best             # (tile configuration, split-K slices; scratch: _setup_splitk)
            elif isinstance(best, tuple):
                d.tune, d.splits = best             # (tile configuration, K splits: 0 = the kernel's own count)
            else:
                d.tune = best
            if key in _TUNE_MS:
                plan._cmd_us[ctypes.addressof(d)] = 1e3 * _TUNE_MS[key]      # measured duration: cost of the scheduler
    plan.tuned = dict(cache)
    if cache is _TUNE_CACHE and len(cache) != n_known:
        _tune_cache_save()
    # the trial launches polluted the statistics accumulators / scratch: reset
    plan.arenas["ws"].tensor.zero_()
    plan.arenas["grad"].tensor.zero_()
    plan.arenas["stats"].tensor.zero_()
    torch.cuda.synchronize()
