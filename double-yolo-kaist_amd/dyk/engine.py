"""Runtime of a compiled YOLO model: owns the parameter store and the plans, and connects the
native forward / backward command lists to torch autograd so that `loss.backward()` and
`torch.optim` in an unchanged train loop (reference train_utils/kaist_train_eval_utils.py:75-108)
keep working.
"""
import ctypes
import os

import torch

from . import lib as L
from .params import ParamStore
from .plan import compile_plan


def _compute_dtype(model):
    forced = getattr(model, "dyk_dtype", None) or os.environ.get("DYK_DTYPE")
    if forced:
        return {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "f32": torch.float32,
                "float32": torch.float32}[str(forced).replace("torch.", "")]
    # the reference trains under torch.cuda.amp.autocast (kaist_train_eval_utils.py:74); any autocast
    # region selects the reduced-precision path, which on MI355X is bf16 MFMA with fp32 accumulation
    if torch.is_autocast_enabled():
        try:
            ac = torch.get_autocast_dtype("cuda")
        except Exception:                      # older torch
            ac = torch.get_autocast_gpu_dtype()
        if ac != torch.bfloat16 and not _compute_dtype.warned:
            _compute_dtype.warned = True
            import warnings
            warnings.warn("dyk: the autocast region asks for %s (the reference trains under fp16 autocast + GradScaler, "
                          "kaist_train_eval_utils.py:74); the MI355X path computes reduced precision in bfloat16 MFMA with "
                          "fp32 accumulation instead -- same storage width, wider exponent, so loss scaling is a no-op.  "
                          "Set model.dyk_dtype = 'fp32' for bit-level comparisons against an fp32 reference run." % ac)
        return torch.bfloat16
    return torch.float32


_compute_dtype.warned = False


class _NetFunction(torch.autograd.Function):
    """forward = native command list; backward = native command list writing parameter gradients
    directly into the flat gradient buffer (ParamStore.G) that every p.grad is a view of."""

    @staticmethod
    def forward(ctx, engine, plan, anchor, x, y):
        engine._run_forward(plan, x, y)
        ctx.engine, ctx.plan = engine, plan
        # fresh tensor objects every call (autograd attaches history to what a Function returns)
        return tuple(p.detach() for p in plan.p_out)

    @staticmethod
    def backward(ctx, *dps):
        ctx.engine._run_backward(ctx.plan, dps)
        return None, None, None, None, None


class Engine:
    def __init__(self, model):
        self.model = model
        self.store = ParamStore(model)
        self.plans = {}
        self.grad_sync = None           # set by dyk.ddp.GradAllReduce
        self.opt_overlap = None         # set by the fused optimizers (dyk/optim.py): segment arithmetic for the early optimizer start
        self._early = self._early_event = None
        self._anchor = None

    # ------------------------------------------------------------------ helpers
    def _prepare(self, x):
        if not x.is_cuda:
            raise L.DykError("YOLO.forward runs on the MI355X HIP path only: move the model and the inputs to a "
                             "cuda device (got %s). There is no CPU fallback." % x.device)
        L.load()
        dev = x.device
        if not self.store.is_adopted(dev):
            first = next(self.model.parameters())
            if first.device != dev:
                raise L.DykError("model parameters are on %s but the input is on %s" % (first.device, dev))
            self.store.adopt(dev)
            self.plans = {}

    def get_plan(self, B, H, W, dtype, training, device):
        # frozen parameters (reference train.py:77-82) change the backward list: no weight gradients for them and no
        # backward at all below the first trainable section
        key = (B, H, W, dtype, bool(training)) + ((self.store.frozen_key(),) if training and self.store.frozen_key() else ())
        plan = self.plans.pop(key, None)
        if plan is None:
            if H % 32 or W % 32:
                raise ValueError("input height/width must be multiples of 32 (reference train.py:49), got %dx%d" % (H, W))
            plan = compile_plan(self.model, self.store, B, H, W, dtype, training, device)
        self.plans[key] = plan                     # (re)insert as most recently used
        self._evict(device)
        return plan

    def _evict(self, device):
        """Plans own their arenas (the target cfg at batch 16: ~20 GB for a training plan).  Multi-scale training
        (reference kaist_train_eval_utils.py:59-71 draws a new image size every few batches) would otherwise keep one
        per size: least-recently-used plans are dropped once the total passes DYK_PLAN_MEM_GB (default: half of the
        device memory).  A plan still referenced by a pending backward stays alive until that backward is done."""
        budget = float(os.environ.get("DYK_PLAN_MEM_GB", "0")) * 1e9
        if budget <= 0:
            budget = 0.5 * torch.cuda.get_device_properties(device).total_memory if device.type == "cuda" else float("inf")

        def nbytes(p):
            return sum(a.size for a in p.arenas.values()) + (getattr(p, "part_bytes", 0) or 0) + (getattr(p, "sk_bytes", 0) or 0)
        keys = list(self.plans)
        total = sum(nbytes(self.plans[k]) for k in keys)
        while total > budget and len(keys) > 1:
            k = keys.pop(0)
            total -= nbytes(self.plans.pop(k))

    # ------------------------------------------------------------------ execution
    def _check_bnfwd(self, plan):
        """one-launch conv + BatchNorm blocks (DYK_BNFWD=1): look at the error words of the PREVIOUS pass without a host sync (copy
        to pinned memory behind the pass, examined here once it has landed) and fail loudly instead of training on tiles that
        were never normalised"""
        pend = plan.__dict__.get("_bnfwd_pending")
        if pend is not None and pend[1].query():
            if int(pend[0].max()) != 0:
                raise L.DykError("a DYK_EPI_BNFWD launch gave up waiting for its own workgroups (more than two such launches "
                                 "in flight, or not all of them resident): rerun with DYK_BNFWD=0")
            plan.__dict__["_bnfwd_pending"] = None

    def _post_bnfwd(self, plan):
        if plan.__dict__.get("_bnfwd_pending") is None:
            words = plan.bnfwd_error_words()
            host = torch.zeros(words.numel(), dtype=torch.int32).pin_memory()
            host.copy_(words, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            plan.__dict__["_bnfwd_pending"] = (host, ev)

    def _run_forward(self, plan, x, y):
        if plan.has_bnfwd:
            self._check_bnfwd(plan)
        stream = torch.cuda.current_stream().cuda_stream
        self.store.compute_weights(plan.dtype, force=False)
        xs = {"x": x, "y": y if y is not None else x}
        keep, prepared = [], {}
        for desc, which in plan.dyn_in:
            t = xs[which]
            if isinstance(desc, L.DykStemDesc):
                # direct stem kernels read the image batch as it comes: float32 0..1, or the loader's uint8 (divided by
                # 255 inside the kernel exactly as `imgs.float() / 255.0` does)
                if t.dtype not in (torch.uint8, torch.float32) or not t.is_contiguous():
                    t = prepared.setdefault((which, "f"), t.float().contiguous())
                desc.img, desc.in_u8 = t.data_ptr(), 1 if t.dtype == torch.uint8 else 0
                keep.append(t)
                continue
            if t.dtype == torch.uint8:
                # the loader's uint8 batch handed over as is: `.float() / 255.0` in one HIP pass
                from .functional import prepare_images
                t = prepared.get((which, "p"))
                if t is None:
                    t = prepared[(which, "p")] = prepare_images(xs[which])
            elif t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            keep.append(t)
            desc.p[0] = t.data_ptr()
        plan._dyn_keep = keep
        if plan.stem_fuse:
            # the stem's BatchNorm-backward apply inside its weight gradient (csrc/stem.hip): possible for THIS call's images?
            lib = L.load()
            for ap, wd in plan.stem_fuse:
                ok = bool(lib.dyk_stem_wgrad_bn_fusable(ctypes.byref(wd)))
                wd.bn_fused = 1 if ok else 0
                ap.flags = (ap.flags | L.EW_SKIP) if ok else (ap.flags & ~L.EW_SKIP)
        plan.run("fwd", stream)
        if plan.has_bnfwd:
            self._post_bnfwd(plan)
        if plan.training and len(self.store.nbt):
            self.store.NBT += 1

    def _run_backward(self, plan, dps):
        stream = torch.cuda.current_stream().cuda_stream
        if self.store.wt_ready is not None:          # transposed packs rebuilt on the optimizer's side stream (dyk/optim.py)
            torch.cuda.current_stream().wait_event(self.store.wt_ready)
            self.store.wt_ready = None
        self.store.attach_grads()
        self.store.grads_dirty = True
        keep = []
        for desc, hi in plan.dyn_dp:
            g = dps[hi]
            if g is None:
                g = torch.zeros_like(plan.p_out[hi])
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.float().contiguous()
            keep.append(g)
            desc.p[0] = g.data_ptr()
        plan._dyn_keep_b = keep
        self._early = None
        if self.grad_sync is None and self.opt_overlap is not None:
            # single GPU, fused optimizer attached (dyk/optim.py): the backward runs in two segments; an event after the first marks
            # the point where the gradients of the deep layers (95 % of the parameters) are final -- optimizer.step() starts on
            # them on a side stream while the second segment (the early layers: most of the TIME of a backward pass) still runs
            segs = self.opt_overlap.segments(plan, fractions=(0.95,))      # (70-95 % equal, 99 % worse: r04_ab_optimizer_overlap.log)
            if len(segs) == 2 and segs[0][2] % 8 == 0 and segs[0][2] > 0:
                (c0, c1, lo, hi), (d0, d1, _, _) = segs
                if getattr(self, "opt_early", False):
                    # DYK_OPT_OVERLAP=early: the update of G[lo:] may start in the MIDDLE of the pass, so the pass is cut there
                    plan.run("bwd", stream, c0, c1)
                    ev = self._early_event = self._early_event or torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                    plan.run("bwd", stream, d0, d1)
                    self._early = (ev, lo)
                else:
                    # default: the optimizer's side stream waits for everything the caller enqueued anyway (dyk/optim.py), so the
                    # pass stays ONE schedule and only the split point of the update is handed over.  (Round 6: a cut joins all
                    # streams, i.e. the critical chain waits for whatever weight-gradient launch is in flight -- with grouped
                    # launches that cost the MobileNetV3 cfg 0.65 ms per step, r6_ab_opt_overlap_c5.log)
                    plan.run("bwd", stream)
                    self._early = (None, lo)
            else:
                plan.run("bwd", stream)
        elif self.grad_sync is None:
            plan.run("bwd", stream)
        else:
            # data parallel: launch the backward in segments and hand every finished gradient bucket to
            # the all-reduce while the remaining (earlier) layers are still being differentiated
            for (c0, c1, lo, hi) in self.grad_sync.segments(plan):
                plan.run("bwd", stream, c0, c1)
                self.grad_sync.bucket_ready(lo, hi)

    # ------------------------------------------------------------------ public
    def forward(self, x, y=None):
        model = self.model
        self._prepare(x)
        B, _, H, W = x.shape
        dtype = _compute_dtype(model)
        training = model.training
        plan = self.get_plan(B, H, W, dtype, training, x.device)
        di = "second_index" in model.net_info and y is not None
        if "second_index" in model.net_info and not di:
            raise L.DykError("this cfg is dual-stream (second_index set): call model(visible, lwir)")
        if training:
            if torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()):
                # the differentiable input that ties the outputs to autograd: a private leaf, so that freezing any
                # prefix of the layers (module_list[0] included) leaves `loss.backward()` working
                anchor = self._anchor
                if anchor is None or anchor.device != x.device:
                    anchor = self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
                outs = _NetFunction.apply(self, plan, anchor, x, y)
                return list(outs)
            self._run_forward(plan, x, y)
            return list(plan.p_out)
        with torch.no_grad():
            self._run_forward(plan, x, y)
        return plan.io, tuple(plan.p_out)
