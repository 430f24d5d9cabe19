"""Fused optimizers over the flat parameter store (one HIP launch per step for the whole model).

Drop-in for the torch.optim.Adam / SGD(nesterov) instances reference train.py:85-91 builds: same update
rule, same constructor arguments, `step()` / `zero_grad()` / `param_groups[0]['lr']` for LR schedulers,
`state_dict()` / `load_state_dict()` in torch.optim's own per-parameter format (the reference checkpoint stores
`optimizer.state_dict()`, train.py:229), and -- like `pg = [p for p in model.parameters() if p.requires_grad]`
(train.py:84) -- no update of parameters frozen with `requires_grad_(False)` (`--freeze-layers`, train.py:77-82).
"""
import ctypes

import torch

from . import lib as L
from .lib import check, load


class _FusedBase(torch.optim.Optimizer):
    _state_names = ()            # torch.optim state entries that map onto the flat buffers (m, v)

    def __init__(self, model, defaults):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("optimizer got an empty parameter list")
        super().__init__(params, defaults)
        self._t = 0
        self._m = self._v = None
        self._mask, self._mask_key = None, None
        self.grad_scale = 1.0
        self.zero_in_step = True
        self._side = None
        self._buckets = None
        self._fresh = True
        import os
        mode = os.environ.get("DYK_OPT_OVERLAP", "1")
        self.early_start = mode == "early"
        if mode != "0":
            from .ddp import GradAllReduce
            model.engine.opt_overlap = GradAllReduce(model, None, attach=False)
            model.engine.opt_early = self.early_start

    def _store(self):
        st = self.model.engine.store
        if st.P is None:
            raise L.DykError("run one forward pass (or model.engine.store.adopt(device)) before optimizer.step()")
        return st

    def zero_grad(self, set_to_none=False):
        """Clears the flat gradient buffer unless it is known to be clean: a fused step that ran AFTER the last
        backward has already zeroed it.  A step skipped by GradScaler (inf/NaN gradients) or by the caller leaves
        `store.grads_dirty` set, so the stale gradients are dropped here as with torch.optim."""
        st = self.model.engine.store
        if st.G is not None:
            st.attach_grads()
            if st.grads_dirty:
                st.G.zero_()
                st.grads_dirty = False

    def _buffers(self, st):
        if self._m is None or self._m.device != st.P.device:
            self._m = torch.zeros_like(st.P)
            self._v = torch.zeros_like(st.P) if "exp_avg_sq" in self._state_names else None
            self._fresh = True            # (their fill runs on the caller's stream: the side stream of _launch must see it)
        return self._m, self._v

    def _trainable_mask(self, st):
        """None when every parameter is trainable, else a byte per element of the flat buffer (1 = update):
        frozen parameters get neither the update nor weight decay nor moments."""
        key = st.frozen_key()
        if key != self._mask_key:
            self._mask_key = key
            if not key:
                self._mask = None
            else:
                mask = torch.zeros(st.total, dtype=torch.uint8)
                for e in st.entries:
                    if e.param.requires_grad:
                        mask[e.offset:e.offset + e.numel] = 1
                self._mask = mask.to(st.P.device)
                self._fresh = True
        return self._mask

    def _take_buckets(self, works, total):
        """GradAllReduce.all_reduce(optimizer=self): [(work, lo, hi)] of the data-parallel exchange; accepted when the ranges tile
        the flat buffer in whole 8-element groups (else the caller waits as usual)"""
        spans = sorted((lo, hi) for _, lo, hi in works)
        ok = spans and spans[0][0] == 0 and spans[-1][1] == total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        if not ok or any(lo % 8 for lo, _ in spans):
            return False
        self._buckets = list(works)
        return True

    def _launch(self, fn, name, d, bf):
        """one launch over the whole flat buffer -- or, when the last backward left an event behind its first segment
        (engine._early: gradients of G[lo:] final, single GPU), two: [lo, total) on a side stream as soon as that event has
        passed, [0, lo) behind the backward on the caller's stream, which then waits for the side stream.  Same arithmetic per
        element; the 0.66 ms HBM-bound step of the target cfg no longer runs alone after the backward's tail."""
        eng = self.model.engine
        early, eng._early = eng._early, None
        buckets, self._buckets = self._buckets, None
        main = torch.cuda.current_stream()
        if buckets is None and (early is None or self.model.engine.grad_sync is not None):
            check(fn(ctypes.byref(d), ctypes.c_void_p(main.cuda_stream)), name)
            return
        n = d.n
        if self._side is None:
            # library stream 3 of the schedule runtime (dyk_sched_stream), not a fifth stream: the HIP runtime maps all streams
            # of the process onto four hardware queues and a fifth one shares a queue with whichever stream the runtime picks
            # (in-call, round 4: 28.39 / 28.47 ms with a stream of its own, 28.19 / 28.31 on library stream 3, 28.39 / 28.32 on 2,
            # 28.60 / 28.47 on 1 -- stream 1 carries the second backbone)
            h = ctypes.c_void_p()
            check(load().dyk_sched_stream(3, ctypes.byref(h)), "dyk_sched_stream")
            self._side = torch.cuda.ExternalStream(h.value)
        side = self._side

        def sub(off, cnt):
            e = L.DykOptimDesc()
            ctypes.memmove(ctypes.byref(e), ctypes.byref(d), ctypes.sizeof(L.DykOptimDesc))
            e.p, e.g, e.m = d.p + 4 * off, d.g + 4 * off, d.m + 4 * off
            e.v = d.v + 4 * off if d.v else None
            e.mask = d.mask + off if d.mask else None
            e.wc = d.wc + 2 * off if d.wc else None
            e.n = cnt
            return e
        if self._fresh:
            # moment buffers / trainable mask were created (filled on the caller's stream) for this very step: the side stream
            # waits for the caller's stream once -- the first step runs without the overlap
            first = torch.cuda.Event()
            first.record(main)
            side.wait_event(first)
            self._fresh = False
        if buckets is not None:
            # data parallel: every bucket's range on the side stream behind ITS all-reduce (Work.wait() orders the current stream
            # behind the collective, it does not block the host); the caller's stream joins at the end
            for w, lo, hi in buckets:
                with torch.cuda.stream(side):
                    w.wait()
                check(fn(ctypes.byref(sub(lo, hi - lo)), ctypes.c_void_p(side.cuda_stream)), name)
            done = torch.cuda.Event()
            done.record(side)
            main.wait_event(done)
            return
        ev, lo = early
        if self.early_start and ev is not None:
            # opt-in (DYK_OPT_OVERLAP=early / optimizer.early_start = True): the side stream waits for the event recorded in the
            # MIDDLE of the backward pass only.  Anything the caller enqueued on its own stream between backward() and step() that
            # touches gradients -- clip_grad_norm_, manual scaling, gradient-norm logging -- is NOT ordered against it (a clip
            # would be ignored for G[lo:], a norm could read zeros).  For loops that call step() straight after backward().
            side.wait_event(ev)
        else:
            # default: behind EVERYTHING the caller has enqueued so far: the two ranges still run side by side
            # and the transposed packs are rebuilt beside the next forward
            now = torch.cuda.Event()
            now.record(main)
            side.wait_event(now)
        check(fn(ctypes.byref(sub(lo, n - lo)), ctypes.c_void_p(side.cuda_stream)), name)
        done = torch.cuda.Event()
        done.record(side)
        check(fn(ctypes.byref(sub(0, lo)), ctypes.c_void_p(main.cuda_stream)), name)
        main.wait_event(done)

    def _desc(self, st):
        m, v = self._buffers(st)
        g = self.param_groups[0]
        d = L.DykOptimDesc()
        d.p, d.g, d.m = st.P.data_ptr(), st.G.data_ptr(), m.data_ptr()
        d.v = v.data_ptr() if v is not None else None
        d.n = st.total
        d.lr, d.weight_decay, d.grad_scale = float(g["lr"]), float(g["weight_decay"]), float(self.grad_scale)
        d.zero_grad = 1 if self.zero_in_step else 0
        mask = self._trainable_mask(st)
        d.mask = mask.data_ptr() if mask is not None else None
        bf = st._compute.get(torch.bfloat16)
        d.wc = bf["Wc"].data_ptr() if bf is not None else None
        return d, bf

    def _check_loss_flag(self):
        """the reference raises IndexError inside compute_loss when a target lies on the right / bottom image edge
        (grid index == grid size: build_utils/utils.py:370 with :248); the HIP loss records that in a device flag
        which dyk.detect copies to pinned host memory asynchronously.  Flags whose copy has completed are examined
        here (no host synchronisation: the error surfaces one optimizer step late at most)."""
        from .detect import raise_if_target_outside_grid
        raise_if_target_outside_grid(self.model, wait=False)

    def _finish(self, st, bf):
        if self.zero_in_step:
            st.grads_dirty = False
        st.mark_dirty()
        if bf is not None:                # the bf16 copy was written by the step kernel: refresh the rest only
            side = self._side
            st.compute_weights(torch.bfloat16, skip_cast=True, side=side)
        for dt in list(st._compute):
            if dt != torch.bfloat16:
                st.compute_weights(dt)

    # ------------------------------------------------------------------ checkpoint format of torch.optim
    def _export_state(self):
        st = self.model.engine.store
        if self._m is None or st.P is None:
            return
        by_param = {id(e.param): e for e in st.entries}
        flats = dict(zip(self._state_names, (self._m, self._v)))
        for p in self.param_groups[0]["params"]:
            e = by_param[id(p)]
            s = {"step": torch.tensor(float(self._t))}
            for name, flat in flats.items():
                s[name] = st._view(flat, e)
            self.state[p] = s

    def state_dict(self):
        self._export_state()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        st = self._store()
        m, v = self._buffers(st)
        flats = dict(zip(self._state_names, (m, v)))
        by_param = {id(e.param): e for e in st.entries}
        steps = set()
        loaded = False
        with torch.no_grad():
            for p, s in list(self.state.items()):
                e = by_param[id(p)]
                for name, flat in flats.items():
                    if name in s and s[name] is not None:
                        st._view(flat, e).copy_(s[name].to(flat.device, torch.float32))
                        loaded = True
                if "step" in s:
                    steps.add(int(float(s["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ (%s): the fused step keeps one counter" % sorted(steps))
        # torch.optim.SGD keeps only 'momentum_buffer' (no 'step'): a checkpoint that carries buffers is past its first
        # step -- with _t = 0 the next step would take the first-step branch (m = g) and drop the loaded momentum
        self._t = steps.pop() if steps else (1 if loaded else 0)
        self.state.clear()               # the flat buffers are the state; views are rebuilt on demand
        self._fresh = True               # (the copies above ran on the caller's stream)


class FusedAdam(_FusedBase):
    _state_names = ("exp_avg", "exp_avg_sq")

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(model, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        st = self._store()
        self._check_loss_flag()
        st.attach_grads()
        self._t += 1
        d, bf = self._desc(st)
        g = self.param_groups[0]
        d.beta1, d.beta2, d.eps, d.step = float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), self._t
        self._launch(load().dyk_adam_step, "dyk_adam_step", d, bf)
        self._finish(st, bf)


class FusedSGD(_FusedBase):
    _state_names = ("momentum_buffer",)

    def __init__(self, model, lr=1e-3, momentum=0.9, weight_decay=0.0, nesterov=True):
        if not nesterov:
            raise NotImplementedError("the reference only uses nesterov=True (train.py:88-89)")
        super().__init__(model, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=True, dampening=0))

    @torch.no_grad()
    def step(self, closure=None):
        st = self._store()
        self._check_loss_flag()
        st.attach_grads()
        self._t += 1
        d, bf = self._desc(st)
        d.beta1, d.step = float(self.param_groups[0]["momentum"]), self._t
        self._launch(load().dyk_sgd_step, "dyk_sgd_step", d, bf)
        self._finish(st, bf)
