"""Fused optimizers over the flat parameter store (one HIP launch per step for the whole model).

Drop-in for the torch.optim.Adam / SGD(nesterov) instances reference train.py:85-91 builds: same update
rule, same constructor arguments, `step()` / `zero_grad()` / `param_groups[0]['lr']` for LR schedulers.
"""
import ctypes

import torch

from . import lib as L
from .lib import check, load


class _FusedBase(torch.optim.Optimizer):
    def __init__(self, model, defaults):
        self.model = model
        params = [p for p in model.parameters()]
        super().__init__(params, defaults)
        self._t = 0
        self._m = self._v = None
        self.grad_scale = 1.0
        self.zero_in_step = True

    def _store(self):
        st = self.model.engine.store
        if st.P is None:
            raise L.DykError("run one forward pass (or model.engine.store.adopt(device)) before optimizer.step()")
        return st

    def zero_grad(self, set_to_none=False):
        st = self.model.engine.store
        if st.G is not None:
            st.attach_grads()
            if not getattr(self, "_grads_clean", False):
                st.G.zero_()
                self._grads_clean = True

    def _desc(self, st):
        if self._m is None or self._m.device != st.P.device:
            self._m = torch.zeros_like(st.P)
            self._v = torch.zeros_like(st.P)
        g = self.param_groups[0]
        d = L.DykOptimDesc()
        d.p, d.g, d.m, d.v = st.P.data_ptr(), st.G.data_ptr(), self._m.data_ptr(), self._v.data_ptr()
        d.n = st.total
        d.lr, d.weight_decay, d.grad_scale = float(g["lr"]), float(g["weight_decay"]), float(self.grad_scale)
        d.zero_grad = 1 if self.zero_in_step else 0
        bf = st._compute.get(torch.bfloat16)
        d.wc = bf["Wc"].data_ptr() if bf is not None else None
        return d, bf

    def _finish(self, st, bf):
        self._grads_clean = bool(self.zero_in_step)
        st.mark_dirty()
        if bf is not None:                # the bf16 copy was written by the step kernel: refresh the rest only
            st.compute_weights(torch.bfloat16, skip_cast=True)
        for dt in list(st._compute):
            if dt != torch.bfloat16:
                st.compute_weights(dt)


class FusedAdam(_FusedBase):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(model, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        st = self._store()
        st.attach_grads()
        self._t += 1
        d, bf = self._desc(st)
        g = self.param_groups[0]
        d.beta1, d.beta2, d.eps, d.step = float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), self._t
        check(load().dyk_adam_step(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "dyk_adam_step")
        self._finish(st, bf)


class FusedSGD(_FusedBase):
    def __init__(self, model, lr=1e-3, momentum=0.9, weight_decay=0.0, nesterov=True):
        if not nesterov:
            raise NotImplementedError("the reference only uses nesterov=True (train.py:88-89)")
        super().__init__(model, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        st = self._store()
        st.attach_grads()
        self._t += 1
        d, bf = self._desc(st)
        d.beta1, d.step = float(self.param_groups[0]["momentum"]), self._t
        check(load().dyk_sgd_step(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "dyk_sgd_step")
        self._finish(st, bf)
