"""Parameter store: every trainable tensor of a YOLO model lives in ONE flat fp32 device buffer
(and its gradient in a second one), conv weights in the tap-major order [kh*kw][Cout][Cin] the
kernels consume.  The nn.Parameters the reference API exposes (`module_list[i].Conv2d.weight`, ...)
are strided *views* into that buffer with the reference's shapes, so state_dict()/load_state_dict()/
torch.optim keep working while
  * the weight-gradient kernel accumulates straight into `.grad` storage,
  * the compute-dtype copy of all weights is one cast launch, the transposed copy one more,
  * a data-parallel gradient all-reduce sees one contiguous buffer in layer order.
"""
import ctypes

import torch

from . import lib as L
from .lib import check, load

ALIGN = 64  # elements


def _round_up(n, a):
    return (n + a - 1) // a * a


class Entry:
    __slots__ = ("name", "param", "kind", "shape", "offset", "numel", "layer", "storage_shape", "perm")


class ParamStore:
    def __init__(self, model):
        self.model = model
        self.wt_ready = None    # event behind a side-stream rebuild of the transposed packs (compute_weights(side=...))
        self.entries = []       # parameters (flat P / G)
        self.rstats = []        # (module, attr, offset, numel)  running_mean / running_var
        self.nbt = []           # BatchNorm modules whose num_batches_tracked we own
        self.by_name = {}
        off = 0
        roff = 0
        import torch.nn as nn
        for li, mod in enumerate(model.module_list):
            for mname, sub in mod.named_modules():
                for pname, p in sub.named_parameters(recurse=False):
                    e = Entry()
                    e.name = "module_list.%d.%s%s" % (li, (mname + ".") if mname else "", pname)
                    e.param, e.layer, e.shape = p, li, tuple(p.shape)
                    e.numel = p.numel()
                    if isinstance(sub, nn.Conv2d) and pname == "weight":
                        co, ci, kh, kw = p.shape
                        if getattr(sub, "_dyk_stem", False):
                            e.kind, e.storage_shape, e.perm = "stem_w", (co, kh, kw, ci), (0, 3, 1, 2)
                        elif sub.groups > 1:
                            if not (sub.groups == sub.in_channels == sub.out_channels):
                                raise NotImplementedError("grouped convolution that is not depthwise (%s)" % e.name)
                            # depthwise: [k*k][C], read as fp32 by the depthwise kernels
                            e.kind, e.storage_shape, e.perm = "dw_w", (kh, kw, co, ci), (2, 3, 0, 1)
                        else:
                            e.kind, e.storage_shape, e.perm = "conv_w", (kh, kw, co, ci), (2, 3, 0, 1)
                    else:
                        e.kind, e.storage_shape, e.perm = "vec", tuple(p.shape), None
                    e.offset = off
                    off += _round_up(e.numel, ALIGN)
                    self.entries.append(e)
                    self.by_name[e.name] = e
                if isinstance(sub, nn.BatchNorm2d):
                    for attr in ("running_mean", "running_var"):
                        n = getattr(sub, attr).numel()
                        self.rstats.append((sub, attr, roff, n))
                        roff += _round_up(n, ALIGN)
                    self.nbt.append(sub)
        self.total = max(off, ALIGN)
        self.rtotal = max(roff, ALIGN)
        self.P = self.G = self.R = self.NBT = None
        self.device = None
        self._compute = {}      # dtype -> dict(Wc, Wt, stems, version)
        self._ttable = None
        self._pad_layout = None
        self._dirty = 0
        self.grads_dirty = False   # set by every backward pass, cleared by zero_grad() / a fused step that zeroes G

    def frozen_key(self):
        """names of the parameters frozen with requires_grad_(False) (reference train.py:77-82): part of the plan
        key (their weight gradients are not computed) and of the fused optimizers' update mask"""
        return tuple(e.name for e in self.entries if not e.param.requires_grad)

    # ------------------------------------------------------------------ adoption
    def _view(self, flat, e):
        s = flat[e.offset:e.offset + e.numel].view(e.storage_shape)
        return s.permute(*e.perm) if e.perm is not None else s

    def is_adopted(self, device):
        if self.P is None or self.device != device:
            return False
        e0, e1 = self.entries[0], self.entries[-1]
        ok = e0.param.data.data_ptr() == self.P.data_ptr() + 4 * e0.offset and \
            e1.param.data.data_ptr() == self.P.data_ptr() + 4 * e1.offset
        if ok and self.rstats:
            m, attr, roff, n = self.rstats[-1]
            ok = getattr(m, attr).data_ptr() == self.R.data_ptr() + 4 * roff
        return ok

    def adopt(self, device):
        """(Re)build the flat buffers on `device` from the current parameter values and make the
        Parameters / BN buffers views into them.  Called lazily, e.g. after model.to(device)."""
        P = torch.zeros(self.total, dtype=torch.float32, device=device)
        G = torch.zeros(self.total, dtype=torch.float32, device=device)
        R = torch.zeros(self.rtotal, dtype=torch.float32, device=device)
        NBT = torch.zeros(max(len(self.nbt), 1), dtype=torch.long, device=device)
        with torch.no_grad():
            for e in self.entries:
                v = self._view(P, e)
                v.copy_(e.param.data.to(device=device, dtype=torch.float32))
                e.param.data = v
                e.param.grad = None
            for (m, attr, roff, n) in self.rstats:
                v = R[roff:roff + n]
                v.copy_(getattr(m, attr).to(device=device, dtype=torch.float32))
                m._buffers[attr] = v
            for i, m in enumerate(self.nbt):
                NBT[i] = int(m.num_batches_tracked)
                m._buffers["num_batches_tracked"] = NBT[i]
        self.P, self.G, self.R, self.NBT, self.device = P, G, R, NBT, device
        self._compute = {}
        self._ttable = None

    def grads_attached(self):
        live = [e for e in self.entries if e.param.requires_grad]
        for e in self.entries:
            if not e.param.requires_grad and e.param.grad is not None:
                return False
        for e in live[:1] + live[-1:]:
            g = e.param.grad
            if g is None or g.data_ptr() != self.G.data_ptr() + 4 * e.offset:
                return False
        return True

    def attach_grads(self):
        """Make every p.grad the matching view of G.  If they were detached (zero_grad(set_to_none),
        first step) G is zeroed first, otherwise the kernels keep accumulating into it."""
        if self.grads_attached():
            return
        self.G.zero_()
        self.grads_dirty = False
        for e in self.entries:
            # frozen parameters have no .grad, as under autograd (the backward skips their weight gradients)
            e.param.grad = self._view(self.G, e) if e.param.requires_grad else None

    # ------------------------------------------------------------------ pointers
    def p_ptr(self, name):
        e = self.by_name[name]
        return self.P.data_ptr() + 4 * e.offset

    def g_ptr(self, name):
        e = self.by_name[name]
        return self.G.data_ptr() + 4 * e.offset

    def r_ptr(self, mod, attr):
        for (m, a, roff, n) in self.rstats:
            if m is mod and a == attr:
                return self.R.data_ptr() + 4 * roff
        raise KeyError(attr)

    # ------------------------------------------------------------------ compute-dtype staging
    def _layout_padded(self):
        """Offsets (elements) of the zero-padded weight packs that channel counts off the GEMM K step need:
        fwd  [t][Cout][ru32(Cin)]  for Cin  % 32 != 0   (forward conv reads K = padded input channels)
        bwd  [t][Cin][ru32(Cout)]  for Cout % 32 != 0   (data-gradient conv reads K = padded output channels)"""
        if getattr(self, "_pad_layout", None) is None:
            fwd, bwd, fo, bo = {}, {}, 0, 0
            for e in self.entries:
                if e.kind != "conv_w":
                    continue
                co, ci, kh, kw = e.shape
                if ci % 32:
                    fwd[e.name] = fo
                    fo += _round_up(kh * kw * co * _round_up(ci, 32), ALIGN)
                if co % 32:
                    bwd[e.name] = bo
                    bo += _round_up(kh * kw * ci * _round_up(co, 32), ALIGN)
            self._pad_layout = (fwd, max(fo, ALIGN), bwd, max(bo, ALIGN))
        return self._pad_layout

    def _transpose_table(self):
        if self._ttable is None:
            _, _, bwd_off, _ = self._layout_padded()
            tabs = []
            for padded in (False, True):
                ents = [e for e in self.entries if e.kind == "conv_w" and bool(e.shape[0] % 32) == padded]
                arr = (L.DykTransposeEntry * max(len(ents), 1))()
                tiles = 0
                for i, e in enumerate(ents):
                    co, ci, kh, kw = e.shape
                    arr[i].src_off = e.offset
                    arr[i].dst_off = bwd_off[e.name] if padded else e.offset
                    arr[i].taps, arr[i].rows, arr[i].cols = kh * kw, co, ci
                    arr[i].dst_ld = _round_up(co, 32) if padded else 0
                    arr[i].tile_begin = tiles
                    tiles += kh * kw * ((co + 31) // 32) * ((ci + 31) // 32)
                raw = bytes(arr)
                t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
                tabs.append((t, len(ents), tiles))
            self._ttable = tabs
        return self._ttable

    def mark_dirty(self):
        self._dirty += 1

    def version_key(self):
        return (sum(e.param._version for e in self.entries), self._dirty)

    def alloc_compute(self, dtype):
        """allocate (without filling) the compute-dtype staging buffers; used by dry plan compilation"""
        st = self._alloc(dtype)
        st["version"] = None
        return st

    def _alloc(self, dtype):
        fwd_off, fwd_n, bwd_off, bwd_n = self._layout_padded()
        st = {"stems": {}, "stems_t": {}, "fwd_pad_off": fwd_off, "bwd_pad_off": bwd_off}
        st["Wc"] = self.P if dtype == torch.float32 else torch.empty(self.total, dtype=dtype, device=self.device)
        st["Wt"] = torch.zeros(self.total, dtype=dtype, device=self.device)
        st["Wc_pad"] = torch.zeros(fwd_n, dtype=dtype, device=self.device)
        st["Wt_pad"] = torch.zeros(bwd_n, dtype=dtype, device=self.device)
        for e in self.entries:
            if e.kind == "stem_w":
                co, ci, kh, kw = e.shape
                st["stems"][e.name] = torch.zeros((co, _round_up(ci * kh * kw, 32)), dtype=dtype, device=self.device)
                # fp32 [kh*kw*ci][co]: the direct stem kernel streams it through scalar loads (csrc/stem.hip)
                st["stems_t"][e.name] = torch.zeros((ci * kh * kw, co), dtype=torch.float32, device=self.device)
        return st

    def compute_weights(self, dtype, force=False, skip_cast=False, side=None):
        """Return dict(Wc=<tensor same offsets as P>, Wt=<transposed per tap>, stems={name: tensor})
        for the compute dtype, refreshed iff the master buffer changed since the last call.
        side: a torch stream for the tap-major transposed packs (Wt / Wt_pad).  Only data gradients read them, so after an
        optimizer step they are rebuilt beside the NEXT forward pass instead of behind the step (0.26 ms per step on the target
        cfg); `wt_ready` is the event the next backward waits for (Engine._run_backward)."""
        st = self._compute.get(dtype)
        # Parameter.data views do not share a version counter with P, so track the parameters'
        # own counters (optimizer steps / load_state_dict bump them) plus an explicit dirty flag
        # for writers that go through `.data` (load_darknet_weights).
        ver = self.version_key()
        if st is not None and st["version"] == ver and not force:
            return st
        lib = load()
        if self.wt_ready is not None:
            # a rebuild of the transposed packs may still be pending on the optimizer's side stream (load_state_dict / mark_dirty /
            # a new dtype between step() and the next backward): this rebuild, on whichever stream, comes after it
            torch.cuda.current_stream().wait_event(self.wt_ready)
            if side is not None:
                side.wait_event(self.wt_ready)
            else:
                self.wt_ready = None
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        code = L.DYK_BF16 if dtype == torch.bfloat16 else L.DYK_F32
        if st is None:
            st = self._alloc(dtype)
            self._compute[dtype] = st
        if dtype != torch.float32 and not skip_cast:
            check(lib.dyk_cast_f32(self.P.data_ptr(), st["Wc"].data_ptr(), self.total, code, stream), "dyk_cast_f32")
        tstream = stream
        if side is not None:
            here = torch.cuda.Event()
            here.record(torch.cuda.current_stream())       # behind the optimizer step that wrote P
            side.wait_event(here)
            tstream = ctypes.c_void_p(side.cuda_stream)
        for (tab, n, tiles), dst in zip(self._transpose_table(), (st["Wt"], st["Wt_pad"])):
            if n:
                check(lib.dyk_transpose_taps(self.P.data_ptr(), dst.data_ptr(), tab.data_ptr(), n, tiles, code, tstream),
                      "dyk_transpose_taps")
        if side is not None:
            self.wt_ready = torch.cuda.Event()
            self.wt_ready.record(side)
        # every K-padded forward pack, the stems' padded packs and their fp32 [27][Cout] transposes: ONE table-driven launch
        # (round 5: the MobileNetV3 cfg has 68 padded packs -- 68 launches of 3 us plus two torch copies, 0.5-0.8 ms of a
        # nearly idle GPU on the caller's stream at every step boundary)
        tab = st.get("pad_table")
        if tab is None:
            esz = 2 if dtype == torch.bfloat16 else 4
            ents = []
            for name, off in st["fwd_pad_off"].items():
                e = self.by_name[name]
                co, ci, kh, kw = e.shape
                ents.append((self.P.data_ptr() + 4 * e.offset, st["Wc_pad"].data_ptr() + esz * off, kh * kw * co, ci, _round_up(ci, 32), 0))
            for e in self.entries:
                if e.kind == "stem_w":
                    co, ci, kh, kw = e.shape
                    t = st["stems"][e.name]
                    ents.append((self.P.data_ptr() + 4 * e.offset, t.data_ptr(), co, ci * kh * kw, t.shape[1], 0))
                    ents.append((self.P.data_ptr() + 4 * e.offset, st["stems_t"][e.name].data_ptr(), co, ci * kh * kw, ci * kh * kw, 1))
            arr = (L.DykPadEntry * max(len(ents), 1))()
            blocks = 0
            for i, (src, dst, rows, cols, cpad, tr) in enumerate(ents):
                arr[i].src, arr[i].dst, arr[i].rows, arr[i].cols, arr[i].cpad = src, dst, rows, cols, cpad
                arr[i].blk_begin, arr[i].transpose_f32 = blocks, tr
                blocks += (rows * (cols if tr else cpad) + 2047) // 2048
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            tab = st["pad_table"] = (dev, len(ents), blocks, self.P.data_ptr())
        if tab[1]:
            assert tab[3] == self.P.data_ptr()
            check(lib.dyk_cast_pad_table(tab[0].data_ptr(), tab[1], tab[2], code, stream), "dyk_cast_pad_table")
        st["version"] = ver
        return st
