"""Oracle (test infrastructure): the dual-stream YOLO forward pass as a functional torch-CPU
interpreter over parsed cfg sections.

Restates reference models.py:7-155 (create_modules: how each cfg section becomes an operator, the
channel bookkeeping, which layer outputs are kept), :158-258 (YOLOLayer reshape + box decode) and
:279-315 (YOLO.forward: the module-list walk with the stream switch at `second_index`), plus the
operators of build_utils/layers.py it instantiates (FeatureConcat :32-44, WeightedFeatureFusion
:47-85, SqueezeExcitation :175-190, DepthwiseSeparableConv2d :218-234).  Parameters live in a plain
dict keyed by the reference's state_dict names, so golden vectors and checkpoints interchange.
Pinned by tests/golden/fwd_*.npz (reference outputs) through tests/test_oracle_golden.py.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-5, 0.1      # torch defaults; the reference never overrides them


def _act(name, x):
    if name == "mish":
        return F.mish(x)
    if name == "relu":
        return F.relu(x)
    if name == "leaky":
        return F.leaky_relu(x, 0.1)
    if name == "relu6":
        return F.relu6(x)
    if name == "hard-sigmoid":
        return F.hardsigmoid(x)
    if name == "hard-swish":
        return F.hardswish(x)
    return x                                   # 'linear' and anything else: no module added (models.py:63-64)


def make_divisible(v, divisor):
    return math.ceil(v / divisor) * divisor    # layers.py:9-11


class OracleNet:
    def __init__(self, module_defs, cfg_name):
        """module_defs: list of dicts as returned by parse_model_cfg, [net] first; cfg_name: the cfg
        path string (its substrings select head strides and the v3/v4 box decode, models.py:124-131)."""
        self.net = dict(module_defs[0])
        self.defs = [dict(d) for d in module_defs[1:]]
        self.cfg = cfg_name
        self.v4 = "yolov4" in cfg_name
        self.second_index = self.net.get("second_index", None)
        out_filters = [3]
        routs = []
        self.layers = []                       # per layer: dict(kind=..., ...)
        self.param_shapes = OrderedDict()      # state_dict name -> shape
        self.yolo_layers = []
        yolo_index = -1
        for i, m in enumerate(self.defs):
            t = m["type"]
            L = {"kind": t}
            filters = out_filters[-1]
            pre = "module_list.%d." % i
            if t == "convolutional":
                bn = m["batch_normalize"]
                filters = m["filters"]
                k = m["size"]
                stride = m["stride"] if "stride" in m else (m["stride_y"], m["stride_x"])
                cin = 3 if (self.second_index is not None and i == self.second_index) else out_filters[-1]
                groups = m["groups"] if "groups" in m else 1
                L.update(bn=bool(bn), cin=cin, cout=filters, k=k, stride=stride, pad=k // 2 if m["pad"] else 0,
                         groups=groups, act=m["activation"])
                self.param_shapes[pre + "Conv2d.weight"] = (filters, cin // groups, k, k)
                if bn:
                    self._bn_shapes(pre + "BatchNorm2d.", filters)
                else:
                    self.param_shapes[pre + "Conv2d.bias"] = (filters,)
                    routs.append(i)
            elif t == "depthwiseconvolutional":
                ks = m["size"] if "size" in m else 3
                filters = m["filters"]
                stride = m["stride"] if "stride" in m else (m["stride_y"], m["stride_x"])
                cin = out_filters[-1]
                L.update(cin=cin, cout=filters, k=ks, stride=stride)
                self.param_shapes[pre + "conv.0.weight"] = (cin, 1, ks, ks)
                self._bn_shapes(pre + "conv.1.", cin)
                self.param_shapes[pre + "conv.3.weight"] = (filters, cin, 1, 1)
                self._bn_shapes(pre + "conv.4.", filters)
            elif t == "se":
                c = out_filters[-1]
                cs = make_divisible(c // m["squeeze_factor"], 8)
                L.update(c=c, cs=cs)
                self.param_shapes[pre + "fc1.weight"] = (cs, c, 1, 1)
                self.param_shapes[pre + "fc1.bias"] = (cs,)
                self.param_shapes[pre + "fc2.weight"] = (c, cs, 1, 1)
                self.param_shapes[pre + "fc2.bias"] = (c,)
            elif t == "maxpool":
                L.update(k=m["size"], stride=m["stride"], pad=(m["size"] - 1) // 2)
            elif t == "upsample":
                L.update(scale=m["stride"])
            elif t == "route":
                layers = m["layers"]
                filters = sum(out_filters[l + 1 if l > 0 else l] for l in layers)
                layers = [i + l if l < 0 else l for l in layers]
                routs.extend(layers)
                L.update(layers=layers)
            elif t == "shortcut":
                layers = [i + l if l < 0 else l for l in m["from"]]
                routs.extend(layers)
                weighted = "weights_type" in m
                L.update(layers=layers, weighted=weighted, n=len(layers) + 1)
                if weighted:
                    self.param_shapes[pre + "w"] = (len(layers) + 1,)
            elif t == "yolo":
                yolo_index += 1
                stride = [8, 16, 32, 64, 128]
                if any(s in cfg_name for s in ["yolov-tiny", "fpn", "yolov3"]):
                    stride = [32, 16, 8]
                anchors = torch.tensor(m["anchors"][m["mask"]], dtype=torch.float32)
                L.update(anchors=anchors, nc=m["classes"], stride=stride[yolo_index], na=len(anchors))
                self.yolo_layers.append(i)
            elif t == "dropout":
                L.update(p=m["probability"])
            elif t == "inception":
                # layers.py:148-172: four branches of ConvBnActivation (Conv2d + BN + LeakyReLU(0.1)) blocks, concatenated;
                # models.py:81-85 leaves `filters` untouched (the cfgs keep sum(branches) == input channels)
                cin = out_filters[-1]
                br = [[(cin, m["n1x1"], 1)],
                      [(cin, m["n3x3_reduce"], 1), (m["n3x3_reduce"], m["n3x3"], 3)],
                      [(cin, m["n5x5_reduce"], 1), (m["n5x5_reduce"], m["n5x5"], 3), (m["n5x5"], m["n5x5"], 3)],
                      [(cin, m["pool_proj"], 1)]]
                L.update(branches=br)
                for bi, convs in enumerate(br):
                    for ci, (a, b, k) in enumerate(convs):
                        q = pre + "branch%d.%d.conv." % (bi + 1, ci + (1 if bi == 3 else 0))
                        self.param_shapes[q + "0.weight"] = (b, a, k, k)
                        self._bn_shapes(q + "1.", b)
            else:
                raise NotImplementedError("oracle: cfg section [%s]" % t)
            self.layers.append(L)
            out_filters.append(filters)
        self.routs = [False] * len(self.defs)
        for r in routs:
            self.routs[r] = True
        self.out_filters = out_filters

    def _bn_shapes(self, pre, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            self.param_shapes[pre + k] = (c,)
        self.param_shapes[pre + "num_batches_tracked"] = ()

    # ------------------------------------------------------------------ parameters
    def anchor_vecs(self):
        return [self.layers[j]["anchors"] / self.layers[j]["stride"] for j in self.yolo_layers]

    def synth_state(self, seed=0):
        """Deterministic synthetic parameters (same procedure is used when generating golden vectors
        with the reference): He-style conv weights, randomised BN affine + running statistics so
        that eval-mode BN is not the identity, reference head-bias initialisation (models.py:135-144)."""
        g = torch.Generator().manual_seed(seed)
        sd = OrderedDict()
        for name, shape in self.param_shapes.items():
            leaf = name.rsplit(".", 1)[1]
            if leaf == "num_batches_tracked":
                sd[name] = torch.zeros((), dtype=torch.long)
            elif leaf == "running_mean":
                sd[name] = torch.randn(shape, generator=g) * 0.1
            elif leaf == "running_var":
                sd[name] = torch.rand(shape, generator=g) + 0.5
            elif name.endswith("BatchNorm2d.weight") or name.endswith("conv.1.weight") or name.endswith("conv.4.weight"):
                sd[name] = torch.rand(shape, generator=g) + 0.5
            elif leaf == "bias":
                sd[name] = torch.randn(shape, generator=g) * 0.1
            elif leaf == "w":
                sd[name] = torch.randn(shape, generator=g) * 0.5
            else:
                fan_in = shape[1] * shape[2] * shape[3]
                sd[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        # head bias init: b[:,4] += -4.5 ; b[:,5:] += log(0.6/(nc-0.99))
        for j in self.yolo_layers:
            L = self.layers[j]
            key = "module_list.%d.Conv2d.bias" % (j - 1)
            if key in sd:
                b = sd[key].view(L["na"], -1)
                b[:, 4] += -4.5
                b[:, 5:] += math.log(0.6 / (L["nc"] - 0.99))
        return sd

    # ------------------------------------------------------------------ forward
    def _bn(self, sd, pre, x, training):
        y = F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"],
                         training, BN_MOMENTUM, BN_EPS)
        if training:
            sd[pre + "num_batches_tracked"] += 1
        return y

    def forward(self, sd, x, y=None, training=False, keep_all=False, emulate_bf16=False, force=None):
        """Returns what reference YOLO.forward returns (models.py:307-315): training -> list of
        [B,na,ny,nx,no]; eval -> (cat(io,1), tuple(p)).  keep_all additionally returns every layer's
        output tensor (for per-layer parity).

        emulate_bf16 (eval only): the same arithmetic with the roundings of the bf16 MFMA path -- conv operands
        (activations and weights) and every stored activation rounded to bfloat16, accumulation / BatchNorm affine /
        activation / pooling in fp32, the Cin=3 stems and the detection heads' outputs in fp32 -- so that a bf16 run of
        the HIP path can be held to a per-layer bound instead of a statistical one.
        force: {section index: tensor}: after section i has been evaluated (and recorded in `every`), its output is
        REPLACED by force[i] for everything downstream -- with the tensors of another implementation this measures every
        section's own error on identical inputs (no accumulation through the depth of the net)."""
        if emulate_bf16:
            assert not training
            rnd = lambda t: t.bfloat16().float()                     # noqa: E731
        else:
            rnd = lambda t: t                                      # noqa: E731
        di = self.second_index is not None and y is not None
        yolo_out, out = [], []
        every = []
        self.raw = {}
        for i, L in enumerate(self.layers):
            t = L["kind"]
            pre = "module_list.%d." % i
            if t == "convolutional":
                if di and i == self.second_index:
                    x = y                                           # models.py:299-301 stream switch
                stem = i == 0 or (self.second_index is not None and i == self.second_index)
                w = sd[pre + "Conv2d.weight"]
                if L["groups"] == 1 and not stem:
                    w = rnd(w)                                      # (stem and depthwise kernels read fp32 weights)
                x = F.conv2d(x, w, sd.get(pre + "Conv2d.bias"), L["stride"], L["pad"], 1, L["groups"])
                if keep_all:
                    self.raw[i] = x
                if L["bn"]:
                    x = self._bn(sd, pre + "BatchNorm2d.", x, training)
                x = _act(L["act"], x)
                if L["bn"]:
                    x = rnd(x)                                      # heads (no BN) stay fp32
            elif t == "depthwiseconvolutional":                     # layers.py:223-231: pad fixed at 1, ReLU6
                x = F.conv2d(x, sd[pre + "conv.0.weight"], None, L["stride"], 1, 1, L["cin"])
                x = F.relu6(self._bn(sd, pre + "conv.1.", x, training))
                x = F.conv2d(x, sd[pre + "conv.3.weight"])
                x = F.relu6(self._bn(sd, pre + "conv.4.", x, training))
            elif t == "se":                                         # layers.py:184-190
                s = F.adaptive_avg_pool2d(x, 1)
                s = F.relu(F.conv2d(s, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"]))
                s = F.hardsigmoid(F.conv2d(s, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"]))
                x = rnd(s * x)
            elif t == "maxpool":
                x = F.max_pool2d(x, L["k"], L["stride"], L["pad"])
            elif t == "inception":
                outs_b = []
                for bi, convs in enumerate(L["branches"]):
                    h = F.max_pool2d(x, 3, 1, 1) if bi == 3 else x
                    for ci, (a, b, k) in enumerate(convs):
                        q = pre + "branch%d.%d.conv." % (bi + 1, ci + (1 if bi == 3 else 0))
                        h = F.conv2d(h, sd[q + "0.weight"], None, 1, k // 2)
                        h = F.leaky_relu(self._bn(sd, q + "1.", h, training), 0.1)
                    outs_b.append(h)
                x = torch.cat(outs_b, 1)
            elif t == "upsample":
                x = F.interpolate(x, scale_factor=L["scale"], mode="nearest")
            elif t == "route":                                      # layers.py:42-44
                ls = L["layers"]
                x = torch.cat([out[j] for j in ls], 1) if len(ls) > 1 else out[ls[0]]
            elif t == "shortcut":                                   # layers.py:63-85
                if L["weighted"]:
                    w = torch.sigmoid(sd[pre + "w"]) * (2 / L["n"])
                    x = x * w[0]
                nx = x.shape[1]
                for q, j in enumerate(L["layers"]):
                    a = out[j] * w[q + 1] if L["weighted"] else out[j]
                    na = a.shape[1]
                    if nx == na:
                        x = x + a
                    elif nx > na:
                        x = torch.cat((x[:, :na] + a, x[:, na:]), 1)
                    else:
                        x = x + a[:, :nx]
                x = rnd(x)
            elif t == "yolo":
                yolo_out.append(self._yolo(L, x, training))
            elif t == "dropout":
                x = F.dropout(x, L["p"], training)
            if keep_all:
                every.append(x)
            if force is not None and i in force:
                x = force[i]
            out.append(x if self.routs[i] else None)
        if training:
            res = yolo_out
        else:
            io, p = zip(*yolo_out)
            res = (torch.cat(io, 1), p)
        return (res, every) if keep_all else res

    def _yolo(self, L, p, training):
        """models.py:218-258."""
        bs, _, ny, nx = p.shape
        na, no = L["na"], L["nc"] + 5
        p = p.view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
        if training:
            return p
        yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float()
        anchor_wh = (L["anchors"] / L["stride"]).view(1, na, 1, 1, 2)
        if self.v4:
            io = p.sigmoid()
            xy = io[..., :2] * 2. - 0.5 + grid
            wh = (io[..., 2:4] * 2) ** 2 * anchor_wh
            io = torch.cat((xy * L["stride"], wh * L["stride"], io[..., 4:]), -1)
        else:
            xy = torch.sigmoid(p[..., :2]) + grid
            wh = torch.exp(p[..., 2:4]) * anchor_wh
            io = torch.cat((xy * L["stride"], wh * L["stride"], torch.sigmoid(p[..., 4:])), -1)
        return io.view(bs, -1, no), p
