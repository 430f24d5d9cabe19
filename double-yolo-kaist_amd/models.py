"""`models.py` of the reference on the MI355X HIP path: same public surface (create_modules,
YOLOLayer, YOLO, load_darknet_weights; reference models.py:7-364), same module-list structure and
state_dict key names, but YOLO.forward executes a compiled native plan (dyk/plan.py, dyk/engine.py)
instead of walking the module list in Python.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from build_utils import torch_utils
from build_utils.layers import *  # noqa: F401,F403  (callers do `from models import *`, reference models.py:1-4)
from build_utils.layers import (ConvBlock, DepthwiseSeparableConv2d, FeatureConcat, Inception, SqueezeExcitation,
                                WeightedFeatureFusion, make_activation)
from build_utils.parse_config import *  # noqa: F401,F403
from build_utils.parse_config import parse_model_cfg
from build_utils.utils import get_yolo_layers


def create_modules(modules_defs: list, img_size, cfg):
    """list of cfg section dicts -> (nn.ModuleList, routs bitmap, [net] dict), following the
    reference's mapping (models.py:7-155): channel bookkeeping, which outputs are kept, head bias
    initialisation, and the cfg-path-dependent head strides / box decode flavour."""
    img_size = [img_size] * 2 if isinstance(img_size, int) else img_size
    net_infos = modules_defs.pop(0)
    out_filters = [3]
    module_list = nn.ModuleList()
    routs = []
    yolo_index = -1
    second = net_infos.get("second_index", None)

    for i, mdef in enumerate(modules_defs):
        kind = mdef["type"]
        filters = out_filters[-1]
        if kind == "convolutional":
            bn = mdef["batch_normalize"]
            filters = mdef["filters"]
            k = mdef["size"]
            stride = mdef["stride"] if "stride" in mdef else (mdef["stride_y"], mdef["stride_x"])
            if not isinstance(k, int):
                raise TypeError("conv2d filter size must be int type")
            cin = 3 if (second is not None and i == second) else out_filters[-1]
            modules = ConvBlock()
            conv = nn.Conv2d(in_channels=cin, out_channels=filters, kernel_size=k, stride=stride,
                             padding=k // 2 if mdef["pad"] else 0, groups=mdef["groups"] if "groups" in mdef else 1,
                             bias=not bn)
            conv._dyk_stem = (cin == 3 and conv.groups == 1 and (i == 0 or i == second))
            modules.add_module("Conv2d", conv)
            if bn:
                modules.add_module("BatchNorm2d", nn.BatchNorm2d(filters))
            else:
                routs.append(i)
            act = make_activation(mdef["activation"])
            if act is not None:
                modules.add_module("activation", act)
                modules.act_name = mdef["activation"]
        elif kind == "depthwiseconvolutional":
            ks = mdef["size"] if "size" in mdef else 3
            filters = mdef["filters"]
            stride = mdef["stride"] if "stride" in mdef else (mdef["stride_y"], mdef["stride_x"])
            modules = DepthwiseSeparableConv2d(in_channels=out_filters[-1], out_channels=filters, kernel_size=ks,
                                               stride=stride)
        elif kind == "dropout":
            modules = nn.Dropout(mdef["probability"])
        elif kind == "inception":
            modules = Inception(in_channels=out_filters[-1], n1x1=mdef["n1x1"], n3x3_reduce=mdef["n3x3_reduce"],
                                n3x3=mdef["n3x3"], n5x5_reduce=mdef["n5x5_reduce"], n5x5=mdef["n5x5"],
                                pool_proj=mdef["pool_proj"])
            filters = mdef["n1x1"] + mdef["n3x3"] + mdef["n5x5"] + mdef["pool_proj"]
        elif kind == "se":
            modules = SqueezeExcitation(in_channels=out_filters[-1], squeeze_factor=mdef["squeeze_factor"])
        elif kind == "maxpool":
            k = mdef["size"]
            modules = nn.MaxPool2d(kernel_size=k, stride=mdef["stride"], padding=(k - 1) // 2)
        elif kind == "avgpool":
            modules = nn.AdaptiveAvgPool2d(output_size=mdef["size"])
        elif kind == "upsample":
            modules = nn.Upsample(scale_factor=mdef["stride"])
        elif kind == "route":
            layers = mdef["layers"]
            filters = sum([out_filters[l + 1 if l > 0 else l] for l in layers])
            layers = [i + l if l < 0 else l for l in layers]
            routs.extend(layers)
            modules = FeatureConcat(layers=layers)
        elif kind == "shortcut":
            layers = [i + l if l < 0 else l for l in mdef["from"]]
            routs.extend(layers)
            modules = WeightedFeatureFusion(layers=layers, weight="weights_type" in mdef)
        elif kind == "yolo":
            yolo_index += 1
            stride = [8, 16, 32, 64, 128]
            if any(x in cfg for x in ["yolov-tiny", "fpn", "yolov3"]):
                stride = [32, 16, 8]
            modules = YOLOLayer(anchors=mdef["anchors"][mdef["mask"]], nc=mdef["classes"], img_size=img_size,
                                stride=stride[yolo_index], bf_type="yolov4" if "yolov4" in cfg else "yolov3")
            try:
                # detection-head bias initialisation of the preceding conv (reference :135-144)
                b = module_list[-1][0].bias.view(modules.na, -1)
                b.data[:, 4] += -4.5
                b.data[:, 5:] += math.log(0.6 / (modules.nc - 0.99))
                module_list[-1][0].bias = torch.nn.Parameter(b.view(-1), requires_grad=True)
            except Exception as e:  # same tolerance as the reference
                print("WARNING: smart bias initialization failure.", e)
        else:
            print("Warning: Unrecognized Layer Type: " + mdef["type"])
            modules = nn.Sequential()
        module_list.append(modules)
        out_filters.append(filters)

    routs_binary = [False] * len(modules_defs)
    for i in routs:
        routs_binary[i] = True
    return module_list, routs_binary, net_infos


class YOLOLayer(nn.Module):
    """Post-processing of one detection head (reference models.py:158-258): training -> the
    [bs, na, ny, nx, no] view of the head conv output; inference -> additionally the decoded boxes."""

    def __init__(self, anchors, nc, img_size, stride, bf_type="yolov3"):
        super(YOLOLayer, self).__init__()
        self.anchors = torch.Tensor(anchors)
        self.stride = stride
        self.na = len(anchors)
        self.nc = nc
        self.no = nc + 5
        self.nx, self.ny, self.ng = 0, 0, (0, 0)
        self.anchor_vec = self.anchors / self.stride
        self.anchor_wh = self.anchor_vec.view(1, self.na, 1, 1, 2)
        self.bf_type = bf_type
        self.grid = None

    def create_grids(self, ng=(13, 13), device="cpu"):
        self.nx, self.ny = ng
        self.ng = torch.tensor(ng, dtype=torch.float)
        if not self.training:
            yv, xv = torch.meshgrid([torch.arange(self.ny, device=device), torch.arange(self.nx, device=device)],
                                    indexing="ij")
            self.grid = torch.stack((xv, yv), 2).view((1, 1, self.ny, self.nx, 2)).float()
        if self.anchor_vec.device != device:
            self.anchor_vec = self.anchor_vec.to(device)
            self.anchor_wh = self.anchor_wh.to(device)

    def forward(self, p):
        from dyk import functional
        bs, _, ny, nx = p.shape
        if (self.nx, self.ny) != (nx, ny) or self.grid is None:
            self.create_grids((nx, ny), p.device)
        return functional.yolo_layer(p, self)


class YOLO(nn.Module):
    def __init__(self, cfg, img_size=(416, 416), verbose=False):
        super(YOLO, self).__init__()
        self.module_defs = parse_model_cfg(cfg)
        self.module_list, self.routs, self.net_info = create_modules(self.module_defs, img_size, cfg)
        self.yolo_layers = get_yolo_layers(self)
        self.cfg = cfg
        self._engine = None
        self.info(verbose)

    def get_yolo_layers(self):
        return [i for i, module in enumerate(self.module_list) if module.__class__.__name__ == "YOLOLayer"]

    def info(self, verbose=False):
        torch_utils.model_info(self, verbose)

    @property
    def engine(self):
        if self._engine is None:
            from dyk.engine import Engine
            object.__setattr__(self, "_engine", Engine(self))
        return self._engine

    def forward(self, x, y=None):
        """x: visible batch f32 [B,3,H,W] in 0..1, y: LWIR batch or None.
        training -> list of [B,na,ny,nx,no] head tensors; eval -> (boxes [B,N,no], tuple of heads)."""
        return self.engine.forward(x, y)


def load_darknet_weights(model, weights, cutoff=-1):
    """Read a Darknet `.weights` file into the model (reference models.py:318-364): header of three
    int32 + one int64, then per [convolutional] section BN bias, BN weight, running mean, running var
    (or the conv bias), followed by the conv weights."""
    assert weights.endswith(".weights"), "weights file must end with '.weights'"
    with open(weights, "rb") as f:
        model.version = np.fromfile(f, dtype=np.int32, count=3)
        model.seen = np.fromfile(f, dtype=np.int64, count=1)
        blob = np.fromfile(f, dtype=np.float32)
    ptr = 0

    def take(dst):
        nonlocal ptr
        n = dst.numel()
        dst.data.copy_(torch.from_numpy(blob[ptr:ptr + n]).view_as(dst))
        ptr += n

    for mdef, module in zip(model.module_defs[:cutoff], model.module_list[:cutoff]):
        if mdef["type"] != "convolutional":
            continue
        conv = module[0]
        if mdef["batch_normalize"]:
            bn = module[1]
            take(bn.bias)
            take(bn.weight)
            take(bn.running_mean)
            take(bn.running_var)
        else:
            take(conv.bias)
        take(conv.weight)
    if model._engine is not None:
        model._engine.store.mark_dirty()
