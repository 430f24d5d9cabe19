"""Operator surface of the reference's build_utils/layers.py on the MI355X HIP path.

Inside `models.YOLO` these modules are *parameter containers and graph nodes*: YOLO.forward compiles
the whole cfg into a native command list (dyk/plan.py) instead of calling them one by one.  Called
on their own (`module(x)` / `module(x, outputs)`) they run the same HIP kernels through
dyk.functional on torch NCHW tensors (inference semantics, no autograd).

Names, constructor signatures and state_dict keys follow the reference (layers.py:9-320) so that
checkpoints and user code interchange.
"""
import math
import sys  # noqa: F401  (re-exported through `from build_utils.layers import *`, as in the reference)

import numpy as np  # noqa: F401
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401


def make_divisible(v, divisor):
    """smallest multiple of `divisor` that is >= v (reference layers.py:9-11)"""
    return math.ceil(v / divisor) * divisor


def _F():
    from dyk import functional
    return functional


class FeatureConcat(nn.Module):
    """[route]: channel concat of earlier layer outputs, or an alias when there is one (ref :32-44)."""

    def __init__(self, layers):
        super(FeatureConcat, self).__init__()
        self.layers = layers
        self.multiple = len(layers) > 1

    def forward(self, x, outputs):
        if not self.multiple:
            return outputs[self.layers[0]]
        return _F().concat([outputs[i] for i in self.layers])


class WeightedFeatureFusion(nn.Module):
    """[shortcut]: x + outputs[from], optionally with learnable sigmoid weights * 2/n (ref :47-85)."""

    def __init__(self, layers, weight=False):
        super(WeightedFeatureFusion, self).__init__()
        self.layers = layers
        self.weight = weight
        self.n = len(layers) + 1
        if weight:
            self.w = nn.Parameter(torch.zeros(self.n), requires_grad=True)

    def forward(self, x, outputs):
        return _F().weighted_fusion(x, [outputs[i] for i in self.layers], self.w if self.weight else None, self.n)


class SqueezeExcitation(nn.Module):
    """channel attention: GAP -> fc1 -> ReLU -> fc2 -> hardsigmoid -> scale * x (ref :175-190)"""

    def __init__(self, in_channels: int, squeeze_factor: int = 4):
        super(SqueezeExcitation, self).__init__()
        squeeze_channel = make_divisible(in_channels // squeeze_factor, 8)
        self.fc1 = nn.Conv2d(in_channels, squeeze_channel, 1)
        self.fc2 = nn.Conv2d(squeeze_channel, in_channels, 1)

    def forward(self, x):
        return _F().squeeze_excitation(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)


class ConvBlock(nn.Sequential):
    """nn.Sequential(Conv2d[, BatchNorm2d][, activation]) of a [convolutional] section
    (reference models.py:28-64); child names give the reference's state_dict keys."""

    act_name = "linear"

    def forward(self, x):
        bn = self.BatchNorm2d if hasattr(self, "BatchNorm2d") else None
        return _F().conv_bn_act(x, self.Conv2d, bn, self.act_name, self.training)


_ACTS = {"mish": nn.Mish, "relu": nn.ReLU, "relu6": nn.ReLU6, "hard-sigmoid": nn.Hardsigmoid, "hard-swish": nn.Hardswish}


def make_activation(name):
    """the activation module the reference would add for a cfg `activation=` value, or None"""
    if name == "leaky":
        return nn.LeakyReLU(0.1, inplace=True)
    if name in _ACTS:
        return _ACTS[name](inplace=True)
    return None


class ConvBnActivation(nn.Module):
    """Conv2d + BN + activation as an nn.ModuleList named `conv` (ref :88-122)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, pad=0, groups=1, activation="leaky", bn=True):
        super(ConvBnActivation, self).__init__()
        self.conv = nn.ModuleList()
        self.conv.append(nn.Conv2d(in_channels, out_channels, kernel_size, stride,
                                   padding=kernel_size // 2 if pad else 0, groups=groups, bias=not bn))
        if bn:
            self.conv.append(nn.BatchNorm2d(out_channels))
        act = make_activation(activation)
        if act is not None:
            self.conv.append(act)
        self.act_name = activation if (act is not None) else "linear"
        self.has_bn = bn

    def forward(self, x):
        return _F().conv_bn_act(x, self.conv[0], self.conv[1] if self.has_bn else None, self.act_name, self.training)


class DepthwiseSeparableConv2d(nn.Module):
    """DW kxk (padding fixed to 1) + BN + ReLU6 -> PW 1x1 + BN + ReLU6 (ref :218-234)"""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1):
        super(DepthwiseSeparableConv2d, self).__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(in_channels, in_channels, kernel_size, stride, 1, groups=in_channels, bias=False),
            nn.BatchNorm2d(in_channels),
            nn.ReLU6(inplace=True),
            nn.Conv2d(in_channels, out_channels, 1, 1, 0, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU6(inplace=True))

    def forward(self, x):
        f = _F()
        x = f.conv_bn_act(x, self.conv[0], self.conv[1], "relu6", self.training)
        return f.conv_bn_act(x, self.conv[3], self.conv[4], "relu6", self.training)


class Inception(nn.Module):
    """four-branch Inception block (ref :148-172)"""

    def __init__(self, in_channels, n1x1, n3x3_reduce, n3x3, n5x5_reduce, n5x5, pool_proj):
        super(Inception, self).__init__()
        self.branch1 = nn.Sequential(ConvBnActivation(in_channels, n1x1, kernel_size=1))
        self.branch2 = nn.Sequential(ConvBnActivation(in_channels, n3x3_reduce, kernel_size=1),
                                     ConvBnActivation(n3x3_reduce, n3x3, kernel_size=3, pad=1))
        self.branch3 = nn.Sequential(ConvBnActivation(in_channels, n5x5_reduce, kernel_size=1),
                                     ConvBnActivation(n5x5_reduce, n5x5, kernel_size=3, pad=1),
                                     ConvBnActivation(n5x5, n5x5, kernel_size=3, pad=1))
        self.branch4 = nn.Sequential(nn.MaxPool2d(kernel_size=3, stride=1, padding=1),
                                     ConvBnActivation(in_channels, pool_proj, kernel_size=1))

    def forward(self, x):
        f = _F()
        b4 = self.branch4[1](f.maxpool(x, 3))
        return f.concat([self.branch1(x), self.branch2(x), self.branch3(x), b4])


# ----------------------------------------------------------------------------------------------
# The remaining names exist in the reference module but are used by none of its cfgs (SURVEY §2.1
# row 3).  They are kept importable; they are plain parameter-free glue, not part of the hot path.
class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class Concat(nn.Module):
    def __init__(self, dimension=1):
        super(Concat, self).__init__()
        self.d = dimension

    def forward(self, x):
        return torch.cat(x, self.d)


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


class HardSwish(nn.Module):
    def forward(self, x):
        return x * torch.clamp(x + 3, 0., 6.) / 6.


class Mish(nn.Module):
    def forward(self, x):
        return x * torch.nn.functional.softplus(x).tanh()


MemoryEfficientSwish = Swish
MemoryEfficientMish = Mish


class ResBlock(nn.Module):
    """n x (1x1 ConvBnActivation -> 3x3 ConvBnActivation [+ input]) (ref :125-145; no cfg section creates it).
    Built from the same HIP-backed ConvBnActivation blocks; the skip add is one dyk_axpby."""

    def __init__(self, in_channels, filter_1, out_channels, block_nums=1, activation="mish", shortcut=True):
        super(ResBlock, self).__init__()
        self.shortcut = shortcut
        self.module_list = nn.ModuleList()
        for _ in range(block_nums):
            self.module_list.append(nn.ModuleList([
                ConvBnActivation(in_channels, filter_1, kernel_size=1, stride=1, pad=1, activation=activation, bn=True),
                ConvBnActivation(filter_1, out_channels, kernel_size=3, stride=1, pad=1, activation=activation, bn=True)]))

    def forward(self, x):
        for pair in self.module_list:
            y = pair[1](pair[0](x))
            x = _F().weighted_fusion(y, [x], None, 2) if self.shortcut else y
        return x


class SEInceptionFusion(nn.Module):
    """FeatureConcat -> 1x1 ConvBnActivation [-> Inception] [-> SqueezeExcitation] (ref :193-215; no cfg section
    creates it -- the `*_seinc` cfgs spell the same thing out as [route] / [convolutional] / [inception] / [se])."""

    def __init__(self, in_channels, out_channels, layers, inception=False, icp_param_list=(), tmse=False, squeeze_factor=4):
        super(SEInceptionFusion, self).__init__()
        self.concat = FeatureConcat(layers)
        self.enhance = nn.ModuleList([ConvBnActivation(in_channels, out_channels, kernel_size=1)])
        if inception:
            self.enhance.append(Inception(out_channels, *icp_param_list))
        if tmse:
            self.enhance.append(SqueezeExcitation(out_channels, squeeze_factor))

    def forward(self, x, outputs):
        y = self.concat(x, outputs)
        for m in self.enhance:
            y = m(y)
        return y


class MixConv2d(nn.Module):
    """Mixed-kernel convolution (ref :237-268): output channels split over kernel sizes k, either evenly
    ('equal_ch') or so that every group has about the same number of weights ('equal_params').  Parameters and
    state_dict keys (`m.<g>.weight/bias`) as in the reference; used by none of its cfgs, so the forward is not built."""

    def __init__(self, in_ch, out_ch, k=(3, 5, 7), stride=1, dilation=1, bias=True, method="equal_params"):
        super(MixConv2d, self).__init__()
        n = len(k)
        if method == "equal_ch":
            idx = torch.linspace(0, n - 1e-6, out_ch).floor()          # ref :246 (same group boundaries)
            ch = [int((idx == g).sum()) for g in range(n)]
        else:
            # channels c_g with c_g * k_g^2 equal for all groups and sum(c_g) = out_ch: c_g = out_ch * k_g^-2 / sum_j k_j^-2
            inv = [1.0 / (kk * kk) for kk in k]
            ch = [int(round(out_ch * v / sum(inv))) for v in inv]
            ch[0] += out_ch - sum(ch)
        self.m = nn.ModuleList([nn.Conv2d(in_ch, ch[g], k[g], stride, k[g] // 2, dilation=dilation, bias=bias) for g in range(n)])

    def forward(self, x):
        raise NotImplementedError("MixConv2d is not created by any cfg section of the reference; its forward is not built")
