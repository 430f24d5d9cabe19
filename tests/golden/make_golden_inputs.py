"""Generates tests/golden/inputs.npz: the reference harness's input path (train_utils/kaist_train_eval_utils.py:54-71)
executed with the reference's own torch calls in the build container.  train_utils itself cannot be imported here
(pycocotools / torch._six are absent), so the four statements are issued directly:
    imgs.float() / 255.0 ;  sf = img_size / max(shape) ;  ns = ceil(x*sf/gs)*gs ;  F.interpolate(bilinear, False)
Run:  python tests/golden/make_golden_inputs.py
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)
gs = 32
out = {}
u8 = torch.randint(0, 256, (1, 3, 64, 80), dtype=torch.uint8)
out["u8"] = u8.numpy()
f = u8.float() / 255.0
out["plain"] = f.numpy()
sizes = []
for img_size in (64, 96, 160, 80):
    sf = img_size / max(f.shape[2:])
    ns = [math.ceil(x * sf / gs) * gs for x in f.shape[2:]] if sf != 1 else list(f.shape[2:])
    sizes.append([img_size] + ns)
    out["ms_%d" % img_size] = F.interpolate(f, size=ns, mode="bilinear", align_corners=False).numpy()
out["sizes"] = np.array(sizes, dtype=np.int64)
# an odd, non-multiple target and a 1-pixel axis exercise the clamps of the index arithmetic
out["odd_37x53"] = F.interpolate(f, size=[37, 53], mode="bilinear", align_corners=False).numpy()
one = torch.rand(1, 2, 1, 7)
out["one_src"] = one.numpy()
out["one_5x3"] = F.interpolate(one, size=[5, 3], mode="bilinear", align_corners=False).numpy()
np.savez_compressed(os.path.join(os.path.dirname(__file__), "inputs.npz"), **out)
print({k: v.shape for k, v in out.items()})
