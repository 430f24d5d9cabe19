"""bf16 results do not depend on the process: the autotuner's tile choices fix the summation order of the
MFMA path; they persist in a file keyed by the library's source digest and the device (dyk/plan.py:_tune_file_path), so a
second process runs the configurations the first one chose and produces the same bits."""
import os
import subprocess
import sys

import pytest

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import hashlib, sys
sys.path[:0] = [%r, %r]
import torch
from build_utils.parse_config import materialize_cfg
from models import YOLO
torch.manual_seed(0)
m = YOLO(materialize_cfg("kaist_dyolov4_fshare_global_concat_se3"))
m.dyk_dtype = "bf16"
m = m.cuda().train()
g = torch.Generator().manual_seed(3)
x, y = torch.rand(2, 3, 128, 160, generator=g).cuda(), torch.rand(2, 3, 128, 160, generator=g).cuda()
out = m(x, y)
h = hashlib.sha256()
for o in out:
    h.update(o.detach().float().cpu().numpy().tobytes())
sum((o * o).sum() for o in out).backward()
h.update(m.engine.store.G.cpu().numpy().tobytes())
from dyk import plan
print("TUNED", len(plan._TUNE_CACHE), plan._TUNE_FILE["path"])
print("BITS", h.hexdigest())
''' % (ROOT, PKG)


def _run(env):
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = {l.split()[0]: l.split()[1:] for l in out.stdout.splitlines() if l.split() and l.split()[0] in ("TUNED", "BITS")}
    return lines["BITS"][0], int(lines["TUNED"][0]), lines["TUNED"][1]


def test_two_fresh_processes_give_bit_identical_bf16_heads_and_gradients(tmp_path):
    env = dict(os.environ)
    env["DYK_TUNE_CACHE_DIR"] = str(tmp_path)
    env.pop("DYK_TUNE_CACHE", None)
    bits1, n1, path1 = _run(env)
    assert n1 > 20 and os.path.exists(path1) and os.path.dirname(path1) == str(tmp_path)
    stamp = os.stat(path1).st_mtime_ns
    bits2, n2, path2 = _run(env)
    assert path2 == path1 and n2 == n1
    assert os.stat(path1).st_mtime_ns == stamp, "the second process tuned again instead of loading the first one's choices"
    assert bits1 == bits2, "bf16 heads / gradients differ between two processes"
