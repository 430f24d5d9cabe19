"""Which aten ops still launch kernels inside one train step of bench.py's loop (torch.profiler, CPU-side op names with the
kernels they launched).  python tools/aten_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-roofline"]
import bench  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

orig_sync = torch.cuda.synchronize
state = {"n": 0, "prof": None}


def hooked_sync(*a, **k):
    orig_sync(*a, **k)
    state["n"] += 1
    if state["n"] == 1:          # after the warm-up: profile the timed steps
        state["prof"] = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True)
        state["prof"].__enter__()
    elif state["n"] == 2 and state["prof"] is not None:
        state["prof"].__exit__(None, None, None)
        ev = state["prof"].events()
        rows = {}
        for e in ev:
            if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.kernels:
                k = (e.name, tuple(sorted({kk.name[:40] for kk in e.kernels})), tuple(s for s in (e.stack or [])[:6] if "site-packages" not in s)[:3])
                rows[k] = rows.get(k, 0) + 1
        for (name, kern, stack), n in sorted(rows.items(), key=lambda kv: -kv[1]):
            print(n, name, kern, "|", " <- ".join(stack), file=sys.stderr)
        state["prof"] = None


torch.cuda.synchronize = hooked_sync
bench.main()
