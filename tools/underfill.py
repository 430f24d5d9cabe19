"""analysis: time during which the kernels in flight do not fill the chip (sum of workgroups of all running kernels < 256), by
kernel name -- from a rocprofv3 kernel trace (csv or csv.gz) of bench.py, last complete step (between the last two optimizer launches)."""
import csv, gzip, io, sys
from collections import defaultdict

path = sys.argv[1]
f = io.TextIOWrapper(gzip.open(path)) if path.endswith(".gz") else open(path)
rows = [r for r in csv.DictReader(f)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
opt = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"] or "sgd_kernel" in r["Kernel_Name"]]
starts = [opt[0]] + [b for a, b in zip(opt, opt[1:]) if int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) > 2_000_000]
lo, hi = starts[-2], starts[-1]
step = rows[lo:hi]
def wgs(r):
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    w = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    return max(1, g // max(w, 1))
ev = []
for i, r in enumerate(step):
    ev.append((int(r["Start_Timestamp"]), 1, i)); ev.append((int(r["End_Timestamp"]), 0, i))
ev.sort()
running, t_prev = {}, ev[0][0]
under = defaultdict(float); idle = 0.0; under_total = 0.0
for t, kind, i in ev:
    dt = (t - t_prev) / 1e3
    if dt > 0:
        tot = sum(running.values())
        if not running:
            idle += dt
        elif tot < 256:
            under_total += dt
            for j in running:
                under[step[j]["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]] += dt / len(running)
    t_prev = t
    if kind == 1: running[i] = wgs(step[i])
    else: running.pop(i, None)
wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
print("step wall %.0f us, %d launches; chip idle %.0f us; under-filled (< 256 workgroups in flight) %.0f us" % (wall, len(step), idle, under_total))
for k, v in sorted(under.items(), key=lambda kv: -kv[1])[:25]:
    print("  %-40s %8.1f us" % (k[:40], v))
