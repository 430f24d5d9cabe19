"""Per-stream occupancy of ONE train step from a rocprofv3 kernel trace (csv): busy time of every HIP stream in 1-ms
windows between the last two optimizer launches.
    python tools/trace_timeline.py gpurun_out/prof/r1_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)) for r in rows)
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2] or "sgd_kernel" in e[2]]
# round 4: the optimizer runs as two launches per step (deep layers early on a side stream, the rest behind the backward):
# the step ends with the SMALLER one
grids = sorted({ev[i][4] for i in adam})
if len(grids) > 1:
    adam = [i for i in adam if ev[i][4] == grids[0]]
ev = [e[:4] for e in ev]
a0, a1 = adam[-2], adam[-1]
seg = ev[a0 + 1:a1 + 1]
t0 = ev[a0][1]
T = (ev[a1][1] - t0) / 1e6
streams = sorted({e[3] for e in seg})
nb = int(T) + 1
busy = {s: [0.0] * nb for s in streams}
for s, e, n, st in seg:
    a, b = (s - t0) / 1e6, (e - t0) / 1e6
    w = int(a)
    while w < nb and w < b:
        busy[st][w] += max(0.0, min(b, w + 1) - max(a, w))
        w += 1
print("step %.2f ms; per-ms busy fraction per stream (%s)" % (T, ", ".join("s" + s for s in streams)))
for w in range(nb):
    print("%3d ms  " % w + "  ".join("%4.0f%%" % (100 * busy[s][w]) for s in streams))
tot = {s: sum(busy[s]) for s in streams}
print("busy ms:", {("s" + s): round(v, 2) for s, v in tot.items()})
