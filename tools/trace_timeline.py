"""Per-stream occupancy of ONE train step from a rocprofv3 kernel trace (csv): busy time of every HIP stream in 1-ms
windows between the last two optimizer launches.
    python tools/trace_timeline.py gpurun_out/prof/r1_kernel_trace.csv"""
import collections
import csv
import gzip
import io
import os
import sys

_f = io.TextIOWrapper(gzip.open(sys.argv[1])) if sys.argv[1].endswith(".gz") else open(sys.argv[1])
rows = list(csv.DictReader(_f))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)) for r in rows)
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2] or "sgd_kernel" in e[2]]
# round 4: the optimizer runs as two launches per step (deep layers early on a side stream, the rest behind the backward):
# the step ends with the SMALLER one
grids = sorted({ev[i][4] for i in adam})
if len(grids) > 1:
    adam = [i for i in adam if ev[i][4] == grids[0]]
ev = [e[:4] for e in ev]
# STEP_BACK=1 takes the step before the last one (the profiler's buffer flush can land in the last step)
_k = int(os.environ.get("STEP_BACK", "0"))
a0, a1 = adam[-2 - _k], adam[-1 - _k]
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
# chip-idle time: the union of all kernel intervals, the largest gaps with the launches either side, and where the loss sits
iv = sorted((s, e, n) for s, e, n, st in seg)
gaps, cur_end, last = [], iv[0][0], iv[0][2]
idle = 0
for s, e, n in iv:
    if s > cur_end:
        gaps.append((s - cur_end, (cur_end - t0) / 1e6, last, n)); idle += s - cur_end
    if e > cur_end:
        cur_end, last = e, n
print("chip idle (no kernel on any stream): %.2f ms in %d gaps" % (idle / 1e6, len(gaps)))
hist = collections.Counter(min(int(g[0] / 1000) // 2 * 2, 20) for g in gaps)
print("gap histogram (us bucket: count, total us):", {k: (v, round(sum(g[0] for g in gaps if min(int(g[0] / 1000) // 2 * 2, 20) == k) / 1e3)) for k, v in sorted(hist.items())})
for g in sorted(gaps, reverse=True)[:12]:
    print("  gap %6.1f us at %6.2f ms  after %-40s before %s" % (g[0] / 1e3, g[1], g[2][:40], g[3][:40]))
for s, e, n, st in seg:
    if "match_loss" in n:
        print("loss kernel at %.2f ms (forward before it, backward after)" % ((s - t0) / 1e6))
