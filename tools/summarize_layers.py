"""Summarise gpurun_out/layers.json (bench.py --dump-layers): time by conv problem."""
import collections
import json
import sys

rows = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/layers.json"))
agg = collections.OrderedDict()
for r in rows:
    key = (r["pass_"], r["op"], r["Cin"], r["Cout"], r["Hg"], r["Wg"], r["taps"], r.get("isy", 1), r.get("osy", 1))
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += r["ms"]
    a[2] += r["tflops"] * r["ms"]
tot = sum(a[1] for a in agg.values())
print("total conv+wgrad ms %.2f" % tot)
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-4s %-5s Cin %4d Cout %4d %3dx%-3d taps %2d isy %d osy %d  n=%2d  ms %.3f (%.1f%%)  avg TF %.0f" % (
        key + (a[0], a[1], 100 * a[1] / tot, a[2] / max(a[1], 1e-9))))
