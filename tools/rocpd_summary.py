"""Per-kernel statistics (the `--stats` table) from a rocprofv3 rocpd SQLite database.
usage: python tools/rocpd_summary.py <results.db> <out.csv> [steps]"""
import csv
import re
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
                 f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size) "
                 f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n if len(n) < 110 else n[:107] + "..."


with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPR", "AGPR", "LDS_bytes"])
    for n, calls, total, mn, mx, vg, ag, lds in rows:
        w.writerow([short(n), calls, total, round(total / calls, 1), round(100.0 * total / tot, 3), mn, mx, vg, ag, lds])
print("kernels: %d, dispatches: %d, total GPU kernel time %.3f ms%s" % (
    len(rows), sum(r[1] for r in rows), tot / 1e6, (" (%.2f ms/step over %d steps)" % (tot / 1e6 / steps, steps)) if steps else ""))
for n, calls, total, mn, mx, vg, ag, lds in rows[:14]:
    print("%6.2f%% %7d calls avg %9.1f us  %s" % (100.0 * total / tot, calls, total / calls / 1e3, short(n)[:100]))
