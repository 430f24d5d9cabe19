"""Compare two `bench.py --dump-cmds` files of the SAME box / call: per (pass, label) isolated time, sorted by difference.
    python tools/cmd_compare.py a.json b.json"""
import collections
import json
import sys


def agg(path):
    out = collections.OrderedDict()
    for r in json.load(open(path)):
        a = out.setdefault((r["pass"], r["label"]), [0, 0.0])
        a[0] += 1
        a[1] += r["us"]
    return out


def main():
    A, B = agg(sys.argv[1]), agg(sys.argv[2])
    fam = collections.OrderedDict()
    rows = []
    for k in A:
        if k in B:
            rows.append((B[k][1] - A[k][1], k, A[k], B[k]))
            f = fam.setdefault(k[1].split()[0] + "/" + k[0], [0.0, 0.0])
            f[0] += A[k][1]
            f[1] += B[k][1]
    print("total %.1f -> %.1f us" % (sum(a[1] for a in A.values()), sum(b[1] for b in B.values())))
    for f, (a, b) in sorted(fam.items(), key=lambda kv: kv[1][1] - kv[1][0]):
        if abs(b - a) > 5:
            print("  %-22s %9.1f -> %9.1f  (%+7.1f)" % (f, a, b, b - a))
    rows.sort()
    for d, k, a, b in rows[:15] + rows[-15:]:
        print("%-4s %-44s n=%3d %8.1f -> %8.1f (%+7.1f)" % (k[0], k[1], a[0], a[1], b[1], d))


if __name__ == "__main__":
    main()
