#!/usr/bin/env python3
"""Aggregates bench.py --profile-csv output by (kind, variant, M, N, K, groups)."""
import collections
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    agg = collections.OrderedDict()
    for line in open(path):
        r = line.strip().split(",")
        key = tuple(int(v) for v in r[:6])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += float(r[6])
    tot = sum(a[1] for a in agg.values())
    print("total MFMA-launch ms per step: %.3f" % (tot / steps))
    print("kind var         M      N      K    G  cnt  ms/step    TF/s   cum%")
    cum = 0.0
    for (kind, var, m, n, k, g), (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        cum += ms
        fl = 2.0 * m * n * k * g * cnt
        print("%d    %d  %9d %6d %6d %4d %4d %8.3f %7.1f %6.1f" % (kind, var, m, n, k, g, cnt // steps, ms / steps,
                                                                 fl / (ms * 1e-3) / 1e12, 100 * cum / tot))


if __name__ == "__main__":
    main()
