#!/usr/bin/env python3
"""GPU busy time, span and the largest idle gaps of a rocprofv3 kernel trace
(p_kernel_trace.csv), per run when the trace holds several (runs are split where
the GPU idles for more than `split_ms`)."""
import csv
import sys


def main():
    path = sys.argv[1]
    split_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    runs, cur = [], [ev[0]]
    for e in ev[1:]:
        if (e[0] - cur[-1][1]) * 1e-6 > split_ms:
            runs.append(cur)
            cur = []
        cur.append(e)
    runs.append(cur)
    for i, run in enumerate(runs):
        if len(run) < 50:
            continue
        span = (run[-1][1] - run[0][0]) * 1e-6
        busy = sum(e[1] - e[0] for e in run) * 1e-6
        gaps = sorted(((run[k + 1][0] - run[k][1]) * 1e-3, run[k][2][:40], run[k + 1][2][:40])
                      for k in range(len(run) - 1))
        big = [g for g in gaps if g[0] > 30.0]
        print("run %d: %d kernels, span %.1f ms, busy %.1f ms, idle %.1f ms; %d gaps > 30 us sum %.1f ms"
              % (i, len(run), span, busy, span - busy, len(big), sum(g[0] for g in big) * 1e-3))
        if "-v" in sys.argv:
            from collections import Counter
            c = Counter()
            for g in big:
                c[(g[1], g[2])] += g[0]
            for (a, b), t in c.most_common(14):
                print("   %8.1f us  %s -> %s" % (t, a, b))


if __name__ == "__main__":
    main()
