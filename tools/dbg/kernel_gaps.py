#!/usr/bin/env python3
"""Idle time between consecutive kernels of the training step, from a rocprofv3 --kernel-trace CSV of a single-stream run
(RN_TOWER_STREAMS=0 RN_WGRAD_STREAMS=0): sum of (start of kernel i+1 - end of kernel i) where positive, per step."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# drop everything before the first opt_adam (set-up + first step's allocations), keep whole steps between optimizer launches
adam = [i for i, r in enumerate(rows) if r[2].startswith("opt_adam")]
lo, hi = adam[0] + 1, adam[-1] + 1
seg = rows[lo:hi]
n_steps = len(adam) - 1
busy = sum(e - s for s, e, _ in seg)
gaps = [max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)]
span = seg[-1][1] - seg[0][0]
print("%d steps, %d kernels per step" % (n_steps, len(seg) // n_steps))
print("per step: span %.2f ms, kernels busy %.2f ms, gaps %.2f ms (%.1f %% of the span)" % (span / n_steps / 1e6, busy / n_steps / 1e6, sum(gaps) / n_steps / 1e6, 100.0 * sum(gaps) / span))
g = sorted(gaps)
print("gap median %.1f us, mean %.1f us, 90th percentile %.1f us, largest %.1f us" % (g[len(g) // 2] / 1e3, sum(g) / len(g) / 1e3, g[int(0.9 * len(g))] / 1e3, g[-1] / 1e3))
big = sorted(((max(0, seg[i + 1][0] - seg[i][1]), seg[i][2][:50], seg[i + 1][2][:50]) for i in range(len(seg) - 1)), reverse=True)[:8]
for gp, a, b in big:
    print("  %.1f us between %s -> %s" % (gp / 1e3, a, b))
