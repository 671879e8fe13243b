"""Timeline of a streaming turn from a rocprofv3 kernel trace (tools/turn_loop.py): per kernel of the turn its start relative to the
turn's first kernel, its duration, and the idle gap in front of it on the critical chain; medians over the turns after warm-up."""
import csv
import glob
import statistics
import sys

d = sys.argv[1]
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ofarn::", "")))
rows.sort()
# split into turns at the grid filter kernel (the last kernel of a turn)
turns, cur = [], []
for s, e, k in rows:
    cur.append((s, e, k))
    if "k_grid_filter" in k:
        turns.append(cur)
        cur = []
turns = [t for t in turns if len(t) == len(turns[-1])][5:]
print(f"{len(turns)} turns of {len(turns[0])} kernels")
n = len(turns[0])
tot = statistics.median(t[-1][1] - t[0][0] for t in turns) / 1e3
busy = statistics.median(sum(e - s for s, e, _ in t) for t in turns) / 1e3
print(f"turn: first kernel start -> last kernel end {tot:.1f} us; sum of kernel durations {busy:.1f} us")
print(f"{'#':>3} {'start':>8} {'dur':>7} {'gap':>7}  kernel")
for i in range(n):
    st = statistics.median(t[i][0] - t[0][0] for t in turns) / 1e3
    du = statistics.median(t[i][1] - t[i][0] for t in turns) / 1e3
    # gap: time since the latest end of any earlier kernel of the turn (0 if it overlaps one)
    gp = statistics.median(max(0, t[i][0] - max(x[1] for x in t[:i])) if i else 0 for t in turns) / 1e3
    print(f"{i:3d} {st:8.1f} {du:7.1f} {gp:7.1f}  {turns[0][i][2][:90]}")
