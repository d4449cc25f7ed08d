"""Summarise a rocprofv3 kernel trace CSV as a timeline: merged runs of the same kernel, with idle gaps.

usage: python tools/trace_timeline.py <kernel_trace.csv> [--from-last N]   (prints the last N ms window)"""
import csv, sys, re

path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
win_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 130.0
t_end = rows[-1][1]
rows = [r for r in rows if r[0] >= t_end - win_ms * 1e6]
t0 = rows[0][0]
short = lambda n: re.sub(r"\(.*", "", n.replace("void tmf::", "").replace("tmf::", ""))[:44]
runs = []
for s, e, n in rows:
    n = short(n)
    if runs and runs[-1][2] == n and s - runs[-1][1] < 200_000:
        runs[-1][1] = e; runs[-1][3] += 1; runs[-1][4] += e - s
    else:
        runs.append([s, e, n, 1, e - s])
prev = None
busy = 0
for s, e, n, c, d in runs:
    gap = (s - prev) / 1e6 if prev else 0.0
    busy += d
    flag = f"   <-- idle {gap:.2f} ms" if gap > 0.3 else ""
    print(f"{(s - t0) / 1e6:9.2f} ms  +{(e - s) / 1e6:7.2f}  busy {d / 1e6:7.2f}  x{c:<4d} {n}{flag}")
    prev = e
print(f"window {((rows[-1][1]) - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms")
