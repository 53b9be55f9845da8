"""From a rocprofv3 kernel trace (csv): per hardware queue the time its kernels ran, and over all
queues the time at least one / at least two kernels ran, for the last `window` milliseconds of the
trace: python tools/queue_overlap.py <trace dir> [window ms]."""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 60e6
end = max(int(r["End_Timestamp"]) for r in rows)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= end - window]
queues = {}
events = []
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = queues.setdefault(r["Queue_Id"], [0, 0])
    q[0] += b - a
    q[1] += 1
    events += [(a, 1), (b, -1)]
events.sort()
depth, last, busy1, busy2 = 0, events[0][0], 0, 0
for t, d in events:
    if depth >= 1:
        busy1 += t - last
    if depth >= 2:
        busy2 += t - last
    depth += d
    last = t
span = events[-1][0] - events[0][0]
for q, (t, n) in sorted(queues.items()):
    print(f"queue {q}: {n} kernels, {t / 1e6:.2f} ms of kernel time")
print(f"span {span / 1e6:.2f} ms; some kernel running {busy1 / 1e6:.2f} ms; two or more {busy2 / 1e6:.2f} ms")
