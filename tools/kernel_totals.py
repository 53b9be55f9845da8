"""Sums a rocprofv3 kernel trace (csv) per kernel and lists the durations, in launch order, of the
launches of one kernel: python tools/kernel_totals.py <trace dir> [kernel substring]."""
import collections
import csv
import glob
import re
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
total, count = collections.Counter(), collections.Counter()


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"<.*", "", name)
    return name.split("::")[-1].split(" ")[-1]


for r in rows:
    n = short(r["Kernel_Name"])
    total[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    count[n] += 1
for n, t in total.most_common(16):
    print(f"{n:34s} {count[n]:6d} launches {t / 1e6:9.3f} ms  {t / 1e3 / count[n]:9.1f} us each")
if len(sys.argv) > 2:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if short(r["Kernel_Name"]) == sys.argv[2]]
    print(sys.argv[2], len(d), "launches; the last 40:", [round(x, 1) for x in d[-40:]])
