"""Prints a slice of a rocprofv3 kernel trace as a timeline: start (us from the slice's first kernel),
duration, queue, kernel.  python tools/timeline.py <trace dir> <fraction of the trace, 0..1> <count>"""
import csv
import glob
import re
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
first = int(float(sys.argv[2]) * len(rows))
rows = rows[first:first + int(sys.argv[3])]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name).split("::")[-1].split(" ")[-1][:28]
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{a / 1e3:9.1f} +{(b - a) / 1e3:7.1f}  q{r['Queue_Id']}  {name}  grid {r['Grid_Size_X']}")
