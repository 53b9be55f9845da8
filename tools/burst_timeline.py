"""Development aid: start, duration and the gap before each kernel of the last pass in a rocprofv3 kernel trace
(the burst of tools/share_rehearsal.py --steps 20 --warmup 5): python3 tools/burst_timeline.py <trace dir>"""
import csv
import glob
import re
import sys

rows = []
for path in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = [i for i, r in enumerate(rows) if "generateKernel" in r["Kernel_Name"]][-1]
t0 = int(rows[last]["Start_Timestamp"])
previous_end = t0
for r in rows[last:]:
    start, end = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    found = re.search(r"(\w+Kernel\w*|__amd_\w+)", r["Kernel_Name"])
    name = found.group(1) if found else r["Kernel_Name"][:40]
    print(f"{(start - t0) / 1e3:9.1f} us  +{(start - previous_end) / 1e3:7.1f} gap  {(end - start) / 1e3:8.1f} us  {name}")
    previous_end = max(previous_end, end)
print(f"total {(previous_end - t0) / 1e3:.1f} us")
