"""Development aid: prints the kernel timeline of the last step in a rocprofv3 kernel-trace CSV
(start offset, duration, queue, kernel) so that overlap between streams can be read off."""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last accumulate kernels mark the end of steps
acc = [i for i, r in enumerate(rows) if "accumulateKernel" in r["Kernel_Name"]]
tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 1
first = acc[-2 * tiles] + 1 if len(acc) >= 2 * tiles else 0
seg = rows[first:acc[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
print("span", (int(seg[-1]["End_Timestamp"]) - t0) / 1e3, "us")
