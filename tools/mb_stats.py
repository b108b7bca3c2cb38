"""Per-step summary of a c5 rocprof run:  python tools/mb_stats.py gpurun_out/<tag>/prof_c5 [timeline]"""
import csv
import sys
from pathlib import Path

d = Path(sys.argv[1])
rows = list(csv.DictReader(open(next(d.glob("*kernel_stats.csv")))))
trace = list(csv.DictReader(open(next(d.glob("*kernel_trace.csv")))))
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(trace) if "collate_augment" in r["Kernel_Name"]]
steps = len(marks)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:56].ljust(56), f"{int(r['Calls']) / steps:6.1f}/step",
          f"{float(r['TotalDurationNs']) / 1e3 / steps:8.0f} us/step  avg {float(r['AverageNs']) / 1e3:7.1f}  max {float(r['MaxNs']) / 1e3:7.1f}")
print(f"kernel time per step {tot / 1e3 / steps:.0f} us, launches per step {sum(int(r['Calls']) for r in rows) / steps:.1f}")
if len(sys.argv) > 2 and sys.argv[2] == "timeline":
    a, b = marks[len(marks) // 2], marks[len(marks) // 2 + 1]
    t0 = int(trace[a]["Start_Timestamp"])
    prev = t0
    for r in trace[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap {(s - prev) / 1e3:6.1f} q{r['Queue_Id']} {nm} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
        prev = max(prev, e)
