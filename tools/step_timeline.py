"""Prints one steady-state step of a rocprofv3 --kernel-trace CSV as a timeline (start offset, duration, queue, kernel)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def nm(r):
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:44]


names = [nm(r) for r in rows]
anchor = sys.argv[2] if len(sys.argv) > 2 else "logmel"
idx = [i for i, n in enumerate(names) if anchor in n]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
for i in range(a, b):
    r = rows[i]
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1000:8.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000:7.1f} '
          f'q{r["Queue_Id"]} {names[i]}')
print(f"step: {(int(rows[b]['Start_Timestamp']) - t0) / 1000:.1f} us, {b - a} launches")
