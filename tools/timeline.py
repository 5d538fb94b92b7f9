"""One step's kernel timeline from a rocprofv3 --kernel-trace CSV of bench.py: per dispatch its queue, start offset, duration
and the gap to the previous dispatch on the same queue.  usage: timeline.py <kernel_trace.csv> [step index from the end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at an rng_fill_kernel dispatch
starts = [i for i, r in enumerate(rows) if "rng_fill" in r["Kernel_Name"]]
i0, i1 = starts[-back], starts[-back + 1]
t0 = int(rows[i0]["Start_Timestamp"])
last_end = {}
print(f"step of {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us, {i1 - i0} dispatches")
qs = sorted({r["Queue_Id"] for r in rows[i0:i1]})
for r in rows[i0:i1]:
    q = r["Queue_Id"]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = s - last_end.get(q, s)
    last_end[q] = e
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    print(f"q{qs.index(q)} {s / 1e3:8.1f} +{(e - s) / 1e3:6.1f} us  gap {gap / 1e3:5.1f}  {nm}  grid={r.get('Grid_Size', '')} wg={r.get('Workgroup_Size', '')}")
