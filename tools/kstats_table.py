"""Side-by-side per-kernel time per STEP (us) of the kstats.sh runs in a directory: calls/step x average duration."""
import csv, glob, os, sys
d = sys.argv[1]
STEPS = 60 + 10 + 64          # bench.py --steps 60 --warmup 10 + the 64-step event window (no profile legs)
files = sorted(glob.glob(os.path.join(d, "[0-9]*.csv")), key=lambda p: int(os.path.basename(p)[:-4]))
tab, names = [], []
for f in files:
    t = {}
    for r in csv.DictReader(open(f)):
        nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        nm = nm.split("(")[0][:58]
        t[nm] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3)
        if nm not in names:
            names.append(nm)
    tab.append(t)
print(f"{'kernel':60s}" + "".join(f"{'['+str(i)+'] n/step  us/step':>22s}" for i in range(len(tab))))
tot = [0.0] * len(tab)
for nm in names:
    if max(t.get(nm, (0, 0))[1] for t in tab) / STEPS < 0.5:
        continue
    line = f"{nm:60s}"
    for i, t in enumerate(tab):
        c, us = t.get(nm, (0, 0.0))
        tot[i] += us / STEPS
        line += f"{c / STEPS:10.1f} {us / STEPS:10.1f} "
    print(line)
print(f"{'sum of kernel time per step':60s}" + "".join(f"{'':10s} {x:10.1f} " for x in tot))
