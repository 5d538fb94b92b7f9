"""MFMA-pipe busy fraction per kernel from one rocprofv3 PMC pass (SQ and GRBM counters use different slot groups):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d out/pmc_mfma -o m -- \
        python3 bench.py --no-graph --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0
    python tools/pmc_mfma_busy.py out/pmc_mfma/m_counter_collection.csv

busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 * GRBM_GUI_ACTIVE): the busy counter is summed over all 1024 SIMDs, GRBM_GUI_ACTIVE
over the 8 XCDs (MI355X_MICROARCH.md, DVFS note), i.e. 128 SIMDs per XCD-cycle.  GRBM_GUI_ACTIVE spans the whole dispatch
(ramp-up and drain included) and profiled runs clock lower, so this reads below FLOPs/time against the nominal peak."""
import collections
import csv
import json
import re
import sys

if __name__ == "__main__":
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
            acc[nm][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                calls[nm] += 1
    out = []
    for nm, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
        g, b = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if b > 0:
            out.append({"kernel": nm, "launches": calls[nm], "mfma_busy_frac": round(b / (128.0 * g), 4)})
    print(json.dumps({"method": __doc__.split("\n\n")[2].replace("\n", " "), "kernels": out}, indent=1))
