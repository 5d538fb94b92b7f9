"""MFMA-pipe busy fraction per kernel, reconciled with the FLOP-derived roofline fraction.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d out/pmc_mfma -o m -- \
        python3 bench.py --no-graph --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 --launch-flops out/launch_flops.json
    python tools/pmc_mfma_busy.py out/pmc_mfma/*/m_counter_collection.csv out/pmc_mfma/*/m_kernel_trace.csv out/launch_flops.json

Three numbers per kernel symbol, per launch on average:

  busy_over_algorithmic  SQ_VALU_MFMA_BUSY_CYCLES / (64 cycles x algorithmic MFMA count), the count being the
                         launch's FLOPs / 4096 (v_mfma_f32_32x32x2_f32 = 4096 FLOP, 64 cycles on its SIMD).  Clock- and
                         time-free: ~1.0 means the counter is exactly the matrix pipe's issue cycles (sum over SIMDs) and
                         the kernel issues no MFMA beyond the algorithmic ones; > 1 = padded tiles.
  busy_frac_time         SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x dispatch duration x 2.4 GHz): the fraction of the
                         chip's matrix-pipe cycles at the NOMINAL clock over the dispatch's own begin..end timestamps
                         (kernel trace).  By construction = achieved / peak x busy_over_algorithmic, so it can only be
                         >= the FLOP-derived fraction measured on the same (profiled) dispatches.
  grbm_span_ratio        GRBM_GUI_ACTIVE / 8 XCDs / (duration x 2.4 GHz).  Round 1 divided by 128 x GRBM_GUI_ACTIVE; for
                         dispatches of tens of microseconds that counter spans more than the kernel (MI355X_MICROARCH.md,
                         DVFS note: "reads high on dispatches shorter than about 0.3 ms"), which is why its busy
                         fractions (0.46-0.52) sat BELOW the FLOP-derived 0.58.  A ratio > 1 is that over-span.
"""
import collections
import csv
import json
import re
import sys

CLK_GHZ, SIMDS, XCDS = 2.4, 1024, 8


def norm(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    return re.sub(r"\s+", "", name.split("(")[0])


def main(counter_csv, trace_csv=None, flops_json=None):
    dur = {}
    if trace_csv:
        with open(trace_csv, newline="") as f:
            for r in csv.DictReader(f):
                dur[r.get("Dispatch_Id") or r.get("Correlation_Id")] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    with open(counter_csv, newline="") as f:
        for r in csv.DictReader(f):
            nm, did = norm(r["Kernel_Name"]), r.get("Dispatch_Id") or r.get("Correlation_Id")
            acc[nm][r["Counter_Name"]] += float(r["Counter_Value"])
            if did not in seen[nm]:
                seen[nm].add(did)
                if "Start_Timestamp" in r and r["Start_Timestamp"]:
                    acc[nm]["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                elif did in dur:
                    acc[nm]["ns"] += dur[did]
    flops = {}
    if flops_json:
        with open(flops_json) as f:
            flops = {norm(k): v for k, v in json.load(f).items()}
    out = []
    for nm, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
        n, busy, grbm, ns = len(seen[nm]), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0), v.get("ns", 0.0)
        if busy <= 0 or n == 0:
            continue
        row = {"kernel": nm, "launches": n, "busy_cycles_per_launch": round(busy / n), "avg_us": round(ns / n / 1e3, 2) if ns else None}
        if ns:
            row["busy_frac_time"] = round(busy / (SIMDS * ns * CLK_GHZ), 4)
            row["grbm_span_ratio"] = round(grbm / XCDS / (ns * CLK_GHZ), 3) if grbm else None
            row["busy_frac_round1_formula"] = round(busy / (128.0 * grbm), 4) if grbm else None
        fl = flops.get(nm)
        if fl:
            gf = fl["gflop_per_step"] / fl["launches_per_step"]
            row["gflop_per_launch"] = round(gf, 4)
            row["busy_over_algorithmic"] = round((busy / n) / (gf * 1e9 / 4096.0 * 64.0), 4)
            if ns:
                row["flop_frac_same_dispatches"] = round(gf * 1e9 / (ns / n * 1e-9) / 157.3e12, 4)
        out.append(row)
    print(json.dumps({"method": " ".join(__doc__.split()), "clock_ghz_nominal": CLK_GHZ, "kernels": out}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
