"""HBM bytes per launch of the dominant kernel (wino3_kernel; argv[3] = another kernel-name regex) from two rocprofv3 PMC passes (one counter per pass, as
MI355X_MICROARCH.md's HBM section prescribes; never combined with trace domains):

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/pmc_fetch -o f -- python3 bench.py --no-graph --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/pmc_write -o w -- python3 bench.py --no-graph ... (same)
    python tools/pmc_traffic.py out/pmc_fetch/f_counter_collection.csv out/pmc_write/w_counter_collection.csv

bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half of wide coalesced reads (same guide)."""
import csv
import json
import re
import sys

PAT = re.compile(sys.argv[3] if len(sys.argv) > 3 else r"wino3_kernel")
NAME = sys.argv[3] if len(sys.argv) > 3 else "wino3_kernel"


def avg(path, counter):
    tot, n = 0.0, 0
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and PAT.search(row["Kernel_Name"]):
                tot += float(row["Counter_Value"])
                n += 1
    return tot / max(n, 1), n


if __name__ == "__main__":
    fetch, nf = avg(sys.argv[1], "FETCH_SIZE")
    write, nw = avg(sys.argv[2], "WRITE_SIZE")
    print(json.dumps({
        "kernel": NAME,
        "launches": nf,
        "fetch_size_kb_avg": fetch,
        "write_size_kb_avg": write,
        "traffic_bytes_per_launch": (2 * fetch + write) * 1024,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --no-graph`; "
                  "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reports half of wide coalesced reads, "
                  "MI355X_MICROARCH.md HBM section)",
    }, indent=1))
