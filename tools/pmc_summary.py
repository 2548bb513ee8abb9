"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per
launch and kernel.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dirF> -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dirW> -- python3 bench.py ...
  python tools/pmc_summary.py <dirF> <dirW> [--by-grid] [--workload TAG] > profiles/<round>_pmc_traffic.json

--workload TAG records which bench.py workload the passes ran (bench.py prints the tag it
looks for as config.workload_tag, e.g. v1_f32_1024x1024): bench.py only quotes traffic
from a summary of its own workload.

--by-grid keeps launches of one kernel with different grid sizes apart (key "name @ grid
size in threads"): for runs that launch a kernel at several problem sizes.

Both counters are reported in KiB.  On gfx950 FETCH_SIZE counts 128-byte read requests
as 64 bytes, i.e. half of the bytes of wide (16 B per lane) coalesced reads, which is
what the chain kernels issue; `fetch_bytes` below is doubled accordingly
(`fetch_bytes_raw` is the counter as reported).  WRITE_SIZE needs no correction.
"""
import collections
import csv
import glob
import json
import os
import sys


BY_GRID = False


def per_kernel(directory, counter):
    out = collections.defaultdict(list)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                key = (row["Kernel_Name"], row["Grid_Size"] if BY_GRID else "")
                out[key].append(float(row["Counter_Value"]))
    return out


def short(name):
    name = name.replace("fpn::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    global BY_GRID
    BY_GRID = "--by-grid" in sys.argv[3:]
    workload = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else None
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for key in sorted(set(fetch) | set(write)):
        name, grid = key
        if "fpn::" not in name:
            continue
        f, w = fetch.get(key, []), write.get(key, [])
        fr = 1024.0 * sum(f) / max(len(f), 1)
        wr = 1024.0 * sum(w) / max(len(w), 1)
        kernels[short(name) + (" @ " + grid if BY_GRID else "")] = {"launches_sampled": max(len(f), len(w)), "fetch_bytes_raw": fr,
                                "fetch_bytes": 2.0 * fr, "write_bytes": wr, "hbm_bytes_per_launch": 2.0 * fr + wr}
    json.dump({"unit": "bytes per launch", "workload": workload, "fetch_correction": "x2 (gfx950, 16 B/lane coalesced reads)",
               "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
