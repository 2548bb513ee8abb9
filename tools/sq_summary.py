"""Summarise rocprofv3 --pmc passes of SQ / GRBM counters into per-kernel averages and a
few ratios.

  python tools/sq_summary.py <dir> [<dir> ...] [--by-grid] [--workload TAG] > profiles/<round>_sq.json

Every <dir> is the -d directory of one `rocprofv3 --pmc ... --output-format csv` pass (a
pass holds at most 8 SQ counters; GRBM_GUI_ACTIVE rides along in each).  Per kernel the
counters are averaged over its launches.  Ratios (chip = 256 CUs, 4 SIMDs each, 8 XCDs;
rocprofv3 reports GRBM_GUI_ACTIVE summed over the XCDs, SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* /
SQ_WAIT_* in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles):
  mfma_busy       = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 * 4)
  valu_busy       = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE/8 * 1024)
  lds_busy        = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE/8 * 256)
  lds_conflict    = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wave_active / wave_wait_any / wave_wait_inst = share of SQ_WAVE_CYCLES a wave spends
                    issuing / parked on a counter or barrier / stalled at issue
  valu_per_mfma   = SQ_INSTS_VALU (MFMAs excluded) / SQ_INSTS_MFMA
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("fpn::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--") and os.path.isdir(a)]
    by_grid = "--by-grid" in sys.argv
    workload = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else None
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if "fpn::" not in row["Kernel_Name"]:
                    continue
                key = short(row["Kernel_Name"]) + (" @ " + row["Grid_Size"] if by_grid else "")
                vals[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kernels = {}
    for key in sorted(vals):
        c = {name: sum(v) / len(v) for name, v in vals[key].items()}
        n = max(len(v) for v in vals[key].values())
        g = c.get("GRBM_GUI_ACTIVE")
        r = {}
        if g:
            cyc = g / 8.0
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                r["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)
            if "SQ_ACTIVE_INST_VALU" in c:
                r["valu_busy"] = 4 * c["SQ_ACTIVE_INST_VALU"] / (cyc * 1024)
            if "SQ_LDS_IDX_ACTIVE" in c:
                r["lds_busy"] = c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            r["lds_conflict"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        w = c.get("SQ_WAVE_CYCLES")
        if w:
            for out, name in (("wave_active", "SQ_ACTIVE_INST_ANY"), ("wave_wait_any", "SQ_WAIT_ANY"),
                              ("wave_wait_inst", "SQ_WAIT_INST_ANY"), ("wave_wait_inst_lds", "SQ_WAIT_INST_LDS")):
                if name in c:
                    r[out] = c[name] / w
        if c.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in c:
            r["valu_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
        kernels[key] = {"launches_sampled": n, "counters": c, "ratios": r}
    json.dump({"unit": "counter value per launch (average)", "workload": workload, "kernels": kernels}, sys.stdout,
              indent=1)
    print()


if __name__ == "__main__":
    main()
