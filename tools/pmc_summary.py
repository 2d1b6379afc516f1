#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc*/<COUNTER>/*/..counter_collection.csv) into
profiles/<name>.json: per kernel the max-launch (fine pass) value of each counter, HBM traffic with the gfx950
correction (FETCH_SIZE counts 64 B per 128-B request -> x2; units KiB; MI355X_MICROARCH.md 'HBM'), MFMA-busy
fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs).
usage: tools/pmc_summary.py <dir with per-counter subdirs> <out.json>"""
import collections
import csv
import glob
import json
import sys


def main(root, out):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{root}/*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "nerfmi" not in k:
                continue
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in per.items():
        e = {c: max(v) for c, v in cs.items()}       # the fine pass (131 072 points) is the largest launch
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            e["hbm_read_bytes"] = 2 * e.get("FETCH_SIZE", 0) * 1024
            e["hbm_write_bytes"] = e.get("WRITE_SIZE", 0) * 1024
            e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
            e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if "SQ_WAIT_ANY" in e and "SQ_WAVE_CYCLES" in e:
            e["wait_any_frac"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
            e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
        res[k] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, e in res.items():
        print(k[:70], {x: (round(v, 4) if v < 10 else f"{v:.4g}") for x, v in e.items() if x in
                       ("hbm_bytes", "mfma_busy_frac", "wait_any_frac", "l2_hit_rate")})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
