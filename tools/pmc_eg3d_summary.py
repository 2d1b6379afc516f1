#!/usr/bin/env python
"""Summarise tools/pmc_eg3d.sh: per tri-plane kernel and per launch, the L2 request / hit / miss counts, the fabric-side
(L2 -> Infinity Cache / HBM) read requests and the FETCH_SIZE / WRITE_SIZE bytes (gfx950 corrections of
MI355X_MICROARCH.md 'HBM': FETCH_SIZE in KiB counts 64 B per 128-B request -> x2).
usage: tools/pmc_eg3d_summary.py <dir with per-counter subdirs> <out.json>"""
import collections
import csv
import glob
import json
import sys

KERNELS = {"dense_query": ("triplane_kernel<1, false>", "largest"),          # 2 097 152 points per launch
           "rays_forward": ("triplane_kernel<", "rays"),                      # FROM_RAYS instances of the renderer
           "backward": ("triplane_backward_kernel", "all")}


def main(root, out):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{root}/*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for name, cs in per.items():
        if "triplane" not in name:
            continue
        key = "dense_query" if ("<1, false>" in name or "<1,false>" in name) else ("backward" if "backward" in name else "rays_forward:" + name.split("(")[0][-24:])
        e = {"kernel": name.split("(")[0], "launches": max(len(v) for v in cs.values())}
        for c, v in cs.items():
            e[c] = sum(v) / len(v)                                              # per-launch average
        if "FETCH_SIZE" in e:
            e["fabric_read_bytes"] = 2 * e["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in e:
            e["fabric_write_bytes"] = e["WRITE_SIZE"] * 1024
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            e["hbm_bytes"] = e.get("fabric_read_bytes", 0) + e.get("fabric_write_bytes", 0)
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
        if "TCC_READ_sum" in e:
            e["l2_request_bytes"] = e["TCC_READ_sum"] * 128                     # one L2 request = one 128-B line
        res[key] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, e in res.items():
        print(k)
        for x, v in sorted(e.items()):
            print(f"   {x:38s} {v if isinstance(v, str) else f'{v:.6g}'}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
