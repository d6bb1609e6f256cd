#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes, MI355X_MICROARCH.md
"HBM" section) into per-kernel average bytes per launch.

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE out.json

gfx950 correction: FETCH_SIZE counts 128-byte requests as 64 bytes for coalesced streaming reads, so
the fetch side is doubled.  Calibration in this repo's own access pattern: ldati_count_kernel reads
exactly 80 B x H x W x frames with one dword per lane; its raw FETCH_SIZE is 0.547 of that (x2 =
1.09), so the x2 correction applies to dword-per-lane streams as well.  WRITE_SIZE is used as is.
Counter values are in KiB.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("v2ce::(anonymous namespace)::", ""))
        per[name.replace(" ", "")].append(float(r["Counter_Value"]) * 1024.0)
    return per


def main():
    fe, wr = load(sys.argv[1]), load(sys.argv[2])
    out = {}
    for k in sorted(fe):
        if not (k.startswith("conv3d") or k.startswith("ldati") or k.startswith("sn_") or k.startswith("events") or k.startswith("pack")):
            continue
        n = len(fe[k])
        f = sum(fe[k]) / n
        w = sum(wr.get(k, [0.0])) / max(1, len(wr.get(k, [])))
        out[k] = {"launches": n, "fetch_raw_bytes": f, "fetch_corrected_bytes": 2 * f, "write_bytes": w,
                  "traffic_bytes": 2 * f + w}
    from v2ce_toolbox_amd import hip
    out["_provenance"] = hip.provenance()          # (bench.py ignores a summary whose source hash is not its own tree's)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    prov = out.pop("_provenance")
    print("provenance:", prov)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches"]):
        print(f"{k:45s} n={v['launches']:3d} fetch_x2={v['fetch_corrected_bytes']/1e6:9.1f} MB write={v['write_bytes']/1e6:9.1f} MB")


if __name__ == "__main__":
    main()
