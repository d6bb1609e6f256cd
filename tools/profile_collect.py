#!/usr/bin/env python3
"""Copy what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/ (tracked):
bench lines, rocprofv3 kernel-stats CSVs and the PMC traffic summaries (tools/pmc_summary.py).

    python tools/profile_collect.py r02_a
"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
for f in glob.glob(os.path.join(src, "bench_*.json")):
    if os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, f"{tag}_{os.path.basename(f)}"))
for wl in ("e2e", "ldati_stress", "ldati_sparse"):
    st = glob.glob(os.path.join(src, f"kt_{wl}", "*", "*kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(dst, f"{tag}_{wl}_kernel_stats.csv"))
    fe, wr = os.path.join(src, f"pmc_{wl}_FETCH_SIZE"), os.path.join(src, f"pmc_{wl}_WRITE_SIZE")
    if glob.glob(fe + "/*/*counter_collection.csv") and glob.glob(wr + "/*/*counter_collection.csv"):
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), fe, wr,
                        os.path.join(dst, f"{tag}_{wl}_pmc_traffic.json")], check=True)
print(sorted(os.listdir(dst)))
