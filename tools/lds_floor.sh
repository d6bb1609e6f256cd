#!/bin/bash
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of random vs consecutive LDS atomics / writes (tools/micro/lds_conflict_floor.hip); run on the GPU box
R=$PWD
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_conflict_floor $R/tools/micro/lds_conflict_floor.hip 2>/dev/null
cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/r05_lds_floor -- /tmp/lds_conflict_floor > $R/gpurun_out/r05_lds_floor.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/r05_lds_floor/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]] = float(r["Counter_Value"])
for k, c in sorted(acc.items()):
    print(k[:40], {n: int(v) for n, v in c.items()}, "conflict/active = %.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
