#!/bin/bash
# Phase-folded decoder kernel diagnostics, run ON the GPU box from the repo root: in-kernel stamps of the four decoder conv1
# launches (libv2ce_hip_upstamp.so) and SQ counters per conv instantiation of the e2e step.  tools/up_probe.sh <tag>
TAG=${1:-r05_up}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
V2CE_HIP_LIB=$PWD/v2ce-toolbox_amd/csrc/libv2ce_hip_upstamp.so N=1 python3 tools/fwd_only.py 2> $OUT/stamps.txt | tail -1
grep "stamp up" $OUT/stamps.txt | tail -4
if [ -n "${COUNTERS:-}" ]; then
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_s$i -- python3 tools/fwd_only.py > $OUT/pmc_s$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("$OUT/pmc_s*")):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            m = re.search(r"(conv3d_\w+<[^>]*>)", n)
            if m: acc[m.group(1).replace(" ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    mb = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(c.get("GRBM_GUI_ACTIVE", 1) / 8, 1)
    bc = c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)
    wa = c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{k:42s} mfma_busy {mb:.3f}  lds_conflict/active {bc:.3f}  lds_active/busy {c.get('SQ_LDS_IDX_ACTIVE',0)/max(c.get('SQ_BUSY_CYCLES',1),1):.3f} wait_any {wa:.3f} gui_active {c.get('GRBM_GUI_ACTIVE',0):.0f}")
PY
fi
