#!/bin/bash
# SQ counters of one conv layer (tools/conv_bench.py) -> gpurun_out/pmc_conv/<set>/...; run on the GPU box
export TMPDIR=/tmp PRECISION=f16x2 TRACK=1
LAYER=${1:-dec3.conv1}
export FUSE=${2-sc}
i=0; mkdir -p gpurun_out/pmc_conv
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_conv/$LAYER/s$i -- python3 tools/conv_bench.py $LAYER > gpurun_out/pmc_conv/$LAYER.s$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_conv/$LAYER/s*")):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f: print(d, "no counters"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "conv3d_f16x2_ws" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"$LAYER {k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
