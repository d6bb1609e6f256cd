#!/bin/bash
# SQ / GRBM counter set of a round, run ON the GPU box from the repo root:  bash tools/profile_counters.sh r04_a
#   LDATI (stress chunk and the e2e regime): instruction counts, VALU / LDS activity, LDS stalls and bank conflicts
#   stage 1 (e2e step): SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE per conv instantiation
# Counters are collected in their own passes (--pmc with --kernel-trace only; the program goes directly behind `--`);
# tools/counters_summary.py turns gpurun_out/<tag>/sq_* into the text files copied to profiles/.
set -u
TAG=${1:-r04_a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
LIGHT="--no-cpu-baseline --no-exact-f32 --no-host-to-host"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  for wl in ldati_stress ${LDATI_E2E:+e2e}; do
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq_${wl}_s$i -- python3 bench.py --workload $wl --steps 2 --warmup 1 $LIGHT > $OUT/sq_${wl}_s$i.log 2>&1
  done
done
# MFMA busy per conv instantiation (the e2e step)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq_e2e_mfma -- python3 bench.py --workload e2e --steps 2 --warmup 1 $LIGHT > $OUT/sq_e2e_mfma.log 2>&1
python3 tools/counters_summary.py $OUT > $OUT/counters_summary.txt 2>&1
tail -60 $OUT/counters_summary.txt
