#!/bin/bash
# the counter passes of the ldati_sparse workload (PMC traffic + SQ), same recipe as profile_round.sh / profile_counters.sh:
#   bash tools/profile_sparse.sh r06_z      (ON the GPU box; then tools/profile_collect.py and the copies as for the other workloads)
set -u
TAG=${1:-r06_z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
LIGHT="--no-cpu-baseline --no-exact-f32 --no-host-to-host"
wl=ldati_sparse
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$wl -- python3 bench.py --workload $wl --steps 3 --warmup 1 $LIGHT > $OUT/kt_$wl.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${wl}_$c -- python3 bench.py --workload $wl --steps 2 --warmup 3 $LIGHT > $OUT/pmc_${wl}_$c.log 2>&1
done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq_${wl}_s$i -- python3 bench.py --workload $wl --steps 2 --warmup 1 $LIGHT > $OUT/sq_${wl}_s$i.log 2>&1
done
python3 tools/counters_summary.py $OUT > $OUT/counters_summary_sparse.txt 2>&1
tail -8 $OUT/counters_summary_sparse.txt
