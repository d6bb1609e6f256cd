#!/bin/bash
# Round profile set, run ON the GPU box from the repo root:  bash tools/profile_round.sh r02_a
# Writes gpurun_out/<tag>/...; copy the summaries into profiles/ afterwards (tools/profile_collect.py).
set -u
TAG=${1:-r02_a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 8 --warmup 2 > $OUT/bench_e2e.json 2> $OUT/bench_e2e.err
for wl in ldati_stress ldati_sparse; do
  python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
done
python3 bench.py --workload pano --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_pano.json 2> $OUT/bench_pano.err
LIGHT="--no-cpu-baseline --no-exact-f32 --no-host-to-host"
for wl in e2e ldati_stress; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$wl -- python3 bench.py --workload $wl --steps 3 --warmup 1 $LIGHT > $OUT/kt_$wl.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${wl}_$c -- python3 bench.py --workload $wl --steps 2 --warmup 3 $LIGHT > $OUT/pmc_${wl}_$c.log 2>&1
  done
done
ls -R $OUT | head -50
