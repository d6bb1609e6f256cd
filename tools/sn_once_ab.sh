#!/bin/bash
# spectral-norm planes packed once (round 6) against the re-pack of every forward: UNet parity tests, then the e2e bench both ways
# on one box.   bash tools/sn_once_ab.sh <tag>
TAG=${1:-sn_once}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_weights.py tests/test_gpu_fullsize.py tests/test_gpu_upfold.py tests/test_gpu_winograd.py -x -q 2>&1 | tail -6
for mode in once repack once repack; do
  if [ $mode = repack ]; then export V2CE_SN_REPACK=1; else unset V2CE_SN_REPACK; fi
  timeout 600 python3 bench.py --workload e2e --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-host-to-host > $OUT/e2e_$mode.log 2>&1
  grep "^{" $OUT/e2e_$mode.log | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$mode', 'ms/step', round(j['ms_per_step'],4), 'value', round(j['value'],1), 'algo flop/pair', j.get('algorithmic_flop_per_pair'))"
done
unset V2CE_SN_REPACK
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_once -- python3 bench.py --workload e2e --steps 10 --warmup 3 --no-cpu-baseline --no-exact-f32 --no-host-to-host > $OUT/prof_once.log 2>&1
f=$(ls $OUT/prof_once/*/*kernel_stats.csv | head -1)
grep -i "sn_batch\|pack\|fold" $f | cut -d, -f1-4 | sed 's/v2ce::(anonymous namespace):://g' | cut -c1-150
