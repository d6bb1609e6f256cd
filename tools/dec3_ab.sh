#!/bin/bash
# dec3 with its shortcut split by source (round 6) against the fused form: parity tests, then the e2e bench both ways on one box.
TAG=${1:-dec3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_upfold.py tests/test_gpu_unet.py tests/test_gpu_fullsize.py tests/test_gpu_weights.py -x -q 2>&1 | tail -6
for mode in split fused split fused; do
  if [ $mode = fused ]; then export V2CE_DEC3_SPLIT=0; else unset V2CE_DEC3_SPLIT; fi
  timeout 600 python3 bench.py --workload e2e --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-host-to-host > $OUT/e2e_$mode.log 2>&1
  grep "^{" $OUT/e2e_$mode.log | python3 -c "
import sys,json; j=json.loads(sys.stdin.readline()); print('$mode', 'ms/step', round(j['ms_per_step'],4), 'value', round(j['value'],1))
for k,v in j['kernels'].items():
    if 'up_kernel<1,1,4' in k or ',9,' in k or '<1,1,1,1,4,3' in k: print('    %.3f ms x%d  %s' % (v['avg_ms'], v['launches'], k))"
done
