#!/bin/bash
# A/B of a conv switch (round 6): parity tests, then the e2e bench both ways on one box.
#   VAR=V2CE_PEPI (default: the last decoder conv's epilogue shared with the producer waves), V2CE_PEPI_SC, V2CE_G4, V2CE_NA9,
#   V2CE_LDATI_STREAM ...; MODES="1 0 1 0"
TAG=${1:-pepi}
VAR=${VAR:-V2CE_PEPI}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -n "$SKIP_TESTS" ] || timeout 1800 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py tests/test_gpu_upfold.py -x -q 2>&1 | tail -4
for mode in ${MODES:-1 0 1 0}; do
  export $VAR=$mode
  timeout 600 python3 bench.py --workload e2e --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-host-to-host > $OUT/e2e_$mode.log 2>&1
  grep "^{" $OUT/e2e_$mode.log | python3 -c "
import sys,json; j=json.loads(sys.stdin.readline()); print('$VAR=$mode', 'ms/step', round(j['ms_per_step'],4), 'value', round(j['value'],1))
for k,v in j['kernels'].items():
    if ',9,1,' in k or ',3,1,' in k or 'ws_kernel<3,2,' in k or 'up_kernel<1,1,4' in k: print('    %.3f ms x%d  %s' % (v['avg_ms'], v['launches'], k))"
done
