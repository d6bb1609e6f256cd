#!/bin/bash
# generic split-half kernel: conflict-free LDS pitches / lane order / de-interleaved stride-2 columns on and off, same box,
# interleaved repeats; then the parity tests that cover the generic kernel.  Run on the GPU box.
for rep in ${REPS:-1 2}; do
for tune in 0 1; do
  echo "== V2CE_LDS_TUNE=$tune (rep $rep)"
  V2CE_LDS_TUNE=$tune V2CE_UP_VERBOSE=$([ $rep = 1 ] && echo 1) python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-exact-f32 --no-host-to-host 2>gpurun_out/lds_tune_$tune.err | python3 -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{\"metric\"')][0]
print(round(d['value']), round(d['ms_per_step'],3))
for k,v in d['kernels'].items():
    if 'up_' not in k: print('  ', k[:110], v.get('launches', v.get('n','')), round(v['avg_ms'],3))"
done; done
grep "ws lds geometry" gpurun_out/lds_tune_1.err | sort -u
[ -n "$NOTEST" ] || python3 -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py tests/test_gpu_upfold.py -x -q 2>&1 | tail -5
