#!/bin/bash
# same-box A/B of two builds of the library on the e2e bench:  bash tools/lib_ab.sh libA.so libB.so [rounds]
export TMPDIR=/tmp
for i in $(seq 1 ${3:-2}); do
  for lib in $1 $2; do
    V2CE_HIP_LIB=v2ce-toolbox_amd/csrc/$lib python3 bench.py --workload e2e --steps 10 --warmup 3 --no-cpu-baseline --no-exact-f32 --no-host-to-host 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$lib', 'ms/step', round(j['ms_per_step'],3), 'stage1', round(j['stage1_mfma_frac_e2e'],4))"
  done
done
