#!/bin/bash
# per-kernel times of the decoder conv1 launches under timing experiments (V2CE_UP_DBG), same box; run on the GPU box
for dbg in ${DBGS:-0 1 2}; do
  echo "== V2CE_UP_DBG=$dbg"
  V2CE_UP_DBG=$dbg python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-host-to-host 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3)); [print('  ', k, round(v['avg_ms'],3)) for k,v in d['kernels'].items() if 'up_' in k]"
done
echo "== V2CE_UPFOLD=0"
V2CE_UPFOLD=0 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-host-to-host 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3)); [print('  ', k, round(v['avg_ms'],3)) for k,v in d['kernels'].items()][:6]"
