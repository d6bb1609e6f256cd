#!/bin/bash
# decoder conv1 as one launch (phase-folded kernel) or two (upsampled channels folded + skip channels on the Winograd-T kernel); same box
for rep in ${REPS:-1 2}; do
for v in 0 1; do
  echo "== V2CE_UP_SPLIT=$v (rep $rep)"
  V2CE_UP_SPLIT=$v python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-exact-f32 --no-host-to-host 2>/dev/null | python3 -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{\"metric\"')][0]
print(round(d['value']), round(d['ms_per_step'],3), 'executed GF/pair', round(d.get('executed_flop_per_pair',0)/1e9,1))
for k,v in d['kernels'].items():
    if 'up_kernel' in k or 'wt_kernel' in k: print('  ', k[:60], v['launches'], round(v['avg_ms'],3))"
done; done
