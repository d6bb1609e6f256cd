#!/bin/bash
# after a small kernel change: LDATI + head tests, then kernel stats of the e2e and stress benches (compare with the last committed ones)
TAG=${1:-micro}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ldati.py tests/test_gpu_upfold.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -4
for wl in e2e ldati_stress; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$wl -- python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-host-to-host > $OUT/$wl.log 2>&1
  grep "^{" $OUT/$wl.log | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['config']['workload'][:30], 'ms/step', round(j['ms_per_step'],4), 'ldati', (j.get('ldati') or {}).get('avg_ms'), (j.get('ldati') or {}).get('count_ms'))"
  f=$(ls $OUT/$wl/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if ('ldati' in n or 'absmax' in n or 'fillBuffer' in n or 'copyBuffer' in n) and 'check' not in n and 'probe' not in n and 'slope_tab' not in n:
        n = n.replace('void ', '').replace('v2ce::(anonymous namespace)::', '').split('(')[0]
        print(f"   {n:44s} {r['Calls']:>4s} {float(r['AverageNs']) / 1e3:9.1f} us")
PY
done
