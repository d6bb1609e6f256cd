#!/bin/bash
# A/B of the LDATI kernels on the GPU box: LDATI tests, then rocprofv3 kernel stats of the stress and e2e benches.
#   bash tools/ldati_ab.sh <tag> [pytest-args]
TAG=${1:-ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests/test_gpu_ldati.py -x -q ${2:-} 2>&1 | tail -4
export TMPDIR=/tmp
for wl in ldati_stress e2e; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$wl -- python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-host-to-host > $OUT/$wl.log 2>&1
  grep "^{" $OUT/$wl.log | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['config']['workload'][:30], 'ms/step', round(j['ms_per_step'],4), 'frac', j['roofline'].get('frac'))"
  f=$(ls $OUT/$wl/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'ldati' in n and 'check' not in n and 'probe' not in n and 'slope_tab' not in n:
        n = n.replace('void ', '').replace('v2ce::(anonymous namespace)::', '').split('(')[0]
        print(f"   {n:44s} {r['Calls']:>4s} {float(r['AverageNs']) / 1e3:9.1f} us")
PY
done
