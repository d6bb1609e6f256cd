#!/bin/bash
# conv A/B in the network: UNet parity tests, then rocprofv3 kernel stats of the e2e bench (GPU box):  bash tools/conv_ab_e2e.sh <tag> [pytest -k expr]
TAG=${1:-cab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_unet.py -x -q ${2:+-k "$2"} 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e -- python3 bench.py --workload e2e --steps 8 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-host-to-host > $OUT/e2e.log 2>&1
grep "^{" $OUT/e2e.log | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print('ms/step', round(j['ms_per_step'],3), 'frac', round(j['roofline']['frac'],4), 'stage1', round(j['stage1_mfma_frac_e2e'],4))"
f=$(ls $OUT/e2e/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'conv3d' in n:
        n = n.replace('void ', '').replace('v2ce::(anonymous namespace)::', '').split('(')[0]
        tot += float(r['TotalDurationNs'])
        print(f"   {n:50s} {r['Calls']:>4s} {float(r['AverageNs']) / 1e3:9.1f} us")
print('   conv total per step (ms):', tot / 1e6 / 10)
PY
