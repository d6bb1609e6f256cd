#!/bin/bash
# timing ablation: conv producers gathering an upsampled source once per source element (WRONG results) against the product
export TMPDIR=/tmp
mkdir -p gpurun_out/ablup
for lib in libv2ce_hip.so libv2ce_hip_ablup.so; do
  export V2CE_HIP_LIB=v2ce-toolbox_amd/csrc/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ablup/$lib -- python3 bench.py --workload e2e --steps 6 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-host-to-host > gpurun_out/ablup/$lib.log 2>&1
  grep "^{" gpurun_out/ablup/$lib.log | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$lib', 'ms/step', round(j['ms_per_step'],3))"
  f=$(ls gpurun_out/ablup/$lib/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'conv3d_f16x2' in n:
        n = n.replace('void ', '').replace('v2ce::(anonymous namespace)::', '').split('(')[0]
        print(f"   {n:50s} {r['Calls']:>4s} {float(r['AverageNs']) / 1e3:9.1f} us")
PY
done
