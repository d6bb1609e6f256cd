#!/bin/bash
# sort workgroup size A/B over the bench workloads (GPU box):  bash tools/sort_threads_ab.sh <tag>
TAG=${1:-st}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for wl in ${WLS:-e2e ldati_stress pano ldati_sparse}; do
  for st in ${STS:-128 256}; do
    export V2CE_LDATI_SORT_THREADS=$st
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${wl}_$st -- python3 bench.py --workload $wl --steps 10 --warmup 4 --no-cpu-baseline --no-exact-f32 --no-host-to-host > $OUT/${wl}_$st.log 2>&1
    f=$(ls $OUT/${wl}_$st/*/*kernel_stats.csv | head -1)
    python3 - "$f" "$wl" "$st" "$OUT/${wl}_$st.log" <<'PY'
import csv, sys, json
ms = None
for l in open(sys.argv[4]):
    if l.startswith('{'):
        ms = json.loads(l)['ms_per_step']
out = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'bucket_sort' in n or 'bucket_scan' in n:
        n = n.replace('void ', '').replace('v2ce::(anonymous namespace)::', '').split('(')[0]
        out.append(f"{n} {float(r['AverageNs']) / 1e3:.1f} us")
print(sys.argv[2], 'threads', sys.argv[3], 'ms/step', ms, '|', ' | '.join(out))
PY
  done
done
