#!/bin/bash
# key span of the small-group sort regime, A/B over the sparse workloads (GPU box):  bash tools/span_ab.sh <tag>
TAG=${1:-span}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for wl in ${WLS:-e2e ldati_sparse}; do
  for sp in ${SPS:-128 256 512}; do
    export V2CE_LDATI_SPAN_KEYS=$sp
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${wl}_$sp -- python3 bench.py --workload $wl --steps 10 --warmup 4 --no-cpu-baseline --no-exact-f32 --no-host-to-host > $OUT/${wl}_$sp.log 2>&1
    f=$(ls $OUT/${wl}_$sp/*/*kernel_stats.csv | head -1)
    python3 - "$f" "$wl" "$sp" "$OUT/${wl}_$sp.log" <<'PY'
import csv, sys, json
ms = None
for l in open(sys.argv[4]):
    if l.startswith('{'):
        j = json.loads(l); ms = j['ms_per_step']; ld = j.get('ldati', {}).get('avg_ms')
out = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'bucket_sort' in n or 'bucket_scan' in n:
        n = n.replace('void ', '').replace('v2ce::(anonymous namespace)::', '').split('(')[0]
        out.append(f"{n} {float(r['AverageNs']) / 1e3:.1f} us")
print(sys.argv[2], 'span', sys.argv[3], 'ms/step', round(ms, 4), 'ldati ms', ld, '|', ' | '.join(out))
PY
  done
done
