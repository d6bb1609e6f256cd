#!/bin/bash
# per-kernel average durations of one bench.py workload under rocprofv3 (GPU box): tools/kstats.sh ldati_stress [extra bench args]
wl=${1:-ldati_stress}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/kstats_$wl
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 $root/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-host-to-host --no-exact-f32 "$@" > $out/bench.json 2> $out/err.txt
cd $root
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us  x{int(r['Calls']):5d}  {float(r['Percentage']):5.1f}%  {r['Name'][:100]}")
PY
