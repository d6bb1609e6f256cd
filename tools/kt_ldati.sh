export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_tmp -- python3 bench.py --workload ${1:-e2e} --steps 3 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-host-to-host > gpurun_out/kt_tmp.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/kt_tmp/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'ldati' in n or 'events' in n or 'sn_' in n:
        print(f"{n[:100]:100s} n={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:9.1f}us")
PY
