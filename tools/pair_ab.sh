#!/bin/bash
# pair-pass dense kernel (round 6) against the per-bin kernel on one GPU box: parity tests, then the tile-pass kernel's
# duration at three densities and the stress / e2e benches with and without it.   bash tools/pair_ab.sh <tag>
TAG=${1:-pair}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ldati.py -x -q -k "pair_pass or dense_tile_kernel_equals or fused_dense or c5_stress or dense_slot" 2>&1 | tail -5
for mode in ${MODES:-onepass pair nopair}; do
  unset V2CE_LDATI_PAIR V2CE_LDATI_ONEPASS
  if [ $mode = pair ]; then export V2CE_LDATI_PAIR=1; fi
  if [ $mode = onepass ]; then export V2CE_LDATI_ONEPASS=1; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dens_$mode -- python3 tools/ldati_density_probe.py > $OUT/dens_$mode.log 2>&1
  grep "^scale" $OUT/dens_$mode.log
  python3 - $OUT/dens_$mode <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'tile_dense' in n or 'tile_pair' in n or 'bucket_sort' in n or 'bucket_scan' in n:
        k = n.split('::')[-1].split('(')[0]
        d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    print('   %-40s' % k, [round(x) for x in v])
PY
  for wl in ldati_stress e2e; do
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${wl}_$mode -- python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-host-to-host > $OUT/${wl}_$mode.log 2>&1
    grep "^{" $OUT/${wl}_$mode.log | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$mode', j['config']['workload'][:30], 'ms/step', round(j['ms_per_step'],4), 'ldati ms', (j.get('ldati') or {}).get('avg_ms'), 'frac', j['roofline'].get('frac'))"
    f=$(ls $OUT/${wl}_$mode/*/*kernel_stats.csv | head -1)
    python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'ldati' in n and 'check' not in n and 'probe' not in n and 'slope_tab' not in n:
        n = n.replace('void ', '').replace('v2ce::(anonymous namespace)::', '').split('(')[0]
        print(f"   {n:44s} {r['Calls']:>4s} {float(r['AverageNs']) / 1e3:9.1f} us")
PY
  done
done
# phase stamps of the tile pass, both kernels (diagnostic library)
for mode in ${MODES:-onepass pair nopair}; do
  unset V2CE_LDATI_PAIR V2CE_LDATI_ONEPASS
  if [ $mode = pair ]; then export V2CE_LDATI_PAIR=1; fi
  if [ $mode = onepass ]; then export V2CE_LDATI_ONEPASS=1; fi
  echo "stamps $mode"
  V2CE_HIP_LIB=v2ce-toolbox_amd/csrc/libv2ce_hip_stamp.so timeout 300 python3 tools/ldati_stamps.py stress 2>&1 | head -12
done
