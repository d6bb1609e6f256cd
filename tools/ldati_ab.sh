#!/bin/bash
# LDATI A/B on the GPU box: bench.py --workload ldati_stress / ldati_sparse under a few settings; one line each.
# usage: tools/ldati_ab.sh "VAR=VAL ..." ["VAR=VAL ..."] ...   (each argument = one environment to try; "" = defaults)
mkdir -p gpurun_out
for envs in "$@"; do
  for wl in ldati_stress ldati_sparse; do
    out=$(env $envs python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1)
    python - "$envs" "$wl" "$out" <<'PY'
import json, sys
envs, wl, out = sys.argv[1:4]
try:
    d = json.loads(out)
    print(f"[{envs or 'default'}] {wl}: {d['ms_per_step']:.3f} ms/step wall, ldati {d['ldati']['avg_ms']:.3f} ms (count {d['ldati']['count_ms']:.3f}), "
          f"frac {d['roofline']['frac']:.3f}, {d['mevents_per_s']:.0f} Mev/s")
except Exception as e:
    print(f"[{envs}] {wl}: FAILED {e}: {out[-300:]}")
PY
  done
done
