cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 1 2 3 4 8; do
  V2CE_LDATI_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl$d -- python3 bench.py --workload ldati_stress --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "dbg=$d: $(find gpurun_out/abl$d -name '*kernel_stats.csv' | head -1 | xargs grep bucket_sort | cut -d, -f1-4 | cut -c60-200)"
done
