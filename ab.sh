cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
python bench.py --steps 8 --warmup 2 > gpurun_out/r01c_bench_e2e.json 2>/dev/null
python bench.py --workload ldati_stress --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r01c_bench_ldati_stress.json 2>/dev/null
python bench.py --workload ldati_sparse --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r01c_bench_ldati_sparse.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01c_prof_e2e -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01c_prof_stress -- python3 $R/bench.py --workload ldati_stress --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r01c_pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; done
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r01c_pmcs_$c -- python3 $R/bench.py --workload ldati_stress --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; done
ls $R/gpurun_out | grep r01c
