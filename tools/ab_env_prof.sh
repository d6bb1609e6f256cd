#!/bin/bash
# per-kernel A/B of an environment switch (rocprofv3 kernel stats of tools/fwd_only.py).  usage: ab_env_prof.sh VAR=VALUE
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
N=6 rocprofv3 --kernel-trace --stats -d gpurun_out/abp_a -o a -- python3 tools/fwd_only.py > gpurun_out/abp_a.log 2>&1
export "$1"
N=6 rocprofv3 --kernel-trace --stats -d gpurun_out/abp_b -o b -- python3 tools/fwd_only.py > gpurun_out/abp_b.log 2>&1
