#!/bin/bash
# in-network A/B of two library builds: forward-only timing, alternating (tools/fwd_only.py)
cd $GRAFT_REPO_ROOT
B=$PWD/v2ce-toolbox_amd/csrc/libv2ce_hip_base.so
for i in 1 2 3; do
  V2CE_HIP_LIB=$B python tools/fwd_only.py 2>&1 | tail -1
  python tools/fwd_only.py 2>&1 | tail -1
done
