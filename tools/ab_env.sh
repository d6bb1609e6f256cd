#!/bin/bash
# in-network A/B of an environment switch: forward-only timing, alternating.  usage: ab_env.sh VAR=VALUE
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  python tools/fwd_only.py 2>&1 | tail -1
  env "$1" python tools/fwd_only.py 2>&1 | tail -1
done
