#!/bin/bash
# Epilogue-free upper bound of the conv kernels: the diagnostic library (conv3d.hip built with
# -DV2CE_ABLATE_EPI, see the comment there) skips every ws-kernel epilogue once V2CE_ABLATE_EPI=1 is set in the
# process, on the activations of the last real forward.  Build (in v2ce-toolbox_amd/csrc):
#   hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DV2CE_ABLATE_EPI -c conv3d.hip -o /tmp/conv3d_ablate.o
#   hipcc --offload-arch=gfx950 -shared -fPIC -o libv2ce_hip_ablate.so common.o ldati.o /tmp/conv3d_ablate.o sn.o preproc.o voxelize.o sampler.o
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export V2CE_HIP_LIB=$PWD/v2ce-toolbox_amd/csrc/libv2ce_hip_ablate.so
for i in 1 2; do
  for m in 0 1 2 3; do echo -n "ABLATE=$m  "; ABLATE=$m python tools/fwd_only.py 2>&1 | tail -1; done
done
if [ -n "$PROFILE_MODE" ]; then
  export ABLATE=$PROFILE_MODE
  N=4 rocprofv3 --kernel-trace --stats -d gpurun_out/abl -o abl -- python3 tools/fwd_only.py > gpurun_out/abl.log 2>&1
fi
