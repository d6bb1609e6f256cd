#!/bin/bash
# Timing ablations of the consumers' tap loops: stamp builds of the direct and the Winograd-T kernel in which
#   1 = the weight (A) fragment ring is never refilled, 2 = the B fragments are never refilled, 3 = both,
#   4 = (Winograd-T) the producers issue a quarter of their gather loads.        WRONG results by construction -- never a product build.
# Build here (no GPU needed), run on the GPU box:   bash tools/tap_ablate.sh build ;  gpurun -- bash tools/tap_ablate.sh run
cd "$(dirname "$0")/../v2ce-toolbox_amd/csrc" || exit 1
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DV2CE_STAMP"
OTHERS="common.o ldati.o conv3d_up.o conv3d_head.o sn.o preproc.o voxelize.o sampler.o"
if [ "$1" = build ]; then
  make libv2ce_hip.so > /dev/null
  for n in 0 1 2 3; do
    $HC -DV2CE_ABLATE_TAPS=$n -c conv3d.hip -o /tmp/abl_c_$n.o && $HC -DV2CE_ABLATE_TAPS=$n -c conv3d_wt.hip -o /tmp/abl_w_$n.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libv2ce_hip_abltaps$n.so $OTHERS /tmp/abl_c_$n.o /tmp/abl_w_$n.o
  done
  $HC -DV2CE_ABLATE_TAPS=4 -c conv3d_wt.hip -o /tmp/abl_w_4.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libv2ce_hip_abltaps4.so $OTHERS /tmp/abl_c_0.o /tmp/abl_w_4.o
  ls -la libv2ce_hip_abltaps*.so
  exit 0
fi
cd ../..
for n in 0 1 2 3 4; do
  export V2CE_HIP_LIB=v2ce-toolbox_amd/csrc/libv2ce_hip_abltaps$n.so
  echo "== ablation $n"
  [ $n -lt 4 ] && PRECISION=f16x2 FUSE=pred RES=1 TRACK=1 python tools/conv_bench.py dec3.conv2 2>&1 | grep "tap loops" | tail -1 | cut -c1-120
  python3 tools/wt_bench.py 5 2>&1 | grep "main loop" | awk "NR%12==1" | sed "s/per workgroup.*main loop/main loop/;s/| producer.*//" | cut -c1-150
done
