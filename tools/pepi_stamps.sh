#!/bin/bash
# in-kernel timing of the 32-channel conv + fused head with its epilogue on the producer waves (V2CE_PEPI) and on the consumers
for lib in ${LIBS:-cstamp}; do
export V2CE_HIP_LIB=v2ce-toolbox_amd/csrc/libv2ce_hip_$lib.so
for pepi in ${PEPIS:-1 0}; do
  echo "== $lib V2CE_PEPI=$pepi"
  V2CE_PEPI=$pepi PRECISION=f16x2 FUSE=pred RES=1 TRACK=1 python tools/conv_bench.py dec3.conv2 2>&1 | grep "pepi\|^dec3" | tail -3 | cut -c1-340
done
done
