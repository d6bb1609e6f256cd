#!/bin/bash
# dense tile kernel: 512-thread (two workgroups per CU) against 1024-thread form at mid densities (GPU box)
export TMPDIR=/tmp
mkdir -p gpurun_out/nw
for nw in 8 16; do
  export V2CE_LDATI_DENSE_NW=$nw
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nw/$nw -- python3 tools/ldati_density_probe.py > gpurun_out/nw/$nw.log 2>&1
  grep scale gpurun_out/nw/$nw.log
  f=$(ls gpurun_out/nw/$nw/*/*kernel_stats.csv | head -1)
  grep dense_kernel $f | cut -d, -f1-4 | cut -c1-160
done
