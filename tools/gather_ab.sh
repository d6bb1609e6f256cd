#!/bin/bash
# The three ways of bringing the records to host memory, on ONE GPU with the collective path forced (RCCL world of one), run on
# the GPU box:  device gather + rank 0's download | host segment, staging + pwrite | host segment, registered (GPU DMA)
export V2CE_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for mode in "device" "host 0" "host 1"; do
  set -- $mode
  export V2CE_GATHER=$1
  if [ "${2:-}" = "0" ]; then export V2CE_HOST_SEGMENT_MB=0; else unset V2CE_HOST_SEGMENT_MB; fi
  python3 bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-exact-f32 --no-host-to-host 2>/dev/null | MODE="$mode" python3 tools/gather_ab_parse.py
done
