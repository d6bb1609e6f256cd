#!/bin/bash
# Winograd-T kernel: forced boxes (pairs,rows,cols) per layer shape, same box; run on the GPU box
fmt() { tail -4 | sed -E 's/.*(C= *[0-9]+ [0-9x]+):.*winograd-T ([0-9.]+) \/ \+res ([0-9.]+) ms.*/\1  \2 \/ \3/' | tr '\n' ';'; echo; }
echo -n "chosen:      "; python3 tools/wt_bench.py 8 2>&1 | fmt
for b in ${BOXES:-1,8,16 1,4,32 2,4,16 1,6,21 1,7,18 2,8,8 1,3,42 1,2,64 4,2,16 1,11,11 1,9,14 2,5,12 1,5,25 4,4,8}; do
  echo -n "box $b:  "; V2CE_WT_BOX=$b python3 tools/wt_bench.py 8 2>&1 | fmt
done
