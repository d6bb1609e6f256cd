for l in cstamp s1 s2 s3; do
echo "== $l (1 = no residual loads, 2 = no stores, 3 = neither)"
export V2CE_HIP_LIB=v2ce-toolbox_amd/csrc/libv2ce_hip_$l.so
PRECISION=f16x2 FUSE=pred python tools/conv_bench.py dec3.conv2 2>&1 | grep stamp | tail -1 | grep -o "ws<[0-9,]*>\|epilogue [0-9]*\|total [0-9]* |"
PRECISION=f16x2 RES=1 python tools/conv_bench.py enc0.conv2 2>&1 | grep stamp | tail -1 | grep -o "ws<[0-9,]*>\|epilogue [0-9]*\|total [0-9]* |"
PRECISION=f16x2 FUSE=sc python tools/conv_bench.py dec3.conv1 2>&1 | grep stamp | tail -1 | grep -o "ws<[0-9,]*>\|epilogue [0-9]*\|total [0-9]* |"
done
