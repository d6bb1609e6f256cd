for lib in libv2ce_hip_base.so libv2ce_hip.so; do
echo "== $lib"
export V2CE_HIP_LIB=v2ce-toolbox_amd/csrc/$lib
PRECISION=f16x2 TRACK=1 python tools/conv_bench.py dec2.conv1 2>&1 | grep -v amdgpu.ids
PRECISION=f16x2 RES=1 TRACK=1 python tools/conv_bench.py enc0.conv2 2>&1 | grep -v amdgpu.ids
done
