for lib in libv2ce_hip_base.so libv2ce_hip.so; do
echo "== $lib"
export V2CE_HIP_LIB=v2ce-toolbox_amd/csrc/$lib
python tools/conv_bench.py dec2.down dec3.down pred enc0.down 2>&1 | grep -v amdgpu.ids
PRECISION=f16x2 python tools/conv_bench.py dec1.down res0.down enc1.down 2>&1 | grep -v amdgpu.ids
done
