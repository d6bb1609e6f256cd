for lib in libv2ce_hip_base.so libv2ce_hip_p1.so libv2ce_hip_p2.so; do
echo "== $lib"
export V2CE_HIP_LIB=v2ce-toolbox_amd/csrc/$lib
PRECISION=f16x2 FUSE=pred TRACK=1 python tools/conv_bench.py dec3.conv2 2>&1 | grep -v amdgpu.ids
PRECISION=f16x2 FUSE=sc TRACK=1 python tools/conv_bench.py dec3.conv1 enc0.conv1 enc1.conv1 enc3.conv1 2>&1 | grep -v amdgpu.ids
PRECISION=f16x2 TRACK=1 python tools/conv_bench.py dec1.down enc1.down 2>&1 | grep -v amdgpu.ids
done
