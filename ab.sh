cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_unet.py -m gpu -q -x -k "split" 2>&1 | tail -12
for pr in f32 f16x2; do echo "== $pr"; PRECISION=$pr timeout 200 python tools/conv_bench.py enc0.conv2 res0.conv1 dec1.conv1 dec3.conv1 enc1.conv1 2>&1 | grep -v amdgpu.ids; done
