export PRECISION=f16x2 TRACK=1
echo "--- enc2.conv1 sc s2 (out 44x33, 128->256, 128 pos)"; FUSE=sc python tools/tile_probe.py enc2.conv1 16,2,4 16,1,8 8,2,8 4,4,8 8,1,16 4,2,16 4,3,8 2,3,11 4,1,22 8,1,11 2>&1 | grep -v amdgpu
echo "--- enc2.conv2 res (44x33, 256->256)"; FUSE= RES=1 python tools/tile_probe.py enc2.conv2 1,11,23 2,11,11 4,3,11 2,6,11 1,11,22 4,4,11 2>&1 | grep -v amdgpu
echo "--- enc1.conv2 res (87x65 128)"; FUSE= RES=1 python tools/tile_probe.py enc1.conv2 16,2,8 8,4,8 16,4,4 4,4,16 8,2,16 2>&1 | grep -v amdgpu
echo "--- dec2.down 1x1 (173x130, 192->64)"; FUSE= python tools/tile_probe.py dec2.down 8,4,16 4,8,16 4,4,32 16,2,16 2,8,32 1,16,32 2>&1 | grep -v amdgpu
echo "--- dec1.down 1x1 (87x65, 384->128)"; FUSE= python tools/tile_probe.py dec1.down 16,2,8 8,4,8 4,4,16 2,8,16 1,8,32 2>&1 | grep -v amdgpu
