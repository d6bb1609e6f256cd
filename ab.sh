cd $GRAFT_REPO_ROOT
for rep in 1 2; do for t in A D E; do
V2CE_HIP_LIB=$PWD/v2ce-toolbox_amd/csrc/libv2ce_hip_$t.so timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$t', round(d['value'],1), round(d['roofline']['all_conv_tflops'],1), {k.replace('conv3d_kernel',''):round(v['tflops'],1) for k,v in list(d['kernels'].items())[:5]})"
done; done
