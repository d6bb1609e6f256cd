cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do
timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['all_conv_tflops'],1), {k.replace('conv3d_kernel',''):round(v['tflops'],1) for k,v in list(d['kernels'].items())}, round(d['ldati']['avg_ms'],2), d['roofline']['traffic'])"
done
