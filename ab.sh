cd $GRAFT_REPO_ROOT
for t in 2048 1024 512 256 128; do for w in ldati_stress ldati_sparse e2e; do V2CE_BUCKET=$t timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bucket=$t $w', round(d['ms_per_step'],2), round(d['ldati']['avg_ms'],2))"; done; done
