set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 > gpurun_out/r6_bench_steps20_a.json 2> gpurun_out/r6_bench_steps20_a.err; tail -3 gpurun_out/r6_bench_steps20_a.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench_steps20_a.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'])
c=d.get('config3',{})
print('config3',c.get('value'),c.get('ms_per_step'),c.get('other_guided_split_settings'))
for r in c.get('roofline_table') or []: print(r)
for k in ('batch1','batch2','bf16','config5','config4','vae_decode'):
    v=d.get(k); print(k, {a:b for a,b in v.items() if not isinstance(b,(list,dict))} if isinstance(v,dict) else v)
PY
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r6_pytest_gpu_a.txt; cat gpurun_out/r6_pytest_gpu_a.txt
