# round 4, call 8: full GPU suite + bench after the small-batch work
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
timeout 900 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/bench.txt; python - <<'PY'
import json
t=open('gpurun_out/r4h/bench.txt').read()
if '{"metric"' in t:
    j=json.loads(t[t.index('{"metric"'):].splitlines()[0])
    print('value', j['value'], 'frac', j['roofline']['frac'])
    print(json.dumps(j.get('small_batch_reported_only'))[:2500])
    print(json.dumps(j.get('prompt_reported_only'))[:3500])
    print({k:(v.get('tokens_per_s'), v.get('max_memory_MiB')) for k,v in j.get('decode',{}).items() if isinstance(v,dict)})
else: print(t[-3000:])
PY
