# round 4, call 4: full GPU suite after the engine/module image sharing + new full-size parity tests; engine memory
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
timeout 600 python bench.py --no-per-shape --no-prefill --no-config4 --no-small-batch --no-cpu-baseline --steps 10 2>&1 | grep -v amdgpu.ids > $O/bench_decode.txt; python - <<'PY'
import json
t=open('gpurun_out/r4d/bench_decode.txt').read()
if '{"metric"' in t:
    j=json.loads(t[t.index('{"metric"'):].splitlines()[0])
    print(j['value'], j['roofline']['frac'])
    for k,v in j.get('decode',{}).items():
        if isinstance(v,dict): print(k, v.get('tokens_per_s'), v.get('max_memory_MiB'))
    print(j.get('tp1_same_workload_as_gpus_gt_1'))
else: print(t[-2000:])
PY
