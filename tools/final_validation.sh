set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2f}; mkdir -p $O
timeout 700 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json.txt 2> $O/bench.err; tail -c 600 $O/bench.json.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-decode --no-cpu-baseline --no-per-shape --no-prefill --no-config4 --no-small-batch > $GRAFT_REPO_ROOT/$O/prof_bench.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$O/pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --eager --no-decode --no-cpu-baseline --no-per-shape --no-prefill --no-config4 --no-small-batch > $GRAFT_REPO_ROOT/$O/pmc_bench.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/prof_engine -- python $GRAFT_REPO_ROOT/tools/profile_engine.py > $GRAFT_REPO_ROOT/$O/prof_engine.txt 2>&1
cd $GRAFT_REPO_ROOT
ST=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$ST" $O/kernel_stats.csv; head -5 $O/kernel_stats.csv
ST=$(find $O/prof_engine -name "*kernel_stats.csv" | head -1); cp "$ST" $O/decode_engine_kernel_stats.csv; head -12 $O/decode_engine_kernel_stats.csv
CC=$(find $O/pmc -name "*counter_collection.csv" | head -1); python tools/pmc_traffic.py "$CC" $O/traffic.json | tail -5
MS=8,16,32,64,128 timeout 300 python tools/bench_stripe_mm.py 2>&1 | grep -v amdgpu.ids > $O/stripe_mm_4bit.txt
BITS=8 MS=16,64 timeout 200 python tools/bench_stripe_mm.py 2>&1 | grep -v amdgpu.ids > $O/stripe_mm_8bit.txt
for sp in 0 500 1500; do timeout 200 python tools/profile_engine.py --start $sp 2>&1 | grep -o "{.*}" | tail -1; done > $O/engine_context.txt
rm -rf $O/prof $O/pmc $O/prof_engine
