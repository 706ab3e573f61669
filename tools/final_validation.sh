set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2f}; mkdir -p $O
timeout 700 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 600 python bench.py > $O/bench.json.txt 2> $O/bench.err; tail -c 600 $O/bench.json.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-decode --no-cpu-baseline --no-per-shape --no-prefill --no-config4 > $GRAFT_REPO_ROOT/$O/prof_bench.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$O/pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --eager --no-decode --no-cpu-baseline --no-per-shape --no-prefill --no-config4 > $GRAFT_REPO_ROOT/$O/pmc_bench.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/prof_engine -- python $GRAFT_REPO_ROOT/tools/profile_engine.py > $GRAFT_REPO_ROOT/$O/prof_engine.txt 2>&1
cd $GRAFT_REPO_ROOT
ST=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$ST" $O/kernel_stats.csv; head -5 $O/kernel_stats.csv
ST=$(find $O/prof_engine -name "*kernel_stats.csv" | head -1); cp "$ST" $O/decode_engine_kernel_stats.csv; head -12 $O/decode_engine_kernel_stats.csv
CC=$(find $O/pmc -name "*counter_collection.csv" | head -1); python tools/pmc_traffic.py "$CC" $O/traffic.json | tail -5
rm -rf $O/prof $O/pmc $O/prof_engine
