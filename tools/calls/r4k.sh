# round 4, call: 3-bit decode through group-of-four dwordx4 loads (G4) -- parity + config-4 timings; profile passes of the 7B pass without the tp1 leg
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "3bit or config4 or three or bits3 or act_order_and_3bit or 8224-3 or stripe16 or widths" 2>&1 | tail -4
timeout 300 python tools/bench_config4.py 2>&1 | grep -v amdgpu.ids > $O/config4.txt; cat $O/config4.txt | cut -c1-250
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
GPTQ_BENCH_NO_TP1=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- python $R/bench.py --steps 5 --warmup 1 --no-decode --no-cpu-baseline --no-per-shape --no-prefill --no-config4 --no-small-batch > $R/$O/prof_bench.txt 2>&1
GPTQ_BENCH_NO_TP1=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $R/$O/pmc -- python $R/bench.py --steps 2 --warmup 1 --eager --no-decode --no-cpu-baseline --no-per-shape --no-prefill --no-config4 --no-small-batch > $R/$O/pmc_bench.txt 2>&1
cd $R
ST=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$ST" $O/kernel_stats.csv
CC=$(find $O/pmc -name "*counter_collection.csv" | head -1); python tools/pmc_traffic.py "$CC" $O/traffic.json | tail -4
grep stripe_gemv $O/kernel_stats.csv | cut -c1-160
tail -2 $O/prof_bench.txt | cut -c1-300
rm -rf $O/prof $O/pmc
