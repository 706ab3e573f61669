# one gpurun call: tests, smoke, bench, rocprofv3 kernel stats, PMC traffic, PMC counters of the prefill tile GEMM, decode engine profile
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4j}; mkdir -p $O
# a sick box (seen once: 'Memory access fault by GPU' in the first torch op of every process) must not burn the GPU budget: smoke first, stop if it fails
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
grep -q 'smoke ok' $O/smoke.txt || { echo 'SMOKE FAILED -- stopping'; exit 3; }
if [ -z "$SKIP_TESTS" ]; then
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
# round 5: the PMC pass FIRST, stamped with the kernel-source hash (tools/pmc_traffic.py), then the bench line takes its traffic from THIS pass
(cd /tmp && export TMPDIR=/tmp && GPTQ_BENCH_NO_TP1=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$O/pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --eager --no-decode --no-cpu-baseline --no-per-shape --no-prefill --no-config4 --no-small-batch > $GRAFT_REPO_ROOT/$O/pmc_bench.txt 2>&1)
CC=$(find $O/pmc -name "*counter_collection.csv" | head -1); python tools/pmc_traffic.py "$CC" $O/traffic.json | tail -4
timeout 1200 python bench.py --pmc-file $O/traffic.json > $O/bench.json.txt 2> $O/bench.err; tail -c 400 $O/bench.json.txt
fi
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
GPTQ_BENCH_NO_TP1=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- python $R/bench.py --steps 5 --warmup 1 --no-decode --no-cpu-baseline --no-per-shape --no-prefill --no-config4 --no-small-batch > $R/$O/prof_bench.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof_engine -- python $R/tools/profile_engine.py > $R/$O/prof_engine.txt 2>&1
for B in 4 16; do
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof_engine_b$B -- python $R/tools/profile_engine.py --batch $B > $R/$O/prof_engine_b$B.txt 2>&1
ST=$(find $R/$O/prof_engine_b$B -name "*kernel_stats.csv" | head -1); cp "$ST" $R/$O/decode_engine_b${B}_kernel_stats.csv; rm -rf $R/$O/prof_engine_b$B
done
fi
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof_gemm -- python $R/tools/run_prefill_once.py 16384 > $R/$O/prof_gemm.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -f csv -d $R/$O/pmc_gemm1 -- python $R/tools/run_prefill_once.py 16384 > $R/$O/pmc_gemm1.txt 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace -f csv -d $R/$O/pmc_gemm2 -- python $R/tools/run_prefill_once.py 16384 > $R/$O/pmc_gemm2.txt 2>&1
cd $R
if [ -z "$SKIP_TESTS" ]; then
ST=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$ST" $O/kernel_stats.csv
ST=$(find $O/prof_engine -name "*kernel_stats.csv" | head -1); cp "$ST" $O/decode_engine_kernel_stats.csv
fi
ST=$(find $O/prof_gemm -name "*kernel_stats.csv" | head -1); cp "$ST" $O/prefill_kernel_stats.csv; cut -c1-150 $O/prefill_kernel_stats.csv | head -6
C1=$(find $O/pmc_gemm1 -name "*counter_collection.csv" | head -1); C2=$(find $O/pmc_gemm2 -name "*counter_collection.csv" | head -1); python tools/pmc_gemm.py $O/gemm8_pmc.json "$C1" "$C2" | tail -30
tail -3 $O/pmc_gemm1.txt
rm -rf $O/prof $O/pmc $O/prof_engine $O/prof_gemm $O/pmc_gemm1 $O/pmc_gemm2
