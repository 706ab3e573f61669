# round 5, call 11: producer-side permutation for act-order checkpoints (tests, engine A/B) + the counters behind the small-batch cells
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_batch.py tests/test_gpu_parity.py -q -k "act_order or attention or engine" 2>&1 | tail -8 > $O/pytest.txt; tail -4 $O/pytest.txt
for PP in 1 0; do
GPTQ_PRODUCER_PERM=$PP timeout 400 python tools/bench_engine_act_order.py 2>/dev/null | sed "s/^/producer_perm=$PP /" | tee -a $O/engine_act_order.txt
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5k
for M in 64 32; do
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -f csv -d $O/pmc_a_$M -- python $R/tools/run_small_batch_once.py 4096 12288 $M 1 > $O/pmc_a_$M.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -f csv -d $O/pmc_b_$M -- python $R/tools/run_small_batch_once.py 4096 12288 $M 1 > $O/pmc_b_$M.txt 2>&1
python - $O/pmc_a_$M $O/pmc_b_$M $M <<'PY' | tee -a $O/small_batch_counters.txt
import csv, sys, collections, glob
acc = collections.defaultdict(list)
for d in sys.argv[1:3]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'stripe_mm' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
avg = {c: sum(v) / len(v) for c, v in acc.items()}
M = int(sys.argv[3])
out = {'shape': '4096x12288', 'M': M, 'kernel': 'round-4 schedule (stripe_mm3_kernel), per launch', 'launches': len(next(iter(acc.values()))) if acc else 0}
out.update({c: round(v) for c, v in avg.items()})
if 'TCC_HIT_sum' in avg: out['L2_hit_rate'] = round(avg['TCC_HIT_sum'] / max(avg['TCC_HIT_sum'] + avg['TCC_MISS_sum'], 1), 4)
if 'TCP_TCC_READ_REQ_sum' in avg: out['L2_read_requests_x_64B_MB'] = round(avg['TCP_TCC_READ_REQ_sum'] * 64 / 1e6, 1)
if 'TCC_EA0_RDREQ_sum' in avg: out['HBM_read_requests'] = round(avg['TCC_EA0_RDREQ_sum'])
print(out)
PY
rm -rf $O/pmc_a_$M $O/pmc_b_$M
done
