#!/bin/bash
# stripe_mmr_kernel as the default route for 17 .. 64 rows on the wide shapes: the stripe_mm parity file, the times, and its counters at 64 rows
O=gpurun_out/r6t; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "stripe_mm or mid_m or layer_decode or wide_layers" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
python tools/bench_mmr.py 2>&1 | grep GPTQ_MMR > $O/mmr_default.txt; cat $O/mmr_default.txt
cd /tmp; export TMPDIR=/tmp
for M in 32 64; do
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum SQ_INSTS_MFMA --kernel-trace -f csv -d $R/$O/pmc_a_$M -- python $R/tools/run_small_batch_once.py 4096 12288 $M 1 > $R/$O/pmc_a_$M.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -f csv -d $R/$O/pmc_b_$M -- python $R/tools/run_small_batch_once.py 4096 12288 $M 1 > $R/$O/pmc_b_$M.txt 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
out = {}
for M in (32, 64):
    for tag in ('a', 'b'):
        for f in glob.glob('gpurun_out/r6t/pmc_%s_%d/**/*counter_collection.csv' % (tag, M), recursive=True):
            acc = collections.defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(f)):
                if 'stripe_mmr' in r['Kernel_Name']:
                    acc[r['Counter_Name']][0] += float(r['Counter_Value']); acc[r['Counter_Name']][1] += 1
            for k, (v, n) in acc.items():
                out.setdefault('M%d' % M, {})[k] = round(v / max(n, 1), 1)
print(json.dumps(out))
open('gpurun_out/r6t/mmr_counters_per_launch.json', 'w').write(json.dumps(out, indent=1))
PY
rm -rf $O/pmc_a_* $O/pmc_b_*
