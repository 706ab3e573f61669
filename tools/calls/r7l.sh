#!/bin/bash
# rocprofv3 kernel stats + counters of the loader / consumer kernel at the final source: 4096 x 12288 at 64 / 128 rows, 11008 x 4096 at 64 rows (K slices + combine)
O=gpurun_out/r7l; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for cfg in "4096 12288 64" "4096 12288 128" "11008 4096 64"; do set -- $cfg; tag=${1}x${2}_M$3
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/st_$tag -- python $R/tools/run_small_batch_once.py $1 $2 $3 3 > $R/$O/st_$tag.txt 2>&1
ST=$(find $R/$O/st_$tag -name "*kernel_stats.csv" | head -1); grep -i "mmr\|Name" "$ST" | cut -c1-260 > $R/$O/kernel_stats_$tag.csv
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum SQ_INSTS_MFMA --kernel-trace -f csv -d $R/$O/pa_$tag -- python $R/tools/run_small_batch_once.py $1 $2 $3 1 > $R/$O/pa_$tag.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -f csv -d $R/$O/pb_$tag -- python $R/tools/run_small_batch_once.py $1 $2 $3 1 > $R/$O/pb_$tag.txt 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
out = {}
for d in glob.glob('gpurun_out/r7l/p[ab]_*'):
    if not d.endswith('.txt'):
        tag = d.split('/')[-1][3:]
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            acc = collections.defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(f)):
                if 'mmr' in r['Kernel_Name']:
                    key = ('combine:' if 'combine' in r['Kernel_Name'] else '') + r['Counter_Name']
                    acc[key][0] += float(r['Counter_Value']); acc[key][1] += 1
            for k, (v, n) in acc.items():
                out.setdefault(tag, {})[k] = round(v / max(n, 1), 1)
open('gpurun_out/r7l/mmr_counters_per_launch.json', 'w').write(json.dumps(out, indent=1, sort_keys=True))
print(json.dumps(out, sort_keys=True)[:1500])
PY
cat $O/kernel_stats_*.csv | cut -c1-200
rm -rf $O/st_*/ $O/pa_*/ $O/pb_*/
