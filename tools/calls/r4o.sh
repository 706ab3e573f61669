# round 4: evidence refresh -- bench line at HEAD, PMC traffic of the LM-head kernel, engine kernel stats, attention timeline at T = 20
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4o; mkdir -p $O
timeout 900 python bench.py > $O/bench.json.txt 2> $O/bench.err; tail -c 300 $O/bench.json.txt
timeout 200 python tools/timeline_attn.py 20 2>&1 | grep -v amdgpu.ids > $O/attn_timeline.txt; cat $O/attn_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof_engine -- python $R/tools/profile_engine.py > $R/$O/prof_engine.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $R/$O/pmc_lm -- python $R/tools/bench_lm_head.py > $R/$O/pmc_lm.txt 2>&1
cd $R
ST=$(find $O/prof_engine -name "*kernel_stats.csv" | head -1); cp "$ST" $O/decode_engine_kernel_stats.csv
CC=$(find $O/pmc_lm -name "*counter_collection.csv" | head -1); python - "$CC" <<'PY' | tee $O/lm_head_traffic.txt
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(list)
for r in rows:
    if r.get('Counter_Name')=='FETCH_SIZE' and 'dense_gemv' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k,v in acc.items():
    b=sum(v)/len(v)*1024*2
    print(k, 'launches', len(v), 'HBM bytes per launch (KiB*1024*2 gfx950 correction)', int(b), 'algorithmic', 32000*4096*2+4096*2+32000*2, 'ratio %.3f'%(b/(32000*4096*2+4096*2+32000*2)))
PY
head -12 $O/decode_engine_kernel_stats.csv | cut -c1-140
rm -rf $O/prof_engine $O/pmc_lm
