cd $GRAFT_REPO_ROOT
O=gpurun_out/r5aa; mkdir -p $O
for P in 0 1 0 1; do
echo "C1_PLAIN $P"
GPTQ_DECODE_C1_PLAIN=$P MS=1 SHAPES= timeout 300 python tools/bench_layer_decode.py 2>/dev/null | grep "12288\|pair"
GPTQ_DECODE_C1_PLAIN=$P timeout 300 python bench.py --steps 30 --warmup 5 --no-decode --no-cpu-baseline --no-per-shape --no-prefill --no-config4 --no-small-batch 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', d['value'], d['ms_per_step'])"
done > $O/c1_plain.txt; cat $O/c1_plain.txt
