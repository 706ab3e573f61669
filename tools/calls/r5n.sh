set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; mkdir -p $O
ALL_ROUTES=1 MS=192,256,384,512,640,768,896,1024,1280,1536,1792,2048,2304,2560 timeout 900 python tools/bench_mid_prefill.py > $O/mid_prefill.txt 2>&1; cat $O/mid_prefill.txt
