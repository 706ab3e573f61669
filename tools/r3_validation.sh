set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3d}; mkdir -p $O
timeout 500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -25 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 600 python bench.py > $O/bench.json.txt 2> $O/bench.err; tail -c 3000 $O/bench.json.txt; tail -5 $O/bench.err
