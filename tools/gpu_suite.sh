#!/bin/bash
# the whole GPU suite + smoke (what the driver runs at round end); output under gpurun_out/$1
O=gpurun_out/${1:-suite}; mkdir -p $O
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
