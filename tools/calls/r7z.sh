#!/bin/bash
# the generate fast-path tests once more after the torch-version guard of the sampling path
O=gpurun_out/r7z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_reference_callers.py -x -q -m gpu -k "fast_path or main_as or default_load or generate" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
