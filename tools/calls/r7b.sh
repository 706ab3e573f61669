#!/bin/bash
# the inference script's __main__ as a process (tokenizer leg included) + the reference-callers file + the new pair route test
O=gpurun_out/r7b; mkdir -p $O
timeout 1200 python -m pytest tests/test_reference_callers.py tests/test_gpu_parity.py -x -q -m gpu -k "reference_callers or loader_consumer or main_as or default_load or autotune" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
