#!/bin/bash
# longer soak of the loader / consumer kernel's instances: 50 000 launches each (GPTQ_SOAK_LAUNCHES / 8), bit for bit, LDS dirtied between blocks of launches
O=gpurun_out/r7n; mkdir -p $O
GPTQ_SOAK_LAUNCHES=400000 timeout 2400 python -m pytest tests/test_gpu_soak.py -q -m gpu -k "short_prompt_tiles" -v > $O/soak_mmr_50000.txt 2>&1; tail -22 $O/soak_mmr_50000.txt | cut -c1-160
