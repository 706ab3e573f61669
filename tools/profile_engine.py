#!/usr/bin/env python3
"""run the DecodeEngine for a few tokens (for rocprofv3 --kernel-trace --stats).  --batch B: DecodeEngine(batch=B); --start P: at depth P; --nofuse"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from quant.decode import build_random_llama, benchmark_decode_engine
fuse = '--nofuse' not in sys.argv
m = build_random_llama('cuda:0')
sp = int(sys.argv[sys.argv.index('--start') + 1]) if '--start' in sys.argv else 0
B = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 1
print(benchmark_decode_engine(m, tokens=40, graph=True, fuse_norm=fuse, fuse_attn=fuse, start_pos=sp, batch=B))
