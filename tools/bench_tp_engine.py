#!/usr/bin/env python3
"""Tensor-parallel decode engine (quant/tp_decode.py) on the LLaMA-7B shape: tok/s per replay, every rank one process.
  torchrun --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_tp_engine.py          (one rank per GPU, RCCL only for setup)
  TP_ONE_DEVICE=1 ... --nproc-per-node 2 ...                                                        (all ranks on cuda:0: the test box)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
local = 0 if os.environ.get('TP_ONE_DEVICE') else int(os.environ.get('LOCAL_RANK', '0'))
dist.init_process_group('gloo', rank=rank, world_size=world)          # handle exchange + barriers only: the data path is the one-shot exchange
torch.cuda.set_device(local)
from quant.decode import build_random_llama, DecodeEngine
from quant.tp_decode import TPDecodeEngine
dev = 'cuda:%d' % local
model = build_random_llama(dev)                                       # the same seed on every rank
tokens = int(os.environ.get('TOKENS', '64'))
eng = TPDecodeEngine(model, t_max=2048).capture()
tok = torch.zeros(1, dtype=torch.long, device=dev)
for _ in range(4):
    eng.decode(tok)
torch.cuda.synchronize(); dist.barrier()
ts = []
for _ in range(tokens):
    t0 = time.perf_counter()
    eng.decode(tok)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
dist.barrier()
med = sorted(ts)[len(ts) // 2]
out = {'ranks': world, 'one_device': bool(os.environ.get('TP_ONE_DEVICE')), 'median_s_per_token': round(med, 6), 'tokens_per_s': round(1.0 / med, 1),
       'exchanges_per_token': 2 * len(eng.layers), 'p2p_status': eng.status()}
if rank == 0:
    full = DecodeEngine(model, t_max=2048).capture()
    for _ in range(4):
        full.decode(tok)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(16):
        full.decode(tok); torch.cuda.synchronize()
    out['single_gpu_engine_tokens_per_s'] = round(16 / (time.perf_counter() - t0), 1)
    print(json.dumps(out))
dist.barrier()
dist.destroy_process_group()
