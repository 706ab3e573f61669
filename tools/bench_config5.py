#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU: LLaMA-65B-shaped 4-bit g128 linears at batch 1 -- the whole matrices, and the
K-shards each of 8 ranks runs in the row-sharded layout (quant/tensor_parallel.py: row_shard_bounds cuts on group
boundaries, 172 groups -> 22,22,22,22,21,21,21,21).  Cold weights (rotation over > 256 MiB of distinct sets in one
hipGraph), SURVEY 8(d) byte model.  The all-reduce itself needs 8 GPUs and is not measured here."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from quant import _native
from quant.tensor_parallel import row_shard_bounds
dev = 'cuda:0'
lib = _native.lib(); ws = _native.workspace(torch.device(dev))
gen = torch.Generator(device=dev); gen.manual_seed(0)
BITS, GS = 4, 128
H, I = 8192, 22016

def make(K, N):
    G = K // GS
    return (torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=gen),
            (torch.rand((G, N), device=dev, generator=gen) * 0.01 + 0.001).half(),
            torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev, generator=gen))

def nbytes(K, N, nsets=1):
    G = K // GS
    return nsets * (4 * (K // 8) * N + 4 * G * (N // 8) + 2 * G * N) + 2 * K + 2 * N

def run_case(label, K, N, fused):
    nb = nbytes(K, N, 2 if fused else 1)
    nsets = int(300e6 // nb) + 1
    sets = [(make(K, N), make(K, N) if fused else None) for _ in range(nsets)]
    x = torch.randn((1, K), device=dev, generator=gen).half()
    y = torch.empty((1, N), device=dev, dtype=torch.float16)
    def launch(i):
        (qw, sc, qz), up = sets[i]
        s = torch.cuda.current_stream().cuda_stream
        if fused:
            rc = lib.gptq_fused_mlp_f16(x.data_ptr(), K, qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, up[0].data_ptr(), up[1].data_ptr(),
                                        up[2].data_ptr(), None, y.data_ptr(), N, 1, K, N, BITS, GS, ws.data_ptr(), ws.numel(), s)
        else:
            rc = lib.gptq_matmul248_f16(x.data_ptr(), K, qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, None, y.data_ptr(), N, 1, K, N,
                                        BITS, GS, ws.data_ptr(), ws.numel(), s)
        _native.check(rc, label)
    for i in range(nsets): launch(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(nsets): launch(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * nsets)
    print(json.dumps({'case': label, 'K': K, 'N': N, 'fused_gate_up': fused, 'us': round(us, 2), 'MB': round(nb / 1e6, 2),
                      'GBps': round(nb / us / 1e3, 1), 'frac_of_8TBps': round(nb / us / 8e6, 4)}), flush=True)
    del sets, g
    torch.cuda.empty_cache()

for label, K, N, fused in [('65B qkv (whole)', H, 3 * H, False), ('65B o_proj (whole)', H, H, False), ('65B gate/up+SiLU (whole)', H, I, True),
                           ('65B down_proj (whole)', I, H, False)]:
    run_case(label, K, N, fused)
# per-rank K shards of the row-sharded layout (rank 0 of 8; uneven group counts for K = 22016)
for label, K, N, fused in [('qkv', H, 3 * H, False), ('o_proj', H, H, False), ('gate/up+SiLU', H, I, True), ('down_proj', I, H, False)]:
    lo, hi = row_shard_bounds(K, GS, BITS, 8)[0]
    run_case('65B %s, K-shard of rank 0/8 (rows %d:%d)' % (label, lo, hi), hi - lo, N, fused)
