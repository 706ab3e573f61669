#!/usr/bin/env python3
"""tools/stress_backward_small_m.py -- root-causing the ONE unexplained abort of round 2 (DESIGN.md "Known and unexplained"):
the GPU test process died once right after the backward product dx[M, K] = dy[M, N] . deq(W)^T had been put on the hipBLASLt route for every
M, suspect: the library on GEMMs with one to three columns.  This script hammers exactly that: M = 1, 2, 3 (and 5, 16) through
gptq_prefill_transpose_matmul248_f16 with the library forced (route 0), tens of thousands of calls over several shapes, a fresh plan-cache
entry per (M, shape), results checked against the LDS-tiled own kernel every 64 calls, the plan cache churned through its LRU bound by
odd leading dimensions, optionally under HIP_LAUNCH_BLOCKING=1 / AMD_LOG_LEVEL.  Output is kept under profiles/r3c_abort/."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd'))
import torch
from quant import _native

lib = _native.lib()
dev = torch.device('cuda:0')
gen = torch.Generator(device=dev)
gen.manual_seed(7)
ITER = int(os.environ.get('ITER', '4000'))
prev = lib.gptq_set_prefill_route(0)
t0 = time.time()
calls = bad = 0
for (K, N) in [(512, 288), (1024, 512), (4096, 4096), (4096, 11008)]:
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev, generator=gen)
    sc = (torch.rand((K // 128, N), device=dev, generator=gen) * 0.01 + 0.001).half()
    for M in (1, 2, 3, 5, 16):
        dy = torch.randn((M, N), device=dev, generator=gen).half()
        ref = torch.empty((M, K), dtype=torch.float16, device=dev)
        rc = lib.gptq_transpose_matmul248_f16(dy.data_ptr(), N, qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, ref.data_ptr(), K, M, K, N, 4, 128,
                                              torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        ws = torch.empty(lib.gptq_prefill_workspace_bytes(M, K, N, 1), dtype=torch.uint8, device=dev)
        for it in range(ITER if K <= 1024 else ITER // 8):
            ldx = K + 8 * (it % 97 if it % 16 == 0 else 0)          # every 16th call: a new leading dimension -> a new plan (LRU churn)
            dx = torch.empty((M, ldx), dtype=torch.float16, device=dev)
            rc = lib.gptq_prefill_transpose_matmul248_f16(dy.data_ptr(), N, qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, dx.data_ptr(), ldx, M, K, N,
                                                          4, 128, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            calls += 1
            if it % 64 == 0:
                torch.cuda.synchronize()
                err = float((dx[:, :K].float() - ref.float()).abs().max() / ref.float().abs().max())
                if not err < 2e-3:
                    bad += 1
                    print('MISMATCH K=%d N=%d M=%d it=%d err=%.3e' % (K, N, M, it, err), flush=True)
        torch.cuda.synchronize()
        print('K=%-5d N=%-5d M=%-2d ok (%d calls so far, %d plans cached, %.0f s)' % (K, N, M, calls, lib.gptq_prefill_plan_count(), time.time() - t0), flush=True)
lib.gptq_set_prefill_route(prev)
print('DONE: %d library calls with 1..16 columns, %d mismatches, plan cache at %d (bound 64), no abort' % (calls, bad, lib.gptq_prefill_plan_count()))
