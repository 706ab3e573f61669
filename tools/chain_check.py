#!/usr/bin/env python3
"""GPU check + timing of the persistent matvec chain (csrc/chain.hip) against the per-op C-ABI path.

  python tools/chain_check.py [--layers 32] [--reps 20]

1. one decoder layer's four ops (RMSNorm+qkv, o_proj+residual, RMSNorm+gate/up+SiLU, down_proj+residual) through
   the chain vs the same ops through gptq_rmsnorm_matmul248_f16 / gptq_matmul248_f16 / gptq_rmsnorm_fused_mlp_f16;
2. the bench.py workload (LLaMA-7B, 4 ops x L layers, distinct weights) as ONE chain vs 4*L launches, both as
   hipGraph replays, HIP-event timed;
3. the per-op timeline of workgroup 0.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd'))
sys.path.insert(0, ROOT)
import torch

from quant import _native
from quant.chain import MatvecChain
import bench as B


def per_op_layer(lib, ws, s, L, h, ln1, ln2, attn, qkv, act, eps):
    """reference: the existing one-launch-per-op entry points"""
    q, o, g, u, d = L['qkv'], L['o'], L['gate'], L['up'], L['down']
    _native.check(lib.gptq_rmsnorm_matmul248_f16(h.data_ptr(), ln1.data_ptr(), eps, q.qweight.data_ptr(), q.scales.data_ptr(), q.qzeros.data_ptr(),
                                                 None, None, qkv.data_ptr(), q.K, q.N, B.BITS, B.GS, ws.data_ptr(), ws.numel(), s), 'norm qkv')
    _native.check(lib.gptq_matmul248_f16(attn.data_ptr(), o.K, o.qweight.data_ptr(), o.scales.data_ptr(), o.qzeros.data_ptr(), None, h.data_ptr(),
                                         h.data_ptr(), o.N, 1, o.K, o.N, B.BITS, B.GS, ws.data_ptr(), ws.numel(), s), 'o')
    _native.check(lib.gptq_rmsnorm_fused_mlp_f16(h.data_ptr(), ln2.data_ptr(), eps, g.qweight.data_ptr(), g.scales.data_ptr(), g.qzeros.data_ptr(), None,
                                                 u.qweight.data_ptr(), u.scales.data_ptr(), u.qzeros.data_ptr(), None, act.data_ptr(), g.K, g.N,
                                                 B.BITS, B.GS, ws.data_ptr(), ws.numel(), s), 'norm mlp')
    _native.check(lib.gptq_matmul248_f16(act.data_ptr(), d.K, d.qweight.data_ptr(), d.scales.data_ptr(), d.qzeros.data_ptr(), None, h.data_ptr(),
                                         h.data_ptr(), d.N, 1, d.K, d.N, B.BITS, B.GS, ws.data_ptr(), ws.numel(), s), 'down')


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-9))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', type=int, default=32)
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    dev = 'cuda:0'
    torch.cuda.set_device(0)
    lib = _native.lib()
    D = B.DecodeLinears(dev, layers=args.layers)
    ws = D.ws
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    eps = 1e-6
    ln1 = (1 + 0.1 * torch.randn(B.HIDDEN, device=dev, generator=gen)).half()
    ln2 = (1 + 0.1 * torch.randn(B.HIDDEN, device=dev, generator=gen)).half()
    h0 = torch.randn(B.HIDDEN, device=dev, generator=gen).half()
    attn = (torch.randn(B.HIDDEN, device=dev, generator=gen) * 0.5).half()

    # ---- 1. one layer, chain vs per-op ----
    L0 = D.layers[0]
    h_ref, qkv_ref, act_ref = h0.clone(), torch.empty(3 * B.HIDDEN, device=dev, dtype=torch.float16), torch.empty(B.INTER, device=dev, dtype=torch.float16)
    per_op_layer(lib, ws, torch.cuda.current_stream().cuda_stream, L0, h_ref, ln1, ln2, attn, qkv_ref, act_ref, eps)
    torch.cuda.synchronize()
    h, qkv, act = h0.clone(), torch.zeros_like(qkv_ref), torch.zeros_like(act_ref)
    ch = MatvecChain(B.BITS, B.GS, dev)
    q, o, g, u, d = L0['qkv'], L0['o'], L0['gate'], L0['up'], L0['down']
    ch.add(h, q.qweight, q.scales, q.qzeros, qkv, norm_weight=ln1, norm_eps=eps)
    ch.add(attn, o.qweight, o.scales, o.qzeros, h, residual=h)
    ch.add(h, g.qweight, g.scales, g.qzeros, act, up=(u.qweight, u.scales, u.qzeros), norm_weight=ln2, norm_eps=eps)
    ch.add(act, d.qweight, d.scales, d.qzeros, h, residual=h)
    ch.finalize()
    ch.run()
    torch.cuda.synchronize()
    print('chain status', ch.status(), ' workgroups', ch.nwg, flush=True)
    print('1-layer chain vs per-op: qkv %.2e  act %.2e  h %.2e' % (rel(qkv, qkv_ref), rel(act, act_ref), rel(h, h_ref)), flush=True)
    print('workspace clean:', bool((ws[:262144] == 0).all()), flush=True)
    # run-to-run reproducibility
    h2, qkv2, act2 = h0.clone(), torch.zeros_like(qkv_ref), torch.zeros_like(act_ref)
    ch2 = MatvecChain(B.BITS, B.GS, dev)
    ch2.add(h2, q.qweight, q.scales, q.qzeros, qkv2, norm_weight=ln1, norm_eps=eps)
    ch2.add(attn, o.qweight, o.scales, o.qzeros, h2, residual=h2)
    ch2.add(h2, g.qweight, g.scales, g.qzeros, act2, up=(u.qweight, u.scales, u.qzeros), norm_weight=ln2, norm_eps=eps)
    ch2.add(act2, d.qweight, d.scales, d.qzeros, h2, residual=h2)
    ch2.finalize(); ch2.run(); torch.cuda.synchronize()
    print('bit-identical rerun:', bool(torch.equal(h, h2) and torch.equal(act, act2) and torch.equal(qkv, qkv2)), 'status', ch2.status(), flush=True)

    # ---- 2. bench workload: one chain of 4*L ops vs 4*L launches ----
    big = MatvecChain(B.BITS, B.GS, dev)
    for L in D.layers:
        big.add(D.x_h.view(-1), L['qkv'].qweight, L['qkv'].scales, L['qkv'].qzeros, D.y_qkv.view(-1))
        big.add(D.x_h.view(-1), L['o'].qweight, L['o'].scales, L['o'].qzeros, D.y_h.view(-1))
        big.add(D.x_h.view(-1), L['gate'].qweight, L['gate'].scales, L['gate'].qzeros, D.y_i.view(-1),
                up=(L['up'].qweight, L['up'].scales, L['up'].qzeros))
        big.add(D.x_i.view(-1), L['down'].qweight, L['down'].scales, L['down'].qzeros, D.y_h.view(-1))
    big.finalize()
    D.step(); torch.cuda.synchronize()
    y_ref = (D.y_qkv.clone(), D.y_h.clone(), D.y_i.clone())
    D.y_qkv.zero_(); D.y_h.zero_(); D.y_i.zero_()
    big.run(); torch.cuda.synchronize()
    print('%d-op chain status %d; last-layer outputs vs per-op: qkv %.2e h %.2e act %.2e' %
          (len(big.ops), big.status(), rel(D.y_qkv, y_ref[0]), rel(D.y_h, y_ref[1]), rel(D.y_i, y_ref[2])), flush=True)

    def time_graph(fn):
        g = torch.cuda.CUDAGraph()
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.reps
    us_ops = time_graph(D.step)
    for depth in (2, 3):
        lib.gptq_set_chain_depth(depth)
        us = time_graph(big.run)
        print('chain depth %d   : %8.1f us/pass  %7.1f GB/s' % (depth, us, D.bytes_per_step / us / 1e3), flush=True)
    lib.gptq_set_chain_depth(4)
    us_chain = time_graph(big.run)
    nb = D.bytes_per_step
    print('per-op launches: %8.1f us/pass  %7.1f GB/s (%.1f%% of 8 TB/s)' % (us_ops, nb / us_ops / 1e3, nb / us_ops / 1e3 / 80), flush=True)
    print('one chain      : %8.1f us/pass  %7.1f GB/s (%.1f%% of 8 TB/s)  status %d' % (us_chain, nb / us_chain / 1e3, nb / us_chain / 1e3 / 80, big.status()), flush=True)
    big.run(timeline=True); torch.cuda.synchronize()
    tl = big.timeline().double() / 100.0   # us
    names = ['qkv', 'o', 'gate/up', 'down']
    slots = [(0, 'dep seen'), (1, 'x staged'), (5, 'cw: x seen'), (6, 'cw: w landed'), (8, 'cw: math1 done'), (7, 'cw: last job handed'),
             (2, 'sw: job1 arrived'), (3, 'sw: job1 atomic back'), (9, 'sw: job1 y acked'), (4, 'sw: last job published')]
    for i in range(8, min(12, len(big.ops))):
        seen = tl[i, :, 0]
        t0 = seen[seen > 0].min()           # workgroups without a job in this op leave no stamps
        print('  op %2d %-7s (t0 = first workgroup sees the dependency; min / median / max over workgroups, us)' % (i, names[i % 4]))
        for k, nm in slots:
            v = tl[i, :, k] - t0
            v = v[tl[i, :, k] > 0]
            if v.numel():
                print('      %-24s %7.2f %7.2f %7.2f' % (nm, v.min(), v.median(), v.max()))
    n = len(big.ops)
    print('  whole chain: %.1f us' % (tl[n - 1, :, 4].max() - tl[0, :, 1].min()), flush=True)


if __name__ == '__main__':
    main()
