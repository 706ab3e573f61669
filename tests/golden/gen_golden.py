#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ FROM THE REFERENCE ITSELF.

Run once in the build container (needs the read-only upstream tree, default
/root/reference; it is never read by tests, smoke() or bench.py):

    python tests/golden/gen_golden.py [--ref /root/reference]

What is recorded (all inputs seeded, all arrays small):

* pack_*.npz   -- reference ``Quantizer.find_params``/``quantize`` per group + the
                  reference's own ``QuantLinear.pack`` (quant/quant_linear.py:325-371).
* fwd_*.npz    -- outputs of the reference's own ``matmul_248_kernel``
                  (quant/quant_linear.py:72-137) executed by Triton's CPU interpreter
                  (TRITON_INTERPRET=1), bypassing only the custom autotuner (which does not run
                  on triton 3.6) by launching ``kernel.fn[grid]`` with one explicit tile config.
* bwd_*.npz    -- ``transpose_matmul_248_kernel`` (quant/quant_linear.py:191-258), same way.
* mlp_*.npz    -- ``fusedmatmul_248_kernel`` (quant/fused_mlp.py:84-168).
* norm_*.npz   -- ``rms_norm_fwd_fused`` (quant/triton_norm.py:7-39).
* rope_*.npz   -- ``rotate_half_kernel`` (quant/fused_attn.py:8-58) with ``tl.libdevice.exp``
                  (removed from triton 3.x) mapped to ``tl.exp`` at generation time.

The reference source is imported, never copied.
"""
import argparse
import math
import os
import sys
import types

os.environ['TRITON_INTERPRET'] = '1'
sys.dont_write_bytecode = True

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference(ref):
    for name in ('texttable', 'toml'):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                m = types.ModuleType(name)
                m.Texttable = object
                sys.modules[name] = m
    sys.path.insert(0, ref)
    import triton
    import triton.language as tl
    if not hasattr(tl, 'libdevice'):
        class _LibdeviceShim:  # resolve at call time so the interpreter's patched tl.exp is used
            def __getattr__(self, name):
                return getattr(tl, name)
        tl.libdevice = _LibdeviceShim()
    import quant  # the REFERENCE's quant package
    assert os.path.abspath(quant.__file__).startswith(os.path.abspath(ref)), quant.__file__
    return quant, triton, tl


def act_order_g_idx(K, gs, rng):
    perm = rng.permutation(K)
    invperm = np.argsort(perm)
    return (np.arange(K) // gs)[invperm].astype(np.int32)


def make_faithful_layer(quant, bits, groupsize, K, N, act_order, sym, seed, bias=False):
    """random fp32 Linear -> reference Quantizer per group -> reference pack()."""
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    gs = K if groupsize == -1 else groupsize
    G = K // gs
    lin = torch.nn.Linear(K, N, bias=bias)
    W = lin.weight.data.clone().float() * 3.0
    g_idx = act_order_g_idx(K, gs, rng) if act_order else (np.arange(K) // gs).astype(np.int32)
    g_t = torch.from_numpy(g_idx.astype(np.int64))
    scale = torch.zeros(N, G)
    zero = torch.zeros(N, G)
    Q = torch.zeros_like(W)
    for g in range(G):
        cols = (g_t == g).nonzero().flatten()
        qz = quant.Quantizer()
        qz.configure(bits, perchannel=True, sym=sym, mse=False)
        # keep every group strictly spanning 0 on the low side so zero >= 1
        Wg = W[:, cols]
        Wg[:, 0] = -Wg[:, 0].abs() - 0.01
        W[:, cols] = Wg
        qz.find_params(Wg, weight=True)
        Q[:, cols] = qz.quantize(Wg)
        scale[:, g] = qz.scale.flatten()
        zero[:, g] = qz.zero.flatten()
    lin.weight.data = Q.clone()
    ql = quant.QuantLinear(bits, groupsize, K, N, bias)
    ql.pack(lin, scale.clone(), zero.clone(), torch.from_numpy(g_idx.copy()))
    return dict(weight_q=Q.numpy(), scales_in=scale.numpy(), zeros_in=zero.numpy(),
                g_idx=g_idx, qweight=ql.qweight.numpy(), qzeros=ql.qzeros.numpy(),
                scales=ql.scales.numpy(), bias=(ql.bias.detach().numpy() if bias else np.zeros(0, np.float16)),
                bits=np.int32(bits), groupsize=np.int32(groupsize))


def make_random_layer(bits, groupsize, K, N, act_order, seed):
    """SURVEY 8(d) synthetic distribution: uniform int32 bit patterns, scales U(.001,.011)."""
    rng = np.random.default_rng(seed)
    gs = K if groupsize == -1 else groupsize
    G = K // gs
    qweight = rng.integers(-2**31, 2**31, size=(K // 32 * bits, N), dtype=np.int64).astype(np.int32)
    qzeros = rng.integers(-2**31, 2**31, size=(G, N // 32 * bits), dtype=np.int64).astype(np.int32)
    scales = rng.uniform(0.001, 0.011, size=(G, N)).astype(np.float16)
    g_idx = act_order_g_idx(K, gs, rng) if act_order else (np.arange(K) // gs).astype(np.int32)
    return dict(qweight=qweight, qzeros=qzeros, scales=scales, g_idx=g_idx,
                bits=np.int32(bits), groupsize=np.int32(groupsize))


def ref_forward(quant, triton, x, L, bm=16, bn=32, bk=32):
    ql = quant.quant_linear
    bits = int(L['bits'])
    qweight = torch.from_numpy(L['qweight'])
    qzeros = torch.from_numpy(L['qzeros'])
    scales = torch.from_numpy(L['scales'])
    g_idx = torch.from_numpy(L['g_idx'])
    M, K = x.shape
    N = qweight.shape[1]
    out = torch.empty((M, N), dtype=torch.float16)
    grid = (triton.cdiv(M, bm) * triton.cdiv(N, bn),)
    ql.matmul_248_kernel.fn[grid](x, qweight, out, scales, qzeros, g_idx, M, N, K, bits, 2**bits - 1,
                                  x.stride(0), x.stride(1), qweight.stride(0), qweight.stride(1),
                                  out.stride(0), out.stride(1), scales.stride(0), qzeros.stride(0),
                                  BLOCK_SIZE_M=bm, BLOCK_SIZE_N=bn, BLOCK_SIZE_K=bk, GROUP_SIZE_M=8)
    return out


def ref_backward(quant, triton, dy, L, bm=16, bn=32, bk=32):
    ql = quant.quant_linear
    bits = int(L['bits'])
    qweight = torch.from_numpy(L['qweight'])
    qzeros = torch.from_numpy(L['qzeros'])
    scales = torch.from_numpy(L['scales'])
    g_idx = torch.from_numpy(L['g_idx'])
    M, N = dy.shape
    K = qweight.shape[0] * 32 // bits
    out = torch.empty((M, K), dtype=torch.float16)
    grid = (triton.cdiv(M, bm) * triton.cdiv(K, bk),)
    ql.transpose_matmul_248_kernel.fn[grid](dy, qweight, out, scales, qzeros, g_idx, M, N, K, bits,
                                            2**bits - 1, dy.stride(0), dy.stride(1),
                                            qweight.stride(0), qweight.stride(1), out.stride(0),
                                            out.stride(1), scales.stride(0), qzeros.stride(0),
                                            BLOCK_SIZE_M=bm, BLOCK_SIZE_N=bn, BLOCK_SIZE_K=bk,
                                            GROUP_SIZE_M=8)
    return out


def ref_fused_mlp(quant, triton, x, A, B, bm=16, bn=32, bk=32):
    fm = quant.fused_mlp
    bits = int(A['bits'])
    t = lambda a: torch.from_numpy(a)
    M, K = x.shape
    N = A['qweight'].shape[1]
    c = torch.empty((M, N), dtype=torch.float16)
    grid = (triton.cdiv(M, bm) * triton.cdiv(N, bn),)
    qa, qb = t(A['qweight']), t(B['qweight'])
    sa, sb = t(A['scales']), t(B['scales'])
    za, zb = t(A['qzeros']), t(B['qzeros'])
    ga, gb = t(A['g_idx']), t(B['g_idx'])
    fm.fusedmatmul_248_kernel.fn[grid](x, c, qa, sa, za, ga, qb, sb, zb, gb, M, N, K, bits,
                                       2**bits - 1, x.stride(0), x.stride(1), qa.stride(0),
                                       qa.stride(1), c.stride(0), c.stride(1), sa.stride(0),
                                       za.stride(0), BLOCK_SIZE_M=bm, BLOCK_SIZE_N=bn,
                                       BLOCK_SIZE_K=bk, GROUP_SIZE_M=8)
    return c


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print('wrote', os.path.relpath(path), os.path.getsize(path), 'bytes')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    args = ap.parse_args()
    quant, triton, tl = import_reference(args.ref)

    # ---- pack fixtures (reference Quantizer + pack) + forward on them --------------------
    faithful_cfgs = [
        # name, bits, groupsize, K, N, act_order, sym, bias
        ('w4g128', 4, 128, 256, 128, False, False, False),
        ('w4g32_act', 4, 32, 128, 64, True, False, False),
        ('w4gall_sym_bias', 4, -1, 128, 64, False, True, True),
        ('w2g64', 2, 64, 128, 64, False, False, False),
        ('w8g128', 8, 128, 256, 64, False, False, False),
    ]
    for i, (name, bits, gs, K, N, act, sym, bias) in enumerate(faithful_cfgs):
        L = make_faithful_layer(quant, bits, gs, K, N, act, sym, seed=100 + i, bias=bias)
        torch.manual_seed(200 + i)
        x = torch.randn(5, K).half()
        y = ref_forward(quant, triton, x, L)
        save('pack_' + name, x=x.numpy(), y=y.numpy(), **L)

    # ---- forward fixtures on the synthetic (uniform bit pattern) distribution -------------
    fwd_cfgs = [
        ('w4g128_m1', 4, 128, 512, 256, False, 1),
        ('w4g128_m33', 4, 128, 256, 128, False, 33),
        ('w4g128_act_m3', 4, 128, 512, 128, True, 3),
        ('w4gall_m2', 4, -1, 256, 128, False, 2),
        ('w2g64_m4', 2, 64, 256, 128, False, 4),
        ('w2g64_act_m1', 2, 64, 256, 64, True, 1),
        ('w8g128_m7', 8, 128, 256, 128, False, 7),
        ('w8g32_act_m1', 8, 32, 128, 64, True, 1),
    ]
    for i, (name, bits, gs, K, N, act, M) in enumerate(fwd_cfgs):
        L = make_random_layer(bits, gs, K, N, act, seed=300 + i)
        torch.manual_seed(400 + i)
        x = torch.randn(M, K).half()
        y = ref_forward(quant, triton, x, L)
        save('fwd_' + name, x=x.numpy(), y=y.numpy(), **L)

    # ---- backward (transposed) -------------------------------------------------------------
    bwd_cfgs = [
        ('w4g128_m3', 4, 128, 256, 128, False, 3),
        ('w4g32_act_m17', 4, 32, 128, 64, True, 17),
        ('w8g128_m2', 8, 128, 256, 64, False, 2),
        ('w2g64_m1', 2, 64, 128, 64, False, 1),
    ]
    for i, (name, bits, gs, K, N, act, M) in enumerate(bwd_cfgs):
        L = make_random_layer(bits, gs, K, N, act, seed=500 + i)
        torch.manual_seed(600 + i)
        dy = torch.randn(M, N).half()
        dx = ref_backward(quant, triton, dy, L)
        save('bwd_' + name, dy=dy.numpy(), dx=dx.numpy(), **L)

    # ---- fused gate/up MLP -----------------------------------------------------------------
    mlp_cfgs = [
        ('w4g128_m1', 4, 128, 256, 128, False, 1),
        ('w4g128_act_m5', 4, 128, 256, 64, True, 5),
        ('w8g128_m2', 8, 128, 128, 64, False, 2),
    ]
    for i, (name, bits, gs, K, N, act, M) in enumerate(mlp_cfgs):
        A = make_random_layer(bits, gs, K, N, act, seed=700 + 2 * i)
        B = make_random_layer(bits, gs, K, N, act, seed=701 + 2 * i)
        torch.manual_seed(800 + i)
        x = torch.randn(M, K).half()
        c = ref_fused_mlp(quant, triton, x, A, B)
        arrs = {('gate_' + k): v for k, v in A.items()}
        arrs.update({('up_' + k): v for k, v in B.items()})
        save('mlp_' + name, x=x.numpy(), c=c.numpy(), **arrs)

    # ---- RMSNorm ---------------------------------------------------------------------------
    from quant.triton_norm import TritonLlamaRMSNorm
    import contextlib
    torch.cuda.device = lambda d: contextlib.nullcontext()  # no GPU here; kernel runs interpreted
    for i, (M, N, eps) in enumerate([(3, 256, 1e-6), (1, 4096, 1e-5), (5, 96, 1e-6)]):
        torch.manual_seed(900 + i)
        x = (torch.randn(M, N) * 1.7).half()
        w = (1.0 + 0.1 * torch.randn(N)).half()
        y = TritonLlamaRMSNorm(w, eps)(x)
        save('norm_%d' % i, x=x.numpy(), w=w.numpy(), y=y.numpy(), eps=np.float32(eps))

    # ---- RoPE ------------------------------------------------------------------------------
    from quant.fused_attn import triton_rotate_half_
    for i, (bsz, seq, heads, hd, past) in enumerate([(1, 1, 4, 128, 17), (2, 5, 2, 64, 0),
                                                     (1, 3, 2, 32, 2000)]):
        torch.manual_seed(1000 + i)
        qkv = torch.randn(bsz, seq, 3, heads, hd).half()
        pos = (torch.arange(seq)[None, :] + past + torch.arange(bsz)[:, None] * 3).long().contiguous()
        before = qkv.clone()
        triton_rotate_half_(qkv[:, :, :2], pos)
        save('rope_%d' % i, qkv_in=before.numpy(), qkv_out=qkv.numpy(), pos=pos.numpy())


if __name__ == '__main__':
    main()
