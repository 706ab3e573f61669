#!/usr/bin/env python3
"""Feasibility probe: one decode pass over the LLaMA-7B linears as TWO half-chains on two streams of ONE GPU (Megatron pairing at
world 2 on one device: qkv and gate/up split by columns, o and down by rows with fp32 partials, one join = sum + rounding after
each pair) captured in one hipGraph, against the single chain of 4 launches per layer.  Data dependencies are real: every op
reads what the previous one wrote."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from bench import PackedSet, alg_bytes, BITS, GS, HIDDEN, INTER
from quant import _native, quant_linear as QL, tensor_parallel as TP
dev = 'cuda:0'; lib = _native.lib()
gen = torch.Generator(device=dev); gen.manual_seed(0)
LAYERS = int(os.environ.get('LAYERS', '32'))
H, I = HIDDEN, INTER
W = 2


def k_shard(w, kb):
    r0, r1, g0, g1 = kb[0] // 8, kb[1] // 8, kb[0] // GS, kb[1] // GS
    return w.qweight[r0:r1].contiguous(), w.scales[g0:g1].contiguous(), w.qzeros[g0:g1].contiguous()


def n_shard(w, nb):
    return (w.qweight[:, nb[0]:nb[1]].contiguous(), w.scales[:, nb[0]:nb[1]].contiguous(), w.qzeros[:, nb[0] // 8:nb[1] // 8].contiguous())


kb_h = TP.row_shard_bounds(H, GS, BITS, W)
kb_i = TP.row_shard_bounds(I, GS, BITS, W)
nb_qkv = TP.col_shard_bounds(3 * H, W)
nb_i = TP.col_shard_bounds(I, W)
full, halves = [], []
for _ in range(LAYERS):
    f = dict(qkv=PackedSet(H, 3 * H, dev, gen), o=PackedSet(H, H, dev, gen), gate=PackedSet(H, I, dev, gen), up=PackedSet(H, I, dev, gen),
             down=PackedSet(I, H, dev, gen))
    full.append(dict(qkv=QL.stripe_copy(f['qkv'].qweight, f['qkv'].scales, f['qkv'].qzeros, BITS, GS),
                     o=QL.stripe_copy(f['o'].qweight, f['o'].scales, f['o'].qzeros, BITS, GS),
                     mlp=QL.stripe_copy(f['gate'].qweight, f['gate'].scales, f['gate'].qzeros, BITS, GS, up=(f['up'].qweight, f['up'].scales, f['up'].qzeros)),
                     down=QL.stripe_copy(f['down'].qweight, f['down'].scales, f['down'].qzeros, BITS, GS)))
    hs = []
    for r in range(W):
        hs.append(dict(qkv=QL.stripe_copy(*n_shard(f['qkv'], nb_qkv[r]), BITS, GS), o=QL.stripe_copy(*k_shard(f['o'], kb_h[r]), BITS, GS),
                       mlp=QL.stripe_copy(*n_shard(f['gate'], nb_i[r]), BITS, GS, up=n_shard(f['up'], nb_i[r])),
                       down=QL.stripe_copy(*k_shard(f['down'], kb_i[r]), BITS, GS)))
    halves.append(hs)
    torch.cuda.synchronize()
    del f
f16, f32 = dict(dtype=torch.float16, device=dev), dict(dtype=torch.float32, device=dev)
x_h = torch.randn((1, H), device=dev, generator=gen).half()
y_qkv, y_o, y_i = torch.empty((1, 3 * H), **f16), torch.empty((1, H), **f16), torch.empty((1, I), **f16)
p_o = [torch.empty((1, H), **f32) for _ in range(W)]
p_d = [torch.empty((1, H), **f32) for _ in range(W)]


def mv(x, st, out, K, N, nsets, s):
    _native.check(lib.gptq_stripe_matvec_f16(x.data_ptr(), K, st.data_ptr(), st.numel(), None, out.data_ptr(), N, 1, K, N, BITS, GS, nsets, None, 0.0, None, s), 'mv')


def part(x, st, out, K, N, s):
    _native.check(lib.gptq_stripe_matvec_partial_f32(x.data_ptr(), st.data_ptr(), st.numel(), out.data_ptr(), K, N, BITS, GS, 1, None, s), 'part')


def single():
    s = torch.cuda.current_stream().cuda_stream
    for L in full:       # x_h -> qkv -> (stand-in for attention: o reads the first H of qkv) -> o -> gate/up -> down -> x_h
        mv(x_h, L['qkv'], y_qkv, H, 3 * H, 1, s)
        mv(y_qkv[:, :H], L['o'], y_o, H, H, 1, s)
        mv(y_o, L['mlp'], y_i, H, I, 2, s)
        mv(y_i, L['down'], x_h, I, H, 1, s)


sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def two_streams():
    main = torch.cuda.current_stream()
    streams = [sA, sB]
    for hs in halves:
        ev0 = torch.cuda.Event(); ev0.record(main)
        evs = []
        for r, st in enumerate(streams):
            st.wait_event(ev0)
            with torch.cuda.stream(st):
                s = st.cuda_stream
                n0, n1 = nb_qkv[r]
                mv(x_h, hs[r]['qkv'], y_qkv[:, n0:n1], H, n1 - n0, 1, s)                  # this half's columns of qkv
                k0, k1 = kb_h[r]
                part(y_qkv[:, n0:n0 + (k1 - k0)], hs[r]['o'], p_o[r], k1 - k0, H, s)        # stand-in for this half's attention output
                e = torch.cuda.Event(); e.record(st); evs.append(e)
        for e in evs:
            main.wait_event(e)
        torch.add(p_o[0], p_o[1], out=p_o[0]); y_o.copy_(p_o[0])                             # join: sum, ONE rounding (2 tiny launches here)
        ev1 = torch.cuda.Event(); ev1.record(main)
        evs = []
        for r, st in enumerate(streams):
            st.wait_event(ev1)
            with torch.cuda.stream(st):
                s = st.cuda_stream
                n0, n1 = nb_i[r]
                mv(y_o, hs[r]['mlp'], y_i[:, n0:n1], H, n1 - n0, 2, s)
                k0, k1 = kb_i[r]
                part(y_i[:, k0:k1], hs[r]['down'], p_d[r], k1 - k0, H, s)
                e = torch.cuda.Event(); e.record(st); evs.append(e)
        for e in evs:
            main.wait_event(e)
        torch.add(p_d[0], p_d[1], out=p_d[0]); x_h.copy_(p_d[0])


def timed(fn, reps=20):
    x_h.normal_(generator=gen)
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


nbytes = LAYERS * (alg_bytes(1, H, 3 * H) + alg_bytes(1, H, H) + alg_bytes(1, H, I, nsets=2) + alg_bytes(1, I, H))
t1 = timed(single)
t2 = timed(two_streams)
print(json.dumps({'layers': LAYERS, 'single_chain_ms': round(t1, 4), 'single_chain_GBps': round(nbytes / t1 / 1e6, 1), 'two_half_chains_ms': round(t2, 4),
                  'two_half_chains_GBps': round(nbytes / t2 / 1e6, 1), 'launches_per_layer': '4 vs 8 + 4 join ops'}))
