#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: "int4 g128 matvec GB/s + decode tokens/s, LLaMA-7B 4-bit".

One "step" = one batch-1 decode pass over ALL quantised linears of a LLaMA-7B-shaped model
(32 layers x [fused qkv 4096x12288, o_proj 4096x4096, fused gate/up+SiLU 2 x 4096x11008,
down_proj 11008x4096], 4-bit, groupsize 128; synthetic random-init packed weights),
issued through the C ABI (include/gptq_mi355x.h) exactly as the drop-in modules issue it after
make_quant_attn / make_fused_mlp: 4 launches per layer, 128 per step, 3.37 GB of distinct
weights per step (> the 256 MiB Infinity Cache, so every step streams from HBM),
issued through the C ABI exactly as the drop-in modules issue it: gptq_layer_forward on handles prepared once at load time
(gptq_layer_prepare builds the stripe16 image from the checkpoint buffers; --kernel rowwave = the split-K kernels on the
checkpoint layout, for A/B runs).  The step is captured once into a hipGraph and replayed; inputs are resident in HBM.

value      = algorithmic GB/s of the whole job (SURVEY 8(d) byte model), all ranks summed.
roofline   = the GEMV kernel family against the 8 TB/s HBM3E spec peak.
cpu_baseline = the C/OpenMP oracle (oracle/gptq_oracle.c, a port of the reference kernel
             arithmetic) on the host cores, on a bounded sample of the same workload.
--gpus 1   = the BASELINE configs[1] pass above (the headline line).
--gpus N>1 = BASELINE config 5 (north_star's multi-GPU mode), the default since round 3: `--tp row --allreduce rccl` -- LLaMA-65B-shaped
             decode linears K-sharded over the N ranks on group boundaries, ONE fp32 all-reduce per linear (RCCL over xGMI, captured into
             the hipGraph), strong scaling; the N-replica figure of the 7B pass (weak scaling, no collective; `--dp` makes it the line)
             rides along as `replicas_reported_only`.  `--tp megatron` N-shards qkv / gate / up and K-shards o / down (2 all-reduces
             per layer); `--allreduce p2p` is the one-shot exchange over IPC peer mappings (csrc/p2p.hip).  quant/tensor_parallel.py is
             the module-level counterpart (world_size-2 gloo tests).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'gptq-for-llama_amd')
sys.path.insert(0, PKG)
sys.path.insert(0, ROOT)

import torch

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

HIDDEN, INTER, LAYERS, BITS, GS = 4096, 11008, 32, 4, 128


def alg_bytes(M, K, N, bits=BITS, gs=GS, nsets=1):
    """qweight + qzeros + scales per weight set, x once, y once."""
    G = -(-K // gs)
    per_set = 4 * (K * bits // 32) * N + 4 * G * (N * bits // 32) + 2 * G * N
    return nsets * per_set + 2 * M * K + 2 * M * N


class PackedSet:
    """random packed weight set on the GPU (uniform bit patterns, scales ~ U(0.001, 0.011))."""

    def __init__(self, K, N, dev, gen):
        G = K // GS
        self.K, self.N = K, N
        self.qweight = torch.randint(-2**31, 2**31 - 1, (K * BITS // 32, N), dtype=torch.int32, device=dev, generator=gen)
        self.qzeros = torch.randint(-2**31, 2**31 - 1, (G, N * BITS // 32), dtype=torch.int32, device=dev, generator=gen)
        self.scales = (torch.rand((G, N), device=dev, generator=gen) * 0.01 + 0.001).half()


class DecodeLinears:
    """the 4 launches/layer x 32 layers of one decode token, as raw C-ABI calls.
    kernel = 'stripe' (default): gptq_layer_forward on handles made by gptq_layer_prepare -- exactly what the drop-in modules
    call (quant/layer.py); at M = 1 its table picks the stripe16 decode kernel (csrc/stripe.hip: no K split, no combine atomics);
    'rowwave': gptq_matmul248_f16 / gptq_fused_mlp_f16 on the checkpoint layout (split-K + fixed-point atomic combine), for A/B."""

    def __init__(self, dev, layers=LAYERS, seed=0, kernel='stripe'):
        from quant import _native, layer as QLayer
        self.native = _native
        self.lib = _native.lib()
        self.dev = dev
        self.kernel = kernel
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        self.layers = []
        for _ in range(layers):
            L = dict(qkv=PackedSet(HIDDEN, 3 * HIDDEN, dev, gen), o=PackedSet(HIDDEN, HIDDEN, dev, gen),
                     gate=PackedSet(HIDDEN, INTER, dev, gen), up=PackedSet(HIDDEN, INTER, dev, gen),
                     down=PackedSet(INTER, HIDDEN, dev, gen))
            if kernel == 'stripe':     # what QuantLinear / QuantLlamaMLP do on their first call: gptq_layer_prepare (quant/layer.py)
                for k in ('qkv', 'o', 'down'):
                    L['pl_' + k] = QLayer.PreparedLayer(((L[k].qweight, L[k].scales, L[k].qzeros, None),), None, BITS, GS, L[k].K, L[k].N)
                L['pl_mlp'] = QLayer.PreparedLayer(((L['gate'].qweight, L['gate'].scales, L['gate'].qzeros, None),
                                                    (L['up'].qweight, L['up'].scales, L['up'].qzeros, None)), None, BITS, GS, HIDDEN, INTER)
                assert all(L[k].stripe is not None for k in ('pl_qkv', 'pl_o', 'pl_down', 'pl_mlp'))
            self.layers.append(L)
        self.lws = torch.zeros(self.lib.gptq_layer_workspace_bytes(), dtype=torch.uint8, device=dev)   # persistent workspace of gptq_layer_forward
        self.x_h = torch.randn((1, HIDDEN), device=dev, generator=gen).half()
        self.x_i = (torch.randn((1, INTER), device=dev, generator=gen) * 0.5).half()
        self.y_qkv = torch.empty((1, 3 * HIDDEN), dtype=torch.float16, device=dev)
        self.y_h = torch.empty((1, HIDDEN), dtype=torch.float16, device=dev)
        self.y_i = torch.empty((1, INTER), dtype=torch.float16, device=dev)
        self.ws = _native.workspace(torch.device(dev))
        self.bytes_per_step = layers * (alg_bytes(1, HIDDEN, 3 * HIDDEN) + alg_bytes(1, HIDDEN, HIDDEN) +
                                        alg_bytes(1, HIDDEN, INTER, nsets=2) + alg_bytes(1, INTER, HIDDEN))
        self.launches_per_step = 4 * layers

    def _layer(self, x, pl, y, stream):
        """the product's ONE call site (include/gptq_mi355x.h "Prepared layers"): the M -> kernel table is inside"""
        rc = self.lib.gptq_layer_forward(pl.handle, x.data_ptr(), pl.K, y.data_ptr(), pl.N, 1, self.lws.data_ptr(), self.lws.numel(), None, 0, stream)
        self.native.check(rc, 'gptq_layer_forward')

    def _mm(self, x, L, name, y, stream):
        w = L[name]
        if self.kernel == 'stripe':
            return self._layer(x, L['pl_' + name], y, stream)
        rc = self.lib.gptq_matmul248_f16(x.data_ptr(), w.K, w.qweight.data_ptr(), w.scales.data_ptr(), w.qzeros.data_ptr(),
                                         None, None, y.data_ptr(), w.N, 1, w.K, w.N, BITS, GS, self.ws.data_ptr(),
                                         self.ws.numel(), stream)
        self.native.check(rc, 'gptq_matmul248_f16')

    def _mlp(self, x, L, y, stream):
        g, u = L['gate'], L['up']
        if self.kernel == 'stripe':
            return self._layer(x, L['pl_mlp'], y, stream)
        rc = self.lib.gptq_fused_mlp_f16(x.data_ptr(), g.K, g.qweight.data_ptr(), g.scales.data_ptr(), g.qzeros.data_ptr(),
                                         None, u.qweight.data_ptr(), u.scales.data_ptr(), u.qzeros.data_ptr(), None,
                                         y.data_ptr(), g.N, 1, g.K, g.N, BITS, GS, self.ws.data_ptr(), self.ws.numel(), stream)
        self.native.check(rc, 'gptq_fused_mlp_f16')

    def step(self):
        s = torch.cuda.current_stream().cuda_stream
        for L in self.layers:
            self._mm(self.x_h, L, 'qkv', self.y_qkv, s)
            self._mm(self.x_h, L, 'o', self.y_h, s)
            self._mlp(self.x_h, L, self.y_i, s)
            self._mm(self.x_i, L, 'down', self.y_h, s)

    def per_shape(self, reps=20):
        """event-timed launches per shape, rotating over the 32 layers' distinct weights (cold)."""
        out = {}
        cs = lambda: torch.cuda.current_stream().cuda_stream   # the capture stream inside torch.cuda.graph
        legs = {
            'qkv_4096x12288': (lambda L: self._mm(self.x_h, L, 'qkv', self.y_qkv, cs()), alg_bytes(1, HIDDEN, 3 * HIDDEN)),
            'o_4096x4096': (lambda L: self._mm(self.x_h, L, 'o', self.y_h, cs()), alg_bytes(1, HIDDEN, HIDDEN)),
            'gate_up_silu_2x4096x11008': (lambda L: self._mlp(self.x_h, L, self.y_i, cs()), alg_bytes(1, HIDDEN, INTER, nsets=2)),
            'down_11008x4096': (lambda L: self._mm(self.x_i, L, 'down', self.y_h, cs()), alg_bytes(1, INTER, HIDDEN)),
        }
        for name, (fn, nbytes) in legs.items():
            g = torch.cuda.CUDAGraph()
            for L in self.layers:
                fn(L)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                for L in self.layers:
                    fn(L)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (reps * len(self.layers))
            out[name] = {'us_per_launch': round(us, 3), 'GBps': round(nbytes / us / 1e3, 1),
                         'frac_of_8TBps': round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}
        return out


def larger_model_shapes(dev, reps=5):
    """side leg (reported only): the same decode kernel on LLaMA-65B-shaped layers (BASELINE config 5 shapes on ONE GPU),
    cold weights (rotation over > 256 MiB of distinct sets inside one hipGraph)."""
    from quant import _native, quant_linear
    lib = _native.lib()
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    H65, I65 = 8192, 22016
    out = {}
    for name, K, N, fused in [('qkv_8192x24576', H65, 3 * H65, False), ('o_8192x8192', H65, H65, False),
                              ('gate_up_silu_2x8192x22016', H65, I65, True), ('down_22016x8192', I65, H65, False)]:
        nb = alg_bytes(1, K, N, nsets=2 if fused else 1)
        nsets = int(300e6 // nb) + 1
        sts = []
        for _ in range(nsets):
            w = PackedSet(K, N, dev, gen)
            u = PackedSet(K, N, dev, gen) if fused else None
            sts.append(quant_linear.stripe_copy(w.qweight, w.scales, w.qzeros, BITS, GS, up=(u.qweight, u.scales, u.qzeros) if fused else None))
            torch.cuda.synchronize()
            del w, u
        x = torch.randn((1, K), device=dev, generator=gen).half()
        y = torch.empty((1, N), dtype=torch.float16, device=dev)

        def launch(i):
            st = sts[i]
            rc = lib.gptq_stripe_matvec_f16(x.data_ptr(), K, st.data_ptr(), st.numel(), None, y.data_ptr(), N, 1, K, N, BITS, GS,
                                            2 if fused else 1, None, 0.0, None, torch.cuda.current_stream().cuda_stream)
            _native.check(rc, name)
        for i in range(nsets):
            launch(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nsets):
                launch(i)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * nsets)
        out[name] = {'us_per_launch': round(us, 3), 'GBps': round(nb / us / 1e3, 1), 'frac_of_8TBps': round(nb / us / 1e3 / HBM_PEAK_GBS, 4)}
        del sts, g
        torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------------------
# BASELINE config 5: LLaMA-65B-shaped 4-bit g128 decode linears, sharded over the ranks (--tp row | megatron)
# ------------------------------------------------------------------------------------------------------------
H65, I65 = 8192, 22016


class TPLayers:
    """this rank's shard of `layers` LLaMA-65B-shaped decoder layers (random packed weights), as stripe16 images.
      mode 'row'      north_star's literal layout: EVERY linear K-(row-)sharded on group boundaries (22016 = 172 groups:
                      22,22,22,22,21,21,21,21 over 8 ranks), fp32 partials, ONE all-reduce per linear (4 per layer; gate and up
                      share one), fp16 rounding and SiLU after the reduce
      mode 'megatron' qkv and gate/up N-(column-)sharded (no collective, SiLU fused), o and down K-sharded: 2 all-reduces per layer
    The collective is torch.distributed.all_reduce (RCCL over xGMI on MI355X; gloo in the CPU tests)."""

    def __init__(self, dev, rank, world, mode, layers, seed=0, allreduce='rccl'):
        from quant import _native, quant_linear, tensor_parallel as TP
        self.native, self.lib, self.QL = _native, _native.lib(), quant_linear
        self.dev, self.rank, self.world, self.mode = dev, rank, world, mode
        self.p2p = None
        if allreduce == 'p2p' and world > 1:       # one-shot exchange through IPC peer mappings (csrc/p2p.hip) instead of RCCL
            from quant.p2p import P2PAllReduce
            self.p2p = P2PAllReduce(2 * I65)
        gen = torch.Generator(device=dev)
        gen.manual_seed(1000 + seed)            # every rank draws the same full matrices, then keeps its slice
        kb_h = TP.row_shard_bounds(H65, GS, BITS, world)[rank]
        kb_i = TP.row_shard_bounds(I65, GS, BITS, world)[rank]
        nb_qkv = TP.col_shard_bounds(3 * H65, world)[rank]
        nb_i = TP.col_shard_bounds(I65, world)[rank]
        self.kb_h, self.kb_i, self.nb_qkv, self.nb_i = kb_h, kb_i, nb_qkv, nb_i

        def k_shard(w, kb):
            r0, r1, g0, g1 = kb[0] // 8, kb[1] // 8, kb[0] // GS, kb[1] // GS
            return w.qweight[r0:r1].contiguous(), w.scales[g0:g1].contiguous(), w.qzeros[g0:g1].contiguous()

        def n_shard(w, nb):
            return (w.qweight[:, nb[0]:nb[1]].contiguous(), w.scales[:, nb[0]:nb[1]].contiguous(),
                    w.qzeros[:, nb[0] // 8:nb[1] // 8].contiguous())
        self.layers = []
        for _ in range(layers):
            full = dict(qkv=PackedSet(H65, 3 * H65, dev, gen), o=PackedSet(H65, H65, dev, gen), gate=PackedSet(H65, I65, dev, gen),
                        up=PackedSet(H65, I65, dev, gen), down=PackedSet(I65, H65, dev, gen))
            L = {}
            if mode == 'row':
                parts = {k: k_shard(full[k], kb_i if k == 'down' else kb_h) for k in full}
            else:
                parts = dict(qkv=n_shard(full['qkv'], nb_qkv), gate=n_shard(full['gate'], nb_i), up=n_shard(full['up'], nb_i),
                             o=k_shard(full['o'], kb_h), down=k_shard(full['down'], kb_i))
            for k in ('qkv', 'o', 'down'):
                L[k] = quant_linear.stripe_copy(*parts[k], BITS, GS)
            L['mlp'] = quant_linear.stripe_copy(*parts['gate'], BITS, GS, up=parts['up'])
            torch.cuda.synchronize()
            self.layers.append(L)
            del full, parts
        f16, f32 = dict(dtype=torch.float16, device=dev), dict(dtype=torch.float32, device=dev)
        self.x_h = torch.randn((1, H65), device=dev, generator=gen).half()
        self.x_i = (torch.randn((1, I65), device=dev, generator=gen) * 0.5).half()
        self.p_qkv, self.p_h, self.p_mlp = torch.empty((1, 3 * H65), **f32), torch.empty((1, H65), **f32), torch.empty((2, I65), **f32)
        self.y_qkv, self.y_h, self.y_i = torch.empty((1, 3 * H65), **f16), torch.empty((1, H65), **f16), torch.empty((1, I65), **f16)
        self.y_qkv_loc = torch.empty((1, nb_qkv[1] - nb_qkv[0]), **f16)
        self.y_i_loc = torch.empty((1, nb_i[1] - nb_i[0]), **f16)
        # algorithmic bytes of the FULL (unsharded) layer stack: the whole job moves them once per step
        self.bytes_per_step = layers * (alg_bytes(1, H65, 3 * H65) + alg_bytes(1, H65, H65) + alg_bytes(1, H65, I65, nsets=2) +
                                        alg_bytes(1, I65, H65))
        self.collectives_per_step = layers * (4 if mode == 'row' else 2)
        self.launches_per_step = 4 * layers

    def _partial(self, x, st, out32, K, N, nsets):
        rc = self.lib.gptq_stripe_matvec_partial_f32(x.data_ptr(), st.data_ptr(), st.numel(), out32.data_ptr(), K, N, BITS, GS, nsets, None,
                                                     torch.cuda.current_stream().cuda_stream)
        self.native.check(rc, 'gptq_stripe_matvec_partial_f32')

    def _full(self, x, st, out16, K, N, nsets):
        rc = self.lib.gptq_stripe_matvec_f16(x.data_ptr(), K, st.data_ptr(), st.numel(), None, out16.data_ptr(), N, 1, K, N, BITS, GS, nsets, None,
                                             0.0, None, torch.cuda.current_stream().cuda_stream)
        self.native.check(rc, 'gptq_stripe_matvec_f16')

    def _reduce(self, t):
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t)

    def _reduce_round(self, part, out16):
        """sum over the ranks, then the ONE fp16 rounding"""
        if self.p2p is not None:
            self.p2p.allreduce(part, out=out16)      # exchange, rank-ordered sum and rounding in one launch
        else:
            self._reduce(part)
            out16.copy_(part)

    def step(self):
        kh = slice(*self.kb_h)
        ki = slice(*self.kb_i)
        Kh, Ki = self.kb_h[1] - self.kb_h[0], self.kb_i[1] - self.kb_i[0]
        for L in self.layers:
            if self.mode == 'row':
                self._partial(self.x_h[:, kh], L['qkv'], self.p_qkv, Kh, 3 * H65, 1)
                self._reduce_round(self.p_qkv, self.y_qkv)
                self._partial(self.x_h[:, kh], L['o'], self.p_h, Kh, H65, 1)
                self._reduce_round(self.p_h, self.y_h)
                self._partial(self.x_h[:, kh], L['mlp'], self.p_mlp, Kh, I65, 2)
                if self.p2p is not None:
                    self.p2p.allreduce_silu_mul(self.p_mlp, out=self.y_i)
                else:
                    self._reduce(self.p_mlp)
                    torch.mul(torch.nn.functional.silu(self.p_mlp[0:1]), self.p_mlp[1:2], out=self.p_mlp[0:1])   # fp32, fused_mlp.py:160-166
                    self.y_i.copy_(self.p_mlp[0:1])
                self._partial(self.x_i[:, ki], L['down'], self.p_h, Ki, H65, 1)
                self._reduce_round(self.p_h, self.y_h)
            else:
                self._full(self.x_h, L['qkv'], self.y_qkv_loc, H65, self.nb_qkv[1] - self.nb_qkv[0], 1)     # this rank's heads
                self._partial(self.x_h[:, kh], L['o'], self.p_h, Kh, H65, 1)
                self._reduce_round(self.p_h, self.y_h)
                self._full(self.x_h, L['mlp'], self.y_i_loc, H65, self.nb_i[1] - self.nb_i[0], 2)           # SiLU fused: columns are local
                self._partial(self.x_i[:, ki], L['down'], self.p_h, Ki, H65, 1)
                self._reduce_round(self.p_h, self.y_h)


def tp_world1_leg(dev, mode, layers, steps=20):
    """The like-for-like origin of the --gpus N > 1 curve: the SAME LLaMA-65B-shaped stack (same layers, same launches, fp32 partials,
    fp16 rounding after the would-be reduce) on ONE GPU with nothing sharded and no collective."""
    work = TPLayers(dev, 0, 1, mode, layers)
    for _ in range(2):
        work.step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        work.step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {'workload': 'the same %d-layer LLaMA-65B-shaped stack, unsharded on one GPU (world 1), hipGraph replay' % layers,
           'GBps': round(work.bytes_per_step * steps / dt / 1e9, 1), 'ms_per_step': round(dt * 1e3 / steps, 4), 'n_gpus': 1,
           'frac_of_8TBps': round(work.bytes_per_step * steps / dt / 1e9 / HBM_PEAK_GBS, 4)}
    del g, work
    torch.cuda.empty_cache()
    return out


def allreduce_latency_us(dev, world, nfloats, reps=50, p2p=None):
    """mean us of one fp32 all-reduce of nfloats elements, back to back on the stream (reported next to the TP line)."""
    if world == 1:
        return 0.0
    import torch.distributed as dist
    t = torch.zeros(nfloats, dtype=torch.float32, device=dev)
    o = torch.zeros(nfloats, dtype=torch.float16, device=dev)
    one = (lambda: p2p.allreduce(t, out=o)) if p2p is not None else (lambda: dist.all_reduce(t))
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        one()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps



def _time_cold(run, nsets, reps=5):
    """us per launch of run(i), i rotating over nsets distinct weight sets inside one hipGraph (cold weights)."""
    for i in range(nsets):
        run(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(nsets):
            run(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * nsets)


def prefill_leg(dev, M=65536, reps=7):
    """BASELINE config 3 (reported only): LLaMA-7B-shaped 4-bit g128 batched matmul at M = 32 x 2048 through the drop-in
    matmul248 (reference kernel quant_linear.py:72-137), TFLOP/s = 2 M N K / t against the 2.5 PFLOP/s dense fp16 MFMA peak.
    Per shape: the product (gptq_layer_forward -> dequantise per call + the hand-written LDS-DMA / MFMA tile GEMM of
    csrc/gemm8.hip), the same entry with the library switch (dequantise per call + hipBLASLt: the reported ceiling, not the
    product), the round-2 fused tile kernel (csrc/gemm_mfma.hip, family='abi'), and hipBLASLt alone on a weight dequantised
    beforehand."""
    from quant import _native, quant_linear as QL
    lib = _native.lib()
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    out = {'M': M, 'peak_TFLOPs': 2500.0, 'route': QL.PREFILL_ROUTE, 'gemm': 'hand-written tile GEMM (csrc/gemm8.hip): LDS-DMA operands, v_mfma_f32_16x16x32_f16, '
           '8-phase schedule; hipBLASLt only below one round of tiles', 'shapes': {}}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(f):
        """median of `reps` individually timed calls (ms-scale kernels: an event pair per call is exact, and a clock / power
        transient on the box -- seen once as a 2.5x slower stretch -- moves single samples, not the median)"""
        f()
        y = f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0.record()
            y = f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2], y

    def with_route(code, f):
        prev = lib.gptq_set_prefill_route(code)
        try:
            return timed(f)
        finally:
            lib.gptq_set_prefill_route(prev)

    for K, N in [(HIDDEN, HIDDEN), (HIDDEN, 3 * HIDDEN), (HIDDEN, INTER), (INTER, HIDDEN)]:
        w = PackedSet(K, N, dev, gen)
        x = torch.randn((M, K), device=dev, generator=gen).half()
        gi = (torch.arange(K, device=dev) // GS).to(torch.int32)
        prod = lambda: QL.matmul248(x, w.qweight, w.scales, w.qzeros, gi, BITS, 15)
        if not out['shapes']:
            # the legs before this one are microsecond kernels: ~50 ms of GEMM first, so that the first timed leg does not start
            # from their power state (leg-to-leg noise on these boxes stays around +-10 %: profiles/r2e_prefill)
            for _ in range(24):
                prod()
        assert lib.gptq_prefill_route_for(M, K, N, 1, 0) == (1 if QL.PREFILL_ROUTE != 'library' else 0)
        ms, y = timed(prod)
        msl, yl = with_route(0, prod)
        msf, yf = timed(lambda: QL.matmul248(x, w.qweight, w.scales, w.qzeros, gi, BITS, 15, family='abi'))
        W = QL.dequantize(w.qweight, w.scales, w.qzeros, None, BITS, GS)
        msd, yd = timed(lambda: x @ W)
        fl = 2.0 * M * N * K / 1e9
        tf, tfl, tff, tfd = fl / ms, fl / msl, fl / msf, fl / msd
        out['shapes']['%dx%d' % (K, N)] = {'ms': round(ms, 3), 'TFLOPs': round(tf, 1), 'frac_of_2.5PF': round(tf / 2500.0, 4),
                                           'library_route_TFLOPs': round(tfl, 1), 'round2_fused_kernel_TFLOPs': round(tff, 1),
                                           'hipblaslt_dense_TFLOPs': round(tfd, 1), 'vs_library_route': round(tf / tfl, 3),
                                           'vs_hipblaslt_dense': round(tf / tfd, 3),
                                           'max_abs_diff_vs_dense': float((y.float() - yd.float()).abs().max()),
                                           'max_abs_diff_vs_library_route': float((y.float() - yl.float()).abs().max())}
        del w, x, y, yl, yf, yd, W
        torch.cuda.empty_cache()
    # the MLP's gate/up pair with SiLU (fused_mlp.fused_gate_up; reference fusedmatmul_248_kernel, fused_mlp.py:84-168)
    from quant import fused_mlp as FM
    K, N = HIDDEN, INTER
    wg, wu = PackedSet(K, N, dev, gen), PackedSet(K, N, dev, gen)
    x = torch.randn((M, K), device=dev, generator=gen).half()
    gi = (torch.arange(K, device=dev) // GS).to(torch.int32)
    pair = lambda: FM.fused_gate_up(x, (wg.qweight, wg.scales, wg.qzeros, gi), (wu.qweight, wu.scales, wu.qzeros, gi), BITS, GS)
    ms, c = timed(pair)
    msl, cl = with_route(0, pair)
    fl = 4.0 * M * N * K / 1e9
    out['gate_up_silu_2x%dx%d' % (K, N)] = {'ms': round(ms, 3), 'TFLOPs': round(fl / ms, 1), 'library_route_TFLOPs': round(fl / msl, 1),
                                           'note': 'product: ONE launch, SiLU on the fp32 accumulators in the GEMM epilogue; library route: fp32 '
                                                   '[rows, 2N] product in chunks + a SiLU * mul pass',
                                           'max_abs_diff_between_routes': float((c.float() - cl.float()).abs().max())}
    del wg, wu, x, c, cl
    torch.cuda.empty_cache()
    return out


def small_batch_leg(dev):
    """Decode batches of 2 .. 128 rows (reported only; reference forward for every M: quant_linear.py:415-419) through the
    drop-in matmul248 on the four LLaMA-7B shapes, cold weights: M <= 4 share the stripe16 decode launch, 5 .. 8 its row
    groups, above that the 16-row MFMA tiles of csrc/stripe_mm.inc (one launch or K slices + reduce kernel; no atomics)."""
    from quant import quant_linear as QL
    gen = torch.Generator(device=dev)
    gen.manual_seed(6)
    out = {'unit': 'us per launch (hipGraph, cold weights)', 'shapes': {}}
    for K, N in [(HIDDEN, HIDDEN), (HIDDEN, 3 * HIDDEN), (HIDDEN, INTER), (INTER, HIDDEN)]:
        nsets = int(300e6 // alg_bytes(1, K, N)) + 1
        sets = [PackedSet(K, N, dev, gen) for _ in range(nsets)]
        gi = (torch.arange(K, device=dev) // GS).to(torch.int32)
        row = {}
        for M in (1, 4, 8, 16, 32, 64, 128):
            x = torch.randn((M, K), device=dev, generator=gen).half()

            def run(i):
                w = sets[i]
                QL.matmul248(x, w.qweight, w.scales, w.qzeros, gi, BITS, 15)
            us = _time_cold(run, nsets, reps=3)
            row['M%d' % M] = {'us': round(us, 2), 'TFLOPs': round(2.0 * M * K * N / us / 1e6, 1), 'GBps': round(alg_bytes(M, K, N) / us / 1e3, 1)}
        out['shapes']['%dx%d' % (K, N)] = row
        del sets
    # the gate | up pair of the fused MLP (fused_mlp.py:84-168: one launch, SiLU(gate) * up on the fp32 sums): 5 .. 16 rows in the decode launch / one row
    # tile, 17 .. 128 the loader / consumer kernel with the consumers split by set (csrc/stripe_mm.inc stripe_mmr_kernel, round 6)
    from quant import fused_mlp as FM
    K, N = HIDDEN, INTER
    npairs = int(300e6 // alg_bytes(1, K, N, nsets=2)) + 1
    pairs = [(PackedSet(K, N, dev, gen), PackedSet(K, N, dev, gen)) for _ in range(npairs)]
    gi = (torch.arange(K, device=dev) // GS).to(torch.int32)
    row = {}
    for M in (1, 4, 8, 16, 32, 64, 128):
        x = torch.randn((M, K), device=dev, generator=gen).half()

        def run_pair(i):
            a, b = pairs[i]
            FM.fused_gate_up(x, (a.qweight, a.scales, a.qzeros, gi), (b.qweight, b.scales, b.qzeros, gi), BITS, GS)
        us = _time_cold(run_pair, npairs, reps=3)
        row['M%d' % M] = {'us': round(us, 2), 'TFLOPs': round(4.0 * M * K * N / us / 1e6, 1), 'GBps': round(alg_bytes(M, K, N, nsets=2) / us / 1e3, 1)}
    out['shapes']['gate_up_silu_2x%dx%d' % (K, N)] = row
    del pairs
    return out


def prompt_leg(dev):
    """A single request's prompt (reported only): batches of 256 .. 3072 rows through gptq_layer_forward on the LLaMA-7B shapes -- the
    fused-dequantise tile GEMM on the stripe16 image (csrc/stripe_mm.inc stripe_gemm_kernel, the product for 129 .. 2048 rows), above
    that dequantise per call + the own tile GEMM (csrc/gemm8.hip) -- against the dense route forced, and against hipBLASLt (reported
    ceiling: since round 4 the default route never reaches the library for K % 128 == 0) on the same prepared layer; us per call from a hipGraph of 8
    calls, TFLOP/s = 2 M N K / t (reference kernel for every M: quant_linear.py:72-137, fused_mlp.py:84-168)."""
    from quant import _native, layer as QLayer
    lib = _native.lib()
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    out = {'unit': 'us per call (hipGraph of 8 calls, warm weights)', 'shapes': {}}

    def timed(fn, calls=8):
        fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(calls):
                fn()
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / calls)
        return sorted(ts)[2]

    for K, N, pair in [(HIDDEN, HIDDEN, False), (HIDDEN, 3 * HIDDEN, False), (INTER, HIDDEN, False), (HIDDEN, INTER, True)]:
        sets = tuple((w.qweight, w.scales, w.qzeros, None) for w in (PackedSet(K, N, dev, gen), PackedSet(K, N, dev, gen))[:2 if pair else 1])
        pl = QLayer.PreparedLayer(sets, None, BITS, GS, K, N)
        row = {}
        for M in (256, 512, 1024, 2048, 3072):
            x = torch.randn((M, K), device=dev, generator=gen).half()
            y = torch.empty((M, N), dtype=torch.float16, device=dev)
            prev = lib.gptq_set_stripe_gemm_max_rows(0)
            t_dense = timed(lambda: pl.forward(x, y))              # dequantise per call + the own tile GEMM (csrc/gemm8.hip)
            prev_route = lib.gptq_set_prefill_route(0)
            t_lib = timed(lambda: pl.forward(x, y))                # dequantise per call + hipBLASLt: the reported ceiling, off the default route
            lib.gptq_set_prefill_route(prev_route)
            lib.gptq_set_stripe_gemm_max_rows(prev)
            route = lib.gptq_layer_route_for(pl.handle, M)
            t = timed(lambda: pl.forward(x, y))                    # the product's own choice
            fl = (4.0 if pair else 2.0) * M * N * K
            row['M%d' % M] = {'us': round(t, 1), 'TFLOPs': round(fl / t / 1e6, 1), 'route': {3: 'fused tile GEMM on the image', 4: 'dequantise + gemm8', 5: 'library'}.get(route, route),
                              'dense_route_us': round(t_dense, 1), 'vs_dense_route': round(t_dense / t, 2), 'library_route_us': round(t_lib, 1),
                              'vs_library_route': round(t_lib / t, 2)}
        out['shapes'][('gate_up_silu_2x%dx%d' if pair else '%dx%d') % (K, N)] = row
        del pl, sets
    return out


def config4_leg(dev):
    """BASELINE config 4 (reported only): LLaMA-7B-shaped 3-bit no-group and 4-bit g128 act-order, batch 1, cold weights,
    through the drop-in matmul248 (3-bit: its own stripe16 image, dwordx3 per 32 k -- an extension, the reference raises for bits == 3,
    quant_linear.py:308-309; act-order: image of the rows sorted by group at load + decode kernel with the x gather fused; round 4: 3-bit
    g128 act-order, the README's `--wbits 3 --groupsize 128 --act-order` flavour, on the same path)."""
    from quant import quant_linear as QL
    gen = torch.Generator(device=dev)
    gen.manual_seed(4)

    def make(bits, gs, K, N, act):
        G = 1 if gs == -1 else K // gs
        qw = torch.randint(-2**31, 2**31 - 1, (K // 32 * bits, N), dtype=torch.int32, device=dev, generator=gen)
        qz = torch.randint(-2**31, 2**31 - 1, (G, N // 32 * bits), dtype=torch.int32, device=dev, generator=gen)
        sc = (torch.rand((G, N), device=dev, generator=gen) * 0.01 + 0.001).half()
        gi = (torch.arange(K, device=dev) // (K if gs == -1 else gs)).to(torch.int32)
        if act:
            gi = gi[torch.argsort(torch.randperm(K, device=dev, generator=gen))].contiguous()
        return qw, sc, qz, gi

    out = {}
    for label, bits, gs, act in [('w3_nogroup', 3, -1, False), ('w4_g128_act_order', 4, 128, True), ('w3_g128_act_order', 3, 128, True)]:
        out[label] = {'parity': 'unpinned (the reference raises NotImplementedError for bits == 3, quant_linear.py:308-309: no reference output exists; '
                               'checked against the oracle\'s own 3-bit restatement and the float64 product at these sizes, tests/test_gpu_parity.py '
                               'test_config4_full_size_batch1)' if bits == 3 else
                               'pinned (golden fwd_w4g128_act_* from the reference kernel; these sizes vs the oracle in test_config4_full_size_batch1)'}
        for K, N in [(HIDDEN, HIDDEN), (HIDDEN, 3 * HIDDEN), (INTER, HIDDEN), (HIDDEN, INTER)]:
            G = 1 if gs == -1 else K // gs
            nb = 4 * (K * bits // 32) * N + 4 * G * (N * bits // 32) + 2 * G * N + 2 * K + 2 * N + (4 * K if act else 0)
            nsets = int(300e6 // nb) + 1
            sets = [make(bits, gs, K, N, act) for _ in range(nsets)]
            x = torch.randn((1, K), device=dev, generator=gen).half()
            us = _time_cold(lambda i: QL.matmul248(x, sets[i][0], sets[i][1], sets[i][2], sets[i][3], bits, 2**bits - 1), nsets)
            out[label]['%dx%d' % (K, N)] = {'us_per_launch': round(us, 3), 'GBps': round(nb / us / 1e3, 1),
                                            'frac_of_8TBps': round(nb / us / 1e3 / HBM_PEAK_GBS, 4)}
            del sets
            torch.cuda.empty_cache()
    return out


def csrc_sha16():
    """identity of the kernels a measurement belongs to: sha256 over the device sources (csrc/*.hip, *.inc, *.h) in name order.  The GPU box
    has no .git (gpurun ships a snapshot), so this -- not a commit id -- is what ties a PMC pass to the build it was taken on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(PKG, 'csrc', '*'))):
        if f.endswith(('.hip', '.inc', '.h')):
            h.update(os.path.basename(f).encode())
            h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(pmc_file=None):
    """HBM bytes per launch from a PMC pass (traffic.json, produced by tools/pmc_traffic.py from `rocprofv3 --pmc FETCH_SIZE` on this same
    command; counters cannot be read from inside the timed process, so this is a separate-pass figure).  --pmc-file names the pass taken in
    the same validation call; otherwise the newest committed profiles/*/traffic.json.  A pass stamped with the kernel sources of ANOTHER
    build (csrc_sha16) is refused: (None, reason)."""
    import glob
    files = [pmc_file] if pmc_file else sorted(glob.glob(os.path.join(ROOT, 'profiles', '*', 'traffic.json')))
    if not files or not os.path.exists(files[-1]):
        return None, None
    try:
        d = json.load(open(files[-1]))
        rel = os.path.relpath(files[-1], ROOT)
        stamp = d.get('csrc_sha16')
        if stamp is not None and stamp != csrc_sha16():
            return None, '%s refused: taken on kernel sources %s, this build is %s' % (rel, stamp, csrc_sha16())
        src = rel + (' (same kernel sources: csrc_sha16 %s)' % stamp if stamp else ' (unstamped: taken before round 5)')
        return int(d['hbm_bytes_per_launch_avg']), src
    except Exception:
        return None, None


def cpu_baseline(budget_s=10.0):
    """the oracle (port of the reference kernel arithmetic) on the host cores: one decoder layer's
    five matvecs (qkv as one 4096x12288, o, gate, up, down), repeated until ~budget_s."""
    import numpy as np
    from oracle import oracle
    rng = np.random.default_rng(0)

    def rand_set(K, N):
        G = K // GS
        return (rng.integers(-2**31, 2**31, size=(K * BITS // 32, N), dtype=np.int64).astype(np.int32),
                rng.uniform(0.001, 0.011, size=(G, N)).astype(np.float16),
                rng.integers(-2**31, 2**31, size=(G, N * BITS // 32), dtype=np.int64).astype(np.int32),
                oracle.trivial_g_idx(K, GS))

    shapes = [(HIDDEN, 3 * HIDDEN), (HIDDEN, HIDDEN), (HIDDEN, INTER), (HIDDEN, INTER), (INTER, HIDDEN)]
    sets = [rand_set(K, N) for K, N in shapes]
    xs = {HIDDEN: rng.standard_normal((1, HIDDEN)).astype(np.float16), INTER: rng.standard_normal((1, INTER)).astype(np.float16)}
    nbytes = sum(alg_bytes(1, K, N) for K, N in shapes)
    t_total, n = 0.0, 0
    while t_total < budget_s and n < 50:
        t0 = time.perf_counter()
        for (K, N), (qw, sc, qz, gi) in zip(shapes, sets):
            oracle.matmul248(xs[K], qw, sc, qz, gi, BITS)
        t_total += time.perf_counter() - t0
        n += 1
    return {'value': round(nbytes * n / t_total / 1e9, 3), 'unit': 'GB/s', 'cores': oracle.num_threads(), 'kind': 'port',
            'note': 'faithful, unoptimised restatement of the reference kernel arithmetic (software fp16 rounding per weight, -ffp-contract=off): '
                    'a stated baseline, not a target -- the GPU / CPU ratio says nothing about kernel quality, roofline.frac does',
            'sample': '%d x one LLaMA-7B decoder layer (5 matvecs, %.1f MB algorithmic) by oracle/gptq_oracle.c (OpenMP)' %
                      (n, nbytes / 1e6), 'seconds': round(t_total, 2)}


def gptq_loop_baseline(dev):
    """north_star: "the GPTQ column-wise Hessian loop in gptq.py is left to the CPU reference and timed on the host
    cores of the GPU box in the same run as the reported-only baseline".  /root/reference does not travel, so the timed
    code is its restatement oracle/gptq_solver.py (numpy + LAPACK, pinned to the reference's own GPTQ class by
    tests/golden/gptq_*.npz): one 4096x4096 layer, 4-bit g128, Hessian from 2 x 2048 random tokens.
    The MI355X solver (gptq-for-llama_amd/gptq.py, one HIP launch per 128-column block) runs on the same data for
    scale.  Reported only -- not part of `value`."""
    import numpy as np
    from oracle import gptq_solver as G
    rng = np.random.default_rng(0)
    K = N = HIDDEN
    W = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    X = [rng.standard_normal((1, 2048, K)).astype(np.float32) for _ in range(2)]
    t0 = time.perf_counter()
    H, n = np.zeros((K, K), np.float32), 0
    for b in X:
        H, n = G.hessian_add_batch(H, n, b)
    t1 = time.perf_counter()
    _, _, _, _, err_cpu = G.fasterquant(W, H, BITS, 128, 0.01, GS, False, False)
    t2 = time.perf_counter()
    out = {'layer': '4096x4096 nn.Linear -> 4-bit g128, Hessian from 2 x 2048 tokens', 'kind': 'port',
           'cpu_add_batch_s': round(t1 - t0, 3), 'cpu_fasterquant_s': round(t2 - t1, 3), 'cpu_loss': round(err_cpu, 2),
           'cores': os.cpu_count()}
    try:
        import gptq as product_gptq
        layer = torch.nn.Linear(K, N, bias=False)
        layer.weight.data = torch.from_numpy(W)
        layer = layer.to(dev)
        xs = [torch.from_numpy(b).to(dev) for b in X]
        for timed in (False, True):         # first pass warms rocSOLVER / hipBLASLt up
            layer.weight.data = torch.from_numpy(W).to(dev)
            g = product_gptq.GPTQ(layer)
            g.quantizer.configure(BITS, perchannel=True, sym=False, mse=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in xs:
                g.add_batch(b, None)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                _, _, _, err_gpu = g.fasterquant(blocksize=128, percdamp=0.01, groupsize=GS, actorder=False, name='bench')
            t2 = time.perf_counter()
            g.free()
        out.update({'gpu_add_batch_s': round(t1 - t0, 4), 'gpu_fasterquant_s': round(t2 - t1, 4), 'gpu_loss': round(err_gpu, 2)})
    except Exception as e:
        out['gpu_error'] = repr(e)[:200]
    return out


def decode_tokens_per_s(dev, tokens=64):
    """full decode step (norms, fused qkv + RoPE, KV cache, attention, o_proj, fused MLP, lm_head) on a
    random-init LLaMA-7B-shaped model built from the drop-in modules exactly as load_quant() builds it.
      hf_eager         the module chain launch by launch (engine hook off) -- protocol of benchmark(), llama.py:385-438
      drop_in_forward  the same protocol through the public surface: model(input_ids[:, i:i+1], past_key_values=cache);
                       one-token forwards are answered by the hipGraph decode engine (quant/engine_hook.py)
      drop_in_generate model.generate(...) as llama_inference.py:119-127 calls it
      engine_graph     quant.decode.DecodeEngine driven directly (one hipGraph replay per token)
      engine_graph_bN  DecodeEngine(batch=N): N sequences per replay, aggregate tokens/s
      engine_graph_[bN_]ctxT  the same engines with ~T tokens of history per row (round 6)
      drop_in_generate_sampling  the script's own SAMPLING call, self-feeding sampling graph against HF's loop (own process: sampling_generate_leg)"""
    from quant.decode import build_random_llama, benchmark_decode, benchmark_decode_engine, benchmark_generate, benchmark_decode_engine_context
    import quant
    model = build_random_llama(dev)
    out = {'hf_eager': benchmark_decode(model, 24, engine_hook=False)}
    # A/B leg first, on the untouched model: the engine next to BOTH copies of the packed weights (checkpoint buffers + stripe16 images;
    # what GPTQ_RELEASE_CHECKPOINT=0 keeps).  Same kernels, same tokens/s: the difference is memory.
    out['engine_graph_two_copies'] = benchmark_decode_engine(model, tokens=tokens, graph=True)
    # the product's default from here on: the first decode step through the hook builds the engine and releases the checkpoint buffers
    # (quant/engine_hook.py RELEASE_CHECKPOINT): ONE copy of the packed weights, like the reference's 4891 MiB for 7B 4-bit g128
    # (README.md:26, protocol llama.py:426-438)
    out['drop_in_forward'] = benchmark_decode(model, tokens)
    out['drop_in_generate'] = benchmark_generate(model)
    from quant.engine_hook import drop_decode_engines
    drop_decode_engines(model)     # the hook's engines (1 GB of K/V cache per row + a graph each) would sit next to the ones measured below
    torch.cuda.empty_cache()
    done = sum(1 for m in model.modules() if getattr(m, '_released', None) is not None)
    kept = sum(1 for m in model.modules() if isinstance(m, (quant.QuantLinear, quant.fused_mlp.QuantLlamaMLP)) and getattr(m, '_released', None) is None)
    out['engine_graph'] = dict(benchmark_decode_engine(model, tokens=tokens, graph=True), released_modules=done, kept_modules=kept,
                               reference_published_MiB=4891)
    out['tokens_per_s'] = out['engine_graph']['tokens_per_s']
    # round 5: decode BATCHES -- B sequences per hipGraph replay (linears at M = B: the decode kernel's row groups / 16-row MFMA tiles with
    # norm and residual fused, per-row positions in the attention launch, ONE pass over the LM head for all rows); aggregate tokens/s
    for B in (2, 4, 5, 8, 16):
        torch.cuda.empty_cache()
        out['engine_graph_b%d' % B] = benchmark_decode_engine(model, tokens=32, graph=True, batch=B)
    # round 6: the same engines at DEPTH -- the reference's protocol steps through 2048 tokens and prints the median (llama.py:385-438), i.e. its
    # median token sees ~1000 tokens of context; the legs above start from an empty cache.  engine_graph_ctx{512,1024,2047}: tokens/s with that
    # much history per row (B = 1, and B = 4 rows).
    for B in (1, 4):
        torch.cuda.empty_cache()
        ctx = benchmark_decode_engine_context(model, contexts=(512, 1024, 2047), batch=B)
        for k in ('ctx512', 'ctx1024', 'ctx2047'):
            out['engine_graph_%s%s' % ('' if B == 1 else 'b%d_' % B, k)] = dict(ctx[k], protocol=ctx['protocol'], attention=ctx['attention'])
    # ... and through the reference's own call site: generate on FOUR left-padded prompts -- [4, 1] steps with per-row positions, answered by the hook's
    # DecodeEngine(batch=4) (HF's per-step host work included)
    torch.cuda.empty_cache()
    out['drop_in_generate_b4_left_padded'] = benchmark_generate(model, batch=4, left_pad=True, new_tokens=64)
    drop_decode_engines(model)
    del model
    torch.cuda.empty_cache()
    out['drop_in_generate_sampling'] = sampling_generate_leg()
    return out


_SAMPLING_LEG = r'''
import json, sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
from quant import decode as D, engine_hook as EH
fill = D.fill_random_quant_
def small(layer, gen):
    fill(layer, gen)
    layer.scales.mul_(0.05)
D.fill_random_quant_ = small
m = D.build_random_llama('cuda:0')
ids = torch.randint(1, 32000, (1, 16), device='cuda:0', generator=torch.Generator(device='cuda:0').manual_seed(0))
with torch.no_grad():
    lg = m(ids).logits[0, -1].float()
res = {'logits_finite': bool(torch.isfinite(lg).all())}
if res['logits_finite']:
    def t(n):
        torch.manual_seed(0); torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            out = m.generate(ids, do_sample=True, max_new_tokens=n, top_p=0.95, temperature=0.8, eos_token_id=None)
        torch.cuda.synchronize(); return time.perf_counter() - t0, out
    toks = {}
    for name, fast in (('sampling_graph', True), ('hf_loop', False)):
        EH.SAMPLE_FAST = fast
        t(4); t1, _ = t(1); tn, out = t(128)
        res[name + '_tokens_per_s'] = round(127 / (tn - t1), 1)
        toks[name] = out[0].tolist()
    res['same_tokens'] = toks['sampling_graph'] == toks['hf_loop']
print('SAMPLING_LEG ' + json.dumps(res), flush=True)
'''


def sampling_generate_leg():
    """llama_inference.py:119-127's own call -- model.generate(do_sample=True, top_p=0.95, temperature=0.8) -- on a LLaMA-7B-shaped random model, tokens/s through
    the hook's self-feeding sampling graph (HF's warpers + torch.multinomial captured behind the engine step: quant/engine_hook.py) and through HF's loop, same
    seed, and whether both drew the same tokens.  In a PROCESS of its own, on a model whose random scales are 20 x smaller: the stock random stack overflows fp16
    to NaN logits, on which torch.multinomial aborts the process (in HF's loop as well) -- an abort here must not take the bench line with it."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, '-c', _SAMPLING_LEG % (PKG, ROOT)], capture_output=True, text=True, timeout=600)
        for line in r.stdout.splitlines():
            if line.startswith('SAMPLING_LEG '):
                return dict(json.loads(line[len('SAMPLING_LEG '):]), call='model.generate(input_ids[1, 16], do_sample=True, max_new_tokens=128, top_p=0.95, temperature=0.8) '
                            '(llama_inference.py:119-127), 7B-shaped random model with scales x 0.05 (finite logits), own process')
        return {'error': 'no result (rc %d): %s' % (r.returncode, r.stderr[-300:])}
    except Exception as e:
        return {'error': repr(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-decode', action='store_true')
    ap.add_argument('--no-per-shape', action='store_true')
    ap.add_argument('--eager', action='store_true', help='time eager launches instead of hipGraph replay')
    ap.add_argument('--kernel', choices=('stripe', 'rowwave'), default='stripe', help='decode matvec kernel family (A/B runs)')
    ap.add_argument('--tp', choices=('row', 'megatron'), default=None,
                    help='BASELINE config 5: LLaMA-65B-shaped decode linears sharded over the --gpus ranks (row = every linear K-sharded, one '
                         'all-reduce per linear: north_star, the DEFAULT for more than one rank; megatron = N-shard qkv/gate/up, K-shard o/down)')
    ap.add_argument('--tp-layers', type=int, default=16)
    ap.add_argument('--allreduce', choices=('rccl', 'p2p'), default='rccl',
                    help='--tp collective: torch.distributed.all_reduce (RCCL) or the one-shot exchange over IPC peer mappings (csrc/p2p.hip)')
    ap.add_argument('--dp', action='store_true', help='N independent replicas of the single-GPU workload (weak scaling) instead of --tp row')
    ap.add_argument('--no-prefill', action='store_true')
    ap.add_argument('--no-config4', action='store_true')
    ap.add_argument('--no-small-batch', action='store_true')
    ap.add_argument('--pmc-file', default=None, help='traffic.json of a `rocprofv3 --pmc FETCH_SIZE` pass over this command (tools/pmc_traffic.py)')
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU under torch.distributed.run,
        # rendezvous on 127.0.0.1) -- the reference spreads over GPUs from one command too (llama.py:328-382).  Rank 0 prints the line.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        if os.environ.get('GPTQ_BENCH_SPAWN_DRY'):      # CPU test hook: show the launch, do not run it
            print(json.dumps({'spawn': cmd}))
            return
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to print a line for the wrong N\n'
                         % (args.gpus, world))
        sys.exit(2)
    import torch.distributed as dist
    distributed = world > 1
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # test hooks (a 1-GPU box): GPTQ_BENCH_BACKEND=gloo + GPTQ_BENCH_ONE_DEVICE=1 run all ranks on cuda:0 through gloo
        backend = os.environ.get('GPTQ_BENCH_BACKEND', 'nccl')
        if os.environ.get('GPTQ_BENCH_ONE_DEVICE'):
            local_rank = 0
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    if world > 1 and args.tp is None and not args.dp:
        args.tp = 'row'            # north_star / BASELINE configs[4]: row-sharded linears, ONE RCCL all-reduce per linear, 1/2/4/8 GPUs

    if args.tp:
        work = TPLayers(dev, rank, world, args.tp, args.tp_layers, allreduce=args.allreduce)
    else:
        work = DecodeLinears(dev, seed=rank, kernel=args.kernel)
    for _ in range(2):
        work.step()
    torch.cuda.synchronize()
    graph = None
    capturable = not (args.tp and distributed and args.allreduce == 'rccl' and os.environ.get('GPTQ_BENCH_BACKEND', 'nccl') != 'nccl')   # gloo collectives cannot be captured
    # test hook: GPTQ_BENCH_FORCE_EAGER_TP=1 makes the capture of a tensor-parallel stack FAIL (an exception inside the capture region, the way a
    # collective that does not support stream capture fails) so that the eager fallback below is exercised, not just written
    force_fail = bool(args.tp and os.environ.get('GPTQ_BENCH_FORCE_EAGER_TP'))
    capture_failed = False
    if not args.eager and (capturable or force_fail):
        try:                       # the collectives are captured with the kernels (RCCL supports stream capture)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                if force_fail:
                    work._partial(work.x_h[:, slice(*work.kb_h)], work.layers[0]['o'], work.p_h, work.kb_h[1] - work.kb_h[0], H65, 1)   # something IS in the capture
                    raise RuntimeError('GPTQ_BENCH_FORCE_EAGER_TP: capture aborted on purpose')
                work.step()
        except Exception:
            capture_failed = True
            if not args.tp:
                raise
            graph = None           # a stack that cannot capture its collectives: eager launches, still correct
            torch.cuda.synchronize()
            try:
                work.step()        # the first launch after an invalidated capture reports (and clears) the runtime's sticky error
            except RuntimeError:
                pass
            torch.cuda.synchronize()
    run = graph.replay if graph is not None else work.step
    launch_mode = 'hipGraph replay' if graph is not None else ('eager (the capture failed: fallback)' if capture_failed else 'eager')

    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if distributed:
        dist.barrier()
    ev_ms = e0.elapsed_time(e1)
    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())

    tp_lat = None
    if args.tp:   # collectives: every rank takes part
        tp_lat = {'32KB_fp32_8192': round(allreduce_latency_us(dev, world, H65), 2), '96KB_fp32_24576': round(allreduce_latency_us(dev, world, 3 * H65), 2),
                  '172KB_fp32_2x22016': round(allreduce_latency_us(dev, world, 2 * I65), 2)}
        if work.p2p is not None:
            tp_lat.update({'p2p_32KB': round(allreduce_latency_us(dev, world, H65, p2p=work.p2p), 2),
                           'p2p_96KB': round(allreduce_latency_us(dev, world, 3 * H65, p2p=work.p2p), 2),
                           'p2p_172KB': round(allreduce_latency_us(dev, world, 2 * I65, p2p=work.p2p), 2),
                           'p2p_status': work.p2p.status()})
    replicas = None
    if args.tp and distributed and not os.environ.get('GPTQ_BENCH_NO_REPLICAS'):
        # reported-only side leg: the N = 1 workload (BASELINE configs[1]) as N independent replicas, no data-path collective
        try:
            del graph
            work_tp, work = work, None
            rep = DecodeLinears(dev, seed=rank)
            for _ in range(2):
                rep.step()
            torch.cuda.synchronize()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                rep.step()
            for _ in range(3):
                g2.replay()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(20):
                g2.replay()
            torch.cuda.synchronize()
            tr = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
            replicas = {'workload': 'BASELINE configs[1] (LLaMA-7B-shaped decode pass) as %d independent replicas' % world,
                        'GBps_whole_job': round(rep.bytes_per_step * world * 20 / float(tr.item()) / 1e9, 1), 'scaling': 'weak'}
            del rep, g2
            work = work_tp
        except Exception as e:
            replicas = {'error': repr(e)[:200]}
            work = work if work is not None else work_tp
    if rank == 0:
        ms_per_step = wall_max * 1e3 / args.steps
        total_bytes = work.bytes_per_step * world
        value = total_bytes / (ms_per_step * 1e-3) / 1e9
        us_per_launch = ev_ms * 1e3 / (args.steps * work.launches_per_step)
        bytes_per_launch = work.bytes_per_step / work.launches_per_step
        achieved = bytes_per_launch / us_per_launch / 1e3
        traffic, traffic_src = pmc_traffic(args.pmc_file)
        if args.tp:
            try:   # rank 0 alone (the other ranks wait in the final barrier): same stack, world 1
                tp1 = tp_world1_leg(dev, args.tp, args.tp_layers) if not os.environ.get('GPTQ_BENCH_NO_TP1') else None
            except Exception as e:
                tp1 = {'error': repr(e)[:200]}
            out = {
                'metric': 'int4 g128 matvec GB/s (LLaMA-65B-shaped 4-bit batch-1 decode linears, sharded over the ranks)',
                'value': round(work.bytes_per_step / (ms_per_step * 1e-3) / 1e9, 1), 'unit': 'GB/s', 'n_gpus': world, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
                'dtype': 'f16 (int4 weights dequantised on the fly, f32 accumulate, f32 partials across ranks)', 'data': 'synthetic',
                'config': {'workload': 'LLaMA-65B-shaped 4-bit g128, %s over %d x MI355X, batch=1 decode (BASELINE configs[4]): %d layers x '
                                       '{qkv 8192x24576, o 8192x8192, gate/up+SiLU 2x8192x22016, down 22016x8192}' %
                                       ('row-sharded linears with one all-reduce per linear' if args.tp == 'row' else
                                        'Megatron pairing (N-shard qkv/gate/up, K-shard o/down: 2 all-reduces per layer)', world, args.tp_layers),
                           'parallelism': 'tp%d %s' % (world, args.tp),
                           'world_size_reported_by_backend': (dist.get_world_size() if distributed else 1),
                           'backend': (dist.get_backend() + (' (RCCL)' if dist.get_backend() == 'nccl' else '')) if distributed else 'none',
                           'collective': ('one-shot push + local sum over IPC peer mappings (gptq_p2p_allreduce_f32), fp32' if work.p2p is not None
                                          else 'torch.distributed.all_reduce (RCCL over xGMI), fp32'),
                           'collectives_per_step': work.collectives_per_step, 'launch_mode': launch_mode,
                           'algorithmic_bytes_per_step': work.bytes_per_step},
                'allreduce_us': tp_lat,
                'tp1_same_workload': tp1,
                'replicas_reported_only': replicas,
                'roofline': {'bound': 'hbm', 'achieved': round(work.bytes_per_step / world / (ms_per_step * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS,
                             'unit': 'GB/s', 'frac': round(work.bytes_per_step / world / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             'traffic': None, 'kernel': 'gptq::stripe_gemv_kernel (per-GPU share of the bytes / wall time incl. collectives)'},
            }
            print(json.dumps(out))
            if distributed:
                dist.barrier()
                dist.destroy_process_group()
            return
        out = {
            'metric': 'int4 g128 matvec GB/s (LLaMA-7B 4-bit batch-1 decode pass over all quantised linears)',
            'value': round(value, 1), 'unit': 'GB/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f16 (int4 weights dequantised on the fly, f32 accumulate)', 'data': 'synthetic',
            'config': {'workload': 'LLaMA-7B-shaped 4-bit g128 matvec, batch=1 seq=1 (BASELINE configs[1]): 32 layers x '
                                   '{qkv 4096x12288, o 4096x4096, gate/up+SiLU 2x4096x11008, down 11008x4096}',
                       'launches_per_step': work.launches_per_step, 'algorithmic_bytes_per_step': work.bytes_per_step,
                       'launch_mode': 'eager' if args.eager else 'hipGraph replay', 'parallelism': 'dp%d replicas' % world,
                       'entry_points': 'gptq_layer_prepare (load) + gptq_layer_forward (per op)',
                       'weight_layout': 'stripe16 image built at load time from the checkpoint buffers (gptq_layer_prepare)'
                                        if args.kernel == 'stripe' else 'checkpoint layout'},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': ('gptq::stripe_gemv_kernel<NU,NS,*> (all 128 launches/step: 96 single-set + 32 fused gate/up)'
                                    if args.kernel == 'stripe' else 'gptq::gemv_rowwave_kernel<4,8,*> (all 128 launches/step)'),
                         'avg_launch_us': round(us_per_launch, 3), 'algorithmic_bytes_per_launch': int(bytes_per_launch)},
        }
        out['memory_MiB'] = {'matvec_pass': round(torch.cuda.max_memory_allocated() / 2**20, 1),
                             'note': 'peak bytes in use by tensors per leg (torch.cuda.reset_peak_memory_stats before each); the matvec pass holds the '
                                     'checkpoint buffers AND their stripe16 images of all 32 layers'}

        def leg(key, fn):
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            try:
                out[key] = fn()
            except Exception as e:        # the headline line must survive a failure of a side leg
                out[key] = {'error': repr(e)[:200]}
            out['memory_MiB'][key] = round(torch.cuda.max_memory_allocated() / 2**20, 1)
        # the side legs run at N = 1 only (the other ranks would sit in the final barrier meanwhile)
        if not args.no_per_shape and world == 1:
            out['per_shape'] = work.per_shape()
            try:
                out['per_shape_llama65b_reported_only'] = larger_model_shapes(dev)
            except Exception as e:
                out['per_shape_llama65b_reported_only'] = {'error': repr(e)[:200]}
        if world == 1:
            work = graph = run = None       # free the 6.8 GB of the matvec pass: every later leg reports its OWN peak
        if not args.no_prefill and world == 1:
            leg('prefill_config3_reported_only', lambda: prefill_leg(dev))
        if not args.no_small_batch and world == 1:
            leg('small_batch_reported_only', lambda: small_batch_leg(dev))
            leg('prompt_reported_only', lambda: prompt_leg(dev))
        if not args.no_config4 and world == 1:
            leg('config4_reported_only', lambda: config4_leg(dev))
        if world == 1 and not os.environ.get('GPTQ_BENCH_NO_TP1'):
            # the N = 1 point of the OTHER curve: `--gpus N > 1` runs BASELINE configs[4] (65B stack, row-sharded); this is that stack at world 1
            leg('tp1_same_workload_as_gpus_gt_1', lambda: tp_world1_leg(dev, 'row', args.tp_layers))
        if not args.no_decode and world == 1:
            leg('decode', lambda: decode_tokens_per_s(dev))
        if not args.no_cpu_baseline and world == 1:
            try:
                out['cpu_baseline'] = cpu_baseline()
            except Exception as e:
                out['cpu_baseline'] = {'error': repr(e)[:200]}
            try:
                out['gptq_loop_reported_only'] = gptq_loop_baseline(dev)
            except Exception as e:
                out['gptq_loop_reported_only'] = {'error': repr(e)[:200]}
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
