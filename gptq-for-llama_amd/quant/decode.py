"""Decode-latency harness for the drop-in modules: the measurement protocol of the reference's
``benchmark()`` (llama.py:385-438) on a random-init LLaMA-shaped model (no network, no
checkpoint): one token per step with a KV cache, a device sync per step, the median step time.

``build_random_llama`` goes through exactly the module surgery the reference's ``load_quant``
performs (llama_inference.py:27-72): ``make_quant_linear`` on every decoder linear (lm_head stays
fp16, :46-48), then ``make_quant_attn``, ``make_quant_norm``, ``make_fused_mlp``.  The packed
buffers are filled with the synthetic distribution of SURVEY 8(d) directly on the GPU.
"""
import time

import numpy as np
import os
import torch

from . import fused_attn, fused_mlp, quant_linear, triton_norm

LLAMA_7B = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                num_key_value_heads=32, vocab_size=32000, max_position_embeddings=2048, rms_norm_eps=1e-6)


def find_layers(module, layers=(torch.nn.Linear,), name=''):
    """same contract as the reference's utils/modelutils.py:7-13 (exact type match)."""
    if type(module) in layers:
        return {name: module}
    res = {}
    for name1, child in module.named_children():
        res.update(find_layers(child, layers=layers, name=name + '.' + name1 if name != '' else name1))
    return res


def fill_random_quant_(layer, gen):
    """uniform int32 bit patterns, scales ~ U(0.001, 0.011) fp16, trivial g_idx."""
    dev = layer.qweight.device
    layer.qweight.copy_(torch.randint(-2**31, 2**31 - 1, layer.qweight.shape, dtype=torch.int32, device=dev, generator=gen))
    layer.qzeros.copy_(torch.randint(-2**31, 2**31 - 1, layer.qzeros.shape, dtype=torch.int32, device=dev, generator=gen))
    layer.scales.copy_((torch.rand(layer.scales.shape, device=dev, generator=gen) * 0.01 + 0.001).half())
    gs = layer.groupsize if layer.groupsize != -1 else layer.infeatures
    layer.g_idx.copy_((torch.arange(layer.infeatures, device=dev) // gs).to(torch.int32))
    if layer.bias is not None:
        layer.bias.zero_()


def build_random_llama(dev='cuda:0', bits=4, groupsize=128, seed=0, fused=True, act_order=False, **overrides):
    """random-init LLaMA (7B shape by default) with every decoder linear replaced by a QuantLinear
    holding random packed weights, then fused attention / norm / MLP like load_quant().
    act_order: every linear gets a non-trivial g_idx of the shape gptq.py:210-216 produces (exactly `groupsize`
    members per group); linears that share their input (q/k/v, gate/up) share the permutation, as in a real
    --act-order checkpoint (same input => same Hessian diagonal)."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding

    cfg = dict(LLAMA_7B)
    cfg.update(overrides)
    config = LlamaConfig(**cfg)
    with torch.device('meta'):
        model = LlamaForCausalLM(config)
    model = model.half().eval()
    layers = find_layers(model)
    for name in ['lm_head']:
        layers.pop(name, None)
    with torch.device('meta'):
        quant_linear.make_quant_linear(model, layers, bits, groupsize)
    model = model.to_empty(device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    with torch.no_grad():
        for _, m in model.named_modules():
            if isinstance(m, quant_linear.QuantLinear):
                fill_random_quant_(m, gen)
        if act_order:
            for layer in model.model.layers:
                K = layer.self_attn.q_proj.infeatures
                gs = groupsize if groupsize != -1 else K

                def random_g_idx(k):
                    inv = torch.argsort(torch.randperm(k, device=dev, generator=gen))
                    return (torch.arange(k, device=dev) // (groupsize if groupsize != -1 else k))[inv].to(torch.int32)
                shared = random_g_idx(K)
                for m in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj, layer.mlp.gate_proj, layer.mlp.up_proj):
                    m.g_idx.copy_(shared)
                layer.self_attn.o_proj.g_idx.copy_(random_g_idx(layer.self_attn.o_proj.infeatures))
                layer.mlp.down_proj.g_idx.copy_(random_g_idx(layer.mlp.down_proj.infeatures))
        for name, p in model.named_parameters():
            if p.dim() == 1:
                p.fill_(1.0)                                  # RMSNorm weights
            else:
                p.copy_((torch.randn(p.shape, device=dev, generator=gen) * 0.02).to(p.dtype))
    model.model.rotary_emb = LlamaRotaryEmbedding(config, device=dev)
    if fused:
        fused_attn.make_quant_attn(model)
        triton_norm.make_quant_norm(model)
        fused_mlp.make_fused_mlp(model)
    model.seqlen = 2048     # reference sets this on the model (llama.py:23)
    return model


def benchmark_decode(model, tokens=64, seed=0, engine_hook=True):
    """llama.py:385-438: feed ``input_ids[:, i:i+1]`` with the growing cache, sync after every
    step, report the median (and peak memory).  Returns a dict.  engine_hook=False switches the transparent
    decode engine of quant/engine_hook.py off for this run (the module chain launch by launch)."""
    from transformers.cache_utils import DynamicCache
    model._gptq_engine_disabled = not engine_hook

    dev = next(model.parameters()).device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    vocab = model.config.vocab_size
    input_ids = torch.randint(0, vocab, (1, tokens), device=dev, generator=gen)
    cache = DynamicCache(config=model.config)
    times = []
    torch.cuda.reset_peak_memory_stats(dev)
    with torch.no_grad():
        for i in range(tokens):
            torch.cuda.synchronize(dev)
            tick = time.perf_counter()
            out = model(input_ids[:, i:i + 1], past_key_values=cache, use_cache=True)
            torch.cuda.synchronize(dev)
            times.append(time.perf_counter() - tick)
            del out
    model._gptq_engine_disabled = False
    med = float(np.median(times[2:])) if len(times) > 4 else float(np.median(times))
    return {'protocol': 'llama.py:385-438 (one token per step, KV cache, sync per step, median)',
            'mode': 'model(input_ids[:, i:i+1], past_key_values=cache) on the drop-in modules, ' +
                    ('decode engine hook (quant/engine_hook.py)' if engine_hook else 'eager module chain'),
            'tokens': tokens, 'median_s_per_token': round(med, 6), 'tokens_per_s': round(1.0 / med, 1),
            'max_memory_MiB': round(torch.cuda.max_memory_allocated(dev) / 1024 / 1024, 1)}


def benchmark_generate(model, prompt_len=16, new_tokens=128, seed=0, batch=1, left_pad=False):
    """``model.generate(input_ids, do_sample=False, max_new_tokens=...)`` as llama_inference.py:119-127 calls it (greedy here
    so that runs are comparable), wall time of the decode part: (t(new_tokens) - t(1 new token)) / (new_tokens - 1).
    batch > 1: that many prompts at once; left_pad: rows of different lengths, left-padded with an attention mask (what a tokenizer
    with padding_side='left' hands to generate); tokens_per_s is the aggregate over the rows."""
    dev = next(model.parameters()).device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    ids = torch.randint(1, model.config.vocab_size, (batch, prompt_len), device=dev, generator=gen)
    mask = torch.ones_like(ids)
    if left_pad:
        for b in range(batch):
            n = (3 * b + 2) % max(prompt_len - 1, 1) if b < batch - 1 or batch == 1 else 0
            ids[b, :n] = 0
            mask[b, :n] = 0

    def run(n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        with torch.no_grad():
            out = model.generate(ids, attention_mask=mask, do_sample=False, max_new_tokens=n, min_new_tokens=n, pad_token_id=0)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, out
    run(4)                                   # warm-up: engine build + graph capture happen here
    t1, _ = run(1)
    tn, out = run(new_tokens)
    per_step = (tn - t1) / (new_tokens - 1)
    from .engine_hook import engine_steps
    return {'call': 'model.generate(input_ids[%d, %d]%s, do_sample=False, max_new_tokens=%d) (llama_inference.py:119-127)' % (
                batch, prompt_len, ', left-padded attention_mask' if left_pad else '', new_tokens),
            'generated': int(out.shape[1] - prompt_len), 's_per_token': round(per_step / batch, 6), 's_per_step': round(per_step, 6),
            'tokens_per_s': round(batch / per_step, 1), 'engine_steps_total': engine_steps(model)}


# ----------------------------------------------------------------------------------------------
# Graph-captured decode engine: the same arithmetic as one HF decoder step
# through the drop-in modules, but issued as a flat list of C-ABI launches on static buffers --
# 10 launches per layer (RMSNorm, qkv GEMV, RoPE+KV append, attention x2, o_proj GEMV + residual,
# RMSNorm, fused gate/up, down GEMV + residual) -- so that ONE hipGraph replay is one token.
# The position lives in device memory and is advanced inside the graph.
# ----------------------------------------------------------------------------------------------
ROPE_TABLE = os.environ.get('GPTQ_ROPE_TABLE', '1') != '0'
# round 6: the batch-1 attention launch leaves its split records to o_proj's decode kernel (gptq_decode_attn_split_f16 + gptq_layer_decode_attn_f16)
# instead of merging them itself; 0: the self-merging launch of rounds 2-5 (A/B runs, and what every other configuration takes anyway)
ATTN_RECORDS = os.environ.get('GPTQ_ATTN_RECORDS', '1') != '0'
LM_HEAD_KERNEL = os.environ.get('GPTQ_LM_HEAD_KERNEL', '1') != '0'   # 0: final norm + torch.matmul (hipBLASLt) as in rounds 1-3 (A/B runs)


MAX_BATCH = 16         # rows of a decode batch one engine serves (the LM head kernel and the decode kernel's row groups end there)


def lm_head_logits(eng, x, logits, s):
    """shared by DecodeEngine and TPDecodeEngine: logits[B, rows] = lm_head(rmsnorm(x[B, hidden])) (+ lm_head.bias).  The hand-written
    launch needs an fp16 [rows, hidden] weight with unit column stride whose shape matches the logits buffer (a resized head would write
    past it: ADVICE r4) -- anything else, or GPTQ_LM_HEAD_KERNEL=0, takes the stand-alone norm + torch.matmul."""
    W = eng.lm_head
    bias = getattr(eng, 'lm_head_bias', None)
    B = x.shape[0]
    ok = (LM_HEAD_KERNEL and W.dtype == torch.float16 and W.dim() == 2 and W.stride(1) == 1 and W.stride(0) % 8 == 0 and W.shape[1] % 8 == 0 and
          W.shape[1] == x.shape[1] and W.shape[0] == logits.shape[-1] and B <= MAX_BATCH and (bias is None or bias.dtype == torch.float16))
    if ok:
        nw = eng.final_norm if eng.fuse_norm else None
        if nw is None:
            eng._norm_rows(x, eng.final_norm, eng.h, s)
        src = x if nw is not None else eng.h
        rc = eng.lib.gptq_dense_matmat_f16(src.data_ptr(), src.stride(0), W.data_ptr(), W.stride(0), eng.native.ptr(bias), logits.data_ptr(),
                                           logits.stride(0), B, W.shape[0], W.shape[1], eng.native.ptr(nw), eng.eps, s)
        if rc != -6:
            eng.native.check(rc, 'gptq_dense_matmat_f16')
            return
    eng._norm_rows(x, eng.final_norm, eng.h, s)
    if bias is not None:
        torch.addmm(bias, eng.h, W.t(), out=logits)
    else:
        torch.matmul(eng.h, W.t(), out=logits)


class DecodeEngine:

    def __init__(self, model, t_max=2048, fuse_norm=True, fuse_attn=True, batch=1):
        """batch (round 5): rows of a decode BATCH -- B sequences advance by one token per step, every row with its own position and its
        own slice of the K/V cache (the reference's one kernel serves any batch, quant_linear.py:263-269; HF generate drives it with
        [B, 1] steps).  The linears run at M = B through gptq_layer_decode_f16 (norm and residual inside the decode kernel's launch up to
        four rows, 16-row MFMA tiles above), attention through gptq_decode_attn_batch_f16, the LM head through gptq_dense_matmat_f16:
        still ONE hipGraph replay per step."""
        from . import _native
        # act-order layers fed by one of the engine's own launches (o_proj <- attention, down_proj <- gate/up + SiLU) get their input written in THEIR
        # sorted order by the producer (round 5) and run the trivial kernel; GPTQ_PRODUCER_PERM=0: the in-kernel gather of round 3 / 4 (A/B)
        self.producer_perm = os.environ.get('GPTQ_PRODUCER_PERM', '1') != '0'
        self.batch = int(batch)
        if not 1 <= self.batch <= MAX_BATCH:
            raise NotImplementedError('DecodeEngine: batch must be 1 .. %d' % MAX_BATCH)
        self.fuse_norm, self.fuse_attn = bool(fuse_norm), bool(fuse_attn)
        self.native = _native
        self.lib = _native.lib()
        self.model = model
        cfg = model.config
        dev = next(model.parameters()).device
        self.dev = dev
        self.t_max = int(t_max)
        self.hidden = cfg.hidden_size
        self.heads = cfg.num_attention_heads
        self.head_dim = self.hidden // self.heads
        self.eps = float(cfg.rms_norm_eps)
        if self.head_dim != 128:
            raise NotImplementedError('DecodeEngine: head_dim must be 128')
        self.embed = model.model.embed_tokens.weight
        self.lm_head = model.lm_head.weight
        self.lm_head_bias = getattr(model.lm_head, 'bias', None)
        self.final_norm = model.model.norm.weight
        self.layers = []
        for layer in model.model.layers:
            attn, mlp = layer.self_attn, layer.mlp
            if not isinstance(attn, fused_attn.QuantLlamaAttention) or not isinstance(mlp, fused_mlp.QuantLlamaMLP):
                raise RuntimeError('DecodeEngine needs make_quant_attn / make_fused_mlp applied first')
            gpack, upack = self._pack_pair(mlp)
            self.layers.append(dict(
                ln1=layer.input_layernorm.weight, ln2=layer.post_attention_layernorm.weight,
                qkv=self._pack(attn.qkv_proj), o=self._pack(attn.o_proj), down=self._pack(mlp.down_proj),
                gate=gpack, up=upack, theta=float(attn.rope_theta)))
        H, I, B = self.hidden, cfg.intermediate_size, self.batch
        f16 = dict(dtype=torch.float16, device=dev)
        self.ids = torch.zeros(B, dtype=torch.int64, device=dev)
        self.pos = torch.zeros(B, dtype=torch.int64, device=dev)       # per row: tokens consumed so far = the position of the next one
        self.x = torch.zeros((B, H), **f16)
        self.x2 = torch.zeros((B, H), **f16)
        self.h = torch.zeros((B, H), **f16)
        self.qkvb = torch.zeros((B, 3 * H), **f16)
        self.ab = torch.zeros((B, H), **f16)
        self.cb = torch.zeros((B, I), **f16)
        # sized from the head itself (a resized / padded lm_head has rows != config.vocab_size: ADVICE r4)
        self.logits = torch.zeros((B, self.lm_head.shape[0]), **f16)
        self.stream_out = torch.zeros(self.t_max + 1, dtype=torch.int64, device=dev)   # greedy mode (batch 1): token chosen after position p
        self.greedy_graph = None
        self.greedy_rows_graph, self.stream_rows, self.stepc, self.sample_graphs = None, None, None, {}
        nl = len(self.layers)
        self.kcb = torch.zeros((nl, B, self.t_max, H), **f16)          # [layer][row][t][heads * head_dim]
        self.vcb = torch.zeros((nl, B, self.t_max, H), **f16)
        self.kc, self.vc = self.kcb[:, 0], self.vcb[:, 0]              # row 0 = THE cache of a batch-1 engine ([layer][t][H] views)
        self.attn_ws = torch.zeros(self.lib.gptq_decode_attn_batch_workspace_bytes(B, self.heads, self.head_dim, self.t_max),
                                   dtype=torch.uint8, device=dev)
        self.ws = _native.workspace(dev)
        self.graph = None
        self.scratch = None
        # batch 1: the attention's splits are merged by o_proj's decode kernel when every o_proj can (trivial g_idx, an image, no bias next to the residual)
        self.attn_records = bool(ATTN_RECORDS and self.fuse_attn and B == 1 and all(
            L['o']['st'] is not None and L['o']['bias'] is None and
            self.lib.gptq_layer_decode_attn_supported(L['o']['_keep'].handle, 1, self.heads, self.head_dim) == 1 for L in self.layers))
        if B > 1:
            need = 0
            for L in self.layers:
                for w in (L['qkv'], L['o'], L['gate'], L['down']):
                    pl = w['_keep']
                    route = self.lib.gptq_layer_route_for(pl.handle, B)
                    if route not in (1, 2) and not (pl.kind == 1 and route > 0):     # stripe decode kernel / 16-row tiles on the image
                        raise NotImplementedError('DecodeEngine(batch=%d): a layer has no stripe16 route for %d rows (route %d)' % (B, B, route))
                    need = max(need, self.lib.gptq_layer_decode_scratch_bytes(pl.handle, B))
            self.scratch = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)

    # -- weights: the engine reads the SAME derived copy as the modules -------------------------------------------------------
    # Every linear is taken from quant/layer.py prepared() with exactly the arguments the module's own forward uses, so the eager
    # path (prefill, anything the hook declines) and the engine share ONE PreparedLayer: one stripe16 image per layer next to the
    # checkpoint buffers -- not the three or four copies of round 3 (module image + the engine's own stripe_copy of every linear +
    # single-set images of gate and up next to the pair's).
    def _from_prepared(self, pl, qweight, scales, qzeros, g_idx, bits, gs, K, N, bias):
        gi = None
        if pl.kind != 0 and g_idx is not None and qweight is not None:
            gi = quant_linear._int32c(g_idx[:K])
        st = pl.stripe
        srt = None
        if st is None and gi is not None and bits in (2, 4, 8):      # no image (GPTQ_STRIPE=0, odd shapes): the sorted-copy kernels
            srt = quant_linear.act_order_sorted(quant_linear._int32c(qweight), gi, K, gs, bits)
        released = qweight is None or qweight.shape[0] == 0
        return dict(qw=None if released else quant_linear._int32c(qweight), sc=None if released else scales,
                    qz=None if released else quant_linear._int32c(qzeros), gi=gi, bits=bits, gs=gs, K=K, N=N, bias=bias, srt=srt, st=st,
                    perm=pl.perm16 if st is not None else None, invperm=pl.invperm32 if st is not None else None, _keep=pl)

    def _pack_raw(self, qweight, scales, qzeros, g_idx, bits, groupsize, K, N, bias=None):
        from .layer import prepared
        gs = groupsize if groupsize != -1 else K
        pl = prepared(((qweight, scales, qzeros, g_idx),), bias, bits, gs, K, N, sort=quant_linear.ACT_ORDER_SORT)
        return self._from_prepared(pl, qweight, scales, qzeros, g_idx, bits, gs, K, N, bias)

    def _pack(self, ql):
        if getattr(ql, '_released', None) is not None:   # memory mode: the PreparedLayer's image is all there is -- and all the decode kernels need
            return self._from_prepared(ql._released, None, None, None, None, ql.bits, ql.groupsize, ql.infeatures, ql.outfeatures, ql.bias)
        return self._pack_raw(ql.qweight, ql.scales, ql.qzeros, ql.g_idx, ql.bits, ql.groupsize, ql.infeatures, ql.outfeatures, ql.bias)

    def _pack_pair(self, mlp):
        """gate and up of a QuantLlamaMLP: ONE PreparedLayer of the pair (the one mlp.hip_llama_mlp uses) -- its image holds both sets,
        of the group-sorted rows when the two share an act-order permutation."""
        from .layer import prepared
        K, N, bits = mlp.infeatures, mlp.intermediate_size, mlp.bits
        gs = mlp.groupsize if mlp.groupsize != -1 else K
        if getattr(mlp, '_released', None) is not None:
            pl = mlp._released
            g = self._from_prepared(pl, None, None, None, None, bits, gs, K, N, None)
            u = dict(g)
        else:
            gate = (mlp.gate_proj_qweight, mlp.gate_proj_scales, mlp.gate_proj_qzeros, mlp.gate_proj_g_idx)
            up = (mlp.up_proj_qweight, mlp.up_proj_scales, mlp.up_proj_qzeros, mlp.up_proj_g_idx)
            pl = prepared((gate, up), None, bits, gs, K, N, sort=quant_linear.ACT_ORDER_SORT)
            g = self._from_prepared(pl, *gate, bits, gs, K, N, None)
            u = self._from_prepared(pl, *up, bits, gs, K, N, None)
        g['st2'], g['perm2'] = pl.stripe, (pl.perm16 if pl.stripe is not None else None)
        g['st'] = u['st'] = None          # the pair's image is not a single-set image
        g['perm'] = u['perm'] = None
        g['pair_sorted'] = False
        if g['st2'] is None and g['gi'] is not None and u['gi'] is not None and bits in (2, 4, 8):
            # no image: gate and up share their input, hence (normally) their act-order permutation -- checked ONCE here
            for w in (g, u):
                w['srt'] = quant_linear.act_order_sorted(w['qw'], w['gi'], K, gs, bits)
            g['pair_sorted'] = g['srt'] is not None and u['srt'] is not None and bool(torch.equal(g['srt'][1], u['srt'][1]))
        return g, u

    # -- launches ------------------------------------------------------------------------------
    def _gemv(self, x, w, y, s, residual=None, x_sorted=False):
        """x_sorted: x was written in this layer's group-sorted order by its producer -- no gather (the image IS the sorted rows)"""
        if w['bias'] is not None and residual is not None:
            # one add slot per launch: the bias rides in the matvec, the residual is a second launch (gptq_layer_decode_f16 does both)
            if self.scratch is None:
                self.scratch = torch.empty(max(256, self.lib.gptq_layer_decode_scratch_bytes(w['_keep'].handle, 1)), dtype=torch.uint8, device=self.dev)
            return self._lin(w, x, y, s, self.native.layer_workspace(self.dev, s), residual=residual)
        b = residual if residual is not None else w['bias']
        ptr = self.native.ptr
        if w['st'] is not None:       # stripe16: no K split, no workspace (act-order: x gathered through perm in the kernel)
            quant_linear.stripe_matvec(x, w['st'], y, w['K'], w['N'], w['bits'], w['gs'], bias=b, perm=None if x_sorted else w['perm'])
            return
        if w['srt'] is not None:      # act-order layer: group-sorted copy + fused x gather
            qs, perm = w['srt']
            rc = self.lib.gptq_matmul248_sorted_f16(x.data_ptr(), w['K'], perm.data_ptr(), qs.data_ptr(), w['sc'].data_ptr(),
                                                    w['qz'].data_ptr(), ptr(b), y.data_ptr(), w['N'], 1, w['K'], w['N'], w['bits'],
                                                    w['gs'], self.ws.data_ptr(), self.ws.numel(), s)
            self.native.check(rc, 'gptq_matmul248_sorted_f16')
            return
        rc = self.lib.gptq_matmul248_f16(x.data_ptr(), w['K'], w['qw'].data_ptr(), w['sc'].data_ptr(), w['qz'].data_ptr(), ptr(w['gi']),
                                         ptr(b), y.data_ptr(), w['N'], 1, w['K'], w['N'], w['bits'], w['gs'], self.ws.data_ptr(),
                                         self.ws.numel(), s)
        self.native.check(rc, 'gptq_matmul248_f16')

    def _norm(self, x, w, y, s):
        rc = self.lib.gptq_rmsnorm_f16(x.data_ptr(), self.hidden, w.data_ptr(), y.data_ptr(), self.hidden, 1, self.hidden, self.eps, s)
        self.native.check(rc, 'gptq_rmsnorm_f16')

    def _norm_gemv(self, x, nw, w, y, s):
        """y = QuantLinear(rmsnorm(x)): one launch when the fused kernel serves the shape, else two."""
        ptr = self.native.ptr
        if self.fuse_norm and w['st'] is not None:      # RMSNorm fused into the staging of x (every workgroup sees all of x)
            quant_linear.stripe_matvec(x, w['st'], y, w['K'], w['N'], w['bits'], w['gs'], bias=w['bias'], norm_weight=nw, eps=self.eps, perm=w['perm'])
            return
        if self.fuse_norm and w['bias'] is None and w['srt'] is not None:      # act-order: norm + gather + GEMV in one launch
            qs, perm = w['srt']
            rc = self.lib.gptq_rmsnorm_sorted_f16(x.data_ptr(), nw.data_ptr(), self.eps, perm.data_ptr(), qs.data_ptr(), w['sc'].data_ptr(),
                                                  w['qz'].data_ptr(), None, None, None, None, y.data_ptr(), w['K'], w['N'], w['bits'],
                                                  w['gs'], self.ws.data_ptr(), self.ws.numel(), s)
            if rc != -6:
                self.native.check(rc, 'gptq_rmsnorm_sorted_f16')
                return
        if self.fuse_norm and w['bias'] is None and w['srt'] is None:
            rc = self.lib.gptq_rmsnorm_matmul248_f16(x.data_ptr(), nw.data_ptr(), self.eps, w['qw'].data_ptr(), w['sc'].data_ptr(),
                                                     w['qz'].data_ptr(), ptr(w['gi']), None, y.data_ptr(), w['K'], w['N'], w['bits'],
                                                     w['gs'], self.ws.data_ptr(), self.ws.numel(), s)
            if rc != -6:   # GPTQ_E_VARIANT: shape needs the generic kernels
                self.native.check(rc, 'gptq_rmsnorm_matmul248_f16')
                return
        self._norm(x, nw, self.h, s)
        self._gemv(self.h, w, y, s)

    def _norm_mlp(self, x, nw, g, u, c, s, out_perm=None):
        """out_perm: store column n of silu(gate) * up at out_perm[n] (down_proj's sorted order); only the stripe16 pair kernel does that"""
        ptr = self.native.ptr
        if g.get('st2') is not None:
            perm = g['perm2']
            if self.fuse_norm:
                quant_linear.stripe_matvec(x, g['st2'], c, g['K'], g['N'], g['bits'], g['gs'], nsets=2, norm_weight=nw, eps=self.eps, perm=perm, out_perm=out_perm)
            else:
                self._norm(x, nw, self.h, s)
                quant_linear.stripe_matvec(self.h, g['st2'], c, g['K'], g['N'], g['bits'], g['gs'], nsets=2, perm=perm, out_perm=out_perm)
            return
        assert out_perm is None
        if g.get('pair_sorted') and self.fuse_norm:
            rc = self.lib.gptq_rmsnorm_sorted_f16(x.data_ptr(), nw.data_ptr(), self.eps, g['srt'][1].data_ptr(), g['srt'][0].data_ptr(),
                                                  g['sc'].data_ptr(), g['qz'].data_ptr(), u['srt'][0].data_ptr(), u['sc'].data_ptr(),
                                                  u['qz'].data_ptr(), None, c.data_ptr(), g['K'], g['N'], g['bits'], g['gs'],
                                                  self.ws.data_ptr(), self.ws.numel(), s)
            if rc != -6:
                self.native.check(rc, 'gptq_rmsnorm_sorted_f16')
                return
        if g.get('pair_sorted'):
            # act-order MLP: gate and up share the permutation of their common input -> sorted fused kernel
            self._norm(x, nw, self.h, s)
            rc = self.lib.gptq_fused_mlp_sorted_f16(self.h.data_ptr(), g['K'], g['srt'][1].data_ptr(), g['srt'][0].data_ptr(),
                                                    g['sc'].data_ptr(), g['qz'].data_ptr(), u['srt'][0].data_ptr(), u['sc'].data_ptr(),
                                                    u['qz'].data_ptr(), c.data_ptr(), g['N'], 1, g['K'], g['N'], g['bits'], g['gs'],
                                                    self.ws.data_ptr(), self.ws.numel(), s)
            if rc != -6:
                self.native.check(rc, 'gptq_fused_mlp_sorted_f16')
                return
        if self.fuse_norm:
            rc = self.lib.gptq_rmsnorm_fused_mlp_f16(x.data_ptr(), nw.data_ptr(), self.eps, g['qw'].data_ptr(), g['sc'].data_ptr(),
                                                     g['qz'].data_ptr(), ptr(g['gi']), u['qw'].data_ptr(), u['sc'].data_ptr(),
                                                     u['qz'].data_ptr(), ptr(u['gi']), c.data_ptr(), g['K'], g['N'], g['bits'], g['gs'],
                                                     self.ws.data_ptr(), self.ws.numel(), s)
            if rc != -6:
                self.native.check(rc, 'gptq_rmsnorm_fused_mlp_f16')
                return
        self._norm(x, nw, self.h, s)
        rc = self.lib.gptq_fused_mlp_f16(self.h.data_ptr(), g['K'], g['qw'].data_ptr(), g['sc'].data_ptr(), g['qz'].data_ptr(), ptr(g['gi']),
                                         u['qw'].data_ptr(), u['sc'].data_ptr(), u['qz'].data_ptr(), ptr(u['gi']), c.data_ptr(), g['N'], 1,
                                         g['K'], g['N'], g['bits'], g['gs'], self.ws.data_ptr(), self.ws.numel(), s)
        self.native.check(rc, 'gptq_fused_mlp_f16')

    def _rope_table(self, theta, s):
        """{cos, sin}[t_max][head_dim / 2] fp32 per rope base (one for a LLaMA), filled once by the kernel that would otherwise
        evaluate them per token and head.  GPTQ_ROPE_TABLE=0: compute in the attention kernel (A/B runs)."""
        if not ROPE_TABLE:
            return None
        tabs = self.__dict__.setdefault('_rope_tabs', {})
        t = tabs.get(float(theta))
        if t is None:
            t = torch.empty((self.t_max, self.head_dim // 2, 2), dtype=torch.float32, device=self.dev)
            self.native.check(self.lib.gptq_rope_table_f32(t.data_ptr(), self.t_max, self.head_dim, float(theta), s), 'gptq_rope_table_f32')
            tabs[float(theta)] = t
        return t

    # -- a decode batch: every linear is ONE gptq_layer_decode_f16 (y = residual + layer(rmsnorm(x)), M = batch) ------------------------
    def _lin(self, w, x, y, s, lws, norm=None, residual=None):
        pl = w['_keep']
        ptr = self.native.ptr
        rc = self.lib.gptq_layer_decode_f16(pl.handle, x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), x.shape[0], ptr(norm), self.eps,
                                            ptr(residual), residual.stride(0) if residual is not None else 0, lws.data_ptr(), lws.numel(),
                                            self.scratch.data_ptr(), self.scratch.numel(), s)
        self.native.check(rc, 'gptq_layer_decode_f16')

    def _lin_next_norm(self, w, x, y, s, lws, residual, next_norm, h):
        """y = residual + layer(x) and, when the launch that combines this batch's K slices can do it for free, h = rmsnorm(y) * next_norm
        (gptq_layer_decode_next_norm_f16) -- returns whether h was written (decided on the host, per shape and batch: the same in every replay)"""
        import ctypes
        pl = w['_keep']
        done = ctypes.c_int(0)
        rc = self.lib.gptq_layer_decode_next_norm_f16(pl.handle, x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), x.shape[0], None, self.eps,
                                                      residual.data_ptr(), residual.stride(0), next_norm.data_ptr(), self.eps, h.data_ptr(), h.stride(0),
                                                      ctypes.byref(done), lws.data_ptr(), lws.numel(), self.scratch.data_ptr(), self.scratch.numel(), s)
        self.native.check(rc, 'gptq_layer_decode_next_norm_f16')
        return bool(done.value)

    def _step_batch(self):
        lib = self.lib
        s = self.native.stream_ptr(self.dev)
        lws = self.native.layer_workspace(self.dev, s)
        B, H = self.batch, self.hidden
        torch.index_select(self.embed, 0, self.ids, out=self.x)
        scale = 1.0 / float(np.sqrt(self.head_dim))
        normed = False          # self.h holds rmsnorm(x) * ln1 of this block already (written by the previous block's down_proj: round 6)
        for li, L in enumerate(self.layers):
            if normed:
                self._lin(L['qkv'], self.h, self.qkvb, s, lws)
            elif self.fuse_norm:
                self._lin(L['qkv'], self.x, self.qkvb, s, lws, norm=L['ln1'])                   # qkv = qkv_proj(rmsnorm(x))
            else:
                self._norm_rows(self.x, L['ln1'], self.h, s)
                self._lin(L['qkv'], self.h, self.qkvb, s, lws)
            tab = self._rope_table(L['theta'], s)
            rc = lib.gptq_decode_attn_batch_f16(self.qkvb.data_ptr(), self.qkvb.stride(0), self.pos.data_ptr(), self.kcb[li].data_ptr(),
                                                self.vcb[li].data_ptr(), self.ab.data_ptr(), self.ab.stride(0), self.attn_ws.data_ptr(),
                                                self.attn_ws.numel(), B, self.heads, self.head_dim, self.t_max, L['theta'], scale,
                                                self.native.ptr(tab), None, s)
            self.native.check(rc, 'gptq_decode_attn_batch_f16')
            self._lin(L['o'], self.ab, self.x2, s, lws, residual=self.x)                       # x2 = x + o_proj(attn)
            if self.fuse_norm:
                self._lin(L['gate'], self.x2, self.cb, s, lws, norm=L['ln2'])                   # c = silu(gate(h)) * up(h), h = rmsnorm(x2)
            else:
                self._norm_rows(self.x2, L['ln2'], self.h, s)
                self._lin(L['gate'], self.h, self.cb, s, lws)
            if self.fuse_norm and li + 1 < len(self.layers):                                   # x = x2 + down(c) (+ the next block's input norm)
                normed = self._lin_next_norm(L['down'], self.cb, self.x, s, lws, self.x2, self.layers[li + 1]['ln1'], self.h)
            else:
                self._lin(L['down'], self.cb, self.x, s, lws, residual=self.x2)
                normed = False
        self._lm_head(s)
        self.pos.add_(1)

    def _norm_rows(self, x, w, y, s):
        rc = self.lib.gptq_rmsnorm_f16(x.data_ptr(), x.stride(0), w.data_ptr(), y.data_ptr(), y.stride(0), x.shape[0], self.hidden, self.eps, s)
        self.native.check(rc, 'gptq_rmsnorm_f16')

    def _step(self):
        if self.batch > 1:
            return self._step_batch()
        lib, ptr = self.lib, self.native.ptr
        s = torch.cuda.current_stream(self.dev).cuda_stream
        H = self.hidden
        torch.index_select(self.embed, 0, self.ids, out=self.x)
        scale = 1.0 / float(np.sqrt(self.head_dim))
        for li, L in enumerate(self.layers):
            self._norm_gemv(self.x, L['ln1'], L['qkv'], self.qkvb, s)       # qkv = qkv_proj(rmsnorm(x))
            # producer-side permutation: o_proj / down_proj of an act-order checkpoint read x in their sorted order -- written that way by the attention /
            # gate-up launch when those are the stripe16 kernels (else: the in-kernel gather)
            # (a consumer with a bias AND a residual goes through gptq_layer_decode_f16, which gathers x itself: no producer-side order for it -- ADVICE r5)
            o_inv = L['o'].get('invperm') if (self.producer_perm and self.fuse_attn and L['o']['st'] is not None and L['o']['bias'] is None) else None
            d_inv = L['down'].get('invperm') if (self.producer_perm and L['gate'].get('st2') is not None and L['down']['st'] is not None and
                                                 L['down']['bias'] is None) else None
            if self.attn_records:
                tab = self._rope_table(L['theta'], s)
                rc = lib.gptq_decode_attn_split_f16(self.qkvb.data_ptr(), self.qkvb.stride(0), self.pos.data_ptr(), self.kc[li].data_ptr(),
                                                    self.vc[li].data_ptr(), self.attn_ws.data_ptr(), self.attn_ws.numel(), 1, self.heads, self.head_dim,
                                                    self.t_max, L['theta'], scale, ptr(tab), 0, s)
                self.native.check(rc, 'gptq_decode_attn_split_f16')
                rc = lib.gptq_layer_decode_attn_f16(L['o']['_keep'].handle, self.attn_ws.data_ptr(), self.attn_ws.numel(), self.pos.data_ptr(), 1,
                                                    self.heads, self.head_dim, self.t_max, 0, self.x2.data_ptr(), self.x2.stride(0), self.x.data_ptr(),
                                                    self.x.stride(0), s)                                   # x2 = x + o_proj(merge(records))
                self.native.check(rc, 'gptq_layer_decode_attn_f16')
                self._norm_mlp(self.x2, L['ln2'], L['gate'], L['up'], self.cb, s, out_perm=d_inv)
                self._gemv(self.cb, L['down'], self.x, s, residual=self.x2, x_sorted=d_inv is not None)
                continue
            if self.fuse_attn:
                tab = self._rope_table(L['theta'], s)
                if o_inv is not None:
                    rc = lib.gptq_decode_attn_batch_f16(self.qkvb.data_ptr(), self.qkvb.stride(0), self.pos.data_ptr(), self.kc[li].data_ptr(),
                                                        self.vc[li].data_ptr(), self.ab.data_ptr(), self.ab.stride(0), self.attn_ws.data_ptr(),
                                                        self.attn_ws.numel(), 1, self.heads, self.head_dim, self.t_max, L['theta'], scale,
                                                        self.native.ptr(tab), o_inv.data_ptr(), s)
                elif tab is not None:
                    rc = lib.gptq_decode_attn_fused_table_f16(self.qkvb.data_ptr(), self.pos.data_ptr(), self.kc[li].data_ptr(),
                                                              self.vc[li].data_ptr(), self.ab.data_ptr(), self.attn_ws.data_ptr(),
                                                              self.attn_ws.numel(), self.heads, self.head_dim, self.t_max, L['theta'], scale,
                                                              tab.data_ptr(), s)
                else:
                    rc = lib.gptq_decode_attn_fused_f16(self.qkvb.data_ptr(), self.pos.data_ptr(), self.kc[li].data_ptr(),
                                                        self.vc[li].data_ptr(), self.ab.data_ptr(), self.attn_ws.data_ptr(),
                                                        self.attn_ws.numel(), self.heads, self.head_dim, self.t_max, L['theta'], scale, s)
                self.native.check(rc, 'gptq_decode_attn_fused_f16')
            else:
                rc = lib.gptq_decode_rope_kv_f16(self.qkvb.data_ptr(), self.pos.data_ptr(), self.kc[li].data_ptr(),
                                                 self.vc[li].data_ptr(), self.heads, self.head_dim, self.t_max, L['theta'], s)
                self.native.check(rc, 'gptq_decode_rope_kv_f16')
                rc = lib.gptq_decode_attn_f16(self.qkvb.data_ptr(), self.kc[li].data_ptr(), self.vc[li].data_ptr(), self.pos.data_ptr(),
                                              self.ab.data_ptr(), self.attn_ws.data_ptr(), self.attn_ws.numel(), self.heads,
                                              self.head_dim, self.t_max, scale, s)
                self.native.check(rc, 'gptq_decode_attn_f16')
            self._gemv(self.ab, L['o'], self.x2, s, residual=self.x, x_sorted=o_inv is not None)        # x2 = x + o_proj(attn)
            self._norm_mlp(self.x2, L['ln2'], L['gate'], L['up'], self.cb, s, out_perm=d_inv)
            self._gemv(self.cb, L['down'], self.x, s, residual=self.x2, x_sorted=d_inv is not None)     # x = x2 + down(silu(gate) * up)
        self._lm_head(s)
        self.pos.add_(1)

    def _lm_head(self, s):
        """logits = lm_head(rmsnorm(x)): the model's final norm and its dense fp16 LM head in ONE hand-written launch (gptq_dense_matmat_f16,
        csrc/dense_gemv.hip: one pass over the weight for all rows of a batch); a head the kernel does not take (odd strides, dtypes) goes
        through the stand-alone norm + torch.matmul."""
        lm_head_logits(self, self.x, self.logits, s)

    def reset(self):
        self.pos.zero_()

    def _greedy_step(self):
        """one decode step whose argmax becomes the next input token, entirely on the device: ids <- argmax(logits),
        and the choice is appended to stream_out[pos] (pos = number of tokens consumed so far)."""
        self._step()                                   # consumes self.ids, advances pos
        torch.argmax(self.logits, dim=-1, out=self.ids)
        self.stream_out.index_copy_(0, self.pos, self.ids)

    def capture_greedy(self):
        """capture the self-feeding greedy step (engine_generate): a replay needs no host input at all."""
        with torch.no_grad():
            pos0, ids0 = self.pos.clone(), self.ids.clone()
            self._greedy_step()
            torch.cuda.synchronize(self.dev)
            self.pos.copy_(pos0); self.ids.copy_(ids0)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._greedy_step()
            self.greedy_graph = g
            self.pos.copy_(pos0); self.ids.copy_(ids0)
        return self

    def _greedy_rows_step(self):
        """the self-feeding greedy step for ANY batch: every row's argmax becomes its next input token, and the choices of step n are row n of
        stream_rows (n = stepc, a device counter the step advances itself)"""
        self._step()
        torch.argmax(self.logits, dim=-1, out=self.ids)
        self.stream_rows.index_copy_(0, self.stepc, self.ids.unsqueeze(0))
        self.stepc.add_(1)

    def capture_greedy_rows(self):
        """capture _greedy_rows_step (quant/engine_hook.py _greedy_fast: greedy model.generate without a host round trip per token)"""
        with torch.no_grad():
            if getattr(self, 'stream_rows', None) is None:
                self.stream_rows = torch.zeros((self.t_max + 1, self.batch), dtype=torch.int64, device=self.dev)
                self.stepc = torch.zeros(1, dtype=torch.int64, device=self.dev)
            pos0, ids0 = self.pos.clone(), self.ids.clone()
            self.stepc.zero_()
            self._greedy_rows_step()
            torch.cuda.synchronize(self.dev)
            self.pos.copy_(pos0); self.ids.copy_(ids0); self.stepc.zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._greedy_rows_step()
            self.greedy_rows_graph = g
            self.pos.copy_(pos0); self.ids.copy_(ids0); self.stepc.zero_()
        return self

    def _sample_rows_step(self, warpers):
        """the self-feeding SAMPLING step: HF's own arithmetic on the step's logits (generation/utils.py `_sample`: fp32 copy, the warpers in HF's order,
        softmax, torch.multinomial) -- inside the graph, where the default generator's Philox offset advances per replay exactly as per eager call"""
        self._step()
        scores = self.logits.to(copy=True, dtype=torch.float32)
        for w in warpers:
            scores = w(None, scores)
        probs = torch.nn.functional.softmax(scores, dim=-1)
        self.ids.copy_(torch.multinomial(probs, num_samples=1).squeeze(1))
        self.stream_rows.index_copy_(0, self.stepc, self.ids.unsqueeze(0))
        self.stepc.add_(1)

    def capture_sample_rows(self, key, warpers):
        """capture _sample_rows_step for one (temperature, top_k, top_p) setting; the generator's state is put back behind the warm-up run"""
        with torch.no_grad():
            if getattr(self, 'stream_rows', None) is None:
                self.stream_rows = torch.zeros((self.t_max + 1, self.batch), dtype=torch.int64, device=self.dev)
                self.stepc = torch.zeros(1, dtype=torch.int64, device=self.dev)
            rng = torch.cuda.get_rng_state(self.dev)
            pos0, ids0 = self.pos.clone(), self.ids.clone()
            self.stepc.zero_()
            self._sample_rows_step(warpers)
            torch.cuda.synchronize(self.dev)
            self.pos.copy_(pos0); self.ids.copy_(ids0); self.stepc.zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._sample_rows_step(warpers)
            self.pos.copy_(pos0); self.ids.copy_(ids0); self.stepc.zero_()
            torch.cuda.synchronize(self.dev)
            torch.cuda.set_rng_state(rng, self.dev)
            self.sample_graphs = {key: g}          # (one setting at a time: a graph holds the temporaries of a vocabulary-sized sort per row)
        return g

    def capture(self):
        """warm up once (module loads, workspace), then capture one decode step into a hipGraph."""
        with torch.no_grad():
            self.reset()
            self._step()
            torch.cuda.synchronize(self.dev)
            self.reset()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step()
            self.graph = g
            self.reset()
        return self

    def decode(self, token):
        """one token per row in ([batch] ids), logits [batch, vocab] out (the static buffer: clone it to keep it)."""
        if torch.is_tensor(token):
            self.ids.copy_(token.reshape(self.batch))
        else:
            self.ids.fill_(int(token))
        with torch.no_grad():
            if self.graph is not None:
                self.graph.replay()
            else:
                self._step()
        return self.logits


def _cache_layer_kv(cache, li):
    """(keys, values) [1, heads, T, head_dim] of layer li from a transformers cache (5.x: cache.layers[li].keys)."""
    layers = getattr(cache, 'layers', None)
    if layers is not None and hasattr(layers[li], 'keys'):
        return layers[li].keys, layers[li].values
    kv = cache[li]
    return kv[0], kv[1]


def engine_generate(model, input_ids, max_new_tokens, eos_token_id=None, engine=None, t_max=2048):
    """Greedy generation: the prompt goes through the HF model once (the drop-in modules' prefill path: MFMA GEMMs,
    fused MLP epilogue), its KV cache is copied into the engine's static cache, and every further token is ONE hipGraph
    replay (DecodeEngine).  Returns the full sequence [1, prompt + generated].  Batch 1; the reference equivalent is
    ``model.generate(input_ids, do_sample=False, max_new_tokens=...)`` through llama_inference.py:109-115."""
    from transformers.cache_utils import DynamicCache
    if input_ids.dim() != 2 or input_ids.shape[0] != 1:
        raise ValueError('engine_generate: batch 1 only')
    dev = input_ids.device
    T = input_ids.shape[1]
    eng = engine if engine is not None else DecodeEngine(model, t_max=t_max).capture()
    if T + max_new_tokens > eng.t_max:
        raise ValueError('engine_generate: prompt + max_new_tokens exceeds the engine cache (%d)' % eng.t_max)
    with torch.no_grad():
        cache = DynamicCache(config=model.config)
        out = model(input_ids, past_key_values=cache, use_cache=True)
        for li in range(len(eng.layers)):
            k, v = _cache_layer_kv(cache, li)
            eng.kc[li, :T].copy_(k[0].transpose(0, 1).reshape(T, -1))
            eng.vc[li, :T].copy_(v[0].transpose(0, 1).reshape(T, -1))
        eng.pos.fill_(T)
        first = out.logits[0, -1].argmax().reshape(1)
        del out, cache
        if eng.greedy_graph is None:
            eng.capture_greedy()
        eng.ids.copy_(first)
        eng.stream_out[T] = first[0]                         # stream_out[p] = token generated after p consumed tokens
        done = 1
        while done < max_new_tokens:
            burst = min(16, max_new_tokens - done)           # the host looks at the stream every 16 tokens only
            for _ in range(burst):
                eng.greedy_graph.replay()
            done += burst
            if eos_token_id is not None and bool((eng.stream_out[T:T + done] == eos_token_id).any()):
                break
        gen = eng.stream_out[T:T + done].clone()
        if eos_token_id is not None:
            hit = (gen == eos_token_id).nonzero()
            if hit.numel():
                gen = gen[:int(hit[0]) + 1]
    return torch.cat([input_ids[0], gen.to(input_ids.dtype)]).unsqueeze(0)


def benchmark_decode_engine(model, tokens=64, t_max=2048, seed=0, graph=True, fuse_norm=True, fuse_attn=True, start_pos=0, batch=1):
    """the llama.py:385-438 protocol on the DecodeEngine (hipGraph replay per token).  batch > 1: B sequences advance together, one
    replay per step; tokens_per_s is the aggregate (B tokens per step)."""
    dev = next(model.parameters()).device
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    eng = DecodeEngine(model, t_max=t_max, fuse_norm=fuse_norm, fuse_attn=fuse_attn, batch=batch)
    if graph:
        eng.capture()
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    input_ids = torch.randint(0, model.config.vocab_size, (batch, tokens), device=dev, generator=gen)
    times = []
    eng.reset()
    if start_pos:      # decode at depth: pretend start_pos tokens are already cached (random K/V rows)
        eng.kcb.normal_(0, 0.5)
        eng.vcb.normal_(0, 0.5)
        eng.pos.fill_(int(start_pos))
    for i in range(tokens):
        torch.cuda.synchronize(dev)
        tick = time.perf_counter()
        eng.decode(input_ids[:, i])
        torch.cuda.synchronize(dev)
        times.append(time.perf_counter() - tick)
    med = float(np.median(times[2:])) if len(times) > 4 else float(np.median(times))
    res = {'protocol': 'llama.py:385-438 (one token per step, KV cache, sync per step, median)',
           'mode': 'DecodeEngine, %s' % ('one hipGraph replay per token' if graph else 'eager launches'), 'tokens': tokens,
           't_max': t_max, 'start_pos': start_pos, 'fused_norm': fuse_norm, 'fused_attention': fuse_attn,
           # per layer: [norm,] qkv, [rope+append, attention partial, merge | one fused launch], o+residual, [norm,] gate/up, down+residual
           'launches_per_token': (9 - 2 * int(fuse_norm) - 2 * int(fuse_attn)) * len(eng.layers) + 4, 'median_s_per_token': round(med, 6),
           'tokens_per_s': round(batch / med, 1),
           # the llama.py:426-438 figure: peak bytes in use by tensors while decoding (model + derived copies + KV cache of t_max + engine buffers)
           'max_memory_MiB': round(torch.cuda.max_memory_allocated(dev) / 1024 / 1024, 1)}
    if batch > 1:
        res['batch'] = batch
        res['mode'] = 'DecodeEngine(batch=%d), %s' % (batch, 'one hipGraph replay per step' if graph else 'eager launches')
        res['median_s_per_step'] = res.pop('median_s_per_token')
        res.pop('launches_per_token')
    return res


def benchmark_decode_engine_context(model, contexts=(512, 1024, 2047), tokens=16, t_max=2048, batch=1, seed=0):
    """the same protocol at DEPTH: for every entry of `contexts` the engine's rows pretend to hold context - tokens cached tokens (random K / V
    rows) and decode `tokens` more, so the median step sees ~context tokens of history -- the regime of the reference's own measurement, which
    steps through --benchmark 2048 tokens and prints the median (llama.py:385-438; README.md:161).  ONE engine / graph for all depths."""
    dev = next(model.parameters()).device
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    eng = DecodeEngine(model, t_max=t_max, batch=batch).capture()
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    input_ids = torch.randint(0, model.config.vocab_size, (batch, tokens), device=dev, generator=gen)
    eng.kcb.normal_(0, 0.5)
    eng.vcb.normal_(0, 0.5)
    out = {}
    for ctx in contexts:
        start = max(0, min(int(ctx), t_max) - tokens)
        eng.pos.fill_(start)
        times = []
        for i in range(tokens):
            torch.cuda.synchronize(dev)
            tick = time.perf_counter()
            eng.decode(input_ids[:, i])
            torch.cuda.synchronize(dev)
            times.append(time.perf_counter() - tick)
        med = float(np.median(times[2:]))
        out['ctx%d' % ctx] = {'positions': [start, start + tokens - 1], 'median_s_per_step': round(med, 6), 'tokens_per_s': round(batch / med, 1)}
    out['protocol'] = 'llama.py:385-438 at depth: rows start with (context - %d) cached tokens, %d steps, sync per step, median' % (tokens, tokens)
    out['batch'] = batch
    out['attention'] = 'split records merged by o_proj' if eng.attn_records else 'self-merging launch'
    return out
