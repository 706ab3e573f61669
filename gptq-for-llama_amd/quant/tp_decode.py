"""Tensor-parallel decode step: BASELINE config 5 at the MODEL level -- one process per GPU, every rank owns a shard of every
quantised linear, one hipGraph replay per token and rank.

The reference's multi-GPU mode places whole LAYERS on different GPUs (llama.py:328-382: one GPU works while the others wait).
Here a token's work is cut the Megatron way, so that every GPU streams 1/P of the weights of every layer:
  qkv_proj   columns of this rank's heads (q | k | v of heads r H/P .. (r+1) H/P), RMSNorm fused, no exchange
  attention  this rank's heads over its own slice of the KV cache
  o_proj     rows of this rank's heads (a K shard on group boundaries): fp32 partial -> exchange, + residual, ONE fp16 rounding
  gate / up  a column shard of the intermediate size (cut on down_proj's group boundaries), SiLU pair fused, no exchange
  down_proj  the matching row shard: fp32 partial -> exchange, + residual
The hidden state, the norm weights, the embedding and lm_head are replicated.  Two exchanges per layer, each ONE launch of the
one-shot all-reduce over IPC peer mappings (csrc/p2p.hip: push to every peer, flag, rank-ordered local sum -- bit-identical on
every rank, which keeps the replicated hidden state replicated), captured in the graph with the kernels.  Every matvec runs on a
stripe16 image of the rank's shard (csrc/stripe.hip).  Requirements: head_dim 128, heads divisible by the world size, shard lengths
that have a stripe image (multiples of the row block).

Round 5:
  * SHARD AT LOAD.  ``TPDecodeEngine(checkpoint=..., config=...)`` reads a checkpoint in the reference's format (the state_dict
    ``load_quant`` loads, llama_inference.py:27-72: ``model.layers.N.self_attn.q_proj.qweight`` ...) through a ``CheckpointSource`` -- a
    dict of CPU tensors or a ``.safetensors`` path, of which only the requested slices are read -- and puts ONLY this rank's rows /
    columns on the device: a 65B model never exists unsharded on any GPU (the constructor from a materialised model slices a full
    copy: fine for tests, 32 GB per rank for 65B).  The shard tensors are dropped once their stripe16 images are built.
  * bias: qkv bias rides in the matvec's epilogue (column shard); o_proj / down_proj bias is added once, after the exchange, with the
    module chain's rounding order (fp16(fp16(sum) + bias), then + residual).
  * act-order: column shards (qkv, gate / up) take the group-sorted image + in-kernel gather of a regular act-order layer; a ROW shard
    of an act-order layer touches every group irregularly (its k range is fixed by the heads / by gate-up's columns), so it runs the
    generic g_idx kernel on the checkpoint rows with the whole layer's scale / zero tables (correct, slower) -- round 6: with an fp32
    partial like every other shard (gptq_matmul248_partial_f32): fp32 partials, one all-reduce, ONE rounding per linear.
"""
import numpy as np
import torch

from . import _native, fused_attn, fused_mlp, quant_linear, tensor_parallel as TP
from .p2p import P2PAllReduce


class CheckpointSource:
    """slices of a reference-format checkpoint without materialising it: ``src`` is a dict of (CPU) tensors or the path of a
    .safetensors file (safe_open + get_slice: only the bytes of the requested rows / columns are read)."""

    def __init__(self, src):
        self._dict, self._st = None, None
        if isinstance(src, dict):
            self._dict = src
        else:
            from safetensors import safe_open
            self._st = safe_open(str(src), framework='pt', device='cpu')
        self.bytes_read = 0            # what a rank actually pulled (tests: ~ 1 / world of the packed weights + the replicated tensors)

    def has(self, key):
        return key in self._dict if self._dict is not None else key in self._st.keys()

    def get(self, key, rows=None, cols=None):
        """contiguous CPU copy of tensor[rows[0]:rows[1], cols[0]:cols[1]] (None = everything along that axis)"""
        if self._dict is not None:
            t = self._dict[key]
            if rows is not None:
                t = t[rows[0]:rows[1]]
            if cols is not None:
                t = t[:, cols[0]:cols[1]]
            t = t.detach().cpu().contiguous().clone() if (rows is not None or cols is not None) else t.detach().cpu().contiguous()
        else:
            sl = self._st.get_slice(key)
            if rows is None and cols is None:
                t = self._st.get_tensor(key)
            elif cols is None:
                t = sl[rows[0]:rows[1]]
            elif rows is None:
                t = sl[:, cols[0]:cols[1]]
            else:
                t = sl[rows[0]:rows[1], cols[0]:cols[1]]
            t = t.contiguous()
        self.bytes_read += t.numel() * t.element_size()
        return t


def _cols(qweight, scales, qzeros, bits, c0, c1):
    """column range [c0, c1) (multiples of 32) of a packed layer"""
    z0, z1 = c0 * bits // 32, c1 * bits // 32
    return qweight[:, c0:c1], scales[:, c0:c1], qzeros[:, z0:z1]


def _rows(qweight, scales, qzeros, bits, groupsize, k0, k1):
    """row range [k0, k1) (multiples of the group size) of a packed layer"""
    r0, r1 = k0 * bits // 32, k1 * bits // 32
    g0, g1 = k0 // groupsize, k1 // groupsize
    return qweight[r0:r1].contiguous(), scales[g0:g1].contiguous(), qzeros[g0:g1].contiguous()


class _ModelShards:
    """this rank's shards out of a materialised (fused) model: slices of the modules' device tensors"""

    def __init__(self, model):
        self.model = model
        cfg = model.config
        self.n_layers = len(model.model.layers)
        self.embed, self.lm_head, self.final_norm = model.model.embed_tokens.weight, model.lm_head.weight, model.model.norm.weight
        self.lm_head_bias = getattr(model.lm_head, 'bias', None)
        self.H, self.I = cfg.hidden_size, cfg.intermediate_size

    def layer(self, li):
        layer = self.model.model.layers[li]
        attn, mlp = layer.self_attn, layer.mlp
        if not isinstance(attn, fused_attn.QuantLlamaAttention) or not isinstance(mlp, fused_mlp.QuantLlamaMLP):
            raise RuntimeError('TPDecodeEngine needs make_quant_attn / make_fused_mlp applied first')
        for m in (attn.qkv_proj, attn.o_proj, mlp.down_proj, mlp):
            if getattr(m, '_released', None) is not None:
                m.restore_checkpoint()                 # the shards are cut out of the checkpoint layout
        q, o, d, H = attn.qkv_proj, attn.o_proj, mlp.down_proj, self.H
        i32 = quant_linear._int32c
        return dict(
            ln1=layer.input_layernorm.weight, ln2=layer.post_attention_layernorm.weight, theta=float(attn.rope_theta), bits=q.bits,
            gs_qkv=q.groupsize if q.groupsize != -1 else q.infeatures, gs_o=o.groupsize if o.groupsize != -1 else o.infeatures,
            gs_mlp=mlp.groupsize if mlp.groupsize != -1 else mlp.infeatures, gs_d=d.groupsize if d.groupsize != -1 else d.infeatures,
            # column getters: (j, c0, c1) -> columns [c0, c1) of q (j = 0) / k / v; gate / up likewise; row getters: (k0, k1)
            qkv_cols=lambda j, c0, c1: _cols(i32(q.qweight), q.scales, i32(q.qzeros), q.bits, j * H + c0, j * H + c1),
            qkv_bias=lambda j, c0, c1: None if q.bias is None else q.bias[j * H + c0:j * H + c1],
            qkv_g_idx=q.g_idx[:q.infeatures],
            o_rows=lambda k0, k1, gs: _rows(i32(o.qweight), o.scales, i32(o.qzeros), o.bits, gs, k0, k1),
            o_all=lambda k0, k1: (i32(o.qweight)[k0 * o.bits // 32:k1 * o.bits // 32].contiguous(), o.scales, i32(o.qzeros)),
            o_g_idx=o.g_idx, o_bias=o.bias,
            gate_cols=lambda c0, c1: _cols(i32(mlp.gate_proj_qweight), mlp.gate_proj_scales, i32(mlp.gate_proj_qzeros), mlp.bits, c0, c1),
            up_cols=lambda c0, c1: _cols(i32(mlp.up_proj_qweight), mlp.up_proj_scales, i32(mlp.up_proj_qzeros), mlp.bits, c0, c1),
            gate_g_idx=mlp.gate_proj_g_idx, up_g_idx=mlp.up_proj_g_idx,
            d_rows=lambda k0, k1, gs: _rows(i32(d.qweight), d.scales, i32(d.qzeros), d.bits, gs, k0, k1),
            d_all=lambda k0, k1: (i32(d.qweight)[k0 * d.bits // 32:k1 * d.bits // 32].contiguous(), d.scales, i32(d.qzeros)),
            d_g_idx=d.g_idx, d_bias=d.bias)


class _CheckpointShards:
    """this rank's shards straight from a reference-format checkpoint: only the slices asked for ever reach the device"""

    def __init__(self, source, config, bits, groupsize, device):
        self.src = source if isinstance(source, CheckpointSource) else CheckpointSource(source)
        self.cfg, self.bits, self.groupsize, self.dev = config, int(bits), int(groupsize), device
        self.n_layers = config.num_hidden_layers
        self.H, self.I = config.hidden_size, config.intermediate_size
        to = lambda k: self.src.get(k).to(device)
        self.embed, self.lm_head, self.final_norm = to('model.embed_tokens.weight').half(), to('lm_head.weight').half(), to('model.norm.weight').half()
        self.lm_head_bias = to('lm_head.bias').half() if self.src.has('lm_head.bias') else None
        rp = getattr(config, 'rope_parameters', None)
        self.theta = float(rp['rope_theta']) if isinstance(rp, dict) and 'rope_theta' in rp else float(getattr(config, 'rope_theta', None) or 10000.0)

    def layer(self, li):
        src, bits, dev, H, I = self.src, self.bits, self.dev, self.H, self.I
        p = 'model.layers.%d.' % li
        gs_h = self.groupsize if self.groupsize != -1 else H
        gs_i = self.groupsize if self.groupsize != -1 else I

        def cols(name, c0, c1):
            z0, z1 = c0 * bits // 32, c1 * bits // 32
            return (src.get(name + '.qweight', cols=(c0, c1)).to(dev), src.get(name + '.scales', cols=(c0, c1)).to(dev).half(),
                    src.get(name + '.qzeros', cols=(z0, z1)).to(dev))

        def rows(name, k0, k1, gs):
            return (src.get(name + '.qweight', rows=(k0 * bits // 32, k1 * bits // 32)).to(dev), src.get(name + '.scales', rows=(k0 // gs, k1 // gs)).to(dev).half(),
                    src.get(name + '.qzeros', rows=(k0 // gs, k1 // gs)).to(dev))

        def rows_all_groups(name, k0, k1):
            return (src.get(name + '.qweight', rows=(k0 * bits // 32, k1 * bits // 32)).to(dev), src.get(name + '.scales').to(dev).half(), src.get(name + '.qzeros').to(dev))

        def opt(name, sl=None):
            if not src.has(name):
                return None
            t = src.get(name)
            return (t if sl is None else t[sl[0]:sl[1]]).to(dev)
        qkv = ('self_attn.q_proj', 'self_attn.k_proj', 'self_attn.v_proj')
        return dict(
            ln1=src.get(p + 'input_layernorm.weight').to(dev).half(), ln2=src.get(p + 'post_attention_layernorm.weight').to(dev).half(), theta=self.theta,
            bits=bits, gs_qkv=gs_h, gs_o=gs_h, gs_mlp=gs_h, gs_d=gs_i,
            qkv_cols=lambda j, c0, c1: cols(p + qkv[j], c0, c1),
            qkv_bias=lambda j, c0, c1: opt(p + qkv[j] + '.bias', (c0, c1)),
            qkv_g_idx=opt(p + qkv[0] + '.g_idx'),
            o_rows=lambda k0, k1, gs: rows(p + 'self_attn.o_proj', k0, k1, gs), o_all=lambda k0, k1: rows_all_groups(p + 'self_attn.o_proj', k0, k1),
            o_g_idx=opt(p + 'self_attn.o_proj.g_idx'), o_bias=opt(p + 'self_attn.o_proj.bias'),
            gate_cols=lambda c0, c1: cols(p + 'mlp.gate_proj', c0, c1), up_cols=lambda c0, c1: cols(p + 'mlp.up_proj', c0, c1),
            gate_g_idx=opt(p + 'mlp.gate_proj.g_idx'), up_g_idx=opt(p + 'mlp.up_proj.g_idx'),
            d_rows=lambda k0, k1, gs: rows(p + 'mlp.down_proj', k0, k1, gs), d_all=lambda k0, k1: rows_all_groups(p + 'mlp.down_proj', k0, k1),
            d_g_idx=opt(p + 'mlp.down_proj.g_idx'), d_bias=opt(p + 'mlp.down_proj.bias'))


def _trivial(g_idx, K, gs):
    return g_idx is None or quant_linear.g_idx_is_trivial(g_idx, K, gs)


class TPDecodeEngine:

    def __init__(self, model=None, t_max=2048, group=None, checkpoint=None, config=None, bits=None, groupsize=None, device=None):
        """model: a materialised fused model (every rank holds all of it; the shards are sliced out) -- or checkpoint = CheckpointSource /
        dict / .safetensors path + config (+ bits, groupsize, device): only this rank's shards are read and placed on the device."""
        import torch.distributed as dist
        from .layer import prepared
        self.native, self.lib = _native, _native.lib()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if model is not None:
            cfg = model.config
            self.dev = next(model.parameters()).device
            shards = _ModelShards(model)
        else:
            if checkpoint is None or config is None or bits is None or groupsize is None:
                raise ValueError('TPDecodeEngine: a model, or checkpoint= with config=, bits=, groupsize=')
            cfg = config
            self.dev = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
            shards = _CheckpointShards(checkpoint, config, bits, groupsize, self.dev)
        self.source = getattr(shards, 'src', None)
        self.t_max = int(t_max)
        self.hidden, self.heads = cfg.hidden_size, cfg.num_attention_heads
        self.head_dim = self.hidden // self.heads
        self.eps = float(cfg.rms_norm_eps)
        self.fuse_norm = True
        if self.head_dim != 128 or self.heads % self.world:
            raise NotImplementedError('TPDecodeEngine: head_dim 128 and heads divisible by the world size')
        self.hl = self.heads // self.world            # this rank's heads
        self.Hl = self.hl * self.head_dim
        H, I, r, P = self.hidden, cfg.intermediate_size, self.rank, self.world
        self.embed, self.lm_head, self.final_norm, self.lm_head_bias = shards.embed, shards.lm_head, shards.final_norm, shards.lm_head_bias
        self.layers = []
        with torch.cuda.device(self.dev):
            for li in range(shards.n_layers):
                S = shards.layer(li)
                bits = S['bits']
                L = dict(ln1=S['ln1'], ln2=S['ln2'], theta=S['theta'], bits=bits, gs_qkv=S['gs_qkv'], gs_o=S['gs_o'], gs_mlp=S['gs_mlp'], gs_d=S['gs_d'], keep=[])
                # ---- qkv: columns of this rank's heads (q | k | v), the same g_idx for all three (they share their input) ----
                parts = [S['qkv_cols'](j, r * self.Hl, (r + 1) * self.Hl) for j in range(3)]
                qkv = tuple(torch.cat([p[i] for p in parts], dim=1).contiguous() for i in range(3))
                bq = [S['qkv_bias'](j, r * self.Hl, (r + 1) * self.Hl) for j in range(3)]
                L['qkv_bias'] = torch.cat(bq).contiguous().half() if bq[0] is not None else None
                gq = S['qkv_g_idx']
                if _trivial(gq, H, S['gs_qkv']):
                    L['qkv'], L['qkv_perm'] = quant_linear.stripe_copy(*qkv, bits, S['gs_qkv']), None
                    L['keep'].append(qkv)            # (the image cache is keyed by these tensors)
                else:
                    pl = prepared(((qkv[0], qkv[1], qkv[2], quant_linear._int32c(gq[:H])),), None, bits, S['gs_qkv'], H, 3 * self.Hl)
                    L['qkv'], L['qkv_perm'] = pl.stripe, pl.perm16
                    L['keep'].append((pl, qkv))
                # ---- o_proj: rows of this rank's heads ----
                k0, k1 = r * self.Hl, (r + 1) * self.Hl
                if self.Hl % S['gs_o']:
                    raise NotImplementedError('TPDecodeEngine: the heads of a rank must cover whole groups of o_proj')
                L['o_bias'] = S['o_bias'].half() if S['o_bias'] is not None else None
                if _trivial(S['o_g_idx'], H, S['gs_o']):
                    o_sh = S['o_rows'](k0, k1, S['gs_o'])
                    L['o'], L['o_generic'] = quant_linear.stripe_copy(*o_sh, bits, S['gs_o']), None
                    L['keep'].append(o_sh)
                else:
                    L['o'], L['o_generic'] = None, S['o_all'](k0, k1) + (quant_linear._int32c(S['o_g_idx'][k0:k1]).contiguous(),)
                # ---- gate / up: a column shard cut on down_proj's group boundaries; down_proj: the matching rows ----
                i0, i1 = TP.row_shard_bounds(I, S['gs_d'], bits, P)[r]
                g_sh = tuple(t.contiguous() for t in S['gate_cols'](i0, i1))
                u_sh = tuple(t.contiguous() for t in S['up_cols'](i0, i1))
                gg, gu = S['gate_g_idx'], S['up_g_idx']
                if _trivial(gg, H, S['gs_mlp']) and _trivial(gu, H, S['gs_mlp']):
                    L['mlp'], L['mlp_perm'] = quant_linear.stripe_copy(*g_sh, bits, S['gs_mlp'], up=u_sh), None
                    L['keep'].append((g_sh, u_sh))
                else:
                    pl = prepared(((g_sh[0], g_sh[1], g_sh[2], quant_linear._int32c(gg[:H])), (u_sh[0], u_sh[1], u_sh[2], quant_linear._int32c(gu[:H]))), None, bits,
                                  S['gs_mlp'], H, i1 - i0)
                    if pl.kind != 1:
                        raise NotImplementedError('TPDecodeEngine: gate and up must share one regular act-order permutation')
                    L['mlp'], L['mlp_perm'] = pl.stripe, pl.perm16
                    L['keep'].append((pl, g_sh, u_sh))
                L['Il'] = i1 - i0
                L['d_bias'] = S['d_bias'].half() if S['d_bias'] is not None else None
                if _trivial(S['d_g_idx'], I, S['gs_d']):
                    d_sh = S['d_rows'](i0, i1, S['gs_d'])
                    L['down'], L['d_generic'] = quant_linear.stripe_copy(*d_sh, bits, S['gs_d']), None
                    L['keep'].append(d_sh)
                else:
                    L['down'], L['d_generic'] = None, S['d_all'](i0, i1) + (quant_linear._int32c(S['d_g_idx'][i0:i1]).contiguous(),)
                if any(L[k] is None for k in ('qkv', 'mlp')) or (L['o'] is None and L['o_generic'] is None) or (L['down'] is None and L['d_generic'] is None):
                    raise NotImplementedError('TPDecodeEngine: a shard has no stripe16 image (lengths must be multiples of the row block)')
                if model is None:
                    # shard at load: the images are all the decode step reads -- the shard tensors they were built from go (the image cache is
                    # keyed weakly by them and dies with them; the image tensors themselves are held by L; a prepared act-order shard releases
                    # its checkpoint rows: its image + permutation are a bijection of them)
                    kept = []
                    for k in L['keep']:
                        if isinstance(k, tuple) and k and hasattr(k[0], 'handle'):
                            k[0].release()
                            kept.append(k[0])
                    L['keep'] = kept
                self.layers.append(L)
                del S
        self.Il_max = max(L['Il'] for L in self.layers)
        f16, f32 = dict(dtype=torch.float16, device=self.dev), dict(dtype=torch.float32, device=self.dev)
        self.ids = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.pos = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.x, self.x2, self.h = torch.zeros((1, H), **f16), torch.zeros((1, H), **f16), torch.zeros((1, H), **f16)
        self.qkvb, self.ab, self.cb = torch.zeros((1, 3 * self.Hl), **f16), torch.zeros((1, self.Hl), **f16), torch.zeros((1, self.Il_max), **f16)
        self.part = torch.zeros((1, H), **f32)
        self.logits = torch.zeros((1, self.lm_head.shape[0]), **f16)
        nl = len(self.layers)
        self.kc, self.vc = torch.zeros((nl, self.t_max, self.Hl), **f16), torch.zeros((nl, self.t_max, self.Hl), **f16)
        self.attn_ws = torch.zeros(self.lib.gptq_decode_attn_workspace_bytes(self.hl, self.head_dim, self.t_max), dtype=torch.uint8, device=self.dev)
        self.rope = {}
        self.ws = _native.workspace(self.dev)
        self.p2p = P2PAllReduce(H, group=group, device=self.dev) if P > 1 else None
        self.graph = None

    def _norm_rows(self, x, w, y, s):
        rc = self.lib.gptq_rmsnorm_f16(x.data_ptr(), x.stride(0), w.data_ptr(), y.data_ptr(), y.stride(0), x.shape[0], self.hidden, self.eps, s)
        self.native.check(rc, 'gptq_rmsnorm_f16')

    def _exchange(self, out, residual, bias, s):
        """out = fp16(sum over the ranks of self.part) [+ bias, rounded] + residual -- the module chain's order: QuantLinear rounds its
        product, adds its bias (quant_linear.py:376), the decoder layer adds the residual"""
        if bias is None:
            if self.p2p is not None:
                self.p2p.allreduce(self.part, out=out, bias=residual)
            else:
                torch.add(self.part.half(), residual, out=out)
            return
        if self.p2p is not None:
            self.p2p.allreduce(self.part, out=out, bias=bias)
        else:
            torch.add(self.part.half(), bias, out=out)
        self.native.check(self.lib.gptq_add_rows_f16(out.data_ptr(), out.stride(0), residual.data_ptr(), residual.stride(0), 1, self.hidden, s), 'gptq_add_rows_f16')

    def _partial(self, x, st, generic, K, N, bits, gs, s):
        if st is not None:
            rc = self.lib.gptq_stripe_matvec_partial_f32(x.data_ptr(), st.data_ptr(), st.numel(), self.part.data_ptr(), K, N, bits, gs, 1, None, s)
            self.native.check(rc, 'gptq_stripe_matvec_partial_f32')
            return
        # a row shard of an act-order layer: its rows point into ALL groups of the layer -- the generic g_idx kernel on the checkpoint rows with
        # the whole layer's {scale, zero} tables, fp32 sums out (round 6: like every other shard -- fp32 partials, ONE rounding after the exchange;
        # until round 5 this launch stored fp16 and cost one extra rounding per rank)
        qw, sc, qz, gi = generic
        rc = self.lib.gptq_matmul248_partial_f32(x.data_ptr(), K, qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), gi.data_ptr(), self.part.data_ptr(), N, 1, K, N, bits,
                                                 sc.shape[0], s)
        self.native.check(rc, 'gptq_matmul248_partial_f32')

    def _step(self):
        lib, H = self.lib, self.hidden
        s = torch.cuda.current_stream(self.dev).cuda_stream
        torch.index_select(self.embed, 0, self.ids, out=self.x)
        scale = 1.0 / float(np.sqrt(self.head_dim))
        for li, L in enumerate(self.layers):
            bits = L['bits']
            quant_linear.stripe_matvec(self.x, L['qkv'], self.qkvb, H, 3 * self.Hl, bits, L['gs_qkv'], bias=L['qkv_bias'], norm_weight=L['ln1'], eps=self.eps,
                                       perm=L['qkv_perm'])
            tab = self.rope.get(L['theta'])
            if tab is None:
                tab = torch.empty((self.t_max, self.head_dim // 2, 2), dtype=torch.float32, device=self.dev)
                self.native.check(lib.gptq_rope_table_f32(tab.data_ptr(), self.t_max, self.head_dim, L['theta'], s), 'gptq_rope_table_f32')
                self.rope[L['theta']] = tab
            rc = lib.gptq_decode_attn_fused_table_f16(self.qkvb.data_ptr(), self.pos.data_ptr(), self.kc[li].data_ptr(), self.vc[li].data_ptr(),
                                                      self.ab.data_ptr(), self.attn_ws.data_ptr(), self.attn_ws.numel(), self.hl, self.head_dim,
                                                      self.t_max, L['theta'], scale, tab.data_ptr(), s)
            self.native.check(rc, 'gptq_decode_attn_fused_table_f16')
            self._partial(self.ab, L['o'], L['o_generic'], self.Hl, H, bits, L['gs_o'], s)
            self._exchange(self.x2, self.x, L['o_bias'], s)                      # x2 = x + o_proj(attn)
            c = self.cb[:, :L['Il']]
            quant_linear.stripe_matvec(self.x2, L['mlp'], c, H, L['Il'], bits, L['gs_mlp'], nsets=2, norm_weight=L['ln2'], eps=self.eps, perm=L['mlp_perm'])
            self._partial(c, L['down'], L['d_generic'], L['Il'], H, bits, L['gs_d'], s)
            self._exchange(self.x, self.x2, L['d_bias'], s)                      # x = x2 + down(silu(gate) * up)
        from .decode import lm_head_logits
        lm_head_logits(self, self.x, self.logits, s)      # replicated dense fp16 head: final norm + matvec in one launch (csrc/dense_gemv.hip)
        self.pos.add_(1)

    def reset(self):
        self.pos.zero_()

    def capture(self):
        """warm up once, then capture one decode step (kernels AND the two exchanges per layer) into a hipGraph.  Collective: every
        rank must call it, and every rank must replay the same number of times."""
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
        """one token in (the same on every rank), logits [1, vocab] out (the same on every rank; a static buffer)."""
        if torch.is_tensor(token):
            self.ids.copy_(token.reshape(1))
        else:
            self.ids.fill_(int(token))
        with torch.no_grad():
            if self.graph is not None:
                self.graph.replay()
            else:
                self._step()
        return self.logits

    def status(self):
        return 0 if self.p2p is None else self.p2p.status()
