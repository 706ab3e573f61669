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
stripe16 image of the rank's shard (csrc/stripe.hip).  Requirements: head_dim 128, heads divisible by the world size, trivial
g_idx, shard lengths that have a stripe image (multiples of the row block)."""
import numpy as np
import torch

from . import _native, fused_attn, fused_mlp, quant_linear, tensor_parallel as TP
from .p2p import P2PAllReduce


def _cols(qweight, scales, qzeros, bits, c0, c1):
    """column range [c0, c1) (multiples of 32) of a packed layer"""
    z0, z1 = c0 * bits // 32, c1 * bits // 32
    return qweight[:, c0:c1], scales[:, c0:c1], qzeros[:, z0:z1]


def _rows(qweight, scales, qzeros, bits, groupsize, k0, k1):
    """row range [k0, k1) (multiples of the group size) of a packed layer"""
    r0, r1 = k0 * bits // 32, k1 * bits // 32
    g0, g1 = k0 // groupsize, k1 // groupsize
    return qweight[r0:r1].contiguous(), scales[g0:g1].contiguous(), qzeros[g0:g1].contiguous()


class TPDecodeEngine:

    def __init__(self, model, t_max=2048, group=None):
        import torch.distributed as dist
        self.native, self.lib = _native, _native.lib()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        cfg = model.config
        self.dev = next(model.parameters()).device
        self.t_max = int(t_max)
        self.hidden, self.heads = cfg.hidden_size, cfg.num_attention_heads
        self.head_dim = self.hidden // self.heads
        self.eps = float(cfg.rms_norm_eps)
        if self.head_dim != 128 or self.heads % self.world:
            raise NotImplementedError('TPDecodeEngine: head_dim 128 and heads divisible by the world size')
        self.hl = self.heads // self.world            # this rank's heads
        self.Hl = self.hl * self.head_dim
        H, I, r, P = self.hidden, cfg.intermediate_size, self.rank, self.world
        self.embed, self.lm_head, self.final_norm = model.model.embed_tokens.weight, model.lm_head.weight, model.model.norm.weight
        self.layers = []
        for layer in model.model.layers:
            attn, mlp = layer.self_attn, layer.mlp
            if not isinstance(attn, fused_attn.QuantLlamaAttention) or not isinstance(mlp, fused_mlp.QuantLlamaMLP):
                raise RuntimeError('TPDecodeEngine needs make_quant_attn / make_fused_mlp applied first')
            q, o, d = attn.qkv_proj, attn.o_proj, mlp.down_proj
            bits, gs = q.bits, (q.groupsize if q.groupsize != -1 else q.infeatures)
            for lin in (q, o, d):
                if lin.bias is not None or not quant_linear.g_idx_is_trivial(lin.g_idx, lin.infeatures, lin.groupsize if lin.groupsize != -1 else lin.infeatures):
                    raise NotImplementedError('TPDecodeEngine: layers without bias and with a trivial g_idx')
            qw, sc, qz = quant_linear._int32c(q.qweight), q.scales, quant_linear._int32c(q.qzeros)
            parts = [_cols(qw, sc, qz, bits, j * H + r * self.Hl, j * H + (r + 1) * self.Hl) for j in range(3)]   # q | k | v of this rank's heads
            qkv = tuple(torch.cat([p[i] for p in parts], dim=1).contiguous() for i in range(3))
            k0, k1 = r * self.Hl, (r + 1) * self.Hl
            gs_o = o.groupsize if o.groupsize != -1 else o.infeatures
            if self.Hl % gs_o:
                raise NotImplementedError('TPDecodeEngine: the heads of a rank must cover whole groups of o_proj')
            o_sh = _rows(quant_linear._int32c(o.qweight), o.scales, quant_linear._int32c(o.qzeros), bits, gs_o, k0, k1)
            gs_d = d.groupsize if d.groupsize != -1 else d.infeatures
            i0, i1 = TP.row_shard_bounds(I, gs_d, bits, P)[r]            # down_proj's K shard = gate / up's column shard
            gsm = mlp.groupsize if mlp.groupsize != -1 else mlp.infeatures
            g_sh = tuple(t.contiguous() for t in _cols(quant_linear._int32c(mlp.gate_proj_qweight), mlp.gate_proj_scales,
                                                       quant_linear._int32c(mlp.gate_proj_qzeros), bits, i0, i1))
            u_sh = tuple(t.contiguous() for t in _cols(quant_linear._int32c(mlp.up_proj_qweight), mlp.up_proj_scales,
                                                       quant_linear._int32c(mlp.up_proj_qzeros), bits, i0, i1))
            d_sh = _rows(quant_linear._int32c(d.qweight), d.scales, quant_linear._int32c(d.qzeros), bits, gs_d, i0, i1)
            L = dict(ln1=layer.input_layernorm.weight, ln2=layer.post_attention_layernorm.weight, theta=float(attn.rope_theta), bits=bits,
                     qkv=quant_linear.stripe_copy(*qkv, bits, gs), gs_qkv=gs,
                     o=quant_linear.stripe_copy(*o_sh, bits, gs_o), gs_o=gs_o,
                     mlp=quant_linear.stripe_copy(*g_sh, bits, gsm, up=u_sh), gs_mlp=gsm,
                     down=quant_linear.stripe_copy(*d_sh, bits, gs_d), gs_d=gs_d, Il=i1 - i0,
                     keep=(qkv, o_sh, g_sh, u_sh, d_sh))     # the images are cached on these tensors
            if any(L[k] is None for k in ('qkv', 'o', 'mlp', 'down')):
                raise NotImplementedError('TPDecodeEngine: a shard has no stripe16 image (lengths must be multiples of the row block)')
            self.layers.append(L)
        self.Il_max = max(L['Il'] for L in self.layers)
        f16, f32 = dict(dtype=torch.float16, device=self.dev), dict(dtype=torch.float32, device=self.dev)
        self.ids = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.pos = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.x, self.x2, self.h = torch.zeros((1, H), **f16), torch.zeros((1, H), **f16), torch.zeros((1, H), **f16)
        self.qkvb, self.ab, self.cb = torch.zeros((1, 3 * self.Hl), **f16), torch.zeros((1, self.Hl), **f16), torch.zeros((1, self.Il_max), **f16)
        self.part = torch.zeros((1, H), **f32)
        self.logits = torch.zeros((1, cfg.vocab_size), **f16)
        nl = len(self.layers)
        self.kc, self.vc = torch.zeros((nl, self.t_max, self.Hl), **f16), torch.zeros((nl, self.t_max, self.Hl), **f16)
        self.attn_ws = torch.zeros(self.lib.gptq_decode_attn_workspace_bytes(self.hl, self.head_dim, self.t_max), dtype=torch.uint8, device=self.dev)
        self.rope = {}
        self.p2p = P2PAllReduce(H, group=group, device=self.dev) if P > 1 else None
        self.graph = None

    def _exchange(self, out, residual):
        """out = fp16(sum over the ranks of self.part) + residual"""
        if self.p2p is not None:
            self.p2p.allreduce(self.part, out=out, bias=residual)
        else:
            torch.add(self.part.half(), residual, out=out)

    def _partial(self, x, st, K, N, bits, gs, s):
        rc = self.lib.gptq_stripe_matvec_partial_f32(x.data_ptr(), st.data_ptr(), st.numel(), self.part.data_ptr(), K, N, bits, gs, 1, None, s)
        self.native.check(rc, 'gptq_stripe_matvec_partial_f32')

    def _step(self):
        lib, H = self.lib, self.hidden
        s = torch.cuda.current_stream(self.dev).cuda_stream
        torch.index_select(self.embed, 0, self.ids, out=self.x)
        scale = 1.0 / float(np.sqrt(self.head_dim))
        for li, L in enumerate(self.layers):
            bits = L['bits']
            quant_linear.stripe_matvec(self.x, L['qkv'], self.qkvb, H, 3 * self.Hl, bits, L['gs_qkv'], norm_weight=L['ln1'], eps=self.eps)
            tab = self.rope.get(L['theta'])
            if tab is None:
                tab = torch.empty((self.t_max, self.head_dim // 2, 2), dtype=torch.float32, device=self.dev)
                self.native.check(lib.gptq_rope_table_f32(tab.data_ptr(), self.t_max, self.head_dim, L['theta'], s), 'gptq_rope_table_f32')
                self.rope[L['theta']] = tab
            rc = lib.gptq_decode_attn_fused_table_f16(self.qkvb.data_ptr(), self.pos.data_ptr(), self.kc[li].data_ptr(), self.vc[li].data_ptr(),
                                                      self.ab.data_ptr(), self.attn_ws.data_ptr(), self.attn_ws.numel(), self.hl, self.head_dim,
                                                      self.t_max, L['theta'], scale, tab.data_ptr(), s)
            self.native.check(rc, 'gptq_decode_attn_fused_table_f16')
            self._partial(self.ab, L['o'], self.Hl, H, bits, L['gs_o'], s)
            self._exchange(self.x2, self.x)                                      # x2 = x + o_proj(attn)
            c = self.cb[:, :L['Il']]
            quant_linear.stripe_matvec(self.x2, L['mlp'], c, H, L['Il'], bits, L['gs_mlp'], nsets=2, norm_weight=L['ln2'], eps=self.eps)
            self._partial(c, L['down'], L['Il'], H, bits, L['gs_d'], s)
            self._exchange(self.x, self.x2)                                      # x = x2 + down(silu(gate) * up)
        W = self.lm_head      # replicated dense fp16 head: final norm + matvec in one hand-written launch (csrc/dense_gemv.hip)
        rc = -6
        if W.dtype == torch.float16 and W.stride(1) == 1 and W.stride(0) % 8 == 0 and W.shape[1] % 8 == 0:
            rc = lib.gptq_dense_matvec_f16(self.x.data_ptr(), W.data_ptr(), W.stride(0), None, self.logits.data_ptr(), W.shape[0], W.shape[1],
                                           self.final_norm.data_ptr(), self.eps, s)
        if rc == -6:
            rc = lib.gptq_rmsnorm_f16(self.x.data_ptr(), H, self.final_norm.data_ptr(), self.h.data_ptr(), H, 1, H, self.eps, s)
            self.native.check(rc, 'gptq_rmsnorm_f16')
            torch.matmul(self.h, self.lm_head.t(), out=self.logits)
        else:
            self.native.check(rc, 'gptq_dense_matvec_f16')
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
