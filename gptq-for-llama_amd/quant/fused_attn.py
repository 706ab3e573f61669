"""LLaMA attention with q/k/v fused into ONE QuantLinear, in-place HIP RoPE on the q and k slices
of the fused output, KV cache update, torch SDPA and o_proj -- drop-in for the reference
``quant/fused_attn.py`` (``triton_rotate_half_`` :61-93, ``QuantLlamaAttention`` :96-161,
``make_quant_attn`` :164-204).

Two call contracts are served:
  * the reference's (transformers 4.28): ``forward(hidden_states, past_key_value=(k, v),
    attention_mask, position_ids, output_attentions, use_cache)`` -> 3-tuple;
  * the installed transformers' decoder layer: keyword call with ``past_key_values=<Cache>``,
    ``position_embeddings``, ``position_ids`` ... -> 2-tuple.
RoPE numerics follow the reference kernel: cos/sin computed on the fly in fp32 from
``position_ids`` (theta from the model config, 10000 upstream), rotation in fp32, fp16 store.
"""
import torch
import torch.nn as nn
from torch.nn import functional as F

from . import _native
from .quant_linear import QuantLinear


def hip_rotate_half_(qk, position_ids, base=10000.0):
    """In-place RoPE on ``qk [bsz, seq, 2, heads, head_dim]`` (a view into the fused qkv output).
    Same layout asserts as the reference (fused_attn.py:69-74)."""
    _native.require_device(qk, 'rotate_half_')
    batch_size, seq_len, qandk, num_heads, head_dim = qk.shape
    assert qandk == 2
    assert qk.dtype == torch.float16
    assert qk.stride(3) == head_dim
    assert qk.stride(4) == 1
    assert qk.stride(2) == num_heads * head_dim
    assert position_ids.shape == (batch_size, seq_len)
    assert position_ids.stride(1) == 1, 'position_ids must be contiguous in the last dimension'
    assert seq_len == 1 or qk.stride(0) == qk.stride(1) * seq_len or batch_size == 1
    if position_ids.dtype != torch.int64:
        position_ids = position_ids.long()
    rc = _native.lib().gptq_rope_f16(qk.data_ptr(), qk.stride(1) if seq_len > 1 else qk.stride(0), position_ids.data_ptr(),
                                     position_ids.stride(0), batch_size, seq_len, num_heads, head_dim, float(base),
                                     _native.stream_ptr(qk.device))
    _native.check(rc, 'gptq_rope_f16')


triton_rotate_half_ = hip_rotate_half_   # reference name


class QuantLlamaAttention(nn.Module):
    """Multi-headed attention from 'Attention Is All You Need' paper"""

    def __init__(self, hidden_size, num_heads, qkv_proj, o_proj, layer_idx=None, rope_theta=10000.0):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.head_dim = hidden_size // num_heads
        if (self.head_dim * num_heads) != self.hidden_size:
            raise ValueError(f"hidden_size must be divisible by num_heads (got `hidden_size`: {self.hidden_size}"
                             f" and `num_heads`: {num_heads}).")
        self.qkv_proj = qkv_proj
        self.o_proj = o_proj
        self.layer_idx = layer_idx
        self.rope_theta = rope_theta

    def forward(self, hidden_states, past_key_value=None, attention_mask=None, position_ids=None, output_attentions=False,
                use_cache=False, past_key_values=None, position_embeddings=None, cache_position=None, **kwargs):
        """Input shape: Batch x Time x Channel"""
        bsz, q_len, _ = hidden_states.size()
        new_api = (past_key_values is not None and hasattr(past_key_values, 'update')) or position_embeddings is not None

        qkv_states = self.qkv_proj(hidden_states)
        qkv_states = qkv_states.view(bsz, q_len, 3, self.num_heads, self.head_dim)

        if new_api and past_key_values is not None:
            past_len = past_key_values.get_seq_length(self.layer_idx)
        elif past_key_value is not None:
            past_len = past_key_value[0].shape[-2]
        else:
            past_len = 0
        if position_ids is None:
            if cache_position is not None:
                position_ids = cache_position.view(1, -1).expand(bsz, -1).contiguous()
            else:
                position_ids = torch.arange(past_len, past_len + q_len, device=hidden_states.device).view(1, -1).expand(
                    bsz, -1).contiguous()
        elif position_ids.dim() == 2 and position_ids.shape[0] == 1 and bsz > 1:
            position_ids = position_ids.expand(bsz, -1).contiguous()     # current transformers hands one row for the whole batch
        elif position_ids.stride(-1) != 1:
            position_ids = position_ids.contiguous()

        # This updates the query and key states in-place, saving VRAM.
        hip_rotate_half_(qkv_states[:, :, :2], position_ids, self.rope_theta)

        query_states, key_states, value_states = (qkv_states[:, :, i].transpose(1, 2) for i in range(3))
        del qkv_states

        if new_api:
            if past_key_values is not None:
                key_states, value_states = past_key_values.update(key_states, value_states, self.layer_idx)
        else:
            if past_key_value is not None:
                key_states = torch.cat([past_key_value[0], key_states], dim=2)
                value_states = torch.cat([past_key_value[1], value_states], dim=2)
            if use_cache:
                # the views keep the whole fused qkv tensor alive; copy like the reference (:145-150)
                key_states, value_states, query_states = (key_states.contiguous(), value_states.contiguous(),
                                                          query_states.contiguous())
            past_key_value = (key_states, value_states) if use_cache else None

        kv_len = key_states.shape[-2]
        mask = None
        if attention_mask is not None and attention_mask.dim() == 4:
            mask = attention_mask[:, :, :, :kv_len]
        if mask is not None:
            attn_output = F.scaled_dot_product_attention(query_states, key_states, value_states, attn_mask=mask)
        elif q_len > 1 and kv_len > q_len:
            causal = torch.ones(q_len, kv_len, dtype=torch.bool, device=query_states.device).tril(kv_len - q_len)
            attn_output = F.scaled_dot_product_attention(query_states, key_states, value_states, attn_mask=causal)
        else:
            attn_output = F.scaled_dot_product_attention(query_states, key_states, value_states, is_causal=q_len > 1)
        del query_states, key_states, value_states

        attn_output = attn_output.transpose(1, 2).reshape(bsz, q_len, self.hidden_size)
        attn_output = self.o_proj(attn_output)

        if new_api:
            return attn_output, None
        return attn_output, None, past_key_value


def _attn_geometry(m):
    """(hidden_size, num_heads, num_kv_heads, layer_idx, rope_theta) for old and new HF modules."""
    cfg = getattr(m, 'config', None)
    if hasattr(m, 'hidden_size') and hasattr(m, 'num_heads'):
        hidden, heads = m.hidden_size, m.num_heads
        kv = getattr(m, 'num_key_value_heads', heads)
    else:
        heads = cfg.num_attention_heads
        hidden = heads * getattr(m, 'head_dim', cfg.hidden_size // heads)
        kv = getattr(cfg, 'num_key_value_heads', heads) or heads
    theta = 10000.0
    if cfg is not None:
        rp = getattr(cfg, 'rope_parameters', None)
        if isinstance(rp, dict) and 'rope_theta' in rp:
            theta = float(rp['rope_theta'])
        elif getattr(cfg, 'rope_theta', None):
            theta = float(cfg.rope_theta)
    return hidden, heads, kv, getattr(m, 'layer_idx', None), theta


def make_quant_attn(model):
    """Replace all LlamaAttention modules with QuantLlamaAttention modules, fusing the q, k, v
    projections (buffers concatenated along N; reference :164-204).  Grouped-query attention
    layers are left untouched (the reference's fused layout assumes num_kv_heads == num_heads)."""
    from transformers.models.llama.modeling_llama import LlamaAttention

    for name, m in list(model.named_modules()):
        if not isinstance(m, LlamaAttention):
            continue
        q_proj, k_proj, v_proj = m.q_proj, m.k_proj, m.v_proj
        if not all(isinstance(p, QuantLinear) for p in (q_proj, k_proj, v_proj)):
            continue
        hidden, heads, kv_heads, layer_idx, theta = _attn_geometry(m)
        if kv_heads != heads:
            continue

        qweights = torch.cat([q_proj.qweight, k_proj.qweight, v_proj.qweight], dim=1)
        qzeros = torch.cat([q_proj.qzeros, k_proj.qzeros, v_proj.qzeros], dim=1)
        scales = torch.cat([q_proj.scales, k_proj.scales, v_proj.scales], dim=1)
        g_idx = torch.cat([q_proj.g_idx, k_proj.g_idx, v_proj.g_idx], dim=0)
        bias = torch.cat([q_proj.bias, k_proj.bias, v_proj.bias], dim=0) if q_proj.bias is not None else None

        qkv_layer = QuantLinear(q_proj.bits, q_proj.groupsize, q_proj.infeatures,
                                q_proj.outfeatures + k_proj.outfeatures + v_proj.outfeatures, q_proj.bias is not None)
        qkv_layer.qweight = qweights
        qkv_layer.qzeros = qzeros
        qkv_layer.scales = scales
        qkv_layer.g_idx = g_idx     # length 3K like the reference; the kernels read the first K
        qkv_layer.bias = bias

        attn = QuantLlamaAttention(hidden, heads, qkv_layer, m.o_proj, layer_idx=layer_idx, rope_theta=theta)

        if '.' in name:
            parent_name, child_name = name.rsplit('.', 1)
            parent = model.get_submodule(parent_name)
        else:
            parent, child_name = model, name
        setattr(parent, child_name, attn)

    # one-token forwards of the fused model (HF generate, the benchmark() protocol) are answered by the hipGraph decode
    # engine from here on; everything else reaches the original forward (quant/engine_hook.py)
    from .engine_hook import install_decode_engine
    install_decode_engine(model)
