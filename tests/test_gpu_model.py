"""Model-level GPU tests: the module surgery of the drop-in package on the INSTALLED transformers
 -- a tiny random LLaMA whose linears are replaced by QuantLinear + fused
attention / norm / MLP must produce the logits of the same model with dense fp16 linears holding
the oracle-dequantised weights and stock HF modules, for prefill and for cached decode steps."""
import numpy as np
import pytest
import torch

import quant
from quant import decode as D
from oracle import oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TINY = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
            vocab_size=512, max_position_embeddings=128)


def dense_twin(qmodel_unfused, cfg_overrides):
    """stock HF model whose nn.Linear weights are the oracle's dequantisation of the packed ones."""
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = dict(D.LLAMA_7B)
    cfg.update(cfg_overrides)
    ref = LlamaForCausalLM(LlamaConfig(**cfg)).half().to(DEV).eval()
    sd = {k: v for k, v in qmodel_unfused.state_dict().items()}
    with torch.no_grad():
        for name, m in qmodel_unfused.named_modules():
            if isinstance(m, quant.QuantLinear):
                W = oracle.dequant(m.qweight.cpu().numpy(), m.qzeros.cpu().numpy(), m.scales.cpu().numpy(),
                                   m.g_idx.cpu().numpy(), m.bits)                     # [K, N] fp16-rounded like the kernel
                ref.get_submodule(name).weight.copy_(torch.from_numpy(np.asarray(W, dtype=np.float32).T).half())
        for name, p in ref.named_parameters():
            if name in sd and sd[name].shape == p.shape and 'proj' not in name:
                p.copy_(sd[name])
    return ref


def run_steps(model, ids, prefill):
    from transformers.cache_utils import DynamicCache
    cache = DynamicCache(config=model.config)
    outs = []
    with torch.no_grad():
        out = model(ids[:, :prefill], past_key_values=cache, use_cache=True)
        outs.append(out.logits[:, -1].float().cpu().numpy())
        for i in range(prefill, ids.shape[1]):
            out = model(ids[:, i:i + 1], past_key_values=cache, use_cache=True)
            outs.append(out.logits[:, -1].float().cpu().numpy())
    return np.stack(outs)


@pytest.mark.parametrize('bits,gs', [(4, 128), (8, 64), (2, 32)])
def test_tiny_llama_logits_match_dense_twin(bits, gs):
    q_unfused = D.build_random_llama(DEV, bits=bits, groupsize=gs, seed=bits, fused=False, **TINY)
    ref = dense_twin(q_unfused, TINY)
    q = D.build_random_llama(DEV, bits=bits, groupsize=gs, seed=bits, fused=True, **TINY)
    kinds = {type(m).__name__ for m in q.modules()}
    assert {'QuantLlamaAttention', 'QuantLlamaMLP', 'TritonLlamaRMSNorm'} <= kinds
    ids = torch.randint(0, TINY['vocab_size'], (1, 9), device=DEV)
    a, b, c = run_steps(q, ids, 5), run_steps(q_unfused, ids, 5), run_steps(ref, ids, 5)
    scale = np.abs(c).max()
    assert np.isfinite(a).all()
    assert np.abs(b - c).max() / scale < 2e-2      # QuantLinear inside stock HF attention / MLP / norm
    assert np.abs(a - c).max() / scale < 2e-2      # + fused qkv/RoPE, fused MLP, HIP RMSNorm
    assert (a.argmax(-1) == c.argmax(-1)).mean() >= 0.75


def test_benchmark_decode_protocol_runs():
    q = D.build_random_llama(DEV, seed=1, **TINY)
    r = D.benchmark_decode(q, tokens=12)
    assert r['tokens_per_s'] > 0 and r['tokens'] == 12
