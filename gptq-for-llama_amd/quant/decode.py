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


def build_random_llama(dev='cuda:0', bits=4, groupsize=128, seed=0, fused=True, **overrides):
    """random-init LLaMA (7B shape by default) with every decoder linear replaced by a QuantLinear
    holding random packed weights, then fused attention / norm / MLP like load_quant()."""
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


def benchmark_decode(model, tokens=64, seed=0):
    """llama.py:385-438: feed ``input_ids[:, i:i+1]`` with the growing cache, sync after every
    step, report the median (and peak memory).  Returns a dict."""
    from transformers.cache_utils import DynamicCache

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
    med = float(np.median(times[2:])) if len(times) > 4 else float(np.median(times))
    return {'protocol': 'llama.py:385-438 (one token per step, KV cache, sync per step, median)', 'mode': 'eager HF decoder + drop-in modules',
            'tokens': tokens, 'median_s_per_token': round(med, 6), 'tokens_per_s': round(1.0 / med, 1),
            'max_memory_MiB': round(torch.cuda.max_memory_allocated(dev) / 1024 / 1024, 1)}
