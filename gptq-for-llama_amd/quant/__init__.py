"""Drop-in ``quant`` package for the MI355X HIP library (``include/gptq_mi355x.h``).

Put the parent directory (``gptq-for-llama_amd/``) on ``sys.path`` / ``PYTHONPATH`` and ``import quant``
exactly as with the reference tree: every public name of the reference's ``quant/__init__.py`` (:1-5) is
re-exported below from the sub-module of the same file name, plus the two spellings of the old-cuda
branch that ``BASELINE.json``'s north_star uses (``make_quant``, ``autotune_warmup``).
"""
from . import quantizer, quant_linear, fused_attn, fused_mlp, triton_norm, tensor_parallel, decode  # noqa: F401

_PUBLIC = {
    quantizer: ('Quantizer', ),
    quant_linear: ('QuantLinear', 'make_quant_linear', 'autotune_warmup_linear', 'make_quant', 'autotune_warmup'),
    fused_attn: ('QuantLlamaAttention', 'make_quant_attn'),
    fused_mlp: ('QuantLlamaMLP', 'make_fused_mlp', 'autotune_warmup_fused'),
    triton_norm: ('TritonLlamaRMSNorm', 'make_quant_norm'),
}
for _mod, _names in _PUBLIC.items():
    for _n in _names:
        globals()[_n] = getattr(_mod, _n)
__all__ = [n for names in _PUBLIC.values() for n in names]
del _mod, _names, _n
