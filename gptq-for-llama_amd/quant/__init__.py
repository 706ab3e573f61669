"""Drop-in ``quant`` package for the MI355X HIP library (``include/gptq_mi355x.h``).

Put the parent directory (``gptq-for-llama_amd/``) on ``sys.path`` / ``PYTHONPATH`` and ``import quant``
exactly as with the reference tree: every public name of the reference's ``quant/__init__.py`` (:1-5) is
re-exported below from the sub-module of the same file name, plus the two spellings of the old-cuda
branch that ``BASELINE.json``'s north_star uses (``make_quant``, ``autotune_warmup``).
"""
from . import layer, quantizer, quant_linear, fused_attn, fused_mlp, triton_norm, tensor_parallel, decode  # noqa: F401

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


def release_checkpoint(model):
    """Memory mode for a whole model (extension; applied when the decode engine is first built unless GPTQ_RELEASE_CHECKPOINT=0): every
    QuantLinear / QuantLlamaMLP whose decode path runs on its stripe16 image frees qweight / scales / qzeros and keeps that ONE copy
    of the packed weights -- the footprint the reference quotes (README.md:23-29: 4891 MiB for 7B 4-bit g128) instead of two
    copies.  ``state_dict()`` is unchanged (tensors reproduced bit-exactly from the images).  Returns (released, kept) counts."""
    import torch
    from .engine_hook import drop_decode_engines
    drop_decode_engines(model)    # engines built before hold references to the buffers about to be freed: rebuilt on the next decode step
    st = getattr(model, '_gptq_engine_state', None)
    if st is not None:
        st.released = True
    done = kept = 0
    for m in model.modules():
        if isinstance(m, (quant_linear.QuantLinear, fused_mlp.QuantLlamaMLP)):
            if m.release_checkpoint():
                done += 1
            else:
                kept += 1
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return done, kept


def restore_checkpoint(model):
    for m in model.modules():
        if isinstance(m, (quant_linear.QuantLinear, fused_mlp.QuantLlamaMLP)):
            m.restore_checkpoint()
