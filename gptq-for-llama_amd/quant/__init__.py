"""Drop-in ``quant`` package: the public names of the reference's ``quant/__init__.py:1-5`` backed
by the MI355X HIP library (``include/gptq_mi355x.h``).  Put the parent directory
(``gptq-for-llama_amd/``) on ``sys.path`` / ``PYTHONPATH`` and ``import quant`` as before."""
from .quantizer import Quantizer
from .fused_attn import QuantLlamaAttention, make_quant_attn
from .fused_mlp import QuantLlamaMLP, make_fused_mlp, autotune_warmup_fused
from .quant_linear import QuantLinear, make_quant_linear, autotune_warmup_linear
from .triton_norm import TritonLlamaRMSNorm, make_quant_norm

# spellings of the old-cuda branch, named by BASELINE.json's north_star
from .quant_linear import make_quant, autotune_warmup

from . import quant_linear, fused_mlp, fused_attn, triton_norm, quantizer, tensor_parallel  # noqa: F401
