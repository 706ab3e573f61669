"""RMSNorm as a plain HIP kernel.  The module keeps the reference's file and class names
(``quant/triton_norm.py``: ``TritonLlamaRMSNorm`` :41-67, ``make_quant_norm`` :70-92) so that
callers importing them keep working; nothing here uses Triton.  Numerics follow the reference
kernel (:22-39): fp32 ``x * rsqrt(mean(x^2)+eps) * w`` with ONE rounding to fp16 (HF's own
LlamaRMSNorm rounds ``x*rstd`` to fp16 first)."""
import torch
from torch import nn

from . import _native


def rms_norm(x, weight, eps):
    _native.require_device(x, 'rms_norm')
    orig_dtype = x.dtype
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype != torch.float16:
        x2 = x2.half()
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M, N = x2.shape
    w = weight if (weight.dtype == torch.float16 and weight.is_contiguous()) else weight.half().contiguous()
    if N * 2 > 65536:
        raise RuntimeError("This layer norm doesn't support feature dim >= 64KB.")
    with _native.on_device(x.device):
        y = torch.empty((M, N), dtype=torch.float16, device=x.device)
        if M:
            rc = _native.lib().gptq_rmsnorm_f16(x2.data_ptr(), x2.stride(0) if M > 1 else N, w.data_ptr(), y.data_ptr(), N,
                                                M, N, float(eps), _native.stream_ptr(x.device))
            _native.check(rc, 'gptq_rmsnorm_f16')
    y = y.reshape(x.shape)
    return y if orig_dtype == torch.float16 else y.to(orig_dtype)


class TritonLlamaRMSNorm(nn.Module):

    def __init__(self, weight, eps=1e-6):
        super().__init__()
        self.weight = weight            # shares the original Parameter (reference :47,:79)
        self.variance_epsilon = eps

    def forward(self, x):
        return rms_norm(x, self.weight, self.variance_epsilon)


HipLlamaRMSNorm = TritonLlamaRMSNorm


def make_quant_norm(model):
    """Replace all LlamaRMSNorm modules by the HIP RMSNorm (reference :70-92)."""
    from transformers.models.llama.modeling_llama import LlamaRMSNorm

    for name, m in list(model.named_modules()):
        if not isinstance(m, LlamaRMSNorm):
            continue
        norm = TritonLlamaRMSNorm(m.weight, m.variance_epsilon)
        if '.' in name:
            parent_name, child_name = name.rsplit('.', 1)
            parent = model.get_submodule(parent_name)
        else:
            parent, child_name = model, name
        setattr(parent, child_name, norm)
