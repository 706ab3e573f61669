"""LLaMA MLP with gate/up fused into one dequant-matmul + SiLU*mul epilogue (HIP), then a
normal QuantLinear for down_proj -- drop-in for the reference ``quant/fused_mlp.py``
(``QuantLlamaMLP`` :177-238, ``make_fused_mlp`` :241-253, ``autotune_warmup_fused`` :256-288)."""
import torch
import torch.nn as nn

from . import _native
from .quant_linear import QuantLinear, _as_rows, _int32c, g_idx_is_trivial


def fused_gate_up(x, gate, up, bits, groupsize, family=None):
    """c = silu(x . deq(gate)) * (x . deq(up)); gate/up = (qweight, scales, qzeros, g_idx) (reference fusedmatmul_248,
    quant/fused_mlp.py:84-168).  The call goes to the pair's prepared handle (quant/layer.py -> gptq_layer_forward with two weight
    sets: decode kernel / row groups / 16-row MFMA tiles on ONE stripe16 image of the pair, the tile GEMM with the SiLU pair
    epilogue at prefill sizes -- SiLU always sees fp32 sums, fused_mlp.py:160-165).  family (tests / A-B runs): 'abi' =
    gptq_fused_mlp_f16 on the checkpoint layout, 'stripe_mm' = the 16-row MFMA tiles."""
    _native.require_device(x, 'fused_gate_up')
    x2 = _as_rows(x.reshape(-1, x.shape[-1]))
    M, K = x2.shape
    N = gate[0].shape[1]
    if family is None:
        from . import quant_linear
        from .layer import prepared
        quant_linear._apply_prefill_route()
        with _native.on_device(x.device):
            c = torch.empty((M, N), device=x.device, dtype=torch.float16)
            if M == 0:
                return c
            return prepared((tuple(gate), tuple(up)), None, bits, groupsize if groupsize != -1 else K, K, N, sort=quant_linear.ACT_ORDER_SORT).forward(x2, c)
    gis = []
    for (qw, sc, qz, gi) in (gate, up):
        gis.append(None if (gi is None or g_idx_is_trivial(gi, K, groupsize)) else _int32c(gi[:K]))
    if family == 'stripe_mm':
        from .quant_linear import stripe_copy, stripe_matmul
        st = None
        if M <= 1024 and all(gi is None for gi in gis):
            st = stripe_copy(_int32c(gate[0]), gate[1], _int32c(gate[2]), bits, groupsize, up=(_int32c(up[0]), up[1], _int32c(up[2])))
        with _native.on_device(x.device):
            c = torch.empty((M, N), device=x.device, dtype=torch.float16)
            if st is not None and stripe_matmul(x2, st, c, K, N, bits, groupsize, nsets=2, strict=False):
                return c
        raise RuntimeError('fused_gate_up: the stripe16 MFMA kernel does not serve this shape')
    with _native.on_device(x.device):
        c = torch.empty((M, N), device=x.device, dtype=torch.float16)
        if M:
            ws = _native.workspace(x.device)
            rc = _native.lib().gptq_fused_mlp_f16(
                x2.data_ptr(), x2.stride(0) if M > 1 else K,
                gate[0].data_ptr(), gate[1].data_ptr(), gate[2].data_ptr(), _native.ptr(gis[0]),
                up[0].data_ptr(), up[1].data_ptr(), up[2].data_ptr(), _native.ptr(gis[1]),
                c.data_ptr(), N, M, K, N, bits, groupsize, ws.data_ptr(), ws.numel(), _native.stream_ptr(x.device))
            _native.check(rc, 'gptq_fused_mlp_f16')
    return c


class QuantLlamaMLP(nn.Module):

    def __init__(self, gate_proj, down_proj, up_proj):
        super().__init__()
        self.register_buffer('gate_proj_qweight', gate_proj.qweight)
        self.register_buffer('gate_proj_scales', gate_proj.scales)
        self.register_buffer('gate_proj_qzeros', gate_proj.qzeros)
        self.register_buffer('gate_proj_g_idx', gate_proj.g_idx)
        self.register_buffer('up_proj_qweight', up_proj.qweight)
        self.register_buffer('up_proj_scales', up_proj.scales)
        self.register_buffer('up_proj_qzeros', up_proj.qzeros)
        self.register_buffer('up_proj_g_idx', up_proj.g_idx)

        self.infeatures = gate_proj.infeatures
        self.intermediate_size = gate_proj.outfeatures
        self.outfeatures = down_proj.outfeatures
        self.bits = gate_proj.bits
        self.maxq = gate_proj.maxq
        self.groupsize = gate_proj.groupsize

        self.down_proj = down_proj
        self._released = None     # memory mode: the pair's PreparedLayer holds the only copy of gate / up (release_checkpoint)

    def forward(self, x):
        return self.down_proj(self.hip_llama_mlp(x))

    # memory mode, see QuantLinear.release_checkpoint: gate | up live on in ONE stripe16 image of the pair
    def release_checkpoint(self):
        if self._released is not None:
            return True
        if not self.gate_proj_qweight.is_cuda:
            return False
        from . import quant_linear
        from .layer import prepared
        pl = prepared(((self.gate_proj_qweight, self.gate_proj_scales, self.gate_proj_qzeros, self.gate_proj_g_idx),
                       (self.up_proj_qweight, self.up_proj_scales, self.up_proj_qzeros, self.up_proj_g_idx)), None, self.bits,
                      self.groupsize if self.groupsize != -1 else self.infeatures, self.infeatures, self.intermediate_size,
                      sort=quant_linear.ACT_ORDER_SORT)
        if not pl.release():
            return False
        self._released = pl
        dev, N = self.gate_proj_qweight.device, self.intermediate_size
        for p in ('gate_proj_', 'up_proj_'):
            setattr(self, p + 'qweight', torch.empty((0, N), dtype=torch.int32, device=dev))
            setattr(self, p + 'qzeros', torch.empty((0, N // 32 * self.bits), dtype=torch.int32, device=dev))
            setattr(self, p + 'scales', torch.empty((0, N), dtype=torch.float16, device=dev))
        return True

    def restore_checkpoint(self):
        if self._released is not None:
            from .layer import RESTORE_EPOCH
            pl, self._released = self._released, None
            for i, p in enumerate(('gate_proj_', 'up_proj_')):
                qw, sc, qz = pl.unpack(i)
                setattr(self, p + 'qweight', qw), setattr(self, p + 'scales', sc), setattr(self, p + 'qzeros', qz)
            RESTORE_EPOCH[0] += 1

    def _apply(self, fn, *args, **kwargs):        # see QuantLinear._apply: a released pair comes back before it really moves / is cast
        from .quant_linear import _moves_or_casts
        if self._released is not None and _moves_or_casts(fn, self.gate_proj_qweight.device):
            self.restore_checkpoint()
        return super()._apply(fn, *args, **kwargs)

    def __getstate__(self):
        self.restore_checkpoint()
        return super().__getstate__() if hasattr(super(), '__getstate__') else self.__dict__

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self._released is not None:
            for i, p in enumerate(('gate_proj_', 'up_proj_')):
                qw, sc, qz = self._released.unpack(i)
                destination[prefix + p + 'qweight'], destination[prefix + p + 'scales'], destination[prefix + p + 'qzeros'] = qw, sc, qz

    def _load_from_state_dict(self, *args, **kwargs):
        self.restore_checkpoint()
        return super()._load_from_state_dict(*args, **kwargs)

    def hip_llama_mlp(self, x):
        out_shape = x.shape[:-1] + (self.intermediate_size, )
        if self._released is not None:
            from . import quant_linear
            _native.require_device(x, 'QuantLlamaMLP')
            x2 = _as_rows(x.reshape(-1, x.shape[-1]))
            quant_linear._apply_prefill_route()
            with _native.on_device(x2.device):
                c = torch.empty((x2.shape[0], self.intermediate_size), device=x2.device, dtype=torch.float16)
                if x2.shape[0]:
                    self._released.forward(x2, c)
            return c.reshape(out_shape)
        c = fused_gate_up(x, (self.gate_proj_qweight, self.gate_proj_scales, self.gate_proj_qzeros, self.gate_proj_g_idx),
                          (self.up_proj_qweight, self.up_proj_scales, self.up_proj_qzeros, self.up_proj_g_idx), self.bits,
                          self.groupsize)
        return c.reshape(out_shape)

    triton_llama_mlp = hip_llama_mlp   # reference method name (fused_mlp.py:206)

    _FUSED = ('gate_proj_qweight', 'gate_proj_scales', 'gate_proj_qzeros', 'gate_proj_g_idx', 'up_proj_qweight',
              'up_proj_scales', 'up_proj_qzeros', 'up_proj_g_idx')

    def fused2cuda(self):
        for n in self._FUSED:
            setattr(self, n, getattr(self, n).cuda())

    def fused2cpu(self):
        for n in self._FUSED:
            setattr(self, n, getattr(self, n).cpu())


def make_fused_mlp(m, parent_name=''):
    """Replace all LlamaMLP modules (whose projections are QuantLinear) with QuantLlamaMLP."""
    from transformers.models.llama.modeling_llama import LlamaMLP

    if isinstance(m, LlamaMLP):
        if all(isinstance(p, QuantLinear) for p in (m.gate_proj, m.down_proj, m.up_proj)):
            return QuantLlamaMLP(m.gate_proj, m.down_proj, m.up_proj)
        return m

    for name, child in m.named_children():
        child = make_fused_mlp(child, parent_name=f"{parent_name}.{name}")
        if isinstance(child, QuantLlamaMLP):
            setattr(m, name, child)
    return m


def autotune_warmup_fused(model):
    """Reference surface (fused_mlp.py:256-288); with the static dispatch table this only makes
    the fused kernels resident for every unique (K, N) at M = 1 .. 2048."""
    from tqdm import tqdm

    kn_values = {}
    for _, m in model.named_modules():
        if not isinstance(m, QuantLlamaMLP):
            continue
        k, n = m.infeatures, m.intermediate_size
        m.fused2cuda()
        if (k, n) not in kn_values:
            kn_values[(k, n)] = m

    print(f'Found {len(kn_values)} unique fused mlp KN values.')
    print('Warming up autotune cache ...')
    with torch.no_grad():
        for m in tqdm(range(0, 12)):
            m = 2**m  # [1, 2048]
            for (k, n), (modules) in kn_values.items():
                a = torch.randn(m, k, dtype=torch.float16, device='cuda')
                modules.hip_llama_mlp(a)
        for (k, n), (modules) in kn_values.items():
            modules.fused2cpu()
    del kn_values
