"""LLaMA MLP with gate/up fused into one dequant-matmul + SiLU*mul epilogue (HIP), then a
normal QuantLinear for down_proj -- drop-in for the reference ``quant/fused_mlp.py``
(``QuantLlamaMLP`` :177-238, ``make_fused_mlp`` :241-253, ``autotune_warmup_fused`` :256-288)."""
import torch
import torch.nn as nn

from . import _native
from .quant_linear import QuantLinear, _as_rows, _int32c, g_idx_is_trivial


PREFILL_SPLIT_M = 64


def _same_perm(a, b):
    """a == b elementwise, decided once per tensor pair (the comparison synchronises: not inside a hipGraph capture).
    The memo holds a WEAK reference to b and is valid only while that very object is alive (object ids and
    addresses are recycled)."""
    import weakref
    from .quant_linear import _ver
    if a is b:
        return True
    memo = getattr(a, '_gptq_same_as', None)
    key = (_ver(a), _ver(b))
    if memo is None or memo[0]() is not b or memo[1] != key:
        memo = (weakref.ref(b), key, bool(torch.equal(a, b)))
        try:
            a._gptq_same_as = memo
        except Exception:  # pragma: no cover
            pass
    return memo[2]


def fused_gate_up(x, gate, up, bits, groupsize, family=None):
    """c = silu(x . deq(gate)) * (x . deq(up)); gate/up = (qweight, scales, qzeros, g_idx).
    family='abi' (tests / A-B runs) skips the stripe16 images and calls gptq_fused_mlp_f16 on the checkpoint layout."""
    _native.require_device(x, 'fused_gate_up')
    x2 = _as_rows(x.reshape(-1, x.shape[-1]))
    M, K = x2.shape
    N = gate[0].shape[1]
    gis = []
    for (qw, sc, qz, gi) in (gate, up):
        gis.append(None if (gi is None or g_idx_is_trivial(gi, K, groupsize)) else _int32c(gi[:K]))
    stripe_rows = 8      # M <= 8: the stripe kernel wins while M rows of x fit in LDS; wider batches -> weight-streaming MFMA kernel
    if family is None and 1 <= M <= (stripe_rows if N <= 4608 else 4) and bits in (2, 3, 4, 8) and all(gi is None for gi in gis):
        # decode (and batches of up to 4 rows): gate and up packed into ONE stripe16 image, silu(gate) * up in the kernel epilogue
        from .quant_linear import stripe_copy, stripe_matvec
        st = stripe_copy(_int32c(gate[0]), gate[1], _int32c(gate[2]), bits, groupsize, up=(_int32c(up[0]), up[1], _int32c(up[2])))
        if st is not None:
            with torch.cuda.device(x.device):
                c = torch.empty((M, N), device=x.device, dtype=torch.float16)
                if stripe_matvec(x2, st, c, K, N, bits, groupsize, nsets=2, strict=False):
                    return c
    if family in (None, 'stripe_mm') and 4 < M <= 128 and bits in (2, 3, 4, 8) and all(gi is None for gi in gis):
        # small batches: the pair image through 16-row MFMA tiles (csrc/stripe_mm.inc), SiLU pair in the (reduce) epilogue
        from .quant_linear import stripe_copy, stripe_matmul
        st = stripe_copy(_int32c(gate[0]), gate[1], _int32c(gate[2]), bits, groupsize, up=(_int32c(up[0]), up[1], _int32c(up[2])))
        if st is not None:
            with torch.cuda.device(x.device):
                c = torch.empty((M, N), device=x.device, dtype=torch.float16)
                if stripe_matmul(x2, st, c, K, N, bits, groupsize, nsets=2, strict=False):
                    return c
    if family == 'stripe_mm':
        raise RuntimeError('fused_gate_up: the stripe16 MFMA kernel does not serve this shape')
    if family is None and M == 1 and bits in (2, 4, 8) and all(gi is not None for gi in gis):
        # act-order MLP at decode: gate and up share their input, hence their act-order permutation -> one x gather,
        # two group-sorted weight copies (cached on the tensors), the trivial-g_idx fused kernel
        from .quant_linear import act_order_sorted, stripe_copy, stripe_matvec
        sg = act_order_sorted(_int32c(gate[0]), gis[0], K, groupsize, bits)
        su = act_order_sorted(_int32c(up[0]), gis[1], K, groupsize, bits)
        if sg is not None and su is not None and _same_perm(sg[1], su[1]):
            st = stripe_copy(sg[0], gate[1], _int32c(gate[2]), bits, groupsize, up=(su[0], up[1], _int32c(up[2])))
            if st is not None:
                with torch.cuda.device(x.device):
                    c = torch.empty((M, N), device=x.device, dtype=torch.float16)
                    stripe_matvec(x2, st, c, K, N, bits, groupsize, nsets=2, perm=sg[1])
                return c
            with torch.cuda.device(x.device):
                c = torch.empty((M, N), device=x.device, dtype=torch.float16)
                ws = _native.workspace(x.device)
                rc = _native.lib().gptq_fused_mlp_sorted_f16(x2.data_ptr(), K, sg[1].data_ptr(), sg[0].data_ptr(), gate[1].data_ptr(),
                                                             gate[2].data_ptr(), su[0].data_ptr(), up[1].data_ptr(), up[2].data_ptr(),
                                                             c.data_ptr(), N, M, K, N, bits, groupsize, ws.data_ptr(), ws.numel(),
                                                             _native.stream_ptr(x.device))
            if rc != -6:   # GPTQ_E_VARIANT: shape not served by the rowwave kernel -> generic path below
                _native.check(rc, 'gptq_fused_mlp_sorted_f16')
                return c
    if family is None and M > PREFILL_SPLIT_M:
        from .quant_linear import _mid_m, _prefill_operand
        if _mid_m(M, N):
            # prefill (gptq_prefill_fused_mlp_f16): gate | up dequantised side by side into ONE [K, 2N] fp16 matrix (our kernel, any
            # width, any g_idx), one hipBLASLt GEMM per chunk of 16 384 rows (bounds the transient [rows, 2N] product: 0.7 GB at
            # 2N = 22 016), then silu(gate) * up in fp32 as a pass of its own (gptq_silu_mul_f16).  The reference's kernel applies
            # SiLU to the fp32 accumulators (fused_mlp.py:160-165); here gate and up are rounded to fp16 first, like its unfused
            # modules do -- inside the parity budget, tested against the oracle.
            lib = _native.lib()
            with torch.cuda.device(x.device):
                xs = _prefill_operand(x2)
                c = torch.empty((M, N), device=x.device, dtype=torch.float16)
                ws = torch.empty(lib.gptq_prefill_workspace_bytes(M, K, N, 2), dtype=torch.uint8, device=x.device)
                sg, su = gate[1] if gate[1].is_contiguous() else gate[1].contiguous(), up[1] if up[1].is_contiguous() else up[1].contiguous()
                rc = lib.gptq_prefill_fused_mlp_f16(xs.data_ptr(), xs.stride(0), _int32c(gate[0]).data_ptr(), sg.data_ptr(), _int32c(gate[2]).data_ptr(),
                                                    _native.ptr(gis[0]), _int32c(up[0]).data_ptr(), su.data_ptr(), _int32c(up[2]).data_ptr(),
                                                    _native.ptr(gis[1]), c.data_ptr(), N, M, K, N, bits, groupsize, ws.data_ptr(), ws.numel(),
                                                    _native.stream_ptr(x.device))
            from .quant_linear import _library_refused
            if not _library_refused(rc, 'gptq_prefill_fused_mlp_f16'):
                return c
        # GPTQ_PREFILL=fused, large prefill: gptq_fused_mlp_f16 runs two MFMA-tile GEMMs, the second applies silu(gate) * up in
        # place in its epilogue -- no intermediates, no extra pass over the [M, N] activations (falls through to the call below)
    with torch.cuda.device(x.device):
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

    def forward(self, x):
        return self.down_proj(self.hip_llama_mlp(x))

    def hip_llama_mlp(self, x):
        out_shape = x.shape[:-1] + (self.intermediate_size, )
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
