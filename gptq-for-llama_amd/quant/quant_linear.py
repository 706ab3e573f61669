"""MI355X-native ``QuantLinear``: drop-in for the reference ``quant/quant_linear.py``.

Same module surface and checkpoint format as the reference --
``QuantLinear(bits, groupsize, infeatures, outfeatures, bias)`` with buffers
``qweight / qzeros / scales / g_idx [/ bias]`` (reference quant/quant_linear.py:306-323), ``pack``
(:325-371), ``forward`` (:373-377), ``matmul248`` / ``transpose_matmul248`` (:263-279),
``QuantLinearFunction`` (:282-301), ``make_quant_linear`` (:380-390) and
``autotune_warmup_linear`` (:393-423) -- but the device code is hand-written HIP for gfx950
behind the C ABI in ``include/gptq_mi355x.h`` (no Triton, no autotuner: a static shape ->
kernel table, see csrc/capi.hip).  There is no CPU forward: a CPU tensor raises.

Extension over the reference: ``bits == 3`` is accepted (dense 96-bit stream layout, see the
header); the reference raises NotImplementedError for it (:308-309).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _native

SUPPORTED_BITS = (2, 3, 4, 8)


# ----------------------------------------------------------------------------------------------
# g_idx bookkeeping: the fast kernels want to know whether g_idx is the trivial k // groupsize map
# (always true unless the checkpoint was quantised with --act-order, reference gptq.py:210-216).
# One device-side check per (buffer, version), cached; done during warm-up / first forward.
# ----------------------------------------------------------------------------------------------
def _infer_groupsize(K, G):
    return K if G <= 1 else -(-K // G)


from torch.utils.weak import WeakTensorKeyDictionary   # identity-keyed, entries die with the tensor: no attributes on tensors

from .layer import _ver, prepared

# derived state of the helper functions below (the engines that drive the stripe kernels themselves use them; the modules go
# through quant.layer.prepared): tensor object -> (validity key, value)
_TRIVIAL, _SORTED, _STRIPE, _U16 = (WeakTensorKeyDictionary() for _ in range(4))


def g_idx_is_trivial(g_idx, K, groupsize):
    """True iff g_idx[:K] == arange(K) // groupsize.  The verdict is memoised per tensor OBJECT
    (validated by its version counter), never by address: the caching allocator hands the address of
    a freed g_idx to the next one."""
    memo = _TRIVIAL.get(g_idx)
    key = (_ver(g_idx), K, groupsize)
    if memo is not None and memo[0] == key:
        return memo[1]
    g = g_idx[:K]
    if g.is_cuda:
        out = torch.empty(1, dtype=torch.int32, device=g.device)
        gi = g if g.dtype == torch.int32 and g.is_contiguous() else g.to(torch.int32).contiguous()
        rc = _native.lib().gptq_g_idx_is_trivial(gi.data_ptr(), K, groupsize, out.data_ptr(),
                                                 _native.stream_ptr(g.device))
        _native.check(rc, 'gptq_g_idx_is_trivial')
        res = bool(out.item())
    else:
        res = bool(torch.equal(g.to(torch.int64), torch.arange(K, dtype=torch.int64) // groupsize))
    _TRIVIAL[g_idx] = (key, res)
    return res


# ----------------------------------------------------------------------------------------------
# Act-order fast path: sort the packed rows by group once (stable argsort of g_idx) so that the
# layer becomes a trivial-g_idx layer of x[perm].  The re-sorted copy of qweight
# and perm are cached per qweight tensor object (validated by the version counters of qweight and g_idx); the
# checkpoint buffers themselves are never modified.  Costs one extra copy of qweight per act-order
# layer; set GPTQ_ACT_ORDER_SORT=0 to keep the generic g_idx-table kernel instead.
# ----------------------------------------------------------------------------------------------
import os as _os

ACT_ORDER_SORT = _os.environ.get('GPTQ_ACT_ORDER_SORT', '1') != '0'


def act_order_sorted(qweight, g_idx, K, groupsize, bits):
    """(qweight_sorted, perm int32 [K]) or None when the layer is not a regular act-order layer
    (every group exactly ``groupsize`` members, as gptq.py:210-216 produces) or bits == 3."""
    if not ACT_ORDER_SORT or bits not in (2, 4, 8) or not qweight.is_cuda:
        return None
    f = 32 // bits
    if groupsize % f != 0 or K % groupsize != 0:
        return None
    memo = _SORTED.get(qweight)
    key = (_ver(qweight), _ver(g_idx), g_idx.data_ptr(), K, groupsize)
    if memo is not None and memo[0] == key:
        return memo[1]
    g = g_idx[:K].to(torch.int64)
    G = K // groupsize
    res = None
    if int(g.min()) >= 0 and int(g.max()) < G and bool((torch.bincount(g, minlength=G) == groupsize).all()):
        perm = torch.argsort(g, stable=True).to(torch.int32).contiguous()
        qs = torch.empty_like(qweight)
        rc = _native.lib().gptq_act_order_repack(qweight.data_ptr(), perm.data_ptr(), K, qweight.shape[1], bits, qs.data_ptr(),
                                                 _native.stream_ptr(qweight.device))
        _native.check(rc, 'gptq_act_order_repack')
        res = (qs, perm)
    _SORTED[qweight] = (key, res)
    return res


# ----------------------------------------------------------------------------------------------
# stripe16 decode path (csrc/stripe.hip): at M == 1 a 4-bit layer is served from a load-time repacked
# copy in which every workgroup's 16 output columns are contiguous (no K split, no combine atomics).
# The copy is built once per weight set (or gate/up pair) by gptq_stripe_repack and cached per qweight
# tensor object, validated by the version counters of the checkpoint buffers -- which stay untouched and remain the
# state_dict.  Costs one extra copy of the packed weights; GPTQ_STRIPE=0 keeps the rowwave kernels.
# ----------------------------------------------------------------------------------------------
STRIPE = _os.environ.get('GPTQ_STRIPE', '1') != '0'


def stripe_copy(qweight, scales, qzeros, bits, groupsize, up=None):
    """uint8 tensor holding the stripe16 image of (qweight, scales, qzeros) -- or of the pair with
    ``up = (qweight, scales, qzeros)`` for the fused gate/up matvec -- or None when the shape is not served
    (K not a multiple of the row block -- 256 / 128 / 128 / 64 k for 2 / 3 / 4 / 8 bits --, group size not a power-of-two multiple of a
    quarter of it ...)."""
    if not STRIPE or bits not in (2, 3, 4, 8) or not qweight.is_cuda:
        return None
    K, N = qweight.shape[0] * 32 // bits, qweight.shape[1]
    lib = _native.lib()
    nsets = 2 if up is not None else 1
    nbytes = lib.gptq_stripe_bytes(K, N, bits, groupsize, nsets)
    if nbytes == 0:
        return None
    srcs = (qweight, scales, qzeros) + (tuple(up) if up is not None else ())
    key = tuple((_ver(t), t.data_ptr(), tuple(t.shape)) for t in srcs) + (groupsize,)
    memo = _STRIPE.get(qweight)
    if memo is not None and memo[0] == key:
        return memo[1]
    if any(t.dtype != d or not t.is_contiguous() for t, d in zip(srcs, (torch.int32, torch.float16, torch.int32) * 2)):
        return None
    with torch.cuda.device(qweight.device):
        st = torch.empty(nbytes, dtype=torch.uint8, device=qweight.device)
        u = up if up is not None else (None, None, None)
        rc = lib.gptq_stripe_repack(qweight.data_ptr(), scales.data_ptr(), qzeros.data_ptr(), _native.ptr(u[0]), _native.ptr(u[1]),
                                    _native.ptr(u[2]), st.data_ptr(), nbytes, K, N, bits, groupsize, _native.stream_ptr(qweight.device))
    _native.check(rc, 'gptq_stripe_repack')
    _STRIPE[qweight] = (key, st)
    return st


def perm_u16(perm):
    """the act-order permutation (int32 [K], act_order_sorted) as the uint16 vector the stripe16 kernels read (cached on the tensor;
    K <= 24576 on that path).  None stays None."""
    if perm is None:
        return None
    if perm.dtype == torch.int16:      # already narrowed (PreparedLayer.perm16)
        return perm
    p16 = _U16.get(perm)
    if p16 is None:
        p16 = perm.to(torch.int16)          # bit pattern of the uint16 value: every index is below 32768
        _U16[perm] = p16
    return p16


def stripe_matvec(x, st, out, K, N, bits, groupsize, nsets=1, bias=None, norm_weight=None, eps=0.0, perm=None, strict=True, out_perm=None):
    """out[M, N] = x[M, K] (1 <= M <= 4) through a stripe16 image (gptq_stripe_matvec_f16) on the current stream.
    strict=False: return False instead of raising when the kernel does not serve the call (GPTQ_E_VARIANT: e.g. four rows of a
    very long K do not fit in LDS) so that the caller can take another kernel family.
    out_perm (int32 [N], round 5): column n of the result is stored at out_perm[n] -- the consumer's sorted order (an act-order layer)."""
    M = x.shape[0]
    if out_perm is not None:
        rc = _native.lib().gptq_stripe_matvec_perm_out_f16(x.data_ptr(), x.stride(0) if M > 1 else K, st.data_ptr(), st.numel(), _native.ptr(bias),
                                                           out.data_ptr(), out.stride(0) if M > 1 else N, M, K, N, bits, groupsize, nsets,
                                                           _native.ptr(norm_weight), float(eps), _native.ptr(perm_u16(perm)), out_perm.data_ptr(),
                                                           _native.stream_ptr(x.device))
        if rc == -6 and not strict:
            return False
        _native.check(rc, 'gptq_stripe_matvec_perm_out_f16')
        return True
    rc = _native.lib().gptq_stripe_matvec_f16(x.data_ptr(), x.stride(0) if M > 1 else K, st.data_ptr(), st.numel(), _native.ptr(bias),
                                              out.data_ptr(), out.stride(0) if M > 1 else N, M, K, N, bits, groupsize, nsets,
                                              _native.ptr(norm_weight), float(eps), _native.ptr(perm_u16(perm)), _native.stream_ptr(x.device))
    if rc == -6 and not strict:
        return False
    _native.check(rc, 'gptq_stripe_matvec_f16')
    return True


def stripe_matmul(x, st, out, K, N, bits, groupsize, nsets=1, bias=None, strict=True):
    """out[M, N] = x[M, K] through a stripe16 image on the matrix core (gptq_stripe_matmul_f16, csrc/stripe_mm.inc): 16-row MFMA tiles up
    to 128 rows, the 2-D tiled fused-dequantise GEMM above (up to gptq_set_stripe_gemm_max_rows, default 1024); exact q - z, fp32 group
    scales.  strict=False: False
    instead of raising on GPTQ_E_VARIANT (the caller takes another kernel family)."""
    M = x.shape[0]
    ws = _native.mm_workspace(x.device)
    rc = _native.lib().gptq_stripe_matmul_f16(x.data_ptr(), x.stride(0) if M > 1 else K, st.data_ptr(), st.numel(), _native.ptr(bias),
                                              out.data_ptr(), out.stride(0) if M > 1 else N, M, K, N, bits, groupsize, nsets,
                                              ws.data_ptr(), ws.numel(), _native.stream_ptr(x.device))
    if rc == -6 and not strict:
        return False
    _native.check(rc, 'gptq_stripe_matmul_f16')
    return True


STRIPE_MAX_M = 16  # rows of x one stripe16 decode launch serves (four per MFMA row group; 8 / 16 rows only while they fit in LDS)



def _as_rows(t):
    """2-D fp16 view whose last dim is contiguous and whose rows are 16-byte aligned."""
    if t.dtype != torch.float16:
        t = t.half()
    if t.stride(-1) != 1 or (t.shape[0] > 1 and t.stride(0) % 8 != 0) or t.data_ptr() % 16 != 0:
        t = t.contiguous()
        if t.data_ptr() % 16 != 0:
            t = t.clone()
    return t


def _int32c(t):
    return t if (t.dtype == torch.int32 and t.is_contiguous()) else t.to(torch.int32).contiguous()


def _prep_weight(input, qweight, scales, qzeros, g_idx, bits):
    _native.require_device(input, 'matmul248')
    if bits not in SUPPORTED_BITS:
        raise NotImplementedError('Only 2,3,4,8 bits are supported.')
    K = qweight.shape[0] * 32 // bits
    N = qweight.shape[1]
    G = scales.shape[0]
    groupsize = _infer_groupsize(K, G)
    qweight, qzeros = _int32c(qweight), _int32c(qzeros)
    scales = scales if (scales.dtype == torch.float16 and scales.is_contiguous()) else scales.half().contiguous()
    gi = None
    if g_idx is not None and not g_idx_is_trivial(g_idx, K, groupsize):
        gi = _int32c(g_idx[:K])
    return K, N, groupsize, qweight, scales, qzeros, gi


# M regimes (DESIGN.md "dispatch"): the table itself lives in the C library since round 3 (gptq_layer_forward, csrc/capi.hip) --
# decode matvec / row groups / 16-row MFMA tiles on the stripe16 image up to 128 rows, above that the layer is dequantised once per
# call and multiplied by the hand-written tile GEMM of csrc/gemm8.hip (hipBLASLt below one full round of its tiles).  What stays
# here are the knobs of tests and A/B runs, all of them C-side switches:
#   GPTQ_PREFILL = auto (default) | library (hipBLASLt for every dense product) | own (the tile GEMM wherever it can run)
#   GPTQ_STRIPE = 0 (no derived copies: checkpoint-layout kernels only), GPTQ_ACT_ORDER_SORT = 0 (generic g_idx kernels)
STREAM_MAX_M = 64
PREFILL_ROUTE = _os.environ.get('GPTQ_PREFILL', 'auto')
if PREFILL_ROUTE == 'fused':      # round-2 spelling
    PREFILL_ROUTE = 'own'
if PREFILL_ROUTE not in ('auto', 'library', 'own'):
    raise RuntimeError("GPTQ_PREFILL must be 'auto', 'library' or 'own', got %r" % PREFILL_ROUTE)
_ROUTE_CODE = {'library': 0, 'auto': 1, 'own': 2}
_route_applied = None


def _apply_prefill_route():
    """push GPTQ_PREFILL (or a test's monkeypatched PREFILL_ROUTE) into the library's switch"""
    global _route_applied
    if _route_applied != PREFILL_ROUTE:
        _native.lib().gptq_set_prefill_route(_ROUTE_CODE[PREFILL_ROUTE])
        _route_applied = PREFILL_ROUTE


def dequantize(qweight, scales, qzeros, g_idx, bits, groupsize=None, out=None):
    """dense fp16 [K, N] weight with the reference's own rounding (fp16(q - z) * fp16 scale -> fp16,
    quant_linear.py:128): bit-identical to what the prefill tile GEMM and the generic GEMV multiply with.  The
    decode kernels (stripe16 / rowwave / stream) keep (q - z) * s in fp32 instead, i.e. they are slightly MORE
    exact than this matrix; all paths sit inside the 1e-3 budget of the parity tests.  ``out``: a [K, N] fp16 view with
    unit column stride (e.g. one half of a [K, 2N] gate | up matrix)."""
    K, N = qweight.shape[0] * 32 // bits, qweight.shape[1]
    groupsize = _infer_groupsize(K, scales.shape[0]) if groupsize is None else groupsize
    gi = None
    if g_idx is not None and not g_idx_is_trivial(g_idx, K, groupsize):
        gi = _int32c(g_idx[:K])
    with torch.cuda.device(qweight.device):
        W = torch.empty((K, N), dtype=torch.float16, device=qweight.device) if out is None else out
        if W.shape != (K, N) or W.dtype != torch.float16 or W.stride(1) != 1 or W.device != qweight.device:
            raise RuntimeError('dequantize: out must be a [%d, %d] fp16 view with unit column stride on %s' % (K, N, qweight.device))
        rc = _native.lib().gptq_dequant_ld_f16(_int32c(qweight).data_ptr(), scales.data_ptr(), _int32c(qzeros).data_ptr(), _native.ptr(gi),
                                               W.data_ptr(), W.stride(0), K, N, bits, groupsize, _native.stream_ptr(qweight.device))
    _native.check(rc, 'gptq_dequant_ld_f16')
    return W


def _prefill_operand(x):
    return x if (x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0) else x.contiguous()


_library_warned = False


def _library_refused(rc, what):
    """GPTQ_E_LIBRARY (-8): hipBLASLt could not be loaded or refused the product -> say so once and let the caller go on to the
    library-free kernels of the C ABI (slower at these sizes, same results); any other code raises."""
    global _library_warned
    if rc != -8:
        _native.check(rc, what)
        return False
    if not _library_warned:
        _library_warned = True
        import warnings
        warnings.warn('%s: %s -- falling back to the fused kernels of the C ABI' % (what, _native.lib().gptq_strerror(rc).decode()))
    return True


def silu_mul(gate, up, out=None):
    """out = fp16(silu(gate) * up) in fp32 math (reference fused_mlp.py:160-165) for two [M, N] fp16 matrices with unit column
    stride (row strides free: the halves of one [M, 2N] product)."""
    _native.require_device(gate, 'silu_mul')
    M, N = gate.shape
    out = torch.empty((M, N), dtype=torch.float16, device=gate.device) if out is None else out
    for t in (gate, up, out):
        if t.shape != (M, N) or t.dtype != torch.float16 or t.stride(1) != 1:
            raise RuntimeError('silu_mul: [M, N] fp16 matrices with unit column stride')
    with torch.cuda.device(gate.device):
        rc = _native.lib().gptq_silu_mul_f16(gate.data_ptr(), gate.stride(0), up.data_ptr(), up.stride(0), out.data_ptr(), out.stride(0), M, N,
                                             _native.stream_ptr(gate.device))
    _native.check(rc, 'gptq_silu_mul_f16')
    return out


_FAMILIES = {'abi': 'gptq_matmul248_f16', 'gemv': 'gptq_gemv_f16', 'skinny': 'gptq_skinny_f16'}


def matmul248(input, qweight, scales, qzeros, g_idx, bits, maxq, bias=None, family=None):
    """``input [M,K] fp16 -> [M,N] fp16`` on the current stream of ``input.device``
    (reference matmul248, quant/quant_linear.py:263-269; ``bias`` is an extension that fuses
    the add of QuantLinear.forward, :376).  The call goes to the layer's prepared handle (quant/layer.py ->
    gptq_layer_forward: the M -> kernel table is in the C library).  ``family`` (tests / benchmarks only) names ONE kernel
    family of the C ABI instead: 'abi' gptq_matmul248_f16, 'gemv', 'skinny', 'stripe' (decode kernel, row groups up to 16
    rows), 'stripe_mm' (16-row MFMA tiles)."""
    _native.require_device(input, 'matmul248')
    if bits not in SUPPORTED_BITS:
        raise NotImplementedError('Only 2,3,4,8 bits are supported.')
    if family is None:
        K, N = qweight.shape[0] * 32 // bits, qweight.shape[1]
        x = _as_rows(input)
        if x.shape[1] != K:
            raise RuntimeError('matmul248: input has %d features, weight expects %d' % (x.shape[1], K))
        M = x.shape[0]
        _apply_prefill_route()
        with _native.on_device(x.device):
            out = torch.empty((M, N), device=x.device, dtype=torch.float16)
            if M == 0:
                return out
            pl = prepared(((qweight, scales, qzeros, g_idx),), bias, bits, _infer_groupsize(K, scales.shape[0]), K, N, sort=ACT_ORDER_SORT)
            return pl.forward(x, out)
    K, N, groupsize, qweight, scales, qzeros, gi = _prep_weight(input, qweight, scales, qzeros, g_idx, bits)
    x = _as_rows(input)
    if x.shape[1] != K:
        raise RuntimeError('matmul248: input has %d features, weight expects %d' % (x.shape[1], K))
    M = x.shape[0]
    with torch.cuda.device(x.device):
        out = torch.empty((M, N), device=x.device, dtype=torch.float16)
        if M == 0:
            return out
        ws = _native.workspace(x.device)
        if family == 'stripe':
            srt = act_order_sorted(qweight, gi, K, groupsize, bits) if gi is not None else None
            if (M == 1 or (gi is None and M <= STRIPE_MAX_M)) and (gi is None or srt is not None):
                st = stripe_copy(srt[0] if srt is not None else qweight, scales, qzeros, bits, groupsize)
                if st is not None and stripe_matvec(x, st, out, K, N, bits, groupsize, bias=bias, perm=srt[1] if srt is not None else None):
                    return out
            raise RuntimeError('matmul248: the stripe16 path does not serve this shape (M <= 16, K a multiple of the row block ...)')
        if family == 'stripe_mm':
            st = stripe_copy(qweight, scales, qzeros, bits, groupsize) if (gi is None and M <= 1024) else None
            if st is not None and stripe_matmul(x, st, out, K, N, bits, groupsize, bias=bias, strict=False):
                return out
            raise RuntimeError('matmul248: the stripe16 MFMA kernels do not serve this shape (M <= 1024, a stripe16 image of the layer)')
        rc = getattr(_native.lib(), _FAMILIES[family])(
            x.data_ptr(), x.stride(0) if M > 1 else K, qweight.data_ptr(), scales.data_ptr(), qzeros.data_ptr(),
            _native.ptr(gi), _native.ptr(bias), out.data_ptr(), N, M, K, N, bits, groupsize,
            ws.data_ptr(), ws.numel(), _native.stream_ptr(x.device))
    _native.check(rc, _FAMILIES[family])
    return out


# The route wins from one row on (31 vs 151 us at M = 1, 0.11 vs 11.6 ms at M = 4096 on 4096^2), but below 16 rows nobody trains, and a
# GEMM with one to a few columns is not a shape the library's kernels see often: one test run that pushed M = 1 ... 3 through it ended in
# an abort that five other runs of the same tests did not reproduce.  The threshold keeps the library on ordinary shapes.
TRANSPOSE_LIBRARY_MIN_M = 16


def transpose_matmul248(input, qweight, scales, qzeros, g_idx, bits, maxq, family=None):
    """``input [M,N] fp16 -> [M,K] fp16`` = input . deq(B)^T (reference transpose_matmul248,
    quant/quant_linear.py:272-279).  From TRANSPOSE_LIBRARY_MIN_M rows on the product takes the prefill route
    (gptq_prefill_transpose_matmul248_f16: dequantise once per call, then the tile GEMM of csrc/gemm8.hip with the roles of K and N
    exchanged -- 1.2-1.3 PF at M >= 4096 -- or hipBLASLt below one round of its tiles); fewer rows or family='abi' keep the
    LDS-tiled kernel of csrc/transpose.hip."""
    K, N, groupsize, qweight, scales, qzeros, gi = _prep_weight(input, qweight, scales, qzeros, g_idx, bits)
    dy = _as_rows(input)
    if dy.shape[1] != N:
        raise RuntimeError('transpose_matmul248: input has %d features, weight has %d output columns' % (dy.shape[1], N))
    M = dy.shape[0]
    with torch.cuda.device(dy.device):
        out = torch.empty((M, K), device=dy.device, dtype=torch.float16)
        if M == 0:
            return out
        if family is None and M >= TRANSPOSE_LIBRARY_MIN_M and K >= 256 and N >= 256:    # ordinary shapes only
            _apply_prefill_route()
            lib = _native.lib()
            dy = _prefill_operand(dy)
            ws = torch.empty(lib.gptq_prefill_workspace_bytes(M, K, N, 1), dtype=torch.uint8, device=dy.device)
            rc = lib.gptq_prefill_transpose_matmul248_f16(dy.data_ptr(), dy.stride(0), qweight.data_ptr(), scales.data_ptr(), qzeros.data_ptr(),
                                                          _native.ptr(gi), out.data_ptr(), K, M, K, N, bits, groupsize, ws.data_ptr(), ws.numel(),
                                                          _native.stream_ptr(dy.device))
            if not _library_refused(rc, 'gptq_prefill_transpose_matmul248_f16'):
                return out
        rc = _native.lib().gptq_transpose_matmul248_f16(
            dy.data_ptr(), dy.stride(0) if M > 1 else N, qweight.data_ptr(), scales.data_ptr(), qzeros.data_ptr(),
            _native.ptr(gi), out.data_ptr(), K, M, K, N, bits, groupsize, _native.stream_ptr(dy.device))
    _native.check(rc, 'gptq_transpose_matmul248_f16')
    return out


class QuantLinearFunction(torch.autograd.Function):
    """autograd wrapper (reference quant/quant_linear.py:282-301): forward = matmul248,
    backward = transpose_matmul248 for grad_input only (the packed weight is frozen)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float16)
    def forward(ctx, input, qweight, scales, qzeros, g_idx, bits, maxq):
        output = matmul248(input, qweight, scales, qzeros, g_idx, bits, maxq)
        ctx.save_for_backward(qweight, scales, qzeros, g_idx)
        ctx.bits, ctx.maxq = bits, maxq
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_output):
        qweight, scales, qzeros, g_idx = ctx.saved_tensors
        grad_input = None
        if ctx.needs_input_grad[0]:
            grad_input = transpose_matmul248(grad_output, qweight, scales, qzeros, g_idx, ctx.bits, ctx.maxq)
        return grad_input, None, None, None, None, None, None


def _pack_fields(fields_u32, bits, axis_rows=True):
    """fields [R*f, C] uint32 (unmasked, as the reference ORs them) -> words [R*bits/…, C]."""
    a = fields_u32 if axis_rows else fields_u32.T
    total, C = a.shape
    if bits == 3:
        blk = a.reshape(total // 32, 32, C).astype(np.uint64) & np.uint64(7)
        out = np.zeros((total // 32, 3, C), dtype=np.uint64)
        for j in range(32):
            bit = 3 * j
            w, o = bit // 32, bit % 32
            out[:, w] |= (blk[:, j] << np.uint64(o)) & np.uint64(0xFFFFFFFF)
            if o + 3 > 32:
                out[:, w + 1] |= blk[:, j] >> np.uint64(32 - o)
        words = out.astype(np.uint32).reshape(total // 32 * 3, C)
    else:
        f = 32 // bits
        words = np.zeros((total // f, C), dtype=np.uint32)
        grouped = a.reshape(total // f, f, C)
        for j in range(f):
            words |= grouped[:, j] << np.uint32(bits * j)
    words = words if axis_rows else np.ascontiguousarray(words.T)
    return words.view(np.int32)


def _moves_or_casts(fn, device):
    """does this Module._apply callback change where / as what a module's fp16 buffers live?  HF and accelerate issue no-op
    ``model.to(same_device)`` / ``.half()`` calls routinely; a released module must not rebuild its checkpoint buffers (and, on the next
    decode step, its image: transiently three copies of the packed weights) for those."""
    try:
        probe = torch.empty(0, dtype=torch.float16, device=device)
        out = fn(probe)
        return out.device != probe.device or out.dtype != probe.dtype
    except Exception:
        return True


class QuantLinear(nn.Module):

    def __init__(self, bits, groupsize, infeatures, outfeatures, bias):
        super().__init__()
        if bits not in SUPPORTED_BITS:
            raise NotImplementedError('Only 2,3,4,8 bits are supported.')
        self.infeatures = infeatures
        self.outfeatures = outfeatures
        self.bits = bits
        self.maxq = 2**self.bits - 1
        self.groupsize = groupsize if groupsize != -1 else infeatures
        groups = math.ceil(infeatures / self.groupsize)

        self.register_buffer('qweight', torch.zeros((infeatures // 32 * self.bits, outfeatures), dtype=torch.int32))
        self.register_buffer('qzeros', torch.zeros((groups, outfeatures // 32 * self.bits), dtype=torch.int32))
        self.register_buffer('scales', torch.zeros((groups, outfeatures), dtype=torch.float16))
        self.register_buffer('g_idx', (torch.arange(infeatures, dtype=torch.int64) // self.groupsize).to(torch.int32))
        if bias:
            self.register_buffer('bias', torch.zeros((outfeatures), dtype=torch.float16))
        else:
            self.bias = None
        self._released = None     # memory mode: the PreparedLayer that now holds the ONLY copy of the packed weights (release_checkpoint)

    # ------------------------------------------------------------------------------ memory mode
    def release_checkpoint(self):
        """Keep ONE copy of the packed weights instead of two (the reference's footprint, README.md:23-29): the stripe16 image the
        decode kernels read is a bijection of qweight / scales / qzeros, so those buffers are freed (they become empty
        placeholders) and forward() goes straight to the prepared handle.  ``state_dict()`` still returns the original tensors
        (reproduced bit-exactly from the image), ``load_state_dict`` restores the buffers first.  Inference only.  Returns False
        -- and changes nothing -- for layers that need the checkpoint layout (irregular act-order g_idx, K not served by the image).  Round 4: regular act-order layers (the image carries the permutation and its inverse; g_idx stays) and 3-bit layers, act-order
        or not, release too."""
        if self._released is not None:
            return True
        if not self.qweight.is_cuda:
            return False
        pl = prepared(((self.qweight, self.scales, self.qzeros, self.g_idx),), self.bias, self.bits, self.groupsize, self.infeatures, self.outfeatures,
                      sort=ACT_ORDER_SORT)     # the SAME registry entry forward() uses (a different `sort` would build a second image)
        if not pl.release():
            return False
        self._released = pl
        dev = self.qweight.device
        self.qweight = torch.empty((0, self.outfeatures), dtype=torch.int32, device=dev)
        self.qzeros = torch.empty((0, self.outfeatures // 32 * self.bits), dtype=torch.int32, device=dev)
        self.scales = torch.empty((0, self.outfeatures), dtype=torch.float16, device=dev)
        return True

    def restore_checkpoint(self):
        """undo release_checkpoint: the buffers come back out of the image (bit-exact)"""
        if self._released is not None:
            from .layer import RESTORE_EPOCH
            pl, self._released = self._released, None
            self.qweight, self.scales, self.qzeros = pl.unpack(0)
            RESTORE_EPOCH[0] += 1

    # a released module holds a device image and a C handle that neither move nor pickle: anything that moves / casts / copies the
    # module (.to(), .cpu(), .half(), copy.deepcopy, pickling) first brings the checkpoint buffers back
    def _apply(self, fn, *args, **kwargs):
        if self._released is not None and _moves_or_casts(fn, self.qweight.device):
            self.restore_checkpoint()
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode=True):
        if mode:
            self.restore_checkpoint()      # training needs the checkpoint layout (backward reads qweight): leave memory mode
        return super().train(mode)

    def __getstate__(self):
        self.restore_checkpoint()
        return super().__getstate__() if hasattr(super(), '__getstate__') else self.__dict__

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self._released is not None:        # the checkpoint format does not change with the memory mode
            qw, sc, qz = self._released.unpack(0)
            destination[prefix + 'qweight'], destination[prefix + 'scales'], destination[prefix + 'qzeros'] = qw, sc, qz

    def _load_from_state_dict(self, *args, **kwargs):
        self.restore_checkpoint()
        return super()._load_from_state_dict(*args, **kwargs)

    # ------------------------------------------------------------------------------------ pack
    def pack(self, linear, scales, zeros, g_idx=None):
        """Pack a grid-valued ``nn.Linear`` (``scales``/``zeros`` are ``[N, G]`` as produced by
        gptq.py:226-228).  Same arithmetic and bit layout as the reference packer
        (quant/quant_linear.py:325-371) -- incl. the division by the fp16-rounded scale, the
        unmasked OR and ``zeros - 1`` -- vectorised on the host, or on the GPU
        (``gptq_pack_f32``) when the layer lives there."""
        self._released = None      # a repacked module starts over on fresh buffers
        self.g_idx = g_idx.clone() if g_idx is not None else self.g_idx
        W = linear.weight.data
        if linear.bias is not None:
            self.bias = linear.bias.clone().half()
        if W.is_cuda:
            return self._pack_gpu(W, scales, zeros)

        K, N, bits = self.infeatures, self.outfeatures, self.bits
        g = self.g_idx.to(torch.int64).cpu()
        st = scales.t().contiguous().float().cpu()              # [G, N]
        zt = zeros.t().contiguous().float().cpu()
        sz = zt * st
        s16 = st.clone().half()
        self.scales = s16
        v = (W.float().cpu().t() + sz[g]) / s16[g]               # fp32 / fp16 -> fp32, like the reference
        intweight = torch.round(v).to(torch.int32).numpy().astype(np.uint32)   # [K, N]
        self.qweight = torch.from_numpy(_pack_fields(intweight, bits, axis_rows=True).copy())
        zi = (zt - 1).numpy().astype(np.int64).astype(np.uint32)               # -1.0 -> 0xFFFFFFFF
        self.qzeros = torch.from_numpy(_pack_fields(zi, bits, axis_rows=False).copy())

    def _pack_gpu(self, W, scales, zeros):
        K, N, bits = self.infeatures, self.outfeatures, self.bits
        dev = W.device
        G = math.ceil(K / self.groupsize)
        Wf = W.float().contiguous()
        sc = scales.to(dev).float().reshape(N, G).contiguous()
        zr = zeros.to(dev).float().reshape(N, G).contiguous()
        gi = _int32c(self.g_idx.to(dev))
        with torch.cuda.device(dev):
            qweight = torch.empty((K // 32 * bits, N), dtype=torch.int32, device=dev)
            qzeros = torch.empty((G, N // 32 * bits), dtype=torch.int32, device=dev)
            s16 = torch.empty((G, N), dtype=torch.float16, device=dev)
            rc = _native.lib().gptq_pack_f32(Wf.data_ptr(), sc.data_ptr(), zr.data_ptr(), gi.data_ptr(), K, N, bits,
                                             self.groupsize, qweight.data_ptr(), qzeros.data_ptr(), s16.data_ptr(),
                                             _native.stream_ptr(dev))
        _native.check(rc, 'gptq_pack_f32')
        self.qweight, self.qzeros, self.scales = qweight, qzeros, s16
        self.g_idx = gi

    # --------------------------------------------------------------------------------- forward
    def forward(self, x):
        out_shape = x.shape[:-1] + (self.outfeatures, )
        x2 = x.reshape(-1, x.shape[-1])
        if self._released is not None and torch.is_grad_enabled() and x2.requires_grad:
            # memory mode is an inference mode, and it is ON by default once the decode engine has run: a caller that goes on to a backward
            # pass (eval -> generate -> train, the reference's autograd path QuantLinearFunction / transpose_matmul248, quant_linear.py:282-301)
            # gets the checkpoint buffers back here instead of an error
            self.restore_checkpoint()
        if self._released is not None:
            _native.require_device(x2, 'QuantLinear.forward')
            xr = _as_rows(x2)
            _apply_prefill_route()
            with _native.on_device(xr.device):
                out = torch.empty((xr.shape[0], self.outfeatures), device=xr.device, dtype=torch.float16)
                if xr.shape[0]:
                    self._released.forward(xr, out)
            return out.reshape(out_shape)
        if torch.is_grad_enabled() and x2.requires_grad:
            out = QuantLinearFunction.apply(x2, self.qweight, self.scales, self.qzeros, self.g_idx, self.bits, self.maxq)
            out = out + self.bias if self.bias is not None else out
        else:
            # inference: bias add fused into the kernel epilogue (same rounding order as the
            # reference's separate add: fp16(fp16(acc) + bias))
            out = matmul248(x2, self.qweight, self.scales, self.qzeros, self.g_idx, self.bits, self.maxq, bias=self.bias)
        return out.reshape(out_shape)


def make_quant_linear(module, names, bits, groupsize, name=''):
    """Replace every ``nn.Linear`` whose qualified name is in ``names`` by a ``QuantLinear``
    (reference quant/quant_linear.py:380-390)."""
    if isinstance(module, QuantLinear):
        return
    for attr in dir(module):
        tmp = getattr(module, attr)
        name1 = name + '.' + attr if name != '' else attr
        if name1 in names:
            delattr(module, attr)
            setattr(module, attr, QuantLinear(bits, groupsize, tmp.in_features, tmp.out_features, tmp.bias is not None))
    for name1, child in module.named_children():
        make_quant_linear(child, names, bits, groupsize, name + '.' + name1 if name != '' else name1)


def autotune_warmup_linear(model, transpose=False):
    """Same call surface as the reference warm-up (quant/quant_linear.py:393-423).  There is no
    autotuner to train: kernels are picked from a static shape table.  The warm-up still walks
    every unique (K, N) at M = 1 .. 2048 so that the g_idx triviality checks, the workspace and
    the kernels' code objects are resident before the first timed token."""
    from tqdm import tqdm

    kn_values = {}
    for _, m in model.named_modules():
        if not isinstance(m, QuantLinear):
            continue
        k, n = m.infeatures, m.outfeatures
        if (k, n) not in kn_values:
            kn_values[(k, n)] = (m.qweight.cuda(), m.scales.cuda(), m.qzeros.cuda(), m.g_idx.cuda(), m.bits, m.maxq)

    print(f'Found {len(kn_values)} unique KN Linear values.')
    print('Warming up autotune cache ...')
    with torch.no_grad():
        for m in tqdm(range(0, 12)):
            m = 2**m  # [1, 2048]
            for (k, n), (qweight, scales, qzeros, g_idx, bits, maxq) in kn_values.items():
                a = torch.randn(m, k, dtype=torch.float16, device='cuda')
                matmul248(a, qweight, scales, qzeros, g_idx, bits, maxq)
                if transpose:
                    a = torch.randn(m, n, dtype=torch.float16, device='cuda')
                    transpose_matmul248(a, qweight, scales, qzeros, g_idx, bits, maxq)
    del kn_values


# old-cuda-branch spellings used by BASELINE.json's north_star
make_quant = make_quant_linear
autotune_warmup = autotune_warmup_linear
