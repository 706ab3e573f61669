"""CPU oracle for the GPTQ-for-LLaMa QuantLinear hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package (``gptq-for-llama_amd/``) never does.

Two layers:

* ``lib`` -- ctypes binding of ``libgptq_oracle.so`` (``gptq_oracle.c``), the C
  restatement of the reference kernels (file:line citations live in the C header).
* ``np_*`` -- a second, independent numpy restatement of the bit layout
  (reference ``quant/quant_linear.py:103-128`` for unpack, ``:325-371`` for pack) used to
  cross-check the C code.

Parity pin: ``tests/golden/*.npz`` were produced by the reference's own Triton kernels run
under ``TRITON_INTERPRET=1`` and by its own ``QuantLinear.pack`` (see
``tests/golden/gen_golden.py``); ``tests/test_oracle_golden.py`` holds this oracle to them.
bits == 3 is an extension the reference rejects (``quant_linear.py:308-309``): parity unpinned.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libgptq_oracle.so')


def build(force=False):
    """(Re)build libgptq_oracle.so with the committed Makefile."""
    src = os.path.join(_HERE, 'gptq_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libgptq_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return _SO


def _load():
    build()
    try:
        return ctypes.CDLL(_SO)
    except OSError:
        build(force=True)
        return ctypes.CDLL(_SO)


lib = _load()

_i32p = ctypes.POINTER(ctypes.c_int32)
_u16p = ctypes.POINTER(ctypes.c_uint16)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(np.asarray(a), dtype=dtype)


def _h(a):
    """fp16 array -> contiguous uint16 view."""
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a))
    if a.dtype != np.float16:
        a = a.astype(np.float16)
    return a.view(np.uint16)


def num_threads():
    return int(lib.oracle_num_threads())


def n_groups(K, groupsize):
    gs = K if groupsize == -1 else groupsize
    return -(-K // gs)


def trivial_g_idx(K, groupsize):
    gs = K if groupsize == -1 else groupsize
    return (np.arange(K) // gs).astype(np.int32)


def dequant(qweight, qzeros, scales, g_idx, bits, faithful=True):
    """W[K,N] float32. faithful=True rounds like the reference kernel (fp16 weight)."""
    qweight = _c(qweight, np.int32)
    qzeros = _c(qzeros, np.int32)
    s = _h(scales)
    g = _c(g_idx, np.int32)
    K = g.shape[0]
    N = qweight.shape[1]
    G = s.shape[0]
    W = np.empty((K, N), dtype=np.float32)
    rc = lib.oracle_dequant(_p(qweight, _i32p), _p(qzeros, _i32p), _p(s, _u16p), _p(g, _i32p),
                            K, N, G, bits, int(bool(faithful)), _p(W, _f32p))
    if rc:
        raise NotImplementedError('oracle: unsupported bits %r' % (bits,))
    return W


def matmul248(x, qweight, scales, qzeros, g_idx, bits, bias=None):
    """Reference-faithful forward; argument order follows reference matmul248()
    (quant/quant_linear.py:263).  x [M,K] fp16 -> y [M,N] fp16."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float16))
    M, K = x.shape
    qweight = _c(qweight, np.int32)
    qzeros = _c(qzeros, np.int32)
    s = _h(scales)
    g = _c(g_idx, np.int32)[:K]
    N = qweight.shape[1]
    G = s.shape[0]
    y = np.empty((M, N), dtype=np.float16)
    b = _h(bias)
    rc = lib.oracle_matmul248(_p(x.view(np.uint16), _u16p), ctypes.c_int64(K), _p(qweight, _i32p),
                              _p(qzeros, _i32p), _p(s, _u16p), _p(g, _i32p), _p(b, _u16p),
                              _p(y.view(np.uint16), _u16p), ctypes.c_int64(N), M, K, N, G, bits)
    if rc:
        raise NotImplementedError('oracle: unsupported bits %r' % (bits,))
    return y


def matmul248_exact(x, qweight, scales, qzeros, g_idx, bits):
    """float64 result with the weight never rounded to fp16."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float16))
    M, K = x.shape
    qweight = _c(qweight, np.int32)
    qzeros = _c(qzeros, np.int32)
    s = _h(scales)
    g = _c(g_idx, np.int32)[:K]
    N = qweight.shape[1]
    G = s.shape[0]
    y = np.empty((M, N), dtype=np.float64)
    rc = lib.oracle_matmul248_exact(_p(x.view(np.uint16), _u16p), ctypes.c_int64(K),
                                    _p(qweight, _i32p), _p(qzeros, _i32p), _p(s, _u16p),
                                    _p(g, _i32p), _p(y, _f64p), M, K, N, G, bits)
    if rc:
        raise NotImplementedError('oracle: unsupported bits %r' % (bits,))
    return y


def transpose_matmul248(dy, qweight, scales, qzeros, g_idx, bits):
    """dX[M,K] = dY[M,N] . deq(B)^T (reference transpose_matmul248, quant_linear.py:272)."""
    dy = np.ascontiguousarray(np.asarray(dy, dtype=np.float16))
    M, N = dy.shape
    qweight = _c(qweight, np.int32)
    qzeros = _c(qzeros, np.int32)
    s = _h(scales)
    K = qweight.shape[0] * 32 // bits
    g = _c(g_idx, np.int32)[:K]
    G = s.shape[0]
    dx = np.empty((M, K), dtype=np.float16)
    rc = lib.oracle_transpose_matmul248(_p(dy.view(np.uint16), _u16p), ctypes.c_int64(N),
                                        _p(qweight, _i32p), _p(qzeros, _i32p), _p(s, _u16p),
                                        _p(g, _i32p), _p(dx.view(np.uint16), _u16p),
                                        ctypes.c_int64(K), M, K, N, G, bits)
    if rc:
        raise NotImplementedError('oracle: unsupported bits %r' % (bits,))
    return dx


def fused_mlp(x, gate, up, bits):
    """silu(x.Wg) * (x.Wu); gate/up = (qweight, scales, qzeros, g_idx) tuples
    (reference fusedmatmul_248_kernel, quant/fused_mlp.py:84-168)."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float16))
    M, K = x.shape
    a = [_c(gate[0], np.int32), _h(gate[1]), _c(gate[2], np.int32), _c(gate[3], np.int32)[:K]]
    b = [_c(up[0], np.int32), _h(up[1]), _c(up[2], np.int32), _c(up[3], np.int32)[:K]]
    N = a[0].shape[1]
    G = a[1].shape[0]
    c = np.empty((M, N), dtype=np.float16)
    rc = lib.oracle_fused_mlp(_p(x.view(np.uint16), _u16p), ctypes.c_int64(K),
                              _p(a[0], _i32p), _p(a[2], _i32p), _p(a[1], _u16p), _p(a[3], _i32p),
                              _p(b[0], _i32p), _p(b[2], _i32p), _p(b[1], _u16p), _p(b[3], _i32p),
                              _p(c.view(np.uint16), _u16p), ctypes.c_int64(N), M, K, N, G, bits)
    if rc:
        raise NotImplementedError('oracle: unsupported bits %r' % (bits,))
    return c


def fused_mlp_exact(x, gate, up, bits):
    """float64 silu(x.Wg) * (x.Wu) with weights never rounded to fp16 (built on matmul248_exact): the yardstick for "at least as
    exact as the reference" checks of the fused-MLP kernels (tests/util.py: assert_not_worse_than_reference)."""
    a = matmul248_exact(x, gate[0], gate[1], gate[2], gate[3], bits)
    b = matmul248_exact(x, up[0], up[1], up[2], up[3], bits)
    return a / (1.0 + np.exp(-a)) * b


def rmsnorm(x, weight, eps):
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float16))
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    M, N = x2.shape
    w = _h(weight)
    y = np.empty((M, N), dtype=np.float16)
    lib.oracle_rmsnorm(_p(x2.view(np.uint16), _u16p), ctypes.c_int64(N), _p(w, _u16p),
                       _p(y.view(np.uint16), _u16p), ctypes.c_int64(N), M, N, ctypes.c_float(eps))
    return y.reshape(shape)


def rope_(qk, position_ids, base=10000.0):
    """In-place RoPE on qk [bsz, seq, 2, heads, head_dim] fp16 view whose (bsz,seq) rows are
    `row_stride` apart (reference triton_rotate_half_, quant/fused_attn.py:61-93).
    qk must be a numpy fp16 array whose last three dims are contiguous."""
    bsz, seq, two, heads, hd = qk.shape
    assert two == 2
    assert qk.strides[4] == 2 and qk.strides[3] == 2 * hd and qk.strides[2] == 2 * hd * heads
    assert qk.strides[0] == qk.strides[1] * seq
    ld = qk.strides[1] // 2
    pos = np.ascontiguousarray(np.asarray(position_ids, dtype=np.int64).reshape(bsz * seq))
    ptr = ctypes.cast(qk.ctypes.data, _u16p)
    lib.oracle_rope(ptr, ctypes.c_int64(ld), _p(pos, _i64p), bsz * seq, 2 * heads, hd,
                    ctypes.c_float(base))
    return qk


def pack(weight, scales, zeros, g_idx, bits, groupsize):
    """C restatement of QuantLinear.pack.  weight [N,K] fp32, scales/zeros [N,G] fp32."""
    weight = _c(weight, np.float32)
    N, K = weight.shape
    G = n_groups(K, groupsize)
    scales = _c(np.asarray(scales).reshape(N, G), np.float32)
    zeros = _c(np.asarray(zeros).reshape(N, G), np.float32)
    g = _c(g_idx if g_idx is not None else trivial_g_idx(K, groupsize), np.int32)
    qweight = np.empty((K // 32 * bits, N), dtype=np.int32)
    qzeros = np.empty((G, N // 32 * bits), dtype=np.int32)
    s16 = np.empty((G, N), dtype=np.uint16)
    rc = lib.oracle_pack(_p(weight, _f32p), _p(scales, _f32p), _p(zeros, _f32p), _p(g, _i32p),
                         K, N, G, bits, _p(qweight, _i32p), _p(qzeros, _i32p), _p(s16, _u16p))
    if rc:
        raise NotImplementedError('oracle: unsupported bits %r' % (bits,))
    return qweight, qzeros, s16.view(np.float16), g


# --------------------------------------------------------------------------------------
# independent numpy restatement of the bit layout (cross-check for the C code)
# --------------------------------------------------------------------------------------

def np_unpack_rows(qweight, bits):
    """[K/32*bits, N] int32 -> integer fields [K, N] (unpack along rows, k axis)."""
    qw = np.asarray(qweight).astype(np.int32).view(np.uint32)
    R, N = qw.shape
    if bits == 3:
        blk = qw.reshape(R // 3, 3, N).astype(np.uint64)
        stream_lo = blk[:, 0] | (blk[:, 1] << np.uint64(32))       # bits 0..63
        stream_hi = blk[:, 2]                                       # bits 64..95
        out = np.empty((R // 3, 32, N), dtype=np.int32)
        for j in range(32):
            b = 3 * j
            if b + 3 <= 64:
                v = (stream_lo >> np.uint64(b)) & np.uint64(7)
            elif b >= 64:
                v = (stream_hi >> np.uint64(b - 64)) & np.uint64(7)
            else:  # straddles bit 64 (j == 21: bits 63..65)
                v = ((stream_lo >> np.uint64(b)) | (stream_hi << np.uint64(64 - b))) & np.uint64(7)
            out[:, j] = v.astype(np.int32)
        return out.reshape(R // 3 * 32, N)
    f = 32 // bits
    sh = (np.arange(f, dtype=np.uint32) * bits)[None, :, None]
    return ((qw[:, None, :] >> sh) & np.uint32((1 << bits) - 1)).astype(np.int32).reshape(R * f, N)


def np_unpack_cols(qzeros, bits):
    """[G, N/32*bits] int32 -> fields [G, N] (unpack along columns, n axis); raw, no +1."""
    return np_unpack_rows(np.ascontiguousarray(np.asarray(qzeros).T), bits).T


def np_dequant(qweight, qzeros, scales, g_idx, bits, faithful=True):
    q = np_unpack_rows(qweight, bits)
    z = np_unpack_cols(qzeros, bits) + 1
    K = q.shape[0]
    g = np.asarray(g_idx)[:K].astype(np.int64)
    s = np.asarray(scales, dtype=np.float16)
    d = (q - z[g])
    if faithful:
        return (d.astype(np.float16) * s[g]).astype(np.float32)
    return d.astype(np.float32) * s[g].astype(np.float32)


def np_pack_fields_rows(fields, bits):
    """Inverse of np_unpack_rows for in-range fields [K,N] -> [K/32*bits, N] int32."""
    fields = np.asarray(fields).astype(np.uint32)
    K, N = fields.shape
    if bits == 3:
        out = np.zeros((K // 32, 3, N), dtype=np.uint64)
        f3 = fields.reshape(K // 32, 32, N).astype(np.uint64) & np.uint64(7)
        for j in range(32):
            b = 3 * j
            w, o = b // 32, b % 32
            out[:, w] |= (f3[:, j] << np.uint64(o)) & np.uint64(0xFFFFFFFF)
            if o + 3 > 32:
                out[:, w + 1] |= f3[:, j] >> np.uint64(32 - o)
        return out.astype(np.uint32).view(np.int32).reshape(K // 32 * 3, N)
    f = 32 // bits
    out = np.zeros((K // f, N), dtype=np.uint32)
    ff = fields.reshape(K // f, f, N)
    for j in range(f):
        out |= ff[:, j] << np.uint32(bits * j)
    return out.view(np.int32)


def np_pack_fields_cols(fields, bits):
    return np.ascontiguousarray(np_pack_fields_rows(np.asarray(fields).T, bits).T)


# --------------------------------------------------------------------------------------
# stripe16: numpy restatement of the load-time repack of csrc/stripe.hip (bits = 4).  The layout is the
# library's own (the reference has no counterpart: its load path ends at load_state_dict,
# llama_inference.py:57-60); this restatement pins gptq_stripe_repack bit for bit and proves the layout
# is a bijection of the checkpoint buffers (stripe16_unpack is its inverse).
# --------------------------------------------------------------------------------------
def stripe_k_of_pos(bits):
    """field position p (bits bits*p ..) of a stripe word holds k = k_of_pos[p] of the packed row: even k low half, odd k high half"""
    F = 32 // bits
    return tuple(2 * p if p < F // 2 else 2 * (p - F // 2) + 1 for p in range(F))


def stripe16_repack(sets, groupsize, bits=4):
    """sets = [(qweight, scales, qzeros)] (one set, or gate and up) in the checkpoint format
    (reference quant_linear.py:316-321) -> uint8 image: R uint32 [N/16][K/(16 KPW)][NS][64][4] followed by
    tab half2 [N/16][NS][G][16] {scale, zero + 1}; KPW = 32 / bits."""
    NS = len(sets)
    if bits == 3:
        return _stripe16_repack3(sets, groupsize)
    kpw = 32 // bits
    qw0 = np.asarray(sets[0][0]).astype(np.int32).view(np.uint32)
    rows, N = qw0.shape
    K = rows * kpw
    assert K % (16 * kpw) == 0 and N % 16 == 0
    nrb, S = rows // 16, N // 16
    gs = K if groupsize in (-1, None) or groupsize >= K else groupsize
    G = K // gs
    R = np.empty((S, nrb, NS, 64, 4), dtype=np.uint32)
    tab = np.empty((S, NS, G, 16), dtype=np.uint32)
    fm = np.uint32((1 << bits) - 1)
    for si, (qw, sc, qz) in enumerate(sets):
        w = np.asarray(qw).astype(np.int32).view(np.uint32)
        o = np.zeros_like(w)
        for pos, k in enumerate(stripe_k_of_pos(bits)):
            o |= ((w >> np.uint32(bits * k)) & fm) << np.uint32(bits * pos)
        # row = rb*16 + rq*4 + j, col = s*16 + c  ->  [s][rb][lane = rq*16 + c][j]
        R[:, :, si] = o.reshape(nrb, 4, 4, S, 16).transpose(3, 0, 1, 4, 2).reshape(S, nrb, 64, 4)
        z = (np_unpack_cols(qz, bits) + 1).astype(np.float16)                # stored + 1, not re-masked
        sc16 = np.asarray(sc, dtype=np.float16)
        e = sc16.view(np.uint16).astype(np.uint32) | (z.view(np.uint16).astype(np.uint32) << np.uint32(16))   # half2 {s, z+1}
        tab[:, si] = e.reshape(G, S, 16).transpose(1, 0, 2)
    return np.concatenate([R.reshape(-1).view(np.uint8), tab.reshape(-1).view(np.uint8)])


def _stripe16_repack3(sets, groupsize):
    """3-bit image (csrc/stripe.hip stripe_repack3_kernel): R uint32 [N/16][K/128][NS][64 lanes][3]; a lane's 32 k = three words, word j =
    k 10j .. 10j+9 as five 3-bit fields per half-word (pair p = bits [3p+2:3p] of the low / high half: k 10j+2p / 10j+2p+1) and bit j of
    k 30 / k 31 in its spare bits 15 / 31; the table as for the other widths."""
    NS = len(sets)
    q0 = np_unpack_rows(sets[0][0], 3)
    K, N = q0.shape
    assert K % 128 == 0 and N % 16 == 0
    nrb, S = K // 128, N // 16
    gs = K if groupsize in (-1, None) or groupsize >= K else groupsize
    G = K // gs
    R = np.empty((S, nrb, NS, 64, 3), dtype=np.uint32)
    tab = np.empty((S, NS, G, 16), dtype=np.uint32)
    for si, (qw, sc, qz) in enumerate(sets):
        q = np_unpack_rows(qw, 3).astype(np.uint32).reshape(nrb, 4, 32, S, 16)          # [rb][rq][k in block][s][c]
        words = np.zeros((nrb, 4, 3, S, 16), dtype=np.uint32)
        for j in range(3):
            for p_ in range(5):
                words[:, :, j] |= q[:, :, 10 * j + 2 * p_] << np.uint32(3 * p_)
                words[:, :, j] |= q[:, :, 10 * j + 2 * p_ + 1] << np.uint32(16 + 3 * p_)
            words[:, :, j] |= ((q[:, :, 30] >> np.uint32(j)) & np.uint32(1)) << np.uint32(15)
            words[:, :, j] |= ((q[:, :, 31] >> np.uint32(j)) & np.uint32(1)) << np.uint32(31)
        # [rb][rq][j][s][c] -> [s][rb][lane = rq*16 + c][j]
        R[:, :, si] = words.transpose(3, 0, 1, 4, 2).reshape(S, nrb, 64, 3)
        z = (np_unpack_cols(qz, 3) + 1).astype(np.float16)
        sc16 = np.asarray(sc, dtype=np.float16)
        e = sc16.view(np.uint16).astype(np.uint32) | (z.view(np.uint16).astype(np.uint32) << np.uint32(16))
        tab[:, si] = e.reshape(G, S, 16).transpose(1, 0, 2)
    return np.concatenate([R.reshape(-1).view(np.uint8), tab.reshape(-1).view(np.uint8)])


def stripe16_unpack3_fields(image, K, N, NS):
    """the integer fields [K, N] per set of a 3-bit image (inverse of the field placement above)."""
    nrb, S = K // 128, N // 16
    nR = S * nrb * NS * 64 * 3
    R = np.asarray(image, dtype=np.uint8)[:nR * 4].view(np.uint32).reshape(S, nrb, NS, 64, 3)
    out = []
    for si in range(NS):
        w = R[:, :, si].reshape(S, nrb, 4, 16, 3)                                          # [s][rb][rq][c][j]
        q = np.zeros((S, nrb, 4, 16, 32), dtype=np.uint32)
        for j in range(3):
            for p_ in range(5):
                q[..., 10 * j + 2 * p_] = (w[..., j] >> np.uint32(3 * p_)) & np.uint32(7)
                q[..., 10 * j + 2 * p_ + 1] = (w[..., j] >> np.uint32(16 + 3 * p_)) & np.uint32(7)
            q[..., 30] |= ((w[..., j] >> np.uint32(15)) & np.uint32(1)) << np.uint32(j)
            q[..., 31] |= ((w[..., j] >> np.uint32(31)) & np.uint32(1)) << np.uint32(j)
        out.append(q.transpose(1, 2, 4, 0, 3).reshape(K, N).astype(np.int32))                # [rb][rq][k][s][c] -> [K][N]
    return out


def stripe16_unpack(image, K, N, groupsize, NS, bits=4):
    """inverse of stripe16_repack: -> [(qweight int32 [K/KPW, N], scales fp16 [G, N], zeros+1 int [G, N])] per set."""
    kpw = 32 // bits
    nrb, S = K // (16 * kpw), N // 16
    gs = K if groupsize in (-1, None) or groupsize >= K else groupsize
    G = K // gs
    nR = S * nrb * NS * 256
    img = np.asarray(image, dtype=np.uint8)
    R = img[:nR * 4].view(np.uint32).reshape(S, nrb, NS, 64, 4)
    tab = img[nR * 4:nR * 4 + S * NS * G * 16 * 4].view(np.uint32).reshape(S, NS, G, 16)
    fm = np.uint32((1 << bits) - 1)
    out = []
    for si in range(NS):
        o = R[:, :, si].reshape(S, nrb, 4, 16, 4).transpose(1, 2, 4, 0, 3).reshape(K // kpw, N)
        w = np.zeros_like(o)
        for pos, k in enumerate(stripe_k_of_pos(bits)):
            w |= ((o >> np.uint32(bits * pos)) & fm) << np.uint32(bits * k)
        e = tab[:, si].transpose(1, 0, 2).reshape(G, N)
        sc = (e & np.uint32(0xFFFF)).astype(np.uint16).view(np.float16)
        z = (e >> np.uint32(16)).astype(np.uint16).view(np.float16).astype(np.int32)
        out.append((w.view(np.int32), sc, z))
    return out


def algorithmic_bytes(M, K, N, bits, groupsize, act_order=False, bias=False):
    """SURVEY.md section 8(d) / BASELINE.md section 3 byte model for one dequant-matmul."""
    G = n_groups(K, groupsize)
    b = 4 * (K * bits // 32) * N + 4 * G * (N * bits // 32) + 2 * G * N + 2 * M * K + 2 * M * N
    if act_order:
        b += 4 * K
    if bias:
        b += 2 * N
    return b
