"""Pins the CPU oracle (oracle/) to the fixtures produced by the REFERENCE ITSELF
(tests/golden/gen_golden.py: the reference's Triton kernels under TRITON_INTERPRET=1 and its
own QuantLinear.pack).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def names(prefix):
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, prefix + '*.npz')))


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def ulp_diff_fp16(a, b):
    """max difference in units of fp16 ulps (monotone integer mapping of fp16 bit patterns)."""
    def key(x):
        u = np.asarray(x, dtype=np.float16).view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, -(u & 0x7FFF), u)
    return int(np.abs(key(a) - key(b)).max())


def test_fixture_inventory():
    assert len(names('pack_')) == 5
    assert len(names('fwd_')) == 8
    assert len(names('bwd_')) == 4
    assert len(names('mlp_')) == 3
    assert len(names('norm_')) == 3
    assert len(names('rope_')) == 3
    assert len(names('gptq_')) == 6


@pytest.mark.parametrize('name', names('pack_'))
def test_pack_bit_exact_vs_reference_pack(name):
    """oracle.pack == reference QuantLinear.pack, bit for bit (quant_linear.py:325-371)."""
    f = load(name)
    bits, gs = int(f['bits']), int(f['groupsize'])
    qweight, qzeros, scales, g_idx = oracle.pack(f['weight_q'], f['scales_in'], f['zeros_in'],
                                                 f['g_idx'], bits, gs)
    assert np.array_equal(qweight, f['qweight'])
    assert np.array_equal(qzeros, f['qzeros'])
    assert np.array_equal(scales.view(np.uint16), f['scales'].view(np.uint16))


@pytest.mark.parametrize('name', names('pack_'))
def test_unpack_inverts_reference_pack(name):
    """dequant(reference pack(Q)) reproduces the grid-valued weight Q the reference packed,
    up to the fp16 rounding of the scale (4.2e-5 at these magnitudes)."""
    f = load(name)
    bits = int(f['bits'])
    W = oracle.dequant(f['qweight'], f['qzeros'], f['scales'], f['g_idx'], bits, faithful=False)
    Wn = oracle.np_dequant(f['qweight'], f['qzeros'], f['scales'], f['g_idx'], bits, faithful=False)
    assert np.array_equal(W, Wn)           # C restatement == numpy restatement
    Q = f['weight_q'].T                     # [K, N]
    s = f['scales'].astype(np.float32)[f['g_idx']]
    # |W - Q| <= |q - z| * |s16 - s| <= maxq * half-ulp(s)
    bound = (2 ** bits) * np.abs(s) * 2.0 ** -11 + 1e-7
    assert np.all(np.abs(W - Q) <= bound)
    # the integer fields themselves are exactly the ones the reference packed
    q = oracle.np_unpack_rows(f['qweight'], bits)
    z = oracle.np_unpack_cols(f['qzeros'], bits) + 1
    intw = np.rint((f['weight_q'].T + (f['scales_in'].T * f['zeros_in'].T)[f['g_idx']]) /
                   f['scales'].astype(np.float32)[f['g_idx']]).astype(np.int64)
    assert np.array_equal(q, intw)
    assert np.array_equal(z, f['zeros_in'].T.astype(np.int64))


@pytest.mark.parametrize('name', names('pack_') + names('fwd_'))
def test_forward_vs_reference_kernel(name):
    """oracle.matmul248 vs the reference's matmul_248_kernel output.  fp32 summation order
    differs (the kernel sums per 32-wide K block), so allow 1 fp16 ulp."""
    f = load(name)
    bits = int(f['bits'])
    y = oracle.matmul248(f['x'], f['qweight'], f['scales'], f['qzeros'], f['g_idx'], bits)
    assert y.shape == f['y'].shape
    assert ulp_diff_fp16(y, f['y']) <= 1
    assert rel_err(y, f['y']) < 1e-3
    # and the float64 "exact" flavour sits inside the stated tolerance of the reference output
    ye = oracle.matmul248_exact(f['x'], f['qweight'], f['scales'], f['qzeros'], f['g_idx'], bits)
    assert rel_err(ye, f['y']) < 1e-3


@pytest.mark.parametrize('name', names('pack_') + names('fwd_'))
def test_dequant_c_equals_numpy(name):
    f = load(name)
    bits = int(f['bits'])
    for faithful in (True, False):
        a = oracle.dequant(f['qweight'], f['qzeros'], f['scales'], f['g_idx'], bits, faithful)
        b = oracle.np_dequant(f['qweight'], f['qzeros'], f['scales'], f['g_idx'], bits, faithful)
        assert np.array_equal(a, b)


@pytest.mark.parametrize('name', names('bwd_'))
def test_backward_vs_reference_kernel(name):
    f = load(name)
    bits = int(f['bits'])
    dx = oracle.transpose_matmul248(f['dy'], f['qweight'], f['scales'], f['qzeros'], f['g_idx'], bits)
    assert dx.shape == f['dx'].shape
    assert ulp_diff_fp16(dx, f['dx']) <= 1
    assert rel_err(dx, f['dx']) < 1e-3


@pytest.mark.parametrize('name', names('mlp_'))
def test_fused_mlp_vs_reference_kernel(name):
    f = load(name)
    bits = int(f['gate_bits'])
    gate = (f['gate_qweight'], f['gate_scales'], f['gate_qzeros'], f['gate_g_idx'])
    up = (f['up_qweight'], f['up_scales'], f['up_qzeros'], f['up_g_idx'])
    c = oracle.fused_mlp(f['x'], gate, up, bits)
    assert c.shape == f['c'].shape
    assert ulp_diff_fp16(c, f['c']) <= 2
    assert rel_err(c, f['c']) < 1e-3


@pytest.mark.parametrize('name', names('norm_'))
def test_rmsnorm_vs_reference_kernel(name):
    f = load(name)
    y = oracle.rmsnorm(f['x'], f['w'], float(f['eps']))
    assert ulp_diff_fp16(y, f['y']) <= 1
    assert rel_err(y, f['y']) < 1e-3


@pytest.mark.parametrize('name', names('rope_'))
def test_rope_vs_reference_kernel(name):
    f = load(name)
    qkv = f['qkv_in'].copy()
    oracle.rope_(qkv[:, :, :2], f['pos'])
    # v untouched, q/k rotated
    assert np.array_equal(qkv[:, :, 2].view(np.uint16), f['qkv_out'][:, :, 2].view(np.uint16))
    assert np.abs(qkv.astype(np.float32) - f['qkv_out'].astype(np.float32)).max() < 4e-3
    assert rel_err(qkv, f['qkv_out']) < 1e-3


def test_bias_is_added_after_fp16_store():
    f = load('pack_w4gall_sym_bias.npz')
    bits = int(f['bits'])
    y0 = oracle.matmul248(f['x'], f['qweight'], f['scales'], f['qzeros'], f['g_idx'], bits)
    y1 = oracle.matmul248(f['x'], f['qweight'], f['scales'], f['qzeros'], f['g_idx'], bits,
                          bias=f['bias'])
    want = (y0.astype(np.float32) + f['bias'].astype(np.float32)[None, :]).astype(np.float16)
    assert np.array_equal(y1.view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize('bits', [2, 3, 4, 8])
@pytest.mark.parametrize('groupsize', [-1, 32, 128])
def test_pack_unpack_roundtrip_all_bits(bits, groupsize):
    """Field-level round trip incl. the 3-bit EXTENSION layout (reference raises
    NotImplementedError for 3 bits, quant_linear.py:308-309 -> parity unpinned there)."""
    rng = np.random.default_rng(bits * 100 + (groupsize % 1000))
    K, N = 256, 96 if bits == 3 else 64
    G = oracle.n_groups(K, groupsize)
    q = rng.integers(0, 2 ** bits, size=(K, N))
    z = rng.integers(1, 2 ** bits + 1, size=(G, N))   # stored as z-1 in [0, maxq]
    qw = oracle.np_pack_fields_rows(q, bits)
    qz = oracle.np_pack_fields_cols(z - 1, bits)
    assert qw.shape == (K // 32 * bits, N) and qz.shape == (G, N // 32 * bits)
    assert np.array_equal(oracle.np_unpack_rows(qw, bits), q)
    assert np.array_equal(oracle.np_unpack_cols(qz, bits) + 1, z)
    s = rng.uniform(0.001, 0.011, size=(G, N)).astype(np.float16)
    g = oracle.trivial_g_idx(K, groupsize)
    W = oracle.dequant(qw, qz, s, g, bits, faithful=False)
    want = (q - z[g]).astype(np.float32) * s[g].astype(np.float32)
    assert np.array_equal(W, want)
    # C packer agrees with the numpy field packer when fed grid-valued weights
    Wt = want.T.copy()                                     # [N, K]
    qw2, qz2, s2, _ = oracle.pack(Wt, s.astype(np.float32).T, z.astype(np.float32).T, g, bits, groupsize)
    assert np.array_equal(qw2, qw) and np.array_equal(qz2, qz)


def test_zero_point_16_is_not_remasked():
    """stored nibble 15 -> z = 16 (quant_linear.py:120-121 adds 1 after the mask)."""
    K, N = 32, 32
    qw = oracle.np_pack_fields_rows(np.zeros((K, N), dtype=np.int64), 4)
    qz = oracle.np_pack_fields_cols(np.full((1, N), 15), 4)
    s = np.ones((1, N), dtype=np.float16)
    W = oracle.dequant(qw, qz, s, np.zeros(K, np.int32), 4, faithful=True)
    assert np.all(W == -16.0)


def test_reference_zero_eq_0_bug_is_replicated():
    """float zero == 0 -> (0 - 1) -> uint32 0xFFFFFFFF OR-ed into the word: every higher field
    of that qzeros word becomes all-ones.  The packer replicates, not fixes."""
    K, N, bits = 32, 32, 4
    W = np.zeros((N, K), dtype=np.float32)
    scales = np.ones((N, 1), dtype=np.float32)
    zeros = np.full((N, 1), 3.0, dtype=np.float32)
    zeros[2, 0] = 0.0
    W[:] = -3.0
    W[2] = 0.0
    qw, qz, s16, g = oracle.pack(W, scales, zeros, None, bits, -1)
    fields = oracle.np_unpack_cols(qz, bits)[0]
    assert list(fields[:2]) == [2, 2]
    assert all(v == 15 for v in fields[2:8])      # rest of the first word: all-ones
    assert all(v == 2 for v in fields[8:])        # other words untouched


# ---------------------------------------------------------------------------------------
# GPTQ solver restatement (oracle/gptq_solver.py) vs the reference's own GPTQ class run on the CPU
# (tests/golden/gen_golden_gptq.py).  LAPACK/BLAS summation order differs between numpy and torch, so the grid
# scale may move by an ulp after the first trailing update; the integer levels must still agree.
# ---------------------------------------------------------------------------------------
def gptq_levels(Q, scale, zero, g_idx):
    return np.rint(Q / scale[:, g_idx]) + zero[:, g_idx]


@pytest.mark.parametrize('name', names('gptq_'))
def test_gptq_solver_vs_reference_gptq(name):
    from oracle import gptq_solver as G
    f = load(name)
    cols = f['W'].shape[1]
    H, n = np.zeros((cols, cols), np.float32), 0
    for batch in f['X']:
        H, n = G.hessian_add_batch(H, n, batch)
    assert rel_err(H, f['H']) < 1e-6                                   # gptq.py:71-96
    Q, scale, zero, g_idx, err = G.fasterquant(f['W'], f['H'], int(f['bits']), int(f['blocksize']), float(f['percdamp']),
                                               int(f['groupsize']), bool(f['actorder']), bool(f['sym']))
    assert np.array_equal(g_idx, f['g_idx'])
    assert np.array_equal(zero, f['zero'])
    assert np.array_equal(scale[:, 0], f['scale'][:, 0]) or bool(f['actorder'])   # first grid: no BLAS involved yet
    assert np.max(np.abs(scale - f['scale']) / f['scale']) < 1e-5
    assert np.mean(gptq_levels(Q, scale, zero, g_idx) != gptq_levels(f['Q'], f['scale'], f['zero'], f['g_idx'])) < 1e-3
    assert rel_err(Q, f['Q']) < 1e-5
    assert abs(err - float(f['error'])) / float(f['error']) < 1e-4


@pytest.mark.parametrize('bits,gs,act', [(4, 64, False), (3, -1, False), (4, 32, True), (2, 64, True)])
def test_gptq_solver_properties(bits, gs, act):
    """size-independent properties of the solver restatement: every output weight lies on its group's grid, g_idx
    has exactly `groupsize` members per group, and the error-feedback solution beats round-to-nearest on the
    calibration inputs (the whole point of gptq.py:171-205)."""
    from oracle import gptq_solver as G
    rng = np.random.default_rng(bits * 100 + (gs if gs > 0 else 7))
    rows, cols = 40, 256
    W = (rng.standard_normal((rows, cols)) * 0.05).astype(np.float32)
    mix = (rng.standard_normal((cols, cols)) * 0.2 + np.eye(cols)).astype(np.float32)
    X = (rng.standard_normal((512, cols)).astype(np.float32) @ mix) * np.exp(rng.standard_normal(cols) * 0.6).astype(np.float32)
    H, n = G.hessian_add_batch(np.zeros((cols, cols), np.float32), 0, X[None])
    Q, scale, zero, g_idx, err = G.fasterquant(W, H, bits, 128, 0.01, gs, act, False)
    maxq = 2 ** bits - 1
    lv = Q / scale[:, g_idx] + zero[:, g_idx]
    assert np.abs(lv - np.rint(lv)).max() < 1e-3 and lv.min() > -1e-3 and lv.max() < maxq + 1e-3
    gsz = cols if gs == -1 else gs
    assert np.array_equal(np.bincount(g_idx, minlength=cols // gsz), np.full(cols // gsz, gsz))
    ref = X @ W.T
    Wg = W.reshape(rows * (cols // gsz), gsz)
    s_r, z_r = G.find_params(Wg, maxq, False)
    rtn = G.quantize(Wg, s_r[:, None], z_r[:, None], maxq).reshape(rows, cols)
    assert np.linalg.norm(X @ Q.T - ref) < np.linalg.norm(X @ rtn.T - ref)
    assert err > 0
