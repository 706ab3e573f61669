"""Shared helpers for the parity tests (test infrastructure; may use the oracle)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def golden_names(prefix):
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith('.npz'))


def act_order_g_idx(K, gs, rng):
    perm = rng.permutation(K)
    return (np.arange(K) // gs)[np.argsort(perm)].astype(np.int32)


def make_random_layer(bits, groupsize, K, N, act_order=False, seed=0):
    """SURVEY 8(d) synthetic distribution: uniform int32 bit patterns (every field uniform, sign
    bit exercised), scales ~ U(0.001, 0.011) fp16, trivial or act-order g_idx."""
    rng = np.random.default_rng(seed)
    gs = K if groupsize == -1 else groupsize
    G = -(-K // gs)
    qweight = rng.integers(-2**31, 2**31, size=(K // 32 * bits, N), dtype=np.int64).astype(np.int32)
    qzeros = rng.integers(-2**31, 2**31, size=(G, N // 32 * bits), dtype=np.int64).astype(np.int32)
    scales = rng.uniform(0.001, 0.011, size=(G, N)).astype(np.float16)
    g_idx = act_order_g_idx(K, gs, rng) if act_order else (np.arange(K) // gs).astype(np.int32)
    return dict(qweight=qweight, qzeros=qzeros, scales=scales, g_idx=g_idx, bits=bits, groupsize=gs)


def rel_err(a, b):
    """max|a-b| / max|b| -- the tolerance definition of SURVEY 0.6 (bar: < 1e-3)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


TOL = 1e-3


def within(name, err, tol):
    """assert err < tol; with GPTQ_TEST_ERRLOG=<file> the observed error is logged (used to set the model-level tolerances)."""
    path = os.environ.get('GPTQ_TEST_ERRLOG')
    if path:
        with open(path, 'a') as f:
            f.write('%s %.3e %.1e\n' % (name, float(err), tol))
    assert err < tol, (name, float(err), tol)
