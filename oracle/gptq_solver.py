"""CPU restatement (numpy, fp32) of the GPTQ column-wise Hessian loop -- the CALLER that produces the packed weights
the hot path consumes.  TEST / BASELINE INFRASTRUCTURE ONLY: imported by tests/, bench.py's reported-only
``cpu_baseline.gptq_loop`` leg and nothing else; the product (gptq-for-llama_amd/gptq.py) never touches it.

Follows the reference line by line in behaviour (not in code):
  * Hessian accumulation          gptq.py:71-96   (H <- H*n/(n+b); n += b; H += (sqrt(2/n) X)(sqrt(2/n) X)^T, fp32)
  * min/max affine grid           quant/quantizer.py:32-76 with perchannel=True, weight=True, mse=False -- the
                                  configuration llama.py:156 uses; (scale, zero) per output row
  * round trip                    quant/quantizer.py:28-32  (clamp(round(x/scale)+zero, 0, maxq) - zero) * scale
  * fasterquant                   gptq.py:128-228: dead columns (:145-147), act-order permutation (:149-152),
                                  damping percdamp*mean(diag H) (:157-159), Hinv = chol(chol_inv(chol(H)), upper) (:160-163),
                                  per block of `blocksize` columns: sequential quantise / error feedback inside the
                                  block (:177-199), trailing update of all later columns (:204).
    Quirk kept on purpose: at a group boundary INSIDE a block the grid is fitted to the global W, which does not yet
    contain the in-block updates (gptq.py:183 reads W, the loop works on the clone W1).
Pinned by tests/test_oracle_golden.py against tests/golden/gptq_*.npz, which tests/golden/gen_golden_gptq.py records
from the reference's own GPTQ class run on the CPU.  LAPACK (scipy potrf/potri) is used where the reference uses
torch.linalg.cholesky / cholesky_inverse; the trailing update is a BLAS GEMM in both, so individual quantised levels
can differ at rounding boundaries (the test bounds the fraction).
"""
import numpy as np
from scipy.linalg import lapack

F = np.float32


def hessian_add_batch(H, nsamples, inp):
    """One reference add_batch call for an nn.Linear: inp [batch, seq, K] or [tokens, K].  Returns (H, nsamples)."""
    inp = np.asarray(inp, dtype=F)
    batch = 1 if inp.ndim == 2 else inp.shape[0]
    x = inp.reshape(-1, inp.shape[-1]).T                      # [K, tokens]
    H = H * F(nsamples / (nsamples + batch))
    nsamples += batch
    x = F(np.sqrt(2.0 / nsamples)) * x
    return (H + x @ x.T).astype(F), nsamples


def find_params(rows, maxq, sym):
    """per-row (scale, zero) of the min/max grid; rows [R, n] fp32."""
    lo = np.minimum(rows.min(axis=1), F(0))
    hi = np.maximum(rows.max(axis=1), F(0))
    if sym:
        hi = np.maximum(np.abs(lo), hi)
        lo = np.where(lo < 0, -hi, lo)
    flat = (lo == 0) & (hi == 0)
    lo = np.where(flat, F(-1), lo).astype(F)
    hi = np.where(flat, F(1), hi).astype(F)
    scale = ((hi - lo) / F(maxq)).astype(F)
    if sym:
        zero = np.full_like(scale, F((maxq + 1) / 2))
    else:
        zero = np.rint(-lo / scale).astype(F)                  # torch.round == round half to even
    return scale, zero


def quantize(x, scale, zero, maxq):
    level = np.clip(np.rint(x / scale) + zero, F(0), F(maxq))
    return (scale * (level - zero)).astype(F)


def inverse_factor(H, percdamp):
    """damped Hessian -> upper Cholesky factor of its inverse (reference gptq.py:157-163), LAPACK fp32 like torch on the CPU."""
    H = np.array(H, dtype=F, copy=True)
    cols = H.shape[0]
    damp = F(percdamp) * np.mean(np.diag(H), dtype=F)
    H[np.arange(cols), np.arange(cols)] += damp
    L, info = lapack.spotrf(H, lower=1)
    assert info == 0, 'Hessian not positive definite'
    Hi, info = lapack.spotri(L, lower=1)
    assert info == 0
    Hi = np.tril(Hi) + np.tril(Hi, -1).T                      # potri fills one triangle
    U, info = lapack.spotrf(Hi, lower=0)
    assert info == 0
    return np.triu(U).astype(F)


def fasterquant(W, H, bits, blocksize=128, percdamp=0.01, groupsize=-1, actorder=False, sym=False):
    """W [rows, cols] fp32 (nn.Linear.weight), H [cols, cols] fp32.  Returns Q (fp32, same shape, original column
    order), scale [rows, groups], zero [rows, groups], g_idx [cols] int32, error (float)."""
    W = np.array(W, dtype=F, copy=True)
    H = np.array(H, dtype=F, copy=True)
    rows, cols = W.shape
    maxq = 2 ** bits - 1
    scale_cur, zero_cur = find_params(W, maxq, sym)           # whole-row grid (the only one when groupsize == -1)

    dead = np.diag(H) == 0
    H[dead, dead] = 1
    W[:, dead] = 0
    perm = None
    if actorder:
        perm = np.argsort(-np.diag(H), kind='stable')
        W = W[:, perm]
        H = H[perm][:, perm]

    Hinv = inverse_factor(H, percdamp)

    Q = np.zeros_like(W)
    loss_total = 0.0
    scales, zeros = [], []
    now_idx = 1
    for i1 in range(0, cols, blocksize):
        i2 = min(i1 + blocksize, cols)
        W1 = W[:, i1:i2].copy()
        Err1 = np.zeros_like(W1)
        Hinv1 = Hinv[i1:i2, i1:i2]
        for i in range(i2 - i1):
            w = W1[:, i]
            d = Hinv1[i, i]
            if groupsize != -1:
                if (i1 + i) % groupsize == 0:
                    scale_cur, zero_cur = find_params(W[:, i1 + i:i1 + i + groupsize], maxq, sym)
                if (i1 + i) // groupsize - now_idx == -1:
                    scales.append(scale_cur)
                    zeros.append(zero_cur)
                    now_idx += 1
            q = quantize(w, scale_cur, zero_cur, maxq)
            Q[:, i1 + i] = q
            loss_total += float(np.sum(((w - q) ** 2 / d ** 2).astype(F), dtype=np.float64)) / 2
            err = ((w - q) / d).astype(F)
            W1[:, i:] -= err[:, None] * Hinv1[i, i:][None, :]
            Err1[:, i] = err
        W[:, i2:] -= Err1 @ Hinv[i1:i2, i2:]

    gs = groupsize if groupsize != -1 else cols
    g_idx = (np.arange(cols) // gs).astype(np.int32)
    if actorder:
        inv = np.argsort(perm, kind='stable')
        Q = Q[:, inv]
        g_idx = g_idx[inv]
    if not scales:
        scales.append(scale_cur)
        zeros.append(zero_cur)
    return Q, np.stack(scales, axis=1), np.stack(zeros, axis=1), g_idx, loss_total
