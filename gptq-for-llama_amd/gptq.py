"""GPTQ solver for MI355X -- same surface as the reference's ``gptq.py`` (``GPTQ(layer)``, ``add_batch``,
``fasterquant`` -> ``(scale, zero, g_idx, error)``, ``free``, ``Observer``), so ``llama.py:95-166`` can drive it
unchanged.  It is the caller that produces the integer weights ``QuantLinear.pack`` packs.

What runs where:
  * Hessian accumulation (reference gptq.py:71-96): one fp32 GEMM per calibration batch on the GPU (torch).
  * damping + Hinv = chol(chol_inv(chol(H)), upper) (gptq.py:157-163): torch.linalg (rocSOLVER).
  * the sequential quantise / error-feedback loop over the columns of a block (gptq.py:177-199) -- ~10 dependent
    torch launches per column in the reference -- is ONE hand-written HIP launch per block
    (``gptq_solver_block_f32``, csrc/gptq_solver.hip): a wave per weight row, the block's columns in registers,
    the reference's fp32 operations one by one.
  * the grid of a group is fitted with ``quant.Quantizer.find_params`` on the global W before the block, exactly where
    the reference fits it (gptq.py:181-183: it reads ``W``, not the in-block clone), so ``sym`` / ``mse`` behave
    as upstream;
  * trailing update W[:, i2:] -= Err1 . Hinv[i1:i2, i2:] (gptq.py:204): one GEMM per block (torch / hipBLASLt).
There is no CPU path: the layer must live on a ROCm device (the CPU restatement used for parity and as the
reported-only baseline is oracle/gptq_solver.py, which this module never imports).
"""
import math
import time

import torch
import torch.nn as nn

import quant
from quant import _native

try:  # transformers is only needed to recognise Conv1D layers (OPT / GPT-2 style checkpoints)
    import transformers
    _Conv1D = transformers.Conv1D
except Exception:  # pragma: no cover
    _Conv1D = ()

KERNEL_MAX_BLOCK = 128   # columns the block kernel keeps in registers (two per lane)


class Observer:
    """keeps the ``topk`` layers with the largest quantisation error (reference gptq.py:15-53)."""

    def __init__(self, topk=32):
        self.loss_list = []
        self.topk = topk

    def submit(self, name, layerid, gptq, error):
        item = (name, layerid, {'gptq': gptq, 'error': error})
        if len(self.loss_list) < self.topk:
            self.loss_list.append(item)
            return
        worst = min(range(len(self.loss_list)), key=lambda i: self.loss_list[i][2]['error'])
        if self.loss_list[worst][2]['error'] < error:
            self.loss_list[worst] = item

    def print(self):
        self.loss_list.sort(key=lambda it: it[2]['error'], reverse=True)
        for name, layerid, info in self.loss_list:
            print('%-32s %.6f' % ('%s.%d' % (name, layerid), info['error']))
        print()

    def items(self):
        return self.loss_list


def _weight_matrix(layer):
    """[out_features, in_features] view of the layer's weight, as the reference flattens it (gptq.py:60-65)."""
    W = layer.weight.data.clone()
    if isinstance(layer, nn.Conv2d):
        W = W.flatten(1)
    if _Conv1D and isinstance(layer, _Conv1D):
        W = W.t()
    return W


def _inverse_factor(hess, percdamp):
    """upper Cholesky factor of (H + percdamp * mean(diag H) * I)^-1, via rocSOLVER (reference gptq.py:157-163)."""
    hess.diagonal().add_(percdamp * hess.diagonal().mean())
    lower = torch.linalg.cholesky(hess)
    return torch.linalg.cholesky(torch.cholesky_inverse(lower), upper=True).contiguous()


class GPTQ:

    def __init__(self, layer, observe=False):
        self.layer = layer
        self.dev = self.layer.weight.device
        W = _weight_matrix(layer)
        self.rows, self.columns = W.shape[0], W.shape[1]
        self.H = torch.zeros((self.columns, self.columns), device=self.dev)
        self.nsamples = 0
        self.quantizer = quant.Quantizer()
        self.observe = observe
        self.inp1 = self.out1 = None

    # -----------------------------------------------------------------------------------------
    def add_batch(self, inp, out):
        """H <- running mean of 2 X X^T over the calibration samples (reference gptq.py:71-96)."""
        if self.observe:
            self.inp1, self.out1 = inp, out
        else:
            self.inp1 = self.out1 = None
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        batch = inp.shape[0]
        if isinstance(self.layer, nn.Conv2d):
            unfold = nn.Unfold(self.layer.kernel_size, dilation=self.layer.dilation, padding=self.layer.padding, stride=self.layer.stride)
            inp = unfold(inp).permute([1, 0, 2]).flatten(1)
        else:
            if inp.dim() == 3:
                inp = inp.reshape((-1, inp.shape[-1]))
            inp = inp.t()
        self.H *= self.nsamples / (self.nsamples + batch)
        self.nsamples += batch
        x = math.sqrt(2 / self.nsamples) * inp.float()
        self.H.addmm_(x, x.t())

    # -----------------------------------------------------------------------------------------
    def _report(self, name, q_weight, weight_error, seconds):
        """assign the quantised weight and print one progress line (reference print_loss, gptq.py:98-126)."""
        self.layer.weight.data = q_weight.reshape(self.layer.weight.shape).to(self.layer.weight.data.dtype)
        fp_snr = q_snr = '-'
        if self.inp1 is not None:
            q8 = quant.Quantizer()
            q8.configure(8, perchannel=False, sym=True, mse=False)
            q8.find_params(self.inp1)
            q_out = self.layer(q8.quantize(self.inp1).type(torch.float16))

            def snr(pred, real):
                pred, real = pred.float().flatten(1), real.float().flatten(1)
                return float((((pred - real) ** 2).sum(-1) / (real ** 2).sum(-1)).mean())
            q_snr = '%.4g' % snr(q_out, self.out1)
            fp_snr = '%.4g' % snr(self.layer(self.inp1), self.out1)
        print('| %-16s | %12.4f | %10s | %10s | %8.3f |' % (name, weight_error, fp_snr, q_snr, seconds))

    def fasterquant(self, blocksize=128, percdamp=.01, groupsize=-1, actorder=False, name=''):
        if self.dev.type != 'cuda':
            raise RuntimeError('GPTQ.fasterquant: layer is on %s -- the MI355X solver has no CPU path' % self.dev)
        lib = _native.lib()
        self.layer.to(self.dev)
        W = _weight_matrix(self.layer).float().contiguous()
        tick = time.time()
        if int(self.quantizer.maxq) < 0:
            raise NotImplementedError('ternary grids (trits) are not supported by the HIP block solver')
        maxq = int(self.quantizer.maxq)

        if not self.quantizer.ready():                      # whole-row grid; the only one without groups (gptq.py:139-140)
            self.quantizer.find_params(W, weight=True)

        hess = self.H
        if not self.observe:
            del self.H
        never_seen = hess.diagonal() == 0                   # input features no calibration token activated (gptq.py:145-147)
        hess.diagonal()[never_seen] = 1
        W[:, never_seen] = 0

        order = None
        if actorder:                                        # most sensitive input features first (gptq.py:149-152)
            order = torch.argsort(hess.diagonal(), descending=True)
            W = W[:, order].contiguous()
            hess = hess[order][:, order]
        rows, cols = self.rows, self.columns
        Hinv = _inverse_factor(hess, percdamp)
        del hess

        gs = cols if groupsize == -1 else groupsize
        n_groups = (cols + gs - 1) // gs
        scale_all = torch.zeros((rows, n_groups), device=self.dev)
        zero_all = torch.zeros((rows, n_groups), device=self.dev)
        if groupsize == -1:
            scale_all[:, 0] = self.quantizer.scale.reshape(-1)
            zero_all[:, 0] = self.quantizer.zero.reshape(-1)
        Q = torch.zeros_like(W)
        Err = torch.empty((rows, KERNEL_MAX_BLOCK), device=self.dev)
        loss_rows = torch.zeros(rows, device=self.dev)
        stream = _native.stream_ptr(self.dev)

        with torch.cuda.device(self.dev):
            for i1 in range(0, cols, blocksize):
                i2 = min(i1 + blocksize, cols)
                if groupsize != -1:
                    # grids of the groups that START inside this block, fitted to W as it is now (gptq.py:181-183)
                    first = -(-i1 // gs) * gs
                    for c in range(first, i2, gs):
                        self.quantizer.find_params(W[:, c:c + gs], weight=True)
                        scale_all[:, c // gs] = self.quantizer.scale.reshape(-1)
                        zero_all[:, c // gs] = self.quantizer.zero.reshape(-1)
                for j1 in range(i1, i2, KERNEL_MAX_BLOCK):   # the kernel keeps <= 128 columns in registers
                    j2 = min(j1 + KERNEL_MAX_BLOCK, i2)
                    rc = lib.gptq_solver_block_f32(W.data_ptr(), cols, Hinv.data_ptr(), cols, rows, cols, j1, j2 - j1, gs, maxq,
                                                   scale_all.data_ptr(), zero_all.data_ptr(), n_groups, Q.data_ptr(), cols,
                                                   Err.data_ptr(), KERNEL_MAX_BLOCK, loss_rows.data_ptr(), stream)
                    _native.check(rc, 'gptq_solver_block_f32')
                    if j2 < cols:
                        W[:, j2:] -= Err[:, :j2 - j1].matmul(Hinv[j1:j2, j2:])
            torch.cuda.synchronize()
        error = float(loss_rows.double().sum().item())

        g_idx = (torch.arange(cols, device=self.dev) // gs).to(torch.int32)
        if order is not None:                               # back to the checkpoint's column order (gptq.py:212-215)
            restore = torch.argsort(order)
            Q, g_idx = Q[:, restore], g_idx[restore]
        if _Conv1D and isinstance(self.layer, _Conv1D):
            Q = Q.t()
        self._report(name, Q, error, time.time() - tick)
        return scale_all, zero_all, g_idx, error

    def free(self):
        self.inp1 = self.out1 = None
        self.H = None
        self.Losses = None
        self.Trace = None
        torch.cuda.empty_cache()
