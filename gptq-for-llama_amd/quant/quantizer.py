"""Min/max affine quantiser used by the offline GPTQ pass -- same public surface as the
reference ``quant/quantizer.py:7-127`` (``configure`` / ``find_params`` / ``quantize`` /
``enabled`` / ``ready`` and the ``maxq`` / ``scale`` / ``zero`` buffers) so the reference's
``gptq.py`` can drive it unchanged.  It is host-side calibration code, not part of the hot path;
written from the behaviour: per-row (or per-tensor) range -> (scale, zero) on a 2**bits grid,
optional symmetric range, optional shrink-the-range search minimising an L_norm error.
"""
import torch
import torch.nn as nn


def _affine_round_trip(x, scale, zero, maxq):
    """x -> grid -> x.  maxq < 0 selects the ternary ("trits") grid of the reference (:29-30)."""
    if maxq < 0:
        hi = (x > scale / 2).float() * scale
        lo = (x < zero / 2).float() * zero
        return hi + lo
    level = torch.clamp(torch.round(x / scale) + zero, 0, maxq)
    return (level - zero) * scale


class Quantizer(nn.Module):

    def __init__(self, shape=1):
        super().__init__()
        self.register_buffer('maxq', torch.tensor(0))
        self.register_buffer('scale', torch.zeros(shape))
        self.register_buffer('zero', torch.zeros(shape))

    def configure(self, bits, perchannel=False, sym=True, mse=False, norm=2.4, grid=100, maxshrink=.8, trits=False):
        self.maxq = torch.tensor(-1 if trits else 2**bits - 1)
        self.perchannel, self.sym, self.mse = perchannel, sym, mse
        self.norm, self.grid, self.maxshrink = norm, grid, maxshrink
        self.scale = torch.zeros_like(self.scale)

    _quantize = staticmethod(_affine_round_trip)

    # ---------------------------------------------------------------------------------------
    @staticmethod
    def _rows(x, perchannel, weight):
        """2-D view with one row per quantisation channel."""
        if not perchannel:
            return x.flatten().unsqueeze(0)
        if weight:
            return x.flatten(1)
        if x.dim() == 4:
            return x.permute([1, 0, 2, 3]).flatten(1)
        if x.dim() == 3:
            return x.reshape((-1, x.shape[-1])).t()
        return x.t()

    def _range_to_params(self, lo, hi):
        if self.maxq < 0:
            return hi, lo
        scale = (hi - lo) / self.maxq
        if self.sym:
            zero = torch.full_like(scale, (self.maxq + 1) / 2)
        else:
            zero = torch.round(-lo / scale)
        return scale, zero

    def find_params(self, x, weight=False):
        dev = x.device
        self.maxq = self.maxq.to(dev)
        shape = x.shape
        rows = self._rows(x, self.perchannel, weight)

        zero_row = torch.zeros(rows.shape[0], device=dev)
        lo = torch.minimum(rows.min(1)[0], zero_row)
        hi = torch.maximum(rows.max(1)[0], zero_row)
        if self.sym:
            hi = torch.maximum(lo.abs(), hi)
            neg = lo < 0
            if torch.any(neg):
                lo[neg] = -hi[neg]
        flat = (lo == 0) & (hi == 0)
        lo[flat] = -1
        hi[flat] = +1

        self.scale, self.zero = self._range_to_params(lo, hi)

        if self.mse:
            best = torch.full([rows.shape[0]], float('inf'), device=dev)
            for step in range(int(self.maxshrink * self.grid)):
                shrink = 1 - step / self.grid
                lo1, hi1 = shrink * lo, shrink * hi
                scale1 = (hi1 - lo1) / self.maxq
                zero1 = self.zero if self.sym else torch.round(-lo1 / scale1)
                err = _affine_round_trip(rows, scale1.unsqueeze(1), zero1.unsqueeze(1), self.maxq)
                err = (err - rows).abs_().pow_(self.norm).sum(1)
                better = err < best
                if torch.any(better):
                    best[better] = err[better]
                    self.scale[better] = scale1[better]
                    self.zero[better] = zero1[better]

        if not self.perchannel:
            reps = shape[0] if weight else (shape[1] if len(shape) != 3 else shape[2])
            self.scale = self.scale.repeat(reps)
            self.zero = self.zero.repeat(reps)

        if weight:
            bshape = [-1] + [1] * (len(shape) - 1)
        elif len(shape) == 4:
            bshape = (1, -1, 1, 1)
        elif len(shape) == 3:
            bshape = (1, 1, -1)
        elif len(shape) == 2:
            bshape = (1, -1)
        else:
            return
        self.scale = self.scale.reshape(bshape)
        self.zero = self.zero.reshape(bshape)

    def quantize(self, x):
        if self.ready():
            return _affine_round_trip(x, self.scale, self.zero, self.maxq)
        return x

    def enabled(self):
        return self.maxq > 0

    def ready(self):
        return torch.all(self.scale != 0)
