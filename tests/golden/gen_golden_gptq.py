#!/usr/bin/env python3
"""Record tests/golden/gptq_*.npz FROM THE REFERENCE's own GPTQ class (gptq.py:56-236) run on the CPU.

    python tests/golden/gen_golden_gptq.py [--ref /root/reference]

Needs the read-only upstream tree (never read by tests, smoke() or bench.py).  ``texttable`` is absent from this
image and only formats the reference's progress line, so a do-nothing stand-in is registered before the import.
Each fixture holds the seeded inputs (nn.Linear weight, calibration batches, settings) and what the reference
returned: the quantised weight it assigned to the layer, scale, zero, g_idx, the loss, and the Hessian.
The reference source is imported, never copied.
"""
import argparse
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name,            rows, cols, bits, groupsize, actorder, sym,  blocksize, tokens, dead column
    ('gptq_w4g64',       96, 256, 4, 64, False, False, 128, 96, None),
    ('gptq_w4g128_act',  64, 256, 4, 128, True, False, 128, 128, None),
    ('gptq_w3gall_sym',  48, 128, 3, -1, False, True, 128, 64, None),
    ('gptq_w4g32',       64, 256, 4, 32, False, False, 128, 96, None),     # group boundaries inside a block
    ('gptq_w2g64_dead',  32, 192, 2, 64, True, False, 64, 80, 17),         # a never-activated input column
    ('gptq_w8g128',      40, 384, 8, 128, False, False, 128, 96, None),
]


def import_reference(ref):
    class _Table:
        def __getattr__(self, name):
            return lambda *a, **k: None

        def draw(self):
            return 'a\nb\nc'
    m = types.ModuleType('texttable')
    m.Texttable = _Table
    sys.modules.setdefault('texttable', m)
    sys.modules.setdefault('toml', types.ModuleType('toml'))
    os.environ.setdefault('TRITON_INTERPRET', '1')
    sys.path.insert(0, ref)
    import gptq  # the REFERENCE's gptq.py
    assert os.path.abspath(gptq.__file__).startswith(os.path.abspath(ref)), gptq.__file__
    return gptq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    args = ap.parse_args()
    gptq = import_reference(args.ref)
    torch.cuda.synchronize = lambda *a, **k: None       # gptq.py:206 synchronises unconditionally; CPU run here
    for seed, (name, rows, cols, bits, gs, act, sym, bs, tokens, dead) in enumerate(CASES):
        torch.manual_seed(100 + seed)
        layer = nn.Linear(cols, rows, bias=False)
        W0 = layer.weight.data.clone()
        # correlated, unevenly scaled activations so that act-order and the error feedback matter
        mix = torch.randn(cols, cols) * 0.15 + torch.eye(cols)
        colscale = torch.exp(torch.randn(cols) * 0.7)
        batches = [((torch.randn(1, tokens, cols) @ mix) * colscale).contiguous() for _ in range(2)]
        if dead is not None:
            for b in batches:
                b[..., dead] = 0
        g = gptq.GPTQ(layer)
        g.quantizer.configure(bits, perchannel=True, sym=sym, mse=False)      # llama.py:156
        for b in batches:
            g.add_batch(b, layer(b))
        H = g.H.clone()
        scale, zero, g_idx, error = g.fasterquant(blocksize=bs, percdamp=0.01, groupsize=gs, actorder=act, name=name)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), W=W0.numpy(), X=torch.stack(batches).numpy(), bits=bits, groupsize=gs,
                            actorder=act, sym=sym, blocksize=bs, percdamp=0.01, H=H.numpy(), Q=layer.weight.data.float().numpy(),
                            scale=scale.numpy(), zero=zero.numpy(), g_idx=g_idx.numpy(), error=np.float64(error))
        print('%-18s rows %3d cols %3d  error %.6g  groups %d' % (name, rows, cols, error, scale.shape[1]))


if __name__ == '__main__':
    main()
