"""One fresh process: prepare a 4-bit 4096 x 4096 layer, run the one-row decode launch as the FIRST kernel launch of the library in this process,
run it again, and compare both against a torch fp32 product on the dequantised weight (round 5: a one-off mismatch of the first test of a pytest
process on one box -- tests/test_gpu_batch.py::test_layer_decode_norm_and_residual[4-4096-4096-128-1-1] -- is looked for here in many processes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'gptq-for-llama_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch
import quant
from quant import _native
from quant.layer import prepared
from util import make_random_layer

DEV = 'cuda:0'
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
bits, gs, K, N = 4, 128, 4096, 4096
L = make_random_layer(bits, gs, K, N, seed=500 + bits)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
sets = ((dev(L['qweight']), dev(L['scales']), dev(L['qzeros']), dev(L['g_idx'])),)
pl = prepared(sets, None, bits, gs, K, N)
x = dev(np.random.default_rng(K + 1).standard_normal((1, K)).astype(np.float16))
lib = _native.lib()
s = _native.stream_ptr(torch.device(DEV))
ws = _native.layer_workspace(torch.device(DEV), s)
scratch = torch.empty(256, dtype=torch.uint8, device=DEV)
ys = []
for i in range(2):
    y = torch.full((1, N), float('nan'), dtype=torch.float16, device=DEV)
    rc = lib.gptq_layer_decode_f16(pl.handle, x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), 1, None, 1e-6, None, 0, ws.data_ptr(), ws.numel(),
                                   scratch.data_ptr(), scratch.numel(), s)
    assert rc == 0, rc
    torch.cuda.synchronize()
    ys.append(y.float().cpu())
W = quant.quant_linear.dequantize(sets[0][0], sets[0][1], sets[0][2], sets[0][3], bits).float()      # [N, K] or [K, N]
ref = (x.float() @ (W if W.shape[0] == K else W.t())).cpu()
d = [float((y - ref).abs().max() / ref.abs().max()) for y in ys]
same = bool(torch.equal(ys[0], ys[1]))
print('seed %d first %.2e second %.2e first == second %s %s' % (seed, d[0], d[1], same, 'OK' if max(d) < 2e-3 and same else 'MISMATCH'), flush=True)
