import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, torch
from quant import quant_linear as QL
from oracle import oracle
from util import make_random_layer, rel_err
dev='cuda:0'
d=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for bits in (2,3,4,8):
    for M in (1,3,9):
        for act in (False, True):
            L = make_random_layer(bits, 128, 512, 256, act_order=act, seed=bits)
            x = np.random.default_rng(M + 20).standard_normal((M, 512)).astype(np.float16)
            y = QL.matmul248(d(x), d(L['qweight']), d(L['scales']), d(L['qzeros']), d(L['g_idx']), bits, 2**bits-1).cpu().numpy()
            ref = oracle.matmul248(x, L['qweight'], L['scales'], L['qzeros'], L['g_idx'], bits)
            ex = oracle.matmul248_exact(x, L['qweight'], L['scales'], L['qzeros'], L['g_idx'], bits)
            bad = np.argwhere(np.abs(y.astype(np.float64)-ref.astype(np.float64)) > 1e-3*np.abs(ref).max())
            print(bits, M, act, 'rel vs faithful %.2e vs exact %.2e  oracle-vs-exact %.2e' % (rel_err(y, ref), rel_err(y, ex), rel_err(ref, ex)), 'nbad', len(bad), bad[:4].tolist())
