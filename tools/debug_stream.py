import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, torch
from quant import quant_linear as QL
from oracle import oracle
from util import make_random_layer, rel_err, load_golden
dev='cuda:0'
d=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def run(tag, x, L):
    bits=int(L['bits'])
    y = QL.matmul248(d(x), d(L['qweight']), d(L['scales']), d(L['qzeros']), d(L['g_idx']), bits, 2**bits-1).cpu().numpy()
    ref = oracle.matmul248(x, L['qweight'], L['scales'], L['qzeros'], L['g_idx'], bits)
    err = np.abs(y.astype(np.float64)-ref.astype(np.float64))
    bad = np.argwhere(err > 1e-3*np.abs(ref).max())
    rows = sorted(set(bad[:,0].tolist())); cols = sorted(set(bad[:,1].tolist()))
    print(tag, 'rel %.2e' % rel_err(y, ref), 'nbad', len(bad), 'rows', rows[:20], 'cols', cols[:40])
f = load_golden('pack_w4g128.npz')
run('pack_w4g128 M5', f['x'], f)
run('pack_w4g128 M1', f['x'][:1], f)
for M in (1, 5, 16, 17, 33):
    for (K,N) in ((256,128),(1024,256)):
        L = make_random_layer(4, 128, K, N, seed=1)
        x = np.random.default_rng(M).standard_normal((M, K)).astype(np.float16)
        run('rand4 K%d N%d M%d' % (K,N,M), x, L)
# x = ones: isolates sums
L = make_random_layer(4, 128, 256, 128, seed=1)
run('ones M1', np.ones((1,256),np.float16), L)
e = np.zeros((1,256),np.float16); e[0,0]=1
run('e0 M1', e, L)
e = np.zeros((1,256),np.float16); e[0,5]=1
run('e5 M1', e, L)
