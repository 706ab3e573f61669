"""A/B of the tile GEMM's workgroup tile (csrc/gemm8.hip, round 5): 192-row against 256-row tiles, the per-launch choice and hipBLASLt on the
LLaMA-7B shapes at prompt sizes above the fused image route (2049 .. 4096 rows).  Same harness as bench.py's prompt leg (hipGraph of 8 calls,
median of 5).  Usage: python tools/bench_gemm8_tile.py [M ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'gptq-for-llama_amd')]
import torch

import bench
from quant import _native, layer as QLayer


def timed(fn, calls=8):
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(calls):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / calls)
    return sorted(ts)[2]


def main():
    dev = 'cuda:0'
    lib = _native.lib()
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    Ms = [int(a) for a in sys.argv[1:]] or [2304, 2560, 3072, 3584, 4096]
    H, I = bench.HIDDEN, bench.INTER
    lib.gptq_set_stripe_gemm_max_rows(0)
    for K, N, pair in [(H, H, False), (H, 3 * H, False), (I, H, False), (H, I, True)]:
        sets = tuple((w.qweight, w.scales, w.qzeros, None) for w in (bench.PackedSet(K, N, dev, gen), bench.PackedSet(K, N, dev, gen))[:2 if pair else 1])
        pl = QLayer.PreparedLayer(sets, None, bench.BITS, bench.GS, K, N)
        for M in Ms:
            x = torch.randn((M, K), device=dev, generator=gen).half()
            y = torch.empty((M, N), dtype=torch.float16, device=dev)
            t = {}
            for tile in (256, 192, 0):
                lib.gptq_set_gemm8_tile(tile)
                t[tile] = timed(lambda: pl.forward(x, y))
            lib.gptq_set_gemm8_tile(0)
            prev = lib.gptq_set_prefill_route(0)
            t_lib = timed(lambda: pl.forward(x, y))
            lib.gptq_set_prefill_route(prev)
            fl = (4.0 if pair else 2.0) * M * N * K
            print('%s%dx%d M=%-5d tile256 %7.1f us  tile192 %7.1f us (%.3f of 256)  chosen %7.1f us %6.1f TF  library %7.1f us  chosen vs library %.3f'
                  % ('pair ' if pair else '', K, N, M, t[256], t[192], t[192] / t[256], t[0], fl / t[0] / 1e6, t_lib, t_lib / t[0]), flush=True)
        del pl, sets


if __name__ == '__main__':
    main()
