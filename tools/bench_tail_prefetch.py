"""Tail prefetch (gptq_layer_forward_next, stripe_kernel.inc PF): the LLaMA-7B decode pass of bench.py with every launch pulling the
first KIB KiB of the next op's stripes into the consuming XCD's L2.  Prints us / pass and GB/s per head size and checks that the
outputs are bit-identical to the plain pass."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def time_pass(work, reps=20):
    for _ in range(2):
        work.step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        work.step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main():
    dev = 'cuda:0'
    work = bench.DecodeLinears(dev)
    ref = None
    kibs = [int(v) for v in (sys.argv[1:] or ['0', '1', '2', '4', '8', '16', '32', '0'])]
    for kib in kibs:
        work.prefetch_kib = kib
        us = time_pass(work)
        outs = torch.cat([work.y_qkv.flatten(), work.y_h.flatten(), work.y_i.flatten()]).clone()
        if ref is None:
            ref = outs
        same = bool(torch.equal(outs, ref))
        print('head %3d KiB  %8.1f us/pass  %6.0f GB/s  %.4f of 8 TB/s  bit-identical %s' % (kib, us, work.bytes_per_step / us / 1e3,
                                                                                            work.bytes_per_step / us / 1e3 / 8000.0, same), flush=True)


if __name__ == '__main__':
    main()
