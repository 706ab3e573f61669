"""MatvecChain -- n dependent batch-1 QuantLinear ops in ONE persistent launch (csrc/chain.hip).

The reference issues one Triton launch per ``QuantLinear.forward`` (quant/quant_linear.py:373-377); at
decode every one of them waits for the previous one's output (qkv -> o_proj, fused_attn.py:117-161;
gate/up -> down_proj, fused_mlp.py:203-218; RMSNorm in front of both blocks, triton_norm.py:50-67).
``MatvecChain`` describes such a sequence once and replays it with ``gptq_chain_run_f16``: the weights of
op i+1 stream while op i is still being combined and published, so HBM does not idle between ops.

    chain = MatvecChain(bits=4, groupsize=128, device=x.device)
    chain.add(x=h, qweight=..., scales=..., qzeros=..., y=qkv, norm_weight=ln1.weight, norm_eps=1e-6)
    chain.add(x=attn_out, ..., y=h, residual=h)                       # o_proj + residual, in place
    chain.add(x=h, ..., up=(qweight_u, scales_u, qzeros_u), y=act, norm_weight=ln2.weight, norm_eps=1e-6)
    chain.add(x=act, ..., y=h, residual=h)
    chain.finalize(); chain.run()                                     # capturable in a hipGraph
"""
import ctypes

import torch

from . import _native


class MatvecChain:
    def __init__(self, bits, groupsize, device):
        self.bits, self.groupsize = int(bits), int(groupsize)
        self.device = torch.device(device)
        self.ops = []
        self._keep = []      # tensors addressed by the device image
        self.state = None
        self.max_k = 0

    def add(self, x, qweight, scales, qzeros, y, up=None, residual=None, norm_weight=None, norm_eps=1e-6, groupsize=None):
        """One op: y = [residual +] fp16(deq(W) applied to [rmsnorm](x)); ``up=(qweight, scales, qzeros)`` makes it
        the fused silu(x.Wg)*(x.Wu) of QuantLlamaMLP.  All tensors fp16/int32 on ``device``, contiguous."""
        if self.state is not None:
            raise RuntimeError('MatvecChain: already finalized')
        tensors = [x, qweight, scales, qzeros, y] + (list(up) if up is not None else []) + [t for t in (residual, norm_weight) if t is not None]
        for t in tensors:
            _native.require_device(t, 'MatvecChain.add')
            if not t.is_contiguous():
                raise RuntimeError('MatvecChain.add: tensors must be contiguous')
        K, N = qweight.shape[0] * 32 // self.bits, qweight.shape[1]
        if x.numel() != K or y.numel() != N:
            raise RuntimeError('MatvecChain.add: x has %d elements (K = %d), y has %d (N = %d)' % (x.numel(), K, y.numel(), N))
        op = _native.ChainOp()
        op.x, op.qweight, op.scales, op.qzeros, op.y = x.data_ptr(), qweight.data_ptr(), scales.data_ptr(), qzeros.data_ptr(), y.data_ptr()
        if up is not None:
            op.qweight_up, op.scales_up, op.qzeros_up = (t.data_ptr() for t in up)
        op.residual = residual.data_ptr() if residual is not None else None
        op.norm_weight = norm_weight.data_ptr() if norm_weight is not None else None
        op.norm_eps = float(norm_eps)
        gs = self.groupsize if groupsize is None else int(groupsize)
        op.K, op.N, op.groupsize = K, N, (K if gs <= 0 else gs)
        self.ops.append(op)
        self._keep.extend(tensors)
        self.max_k = max(self.max_k, K)
        return self

    def finalize(self):
        L = _native.lib()
        n = len(self.ops)
        if n == 0:
            raise RuntimeError('MatvecChain: empty chain')
        with torch.cuda.device(self.device):
            self.nwg = L.gptq_query(5)
        nbytes = L.gptq_chain_state_bytes(n)
        host = (ctypes.c_ubyte * nbytes)()
        arr = (_native.ChainOp * n)(*self.ops)
        _native.check(L.gptq_chain_encode(ctypes.cast(arr, ctypes.c_void_p), n, self.bits, self.nwg, ctypes.cast(host, ctypes.c_void_p), nbytes),
                      'gptq_chain_encode')
        self.state = torch.frombuffer(host, dtype=torch.uint8).clone().to(self.device)
        self._status_off = L.gptq_chain_status_offset(n)
        self.ws = _native.workspace(self.device)
        return self

    def run(self, timeline=False):
        """Enqueue the chain on the current stream.  timeline=True records per-(op, workgroup) stamps (development)."""
        if self.state is None:
            self.finalize()
        L = _native.lib()
        with torch.cuda.device(self.device):
            if timeline:
                self._tl = torch.zeros((len(self.ops), self.nwg, 16), dtype=torch.int64, device=self.device)
                prev = L.gptq_set_debug_buffer(self._tl.data_ptr())
            rc = L.gptq_chain_run_f16(self.state.data_ptr(), len(self.ops), self.bits, self.max_k, self.nwg, 1 if timeline else 0,
                                      self.ws.data_ptr(), self.ws.numel(), _native.stream_ptr(self.device))
            if timeline:
                L.gptq_set_debug_buffer(prev)
        _native.check(rc, 'gptq_chain_run_f16')

    def status(self):
        """0 unless a bounded wait expired during the last run (synchronises)."""
        return int(self.state[self._status_off:self._status_off + 4].view(torch.int32).item())

    def timeline(self):
        """[n_ops, workgroups, 16] int64 s_memrealtime ticks (100 MHz) of the last run(timeline=True); slots: csrc/chain.hip."""
        return self._tl.cpu()
