"""One-shot all-reduce over HIP IPC peer mappings (csrc/p2p.hip) for the row-sharded layout of BASELINE config 5.

``P2PAllReduce(n_max)`` is collective: every rank of the process group allocates its exchange buffer through the C ABI,
the 64-byte IPC handles travel through ``torch.distributed.all_gather_object`` (any backend), and every rank maps the
peers' buffers.  ``allreduce(partial_f32, out=...)`` then is ONE kernel launch per rank on the current stream: write the
partial into every peer, flag, sum the slots in rank order, round to fp16 once (+ bias).  One process per GPU; two
processes sharing one GPU work as well (the test box).  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC) in the environment the
PROCESS STARTED WITH (the image exports it; setting it after HIP initialised has no effect, so the constructor only checks).
The RCCL path (``torch.distributed.all_reduce``) stays the default of ``tensor_parallel``; this is the latency-optimised
alternative for 32-176 KB messages.

Failure reporting: a workgroup that gives up on a peer stores NaN instead of its sums and sets a status word; ``status()``
reads it (host sync), ``self_check()`` -- run once by the constructor unless GPTQ_P2P_SELFCHECK=0 -- all-reduces known integer
patterns through both slot sets and raises when a single element is off (first line of defence on a new topology: the
protocol has only ever crossed a real xGMI link on the driver's 8-GPU tier)."""
import ctypes
import os

import torch

from . import _native


class P2PAllReduce:

    def __init__(self, n_max, group=None, device=None):
        import torch.distributed as dist
        if os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') != '0':     # absent counts as wrong: the runtime's default is the legacy mode
            raise RuntimeError('P2PAllReduce: HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC) must be in the environment when the process starts '
                               '(found %r); the peer mappings cannot be opened otherwise' % os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))
        self.lib = _native.lib()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n_max = (int(n_max) + 3) // 4 * 4
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(self.device):
            own, handle = ctypes.c_void_p(), (ctypes.c_ubyte * 64)()
            _native.check(self.lib.gptq_p2p_create(self.world, self.n_max, ctypes.byref(own), handle), 'gptq_p2p_create')
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(handle), group=group)
            self._own = own.value
            self._opened = []
            ptrs = []
            for r, h in enumerate(handles):
                if r == self.rank:
                    ptrs.append(own.value)
                    continue
                p = ctypes.c_void_p()
                buf = (ctypes.c_ubyte * 64).from_buffer_copy(h)
                _native.check(self.lib.gptq_p2p_open(buf, ctypes.byref(p)), 'gptq_p2p_open')
                self._opened.append(p.value)
                ptrs.append(p.value)
        self._peers = (ctypes.c_void_p * self.world)(*ptrs)     # HOST array, read at launch time
        dist.barrier(group=group)                                 # every rank has mapped every buffer before the first call
        if os.environ.get('GPTQ_P2P_SELFCHECK', '1') != '0':
            self.self_check()

    def self_check(self, rounds=4):
        """collective: all-reduce rank-dependent integer patterns (exact in fp32) over the full exchange width, ``rounds`` times (both
        slot parities twice), and compare EVERY element with the closed form on the host.  Raises on any mismatch or status word."""
        import torch.distributed as dist
        n = self.n_max
        idx = torch.arange(n, device=self.device, dtype=torch.float32)
        for r in range(rounds):
            base = (idx % 1021.0) + float(r + 1)                       # < 2^11: (world (world + 1) / 2) * base is exact in fp32
            part = base * float(self.rank + 1)
            out = torch.empty(n, dtype=torch.float32, device=self.device)
            self.allreduce(part, out=out)
            expect = base * float(self.world * (self.world + 1) // 2)
            bad = int((out != expect).sum().item())
            st = self.status()
            if bad or st:
                raise RuntimeError('P2PAllReduce self-check failed on rank %d (round %d): %d of %d elements wrong, status word %d '
                                   '(1 + the rank a workgroup gave up waiting for)' % (self.rank, r, bad, n, st))
        dist.barrier(group=self.group)

    def allreduce(self, partial, out=None, bias=None):
        """partial: fp32 [n] (or [.., n], flattened) on this rank.  out: fp16 tensor (rounded sum, + bias) or fp32 tensor
        (the exact sum); allocated as fp16 when omitted.  Returns out."""
        part = partial.reshape(-1)
        if part.dtype != torch.float32 or not part.is_contiguous():
            part = part.float().contiguous()
        n = part.numel()
        if bias is not None and bias.numel() != n:
            # the kernel indexes the bias over the FLATTENED length: a [M, N] partial with an [N] bias would read past its end
            raise RuntimeError('P2PAllReduce.allreduce: bias has %d elements, the partial %d (M > 1 with a bias: add it after the reduce)' % (bias.numel(), n))
        if out is None:
            out = torch.empty(partial.shape, dtype=torch.float16, device=part.device)
        y16 = out.data_ptr() if out.dtype == torch.float16 else None
        y32 = out.data_ptr() if out.dtype == torch.float32 else None
        rc = self.lib.gptq_p2p_allreduce_f32(part.data_ptr(), self._peers, self.rank, self.world, n, self.n_max, y16, y32, _native.ptr(bias),
                                             _native.stream_ptr(part.device))
        _native.check(rc, 'gptq_p2p_allreduce_f32')
        return out

    def allreduce_silu_mul(self, partial_pair, out=None):
        """partial_pair: fp32 [2, n] (gate | up partials of a K-sharded fused MLP) -> fp16 [1, n] = silu(sum gate) * sum up."""
        assert partial_pair.dtype == torch.float32 and partial_pair.is_contiguous() and partial_pair.shape[0] == 2
        n = partial_pair.shape[-1]
        if out is None:
            out = torch.empty((1, n), dtype=torch.float16, device=partial_pair.device)
        rc = self.lib.gptq_p2p_allreduce_silu_mul_f32(partial_pair.data_ptr(), self._peers, self.rank, self.world, n, self.n_max, out.data_ptr(),
                                                      _native.stream_ptr(partial_pair.device))
        _native.check(rc, 'gptq_p2p_allreduce_silu_mul_f32')
        return out

    def status(self):
        return self.lib.gptq_p2p_status(self._own, self.world, self.n_max, _native.stream_ptr(self.device))

    def close(self):
        for p in self._opened:
            self.lib.gptq_p2p_close(p, 1)
        self._opened = []
        if self._own:
            self.lib.gptq_p2p_close(self._own, 0)
            self._own = None
