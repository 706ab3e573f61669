"""Row-sharded (K-split) and column-sharded (N-split) QuantLinear over torch.distributed.

One process per GPU; the backend is whatever the process group was created with ("nccl" = RCCL
over xGMI on MI355X, "gloo" in the CPU tests).  This is new functionality -- the reference has
no collective anywhere (its multi-GPU mode is layer placement, llama.py:328-382); BASELINE
config 5 asks for "row-sharded linears ... with a single RCCL all-reduce per layer".

Row sharding:  y = sum_r x[:, K_r] . deq(B[K_r, :])   -> each rank computes a full-width partial
               from ITS slice of qweight rows (plus the scales/qzeros rows of the groups it
               touches), then ONE all-reduce(sum) of the [M, N] partial.
Column sharding: y[:, N_r] = x . deq(B[:, N_r])       -> no reduction, one all-gather.

Cuts are made on quantisation-group boundaries (and on 32-row / 32-column packing boundaries),
so shards are plain slices of the checkpoint tensors; uneven group counts are allowed
(LLaMA-65B down_proj: 172 groups over 8 ranks -> 22,22,22,22,21,21,21,21).
With act-order (non-trivial g_idx) a K-shard may reference any group, so it keeps all scales /
qzeros rows and its slice of g_idx.
"""
import math

import torch
import torch.nn as nn

from .quant_linear import QuantLinear, g_idx_is_trivial, matmul248


def split_counts(total, world):
    base, rem = divmod(total, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def row_shard_bounds(K, groupsize, bits, world):
    """[(k0, k1)] per rank, cut on group boundaries (groupsize must be a multiple of 32)."""
    gs = K if groupsize == -1 else groupsize
    unit = gs if (gs % 32 == 0 and K % gs == 0) else 32
    counts = split_counts(K // unit, world)
    bounds, k = [], 0
    for c in counts:
        bounds.append((k, k + c * unit))
        k += c * unit
    assert k == K
    return bounds


def col_shard_bounds(N, world):
    counts = split_counts(N // 32, world)
    bounds, n = [], 0
    for c in counts:
        bounds.append((n, n + 32 * c))
        n += 32 * c
    return bounds


def shard_rows(layer, rank, world, trivial_g_idx=True):
    """Buffers of rank's K-shard of a QuantLinear as a new QuantLinear (views, no copies)."""
    K, N, bits, gs = layer.infeatures, layer.outfeatures, layer.bits, layer.groupsize
    k0, k1 = row_shard_bounds(K, gs, bits, world)[rank]
    f = 32 // bits if bits != 3 else None
    r0, r1 = (k0 // 32 * bits, k1 // 32 * bits)
    shard = QuantLinear(bits, gs if gs < K else (k1 - k0), max(k1 - k0, 32), N, layer.bias is not None and rank == 0)
    shard.infeatures = k1 - k0
    shard.qweight = layer.qweight[r0:r1]
    if trivial_g_idx and gs < K and k0 % gs == 0:
        g0, g1 = k0 // gs, math.ceil(k1 / gs)
        shard.qzeros = layer.qzeros[g0:g1]
        shard.scales = layer.scales[g0:g1]
        shard.g_idx = layer.g_idx[k0:k1] - g0
        shard.groupsize = gs
    else:
        shard.qzeros = layer.qzeros
        shard.scales = layer.scales
        shard.g_idx = layer.g_idx[k0:k1]
        shard.groupsize = gs if gs < K else K
    if layer.bias is not None and rank == 0:
        shard.bias = layer.bias       # added once, by rank 0's partial
    else:
        shard.bias = None
    return shard, (k0, k1)


def shard_cols(layer, rank, world):
    K, N, bits, gs = layer.infeatures, layer.outfeatures, layer.bits, layer.groupsize
    n0, n1 = col_shard_bounds(N, world)[rank]
    shard = QuantLinear(bits, gs, K, max(n1 - n0, 32), layer.bias is not None)
    shard.outfeatures = n1 - n0
    shard.qweight = layer.qweight[:, n0:n1].contiguous()
    shard.qzeros = layer.qzeros[:, n0 // 32 * bits:n1 // 32 * bits].contiguous()
    shard.scales = layer.scales[:, n0:n1].contiguous()
    shard.g_idx = layer.g_idx
    shard.bias = layer.bias[n0:n1].contiguous() if layer.bias is not None else None
    return shard, (n0, n1)


def _default_matmul(x2, s):
    return matmul248(x2, s.qweight, s.scales, s.qzeros, s.g_idx, s.bits, s.maxq, bias=s.bias)


def _default_partial(x2, s):
    """fp32 partial [M, N] of a K-shard: the stripe16 decode kernel stores its fp32 sums unrounded for up to four rows
    (gptq_stripe_matvec_partial_f32 at M == 1, gptq_stripe_matmul_partial_f32 for 2..4 rows -- an act-order shard gathers x[:, perm]
    first); larger batches go through the fp16 kernels (one extra rounding per shard)."""
    from . import _native
    from .quant_linear import _as_rows, _int32c, act_order_sorted, perm_u16, stripe_copy
    M = x2.shape[0]
    if 1 <= M <= 4 and s.bits in (2, 3, 4, 8) and x2.is_cuda:
        K, N = s.qweight.shape[0] * 32 // s.bits, s.qweight.shape[1]
        gs = s.groupsize if s.groupsize != -1 else K
        qw, perm = _int32c(s.qweight), None
        if not g_idx_is_trivial(s.g_idx, K, gs):
            srt = act_order_sorted(qw, _int32c(s.g_idx[:K]), K, gs, s.bits)
            qw, perm = (srt if srt is not None else (None, None))
        st = stripe_copy(qw, s.scales, _int32c(s.qzeros), s.bits, gs) if qw is not None else None
        if st is not None:
            lib = _native.lib()
            with torch.cuda.device(x2.device):
                part = torch.empty((M, N), dtype=torch.float32, device=x2.device)
                if M == 1:
                    x = _as_rows(x2)
                    rc = lib.gptq_stripe_matvec_partial_f32(x.data_ptr(), st.data_ptr(), st.numel(), part.data_ptr(), K, N, s.bits, gs, 1,
                                                            _native.ptr(perm_u16(perm)), _native.stream_ptr(x.device))
                else:
                    x = _as_rows(x2 if perm is None else x2[:, perm.long()])
                    rc = lib.gptq_stripe_matmul_partial_f32(x.data_ptr(), x.stride(0), st.data_ptr(), st.numel(), part.data_ptr(), M, K, N, s.bits, gs, 1,
                                                            _native.stream_ptr(x.device))
            if rc == 0:
                return part
            if rc != -6:    # GPTQ_E_VARIANT: the row groups do not fit (K too long for four rows of x in LDS) -> the fp16 kernels below
                _native.check(rc, 'gptq_stripe_matmul_partial_f32')
    return matmul248(x2, s.qweight, s.scales, s.qzeros, s.g_idx, s.bits, s.maxq).float()


class RowShardedQuantLinear(nn.Module):
    """This rank's K-slice of a QuantLinear + ONE all-reduce per forward (BASELINE config 5).  The partial sums travel in
    fp32 and are rounded to fp16 once, after the reduce -- the same rounding order as the unsharded layer (fp16 partials
    would add one rounding per shard: ~2e-3 instead of the 1e-3 budget); the bias is added after that, as in
    QuantLinear.forward (reference quant_linear.py:376)."""

    def __init__(self, layer, rank=None, world=None, group=None, matmul_fn=None, p2p=None):
        """p2p: a quant.p2p.P2PAllReduce shared by the layers of this rank -> the one-shot all-reduce over IPC peer mappings
        (one launch: push, flag, local sum, fp16 rounding + bias fused) instead of torch.distributed.all_reduce."""
        super().__init__()
        import torch.distributed as dist
        self.group = group
        self.p2p = p2p
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        trivial = g_idx_is_trivial(layer.g_idx, layer.infeatures, layer.groupsize)   # act-order shards keep every group
        self.shard, (self.k0, self.k1) = shard_rows(layer, self.rank, self.world, trivial_g_idx=trivial)
        self.shard.bias = None              # added once, after the reduce
        self.bias = layer.bias
        self.outfeatures = layer.outfeatures
        self.infeatures = layer.infeatures
        self._partial = matmul_fn or _default_partial

    def forward(self, x):
        import torch.distributed as dist
        out_shape = x.shape[:-1] + (self.outfeatures, )
        x2 = x.reshape(-1, x.shape[-1])[:, self.k0:self.k1]
        part = self._partial(x2, self.shard).float()
        if self.world > 1 and self.p2p is not None and part.is_cuda and part.numel() % 4 == 0 and part.numel() <= self.p2p.n_max:
            if self.bias is None or part.shape[0] == 1:       # the kernel's bias epilogue indexes the flattened vector: one row only
                return self.p2p.allreduce(part, bias=self.bias).reshape(out_shape)
            return (self.p2p.allreduce(part) + self.bias).reshape(out_shape)
        if self.world > 1:
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
        y = part.half()
        if self.bias is not None:
            y = y + self.bias
        return y.reshape(out_shape)


class ColShardedQuantLinear(nn.Module):
    """This rank's N-slice of a QuantLinear; ``gather=True`` all-gathers the full output."""

    def __init__(self, layer, rank=None, world=None, group=None, gather=True, matmul_fn=None):
        super().__init__()
        import torch.distributed as dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.shard, (self.n0, self.n1) = shard_cols(layer, self.rank, self.world)
        self.bounds = col_shard_bounds(layer.outfeatures, self.world)
        self.outfeatures = layer.outfeatures
        self.gather = gather
        self._matmul = matmul_fn or _default_matmul

    def forward(self, x):
        import torch.distributed as dist
        x2 = x.reshape(-1, x.shape[-1])
        part = self._matmul(x2, self.shard)
        if not self.gather or self.world == 1:
            return part.reshape(x.shape[:-1] + (part.shape[-1], ))
        pieces = [torch.empty((x2.shape[0], b - a), dtype=part.dtype, device=part.device) for a, b in self.bounds]
        dist.all_gather(pieces, part.contiguous(), group=self.group)
        return torch.cat(pieces, dim=1).reshape(x.shape[:-1] + (self.outfeatures, ))


def shard_linears_rowwise(model, group=None, matmul_fn=None):
    """Replace every QuantLinear of ``model`` by its row-sharded counterpart (BASELINE config 5)."""
    for name, m in list(model.named_modules()):
        if not isinstance(m, QuantLinear):
            continue
        sharded = RowShardedQuantLinear(m, group=group, matmul_fn=matmul_fn)
        if '.' in name:
            parent_name, child_name = name.rsplit('.', 1)
            parent = model.get_submodule(parent_name)
        else:
            parent, child_name = model, name
        setattr(parent, child_name, sharded)
    return model
