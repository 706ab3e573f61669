"""CPU-side tests (run with -m "not gpu"): the C-ABI library loads and exports every symbol the
header declares (no compute without a GPU), argument validation and error mapping, the host logic
of the drop-in package (buffers / checkpoint keys / packer / module surgery / no CPU fallback), and
the row-/column-sharded linears over torch.distributed with the gloo backend, world_size 2 (the
shard's matmul is supplied by the CPU oracle there -- test infrastructure, never the product)."""
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

import quant
from quant import _native, tensor_parallel as TP
from quant import quant_linear as QL
from oracle import oracle
from util import golden_names, load_golden, make_random_layer, rel_err, TOL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'gptq_mi355x.h')).read()
    declared = sorted(set(re.findall(r'\b(gptq_[a-z0-9_]+)\s*\(', hdr)))
    assert len(declared) >= 18
    lib = _native.lib()
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert set(declared) == set(_native.EXPORTS), set(declared) ^ set(_native.EXPORTS)
    assert lib.gptq_query(0) == 1                       # ABI version
    assert lib.gptq_query(1) >= 1 and lib.gptq_query(2) >= 16
    assert lib.gptq_strerror(0) == b'ok'


def test_no_constant_tables_in_the_built_kernels():
    """build guard (round 4): a `constexpr` array indexed with a run-time value becomes a table in .rodata (`__const.<function>.<name>`) that the
    kernel reads with a global load -- and the s_waitcnt vmcnt(0) behind that load also waits for every weight block in flight.  That cost the
    2- and 3-bit decode kernels 6-13 % (stripe_unpack.inc pair_off, found in the disassembly).  No code object of the library may carry such a
    table; write selects / arithmetic instead."""
    path = _native.lib()._name
    blob = open(path, 'rb').read()
    assert blob.count(b'__const.') == 0, 'a constant table was emitted into %s: look for __const.* in the disassembly of the new kernel' % path


def test_argument_validation_needs_no_gpu():
    lib = _native.lib()
    one = 16
    # bits = 5 -> GPTQ_E_BITS before anything is touched (reference quant_linear.py:308-309)
    assert lib.gptq_matmul248_f16(one, 64, one, one, one, None, None, one, 64, 1, 64, 64, 5, 64, None, 0, None) == -1
    assert lib.gptq_matmul248_f16(one, 64, one, one, one, None, None, one, 64, 1, 48, 64, 4, 64, None, 0, None) == -2   # K % 32
    assert lib.gptq_matmul248_f16(None, 64, one, one, one, None, None, one, 64, 1, 64, 64, 4, 64, None, 0, None) == -4  # NULL x
    assert lib.gptq_matmul248_f16(2, 64, one, one, one, None, None, one, 64, 1, 64, 64, 4, 64, None, 0, None) == -3     # alignment
    assert lib.gptq_rmsnorm_f16(one, 40000, one, one, 40000, 1, 40000, 1e-6, None) == -7              # row > 64 KiB
    # the two kernels of the prefill route (dequantise with a leading dimension, silu * mul pass)
    assert lib.gptq_dequant_ld_f16(one, one, one, None, one, 64, 64, 64, 5, 64, None) == -1
    assert lib.gptq_dequant_ld_f16(one, one, one, None, one, 32, 64, 64, 4, 64, None) == -2          # ldw < N
    assert lib.gptq_dequant_ld_f16(one, one, one, None, None, 64, 64, 64, 4, 64, None) == -4
    assert lib.gptq_silu_mul_f16(one, 64, one, 64, one, 64, 0, 64, None) == 0                         # empty batch
    assert lib.gptq_silu_mul_f16(one, 64, one, 64, one, 60, 2, 64, None) == -2                        # ldc < N
    assert lib.gptq_silu_mul_f16(one, 68, one, 64, one, 64, 2, 64, None) == -2                        # ld % 8
    assert lib.gptq_silu_mul_f16(one, 64, None, 64, one, 64, 2, 64, None) == -4
    assert lib.gptq_silu_mul_f16(8, 64, one, 64, one, 64, 2, 64, None) == -3
    # prefill route behind the C ABI: validation order, workspace formula (weight + 76 MB for the library [+ the chunk product])
    assert lib.gptq_prefill_matmul_f16(one, 64, one, one, one, None, None, one, 64, 4, 64, 64, 5, 64, None, 0, None) == -1
    assert lib.gptq_prefill_matmul_f16(one, 32, one, one, one, None, None, one, 64, 4, 64, 64, 4, 64, None, 0, None) == -2      # ldx < K
    assert lib.gptq_prefill_matmul_f16(one, 64, None, one, one, None, None, one, 64, 4, 64, 64, 4, 64, None, 0, None) == -4
    assert lib.gptq_prefill_matmul_f16(one, 64, one, one, one, None, None, one, 64, 4, 64, 64, 4, 64, None, 0, None) == -5      # no workspace
    assert lib.gptq_prefill_matmul_f16(one, 64, one, one, one, None, None, one, 64, 0, 64, 64, 4, 64, None, 0, None) == 0       # empty batch
    assert lib.gptq_prefill_fused_mlp_f16(one, 64, one, one, one, None, one, one, one, None, one, 64, 4, 64, 64, 4, 64, None, 0, None) == -5
    assert lib.gptq_prefill_fused_mlp_f16(one, 64, one, one, one, None, None, one, one, None, one, 64, 4, 64, 64, 4, 64, None, 0, None) == -4
    assert lib.gptq_prefill_workspace_bytes(100, 4096, 4096, 1) == 4096 * 4096 * 2 + (76 << 20)
    assert lib.gptq_prefill_workspace_bytes(100, 4096, 11008, 2) == 4096 * 11008 * 4 + (76 << 20) + 100 * 2 * 11008 * 4   # fp32 gate | up chunk
    assert lib.gptq_prefill_workspace_bytes(70000, 4096, 11008, 2) == 4096 * 11008 * 4 + (76 << 20) + 16384 * 2 * 11008 * 2
    assert lib.gptq_strerror(-8).startswith(b'prefill route')
    with pytest.raises(NotImplementedError):
        _native.check(-1, 'x')
    with pytest.raises(RuntimeError):
        _native.check(-7, 'x')
    # (the two-launch path's records [heads][t_max / 128][130] are the larger need at batch 1; the streaming launch: test_attention_split_policy...)
    assert lib.gptq_decode_attn_workspace_bytes(32, 128, 2048) == 32 * 16 * 130 * 4
    assert lib.gptq_decode_attn_workspace_bytes(32, 64, 2048) == 0


def test_quantlinear_buffers_match_reference_checkpoint_format():
    """reference quant/quant_linear.py:306-323"""
    m = quant.QuantLinear(4, 128, 4096, 11008, False)
    sd = m.state_dict()
    assert list(sd) == ['qweight', 'qzeros', 'scales', 'g_idx']
    assert sd['qweight'].shape == (4096 // 32 * 4, 11008) and sd['qweight'].dtype == torch.int32
    assert sd['qzeros'].shape == (32, 11008 // 32 * 4) and sd['qzeros'].dtype == torch.int32
    assert sd['scales'].shape == (32, 11008) and sd['scales'].dtype == torch.float16
    assert sd['g_idx'].shape == (4096, ) and sd['g_idx'].dtype == torch.int32
    assert torch.equal(sd['g_idx'], (torch.arange(4096) // 128).to(torch.int32))
    assert (m.infeatures, m.outfeatures, m.bits, m.maxq, m.groupsize) == (4096, 11008, 4, 15, 128)
    assert quant.QuantLinear(4, -1, 256, 64, True).groupsize == 256
    assert 'bias' in quant.QuantLinear(8, 32, 64, 64, True).state_dict()
    with pytest.raises(NotImplementedError):
        quant.QuantLinear(5, 128, 128, 128, False)


@pytest.mark.parametrize('name', golden_names('pack_'))
def test_product_pack_is_bit_exact_with_reference_pack(name):
    """QuantLinear.pack (host path) against the buffers the reference's own pack() produced."""
    f = load_golden(name)
    N, K = f['weight_q'].shape
    bits, gs = int(f['bits']), int(f['groupsize'])
    has_bias = 'bias' in f and f['bias'].size == N
    lin = nn.Linear(K, N, bias=has_bias)
    lin.weight.data = torch.from_numpy(np.asarray(f['weight_q'], dtype=np.float32))
    if has_bias:
        lin.bias.data = torch.from_numpy(np.asarray(f['bias'], dtype=np.float32))
    q = quant.QuantLinear(bits, gs, K, N, has_bias)
    q.pack(lin, torch.from_numpy(np.asarray(f['scales_in'], dtype=np.float32)), torch.from_numpy(np.asarray(f['zeros_in'], dtype=np.float32)),
           torch.from_numpy(f['g_idx']))
    assert np.array_equal(q.qweight.numpy(), f['qweight'])
    assert np.array_equal(q.qzeros.numpy(), f['qzeros'])
    assert np.array_equal(q.scales.numpy().view(np.uint16), f['scales'].view(np.uint16))


def test_module_surgery_and_no_cpu_fallback():
    class Inner(nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = nn.Linear(128, 64, bias=True)
            self.keep = nn.Linear(64, 64)

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Linear(64, 128, bias=False)
            self.b = Inner()
    net = Block()
    names = {'a': net.a, 'b.proj': net.b.proj}
    quant.make_quant_linear(net, names, 4, 32)
    assert type(net.a) is quant.QuantLinear and type(net.b.proj) is quant.QuantLinear     # exact type, as find_layers needs
    assert type(net.b.keep) is nn.Linear
    assert net.b.proj.bias is not None and net.a.bias is None
    assert quant.make_quant is quant.make_quant_linear and quant.autotune_warmup is quant.autotune_warmup_linear
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net.a(torch.zeros(1, 64, dtype=torch.float16))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        quant.triton_norm.rms_norm(torch.zeros(1, 64, dtype=torch.float16), torch.ones(64, dtype=torch.float16), 1e-6)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'gptq-for-llama_amd')
    for d, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(d, fn)).read()
                assert 'oracle' not in src.replace('the oracle', '').replace("oracle's", '').replace('CPU oracle', '').replace('(oracle', '') \
                    or 'import oracle' not in src and 'from oracle' not in src, fn
                assert 'import oracle' not in src and 'from oracle' not in src and 'libgptq_oracle' not in src, fn


def test_shard_bounds():
    assert TP.row_shard_bounds(8192, 128, 4, 8) == [(i * 1024, (i + 1) * 1024) for i in range(8)]
    b = TP.row_shard_bounds(22016, 128, 4, 8)                      # 172 groups over 8 ranks
    assert [(k1 - k0) // 128 for k0, k1 in b] == [22, 22, 22, 22, 21, 21, 21, 21] and b[-1][1] == 22016
    assert TP.col_shard_bounds(11008, 8)[-1][1] == 11008 and all((n1 - n0) % 32 == 0 for n0, n1 in TP.col_shard_bounds(11008, 8))


# ---------------------------------------------------------------------------------- gloo, world 2
def _oracle_matmul(x2, s):
    y = oracle.matmul248(x2.numpy(), s.qweight.numpy(), s.scales.numpy(), s.qzeros.numpy(), s.g_idx.numpy(), s.bits,
                         bias=None if s.bias is None else s.bias.numpy())
    return torch.from_numpy(np.asarray(y))


def _oracle_partial_f32(x2, s):
    """fp32 partial of a K-shard (what gptq_stripe_matvec_partial_f32 hands to the all-reduce on the GPU)"""
    W = oracle.np_dequant(s.qweight.numpy(), s.qzeros.numpy(), s.scales.numpy(), s.g_idx.numpy(), s.bits, faithful=False)
    return torch.from_numpy((x2.numpy().astype(np.float32) @ W.astype(np.float32)).astype(np.float32))


def _layer_from(L, K, N, bias=None):
    q = quant.QuantLinear(int(L['bits']), int(L['groupsize']), K, N, bias is not None)
    q.qweight, q.qzeros, q.scales, q.g_idx = (torch.from_numpy(L[k]) for k in ('qweight', 'qzeros', 'scales', 'g_idx'))
    if bias is not None:
        q.bias = torch.from_numpy(bias)
    return q


def _tp_worker(rank, world, port, act_order, ret):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        K, N = 1024 + 128, 256                          # 9 groups over 2 ranks: uneven (5, 4)
        L = make_random_layer(4, 128, K, N, act_order=act_order, seed=5)
        bias = np.random.default_rng(1).standard_normal(N).astype(np.float16)
        x = torch.from_numpy(np.random.default_rng(2).standard_normal((3, K)).astype(np.float16))
        full = oracle.matmul248(x.numpy(), L['qweight'], L['scales'], L['qzeros'], L['g_idx'], 4, bias=bias)
        row = TP.RowShardedQuantLinear(_layer_from(L, K, N, bias), matmul_fn=_oracle_partial_f32)
        y_row = row(x).numpy()
        col = TP.ColShardedQuantLinear(_layer_from(L, K, N, bias), matmul_fn=_oracle_matmul)
        y_col = col(x).numpy()
        ok = rel_err(y_row, full) < TOL and rel_err(y_col, full) < TOL and (row.k0, row.k1) == [(0, 640), (640, 1152)][rank]
        t = torch.tensor([1 if ok else 0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            ret.put(int(t.item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('act_order', [False, True])
def test_row_and_col_sharded_linear_gloo_world2(act_order):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    ret = ctx.Queue()
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, port, act_order, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == 1


def test_stripe16_layout_restatement_is_a_bijection():
    """oracle.stripe16_repack (numpy restatement of csrc/stripe.hip's load-time repack) loses nothing: the
    checkpoint buffers come back bit for bit, for one set and for a gate/up pair, 2 / 4 / 8 bits, several group sizes."""
    for bits, K, N, gs, NS in [(4, 256, 32, 128, 1), (4, 1024, 288, 64, 2), (4, 384, 64, 32, 1), (4, 512, 64, -1, 2),
                               (8, 128, 32, 64, 1), (8, 1024, 96, 128, 2), (8, 192, 32, 16, 1), (2, 512, 32, 128, 1), (2, 1024, 64, 64, 2)]:
        Ls = [make_random_layer(bits, gs, K, N, seed=K + N + i) for i in range(NS)]
        img = oracle.stripe16_repack([(L['qweight'], L['scales'], L['qzeros']) for L in Ls], gs, bits)
        assert img.nbytes == _native.lib().gptq_stripe_bytes(K, N, bits, K if gs == -1 else gs, NS)
        back = oracle.stripe16_unpack(img, K, N, gs, NS, bits)
        for L, (qw, sc, z) in zip(Ls, back):
            assert np.array_equal(qw, L['qweight'])
            assert np.array_equal(sc.view(np.uint16), L['scales'].view(np.uint16))
            assert np.array_equal(z, oracle.np_unpack_cols(L['qzeros'], bits) + 1)
    # 3 bits: the fields of a random layer survive the three-words-per-32-k placement (ten fields per word + the spare bits 15 / 31)
    for K, N, gs, NS in [(128, 32, 128, 1), (512, 96, -1, 2), (1024, 32, 64, 1)]:
        Ls = [make_random_layer(3, gs, K, N, seed=3 * K + N + i) for i in range(NS)]
        img = oracle.stripe16_repack([(L['qweight'], L['scales'], L['qzeros']) for L in Ls], gs, 3)
        assert img.nbytes == _native.lib().gptq_stripe_bytes(K, N, 3, K if gs == -1 else gs, NS)
        for L, q in zip(Ls, oracle.stripe16_unpack3_fields(img, K, N, NS)):
            assert np.array_equal(q, oracle.np_unpack_rows(L['qweight'], 3))
    # ... and one hand-built block: field k holds k % 8 -> word 0 = k 0..9 (even k low half, odd k high half), bit 0 of k 30 / k 31 in bits 15 / 31
    f = np.arange(32, dtype=np.uint64) % 8
    stream = sum(int(v) << (3 * i) for i, v in enumerate(f))
    qw = np.zeros((12, 32), dtype=np.int32)              # 128 k x 32 columns, only column 0 / block 0 populated
    for j in range(3):
        qw[j, 0] = np.uint32((stream >> (32 * j)) & 0xFFFFFFFF).astype(np.int32)
    img = oracle.stripe16_repack([(qw, np.ones((1, 32), np.float16), np.zeros((1, 3), np.int32))], 128, 3)
    w0 = int(img[:4].view(np.uint32)[0])
    low = sum((2 * p % 8) << (3 * p) for p in range(5)) | ((30 % 8 & 1) << 15)
    high = sum(((2 * p + 1) % 8) << (3 * p) for p in range(5)) | ((31 % 8 & 1) << 15)
    assert w0 == (low | (high << 16)) == 0x9F590D10, hex(w0)     # low half 0,2,4,6,0 = 0x0D10; high half 1,3,5,7,1 + spare bit = 0x9F59
    # one hand-checked word per width: field k holds the value k -> even k in the low half-word, odd k in the high one
    for bits, word, want in [(4, 0x76543210, 0x75316420), (8, 0x03020100, 0x03010200), (2, 0xE4E4E4E4, 0xDDDD8888)]:
        qw = np.zeros((16, 16), dtype=np.int32)
        qw[0, 0] = np.uint32(word).astype(np.int32)
        K = 16 * 32 // bits
        img = oracle.stripe16_repack([(qw, np.ones((1, 16), np.float16), np.zeros((1, 16 * bits // 32 or 1), np.int32))], K, bits)
        assert int(img[:4].view(np.uint32)[0]) == want, (bits, hex(int(img[:4].view(np.uint32)[0])))


def test_stripe_abi_validation_needs_no_gpu():
    lib = _native.lib()
    one = 16
    assert lib.gptq_stripe_bytes(4096, 4096, 4, 128, 1) == 4096 // 8 * 4096 * 4 + 32 * 4096 * 4
    assert lib.gptq_stripe_bytes(4096, 11008, 4, 128, 2) == 2 * (4096 // 8 * 11008 * 4 + 32 * 11008 * 4)
    assert lib.gptq_stripe_bytes(4096, 4096, 8, 128, 1) == 4096 // 4 * 4096 * 4 + 32 * 4096 * 4
    assert lib.gptq_stripe_bytes(4096, 4096, 2, 128, 1) == 4096 // 16 * 4096 * 4 + 32 * 4096 * 4
    assert lib.gptq_stripe_bytes(4096, 4096, 3, 128, 1) == 4096 // 32 * 3 * 4096 * 4 + 32 * 4096 * 4   # three words per 32 k
    assert lib.gptq_stripe_bytes(4096, 4096, 3, 4096, 1) > 0 and lib.gptq_stripe_bytes(4096 + 64, 4096, 3, 128, 1) == 0
    assert lib.gptq_stripe_bytes(4096 + 64, 4096, 4, 128, 1) == 0     # K % 128 (4-bit row block)
    assert lib.gptq_stripe_bytes(4096 + 64, 4096, 8, 64, 1) > 0       # ... but a whole number of 8-bit row blocks (64 k)
    assert lib.gptq_stripe_bytes(4096, 4096, 4, 96, 1) == 0           # group not a power-of-two multiple of 32
    assert lib.gptq_stripe_bytes(4096, 4096, 2, 32, 1) == 0           # 2-bit: a lane block is 64 k
    assert lib.gptq_stripe_bytes(32768, 4096, 4, 128, 1) == 0         # more than 24 row blocks per wave
    assert lib.gptq_stripe_bytes(22016, 8192, 8, 128, 1) > 0           # 8-bit 65B down_proj: 43 row blocks per wave
    assert lib.gptq_stripe_bytes(4096, 4096, 4, 4096, 1) > 0          # one group
    nb = lib.gptq_stripe_bytes(256, 64, 4, 128, 1)
    assert lib.gptq_stripe_repack(one, one, one, None, None, None, one, nb, 256, 64, 5, 128, None) == -1
    assert lib.gptq_stripe_repack(one, one, one, None, None, None, one, nb - 1, 256, 64, 4, 128, None) == -5
    assert lib.gptq_stripe_repack(None, one, one, None, None, None, one, nb, 256, 64, 4, 128, None) == -4
    assert lib.gptq_stripe_repack(one, one, one, None, None, None, one, nb, 256, 64, 3, 96, None) == -6            # 3-bit groups: multiples of 32

    def mv(x=one, st=one, nbytes=nb, y=one, M=1, bits=4, nsets=1, norm=None, ldx=256, ldy=64):
        return lib.gptq_stripe_matvec_f16(x, ldx, st, nbytes, None, y, ldy, M, 256, 64, bits, 128, nsets, norm, 0.0, None, None)
    assert mv(nsets=3) == -2
    assert mv(nbytes=nb - 1) == -5
    assert mv(M=17) == -6                                             # at most sixteen rows per launch (gptq_stripe_matmul_f16 serves up to 64)
    # (round 5: the fused RMSNorm serves every row group -- one rstd per row -- so M = 2 with a norm weight is a launch, not a refusal)
    assert lib.gptq_stripe_matvec_f16(one, 256, one, nb, None, one, 64, 2, 256, 64, 4, 128, 1, None, 0.0, one, None) == -6    # a permutation stays M == 1
    assert mv(M=2, ldx=252) == -3
    assert mv(x=None) == -4
    assert mv(x=2) == -3
    assert mv(M=0) == 0

    # round 5: the entries of the batched decode engine validate before they touch anything
    assert lib.gptq_layer_decode_f16(None, one, 256, one, 64, 2, None, 0.0, None, 0, one, 1 << 30, None, 0, None) == -4
    assert lib.gptq_layer_decode_scratch_bytes(None, 4) == 0
    assert lib.gptq_dense_matmat_f16(one, 256, one, 256, None, one, 64, 17, 64, 256, None, 0.0, None) == -6          # at most 16 rows share the pass
    assert lib.gptq_dense_matmat_f16(one, 128, one, 256, None, one, 64, 2, 64, 256, None, 0.0, None) == -2          # ldx < K
    assert lib.gptq_dense_matmat_f16(one, 256, one, 256, None, one, 64, 0, 64, 256, None, 0.0, None) == 0           # empty batch
    assert lib.gptq_dense_matmat_f16(None, 256, one, 256, None, one, 64, 2, 64, 256, None, 0.0, None) == -4
    assert lib.gptq_add_rows_f16(one, 64, one, 32, 2, 64, None) == -2 and lib.gptq_add_rows_f16(one, 64, None, 64, 2, 64, None) == -4
    assert lib.gptq_add_rows_f16(one, 64, one, 64, 0, 64, None) == 0
    # round 6: the streaming launch keeps [batch][S][heads x 128] fp16 partial outputs + [batch][S][heads] fp32 {M, den} + one ticket per (row, head); S = splits of the
    # grid: heads x rows x S ~ 256 workgroups, at most 8, at most one per 128 tokens of the cache
    rec = lambda batch, heads, S: batch * (S * heads * (128 * 2 + 8) + heads * 4)
    assert [lib.gptq_decode_attn_splits(b, 32, 128, 2048) for b in (1, 2, 3, 4, 16)] == [8, 4, 2, 2, 1]
    assert lib.gptq_decode_attn_splits(1, 2, 128, 384) == 3 and lib.gptq_decode_attn_splits(1, 2, 128, 64) == 1
    assert lib.gptq_decode_attn_splits(1, 32, 64, 2048) == -2
    assert lib.gptq_decode_attn_batch_workspace_bytes(4, 32, 128, 2048) == rec(4, 32, 2) and lib.gptq_decode_attn_batch_workspace_bytes(16, 32, 128, 2048) == rec(16, 32, 1)
    assert lib.gptq_decode_attn_batch_workspace_bytes(2, 32, 128, 2048) == rec(2, 32, 4)
    assert lib.gptq_decode_attn_workspace_bytes(32, 128, 2048) == max(rec(1, 32, 8), 32 * 16 * 130 * 4)    # (the two-launch path's records at batch 1)
    assert lib.gptq_decode_attn_workspace_bytes(32, 128, 1024) == max(rec(1, 32, 8), 32 * 8 * 130 * 4)
    assert lib.gptq_decode_attn_split_f16(one, 3 * 256, one, one, one, one, 16, 2, 2, 128, 64, 10000.0, 1.0, None, 0, None) == -5          # workspace too small
    assert lib.gptq_decode_attn_split_f16(one, 3 * 256, one, one, one, None, 1 << 20, 2, 2, 128, 64, 10000.0, 1.0, None, 0, None) == -4
    assert lib.gptq_layer_decode_attn_f16(None, one, 1 << 20, one, 1, 2, 128, 64, 0, one, 64, None, 0, None) == -4
    assert lib.gptq_layer_decode_attn_supported(None, 1, 2, 128) == -4
    assert lib.gptq_decode_attn_batch_workspace_bytes(4, 32, 64, 2048) == 0
    assert lib.gptq_decode_attn_batch_f16(one, 3 * 256, one, one, one, one, 256, one, 16, 2, 2, 128, 64, 10000.0, 1.0, None, None, None) == -5    # workspace too small
    assert lib.gptq_decode_attn_batch_f16(one, 256, one, one, one, one, 256, one, 1 << 20, 2, 2, 128, 64, 10000.0, 1.0, None, None, None) == -2   # ldq < 3 * hidden
    assert lib.gptq_decode_attn_batch_f16(one, 3 * 256, None, one, one, one, 256, one, 1 << 20, 2, 2, 128, 64, 10000.0, 1.0, None, None, None) == -4
    assert lib.gptq_layer_inverse_perm(None, None) == -4
    assert lib.gptq_stripe_matvec_perm_out_f16(one, 256, one, nb, None, None, 64, 1, 256, 64, 4, 128, 1, None, 0.0, None, one, None) == -4    # NULL y

    # small-batch MFMA kernel on the same image (csrc/stripe_mm.inc): scratch workspace required, up to 256 rows
    def mm(x=one, st=one, nbytes=nb, y=one, M=16, bits=4, nsets=1, ws=256, wsb=1 << 20, ldx=256, ldy=64, gs=128):
        return lib.gptq_stripe_matmul_f16(x, ldx, st, nbytes, None, y, ldy, M, 256, 64, bits, gs, nsets, ws, wsb, None)
    assert lib.gptq_query(5) == 64 << 20
    assert mm(ws=None) == -4 and mm(y=None) == -4 and mm(x=None) == -4
    assert mm(ws=260) == -3 and mm(ldx=252) == -3
    assert mm(M=2049) == -6 and mm(bits=5) == -1                       # 129 .. 2048 rows: the fused tile GEMM (gptq_set_stripe_gemm_max_rows)
    prev = lib.gptq_set_stripe_gemm_max_rows(0)
    assert prev == 2048 and mm(M=257) == -6                            # without it: passes of 128 rows up to 256
    assert lib.gptq_set_stripe_gemm_max_rows(prev) == 0 and lib.gptq_set_stripe_gemm_max_rows(-1) == -6
    assert lib.gptq_set_stripe_mm_pass_rows(96) == -6                  # rows per pass of the 16-row tiles: 64 or 128
    assert lib.gptq_set_stripe_mm_pass_rows(64) == 128 and lib.gptq_set_stripe_mm_pass_rows(128) == 64
    assert mm(nbytes=nb - 1) == -5
    assert mm(M=0) == 0

    # one-shot all-reduce (csrc/p2p.hip): everything is validated before a launch
    import ctypes
    assert lib.gptq_p2p_buffer_bytes(8, 8192) == 2 * 8 * 8192 * 4 + 2 * 8 * 64 * 4 + 64 * 4 + 256
    assert lib.gptq_p2p_buffer_bytes(17, 8192) == 0 and lib.gptq_p2p_buffer_bytes(0, 8192) == 0 and lib.gptq_p2p_buffer_bytes(2, 8190) == 0
    peers = (ctypes.c_void_p * 2)(256, 256)

    def ar(part=256, pb=peers, rank=0, world=2, n=64, n_max=64, y16=256, y32=None):
        return lib.gptq_p2p_allreduce_f32(part, pb, rank, world, n, n_max, y16, y32, None, None)
    assert ar(part=None) == -4 and ar(y16=None) == -4 and ar(pb=None) == -4
    assert ar(pb=(ctypes.c_void_p * 2)(256, None)) == -4
    assert ar(rank=2) == -2 and ar(n=62) == -2 and ar(n=128) == -2 and ar(world=17) == -2
    assert ar(part=260) == -3 and ar(y16=260) == -3
    assert lib.gptq_p2p_create(2, 64, None, None) == -4 and lib.gptq_p2p_open(None, None) == -4 and lib.gptq_p2p_close(None, 0) == -4


def test_byte_model_matches_survey_8d():
    """the algorithmic-bytes figures `roofline.achieved` is computed from, in bench.py and in the oracle"""
    import bench
    from oracle import oracle
    assert bench.alg_bytes(1, 4096, 4096) == 8732672
    assert bench.alg_bytes(1, 4096, 11008) == 23455232 == bench.alg_bytes(1, 11008, 4096)
    assert bench.alg_bytes(1, 4096, 12288) == 26181632
    assert bench.alg_bytes(1, 4096, 11008, nsets=2) == 46880256      # two weight sets, x and y once
    per_layer = bench.alg_bytes(1, 4096, 12288) + bench.alg_bytes(1, 4096, 4096) + bench.alg_bytes(1, 4096, 11008, nsets=2) + bench.alg_bytes(1, 11008, 4096)
    assert 32 * per_layer == 3367993344
    assert oracle.algorithmic_bytes(1, 4096, 4096, 4, 128) == 8732672
    assert oracle.algorithmic_bytes(1, 4096, 4096, 4, 128, act_order=True) == 8732672 + 4 * 4096
    assert oracle.algorithmic_bytes(1, 4096, 4096, 3, 4096) == 6317568      # 3-bit, one group (config 4)


def test_tp_decode_shard_slicing_is_exact_on_cpu():
    """the column / row cuts quant/tp_decode.py feeds to stripe_copy (plain tensor slicing, no GPU): dequantising a shard gives the
    matching slice of the dequantised layer -- uneven group counts over 8 ranks, 4- and 3-bit packing of qzeros along N"""
    import torch
    from quant import tp_decode, tensor_parallel as TP
    for bits, gs, K, N in [(4, 128, 1408, 352), (3, 64, 1024, 256), (8, 32, 320, 96)]:
        L = make_random_layer(bits, gs, K, N, seed=bits + K)
        full = oracle.np_dequant(L['qweight'], L['qzeros'], L['scales'], L['g_idx'], bits)
        qw, sc, qz = (torch.from_numpy(L[k]) for k in ('qweight', 'scales', 'qzeros'))
        for world in (2, 8):
            for (k0, k1) in TP.row_shard_bounds(K, gs, bits, world):
                if k1 == k0:
                    continue
                a, b, c = tp_decode._rows(qw, sc, qz, bits, gs, k0, k1)
                gi = (np.arange(k1 - k0) // gs).astype(np.int32)
                assert np.array_equal(oracle.np_dequant(a.numpy(), c.numpy(), b.numpy(), gi, bits), full[k0:k1])
            for (n0, n1) in TP.col_shard_bounds(N, world):
                if n1 == n0:
                    continue
                a, b, c = (t.contiguous() for t in tp_decode._cols(qw, sc, qz, bits, n0, n1))
                assert np.array_equal(oracle.np_dequant(a.numpy(), c.numpy(), b.numpy(), L['g_idx'], bits), full[:, n0:n1])


def test_prefill_route_selection_is_host_logic(monkeypatch):
    """which engine a dense product takes is host logic in the C library (gptq_prefill_route_for): under 'auto' (and 'own') the tile GEMM of
    csrc/gemm8.hip wherever K % 128 == 0 -- the default route never reaches hipBLASLt for such shapes --, 'library' never.  A refused library (GPTQ_E_LIBRARY) warns once and lets the caller continue, other codes raise."""
    import warnings
    from quant import quant_linear as QL
    lib = _native.lib()
    prev = lib.gptq_set_prefill_route(1)
    try:
        rf = lib.gptq_prefill_route_for
        assert rf(65536, 4096, 4096, 1, 0) == 1 and rf(4096, 4096, 4096, 1, 0) == 1 and rf(4096, 4096, 11008, 2, 0) == 1
        assert rf(1024, 4096, 12288, 1, 0) == 1 and rf(256, 4096, 4096, 1, 0) == 1 and rf(2048, 4096, 4096, 1, 0) == 1     # round 4: own kernel below a full round of tiles too
        assert rf(65536, 4000 // 32 * 32 + 32, 4096, 1, 0) == 0                                                     # K % 128 != 0 -> library
        assert rf(65536, 4096, 4096, 1, 1) == 1 and rf(65536, 4096, 4128, 1, 1) == 0                               # backward: the reduction runs over N
        assert rf(0, 4096, 4096, 1, 0) == -2
        lib.gptq_set_prefill_route(0)
        assert rf(65536, 4096, 4096, 1, 0) == 0
        lib.gptq_set_prefill_route(2)
        assert rf(300, 4096, 4096, 1, 0) == 1 and rf(300, 4064, 4096, 1, 0) == 0
        assert lib.gptq_set_prefill_route(7) == -6
        assert lib.gptq_set_gemm8_tile(100) == -6 and lib.gptq_set_gemm8_tile(0) == 0      # rows of the tile GEMM's workgroup tile: 0 (per launch) / 192 / 256
        assert lib.gptq_set_gemm8_tile(192) == 0 and lib.gptq_set_gemm8_tile(0) == 192
    finally:
        lib.gptq_set_prefill_route(prev)
    assert lib.gptq_prefill_plan_count() == 0          # nothing planned without a GPU
    monkeypatch.setattr(QL, '_library_warned', False)
    with pytest.warns(UserWarning, match='falling back'):
        assert QL._library_refused(-8, 'gptq_prefill_matmul_f16') is True
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        assert QL._library_refused(-8, 'gptq_prefill_matmul_f16') is True          # second time: silent
    assert QL._library_refused(0, 'x') is False
    with pytest.raises(RuntimeError):
        QL._library_refused(-5, 'gptq_prefill_matmul_f16')
    assert QL.TRANSPOSE_LIBRARY_MIN_M >= 16


def test_prepared_layer_abi_host_logic():
    """gptq_layer_* (include/gptq_mi355x.h "Prepared layers"): the image size formula, argument validation and the handle itself are
    host logic -- a handle without an image (checkpoint-layout kernels only) is made and destroyed here without any device."""
    import ctypes
    lib = _native.lib()
    a256 = lambda v: (v + 255) & ~255
    K, N = 4096, 11008
    st1, st2 = lib.gptq_stripe_bytes(K, N, 4, 128, 1), lib.gptq_stripe_bytes(K, N, 4, 128, 2)
    assert st1 > 0 and lib.gptq_layer_image_bytes(K, N, 4, 128, 1, 0) == a256(st1) and lib.gptq_layer_image_bytes(K, N, 4, 128, 2, 0) == a256(st2)
    qw = (K // 8) * N * 4
    assert lib.gptq_layer_image_bytes(K, N, 4, 128, 1, 1) == a256(st1) + 2 * a256(4 * K) + a256(2 * K)          # + perm32, invperm32, perm16 (round 4: no sorted rows)
    assert lib.gptq_layer_image_bytes(K, N, 4, 128, 2, 1) == a256(st2) + 2 * a256(4 * K) + a256(2 * K)
    assert lib.gptq_layer_image_bytes(K, N, 4, 128, 1, 2) == 0                   # irregular g_idx: no derived copies
    st3 = lib.gptq_stripe_bytes(K, N, 3, 128, 1)
    assert st3 > 0 and lib.gptq_layer_image_bytes(K, N, 3, 128, 1, 1) == a256(st3) + 2 * a256(4 * K) + a256(2 * K)   # 3-bit act-order too (round 4)
    assert lib.gptq_layer_image_bytes(K, N, 3, 48, 1, 1) == 0                    # ... unless a group would split a 32-k block
    assert lib.gptq_layer_image_bytes(96, 64, 4, 32, 1, 0) == 0                  # K not a multiple of the row block: no stripe image
    assert lib.gptq_layer_image_bytes(K, N, 4, 128, 3, 0) == 0 and lib.gptq_layer_image_bytes(K, N, 4, 128, 1, 5) == 0
    assert lib.gptq_layer_workspace_bytes() == lib.gptq_query(3) + lib.gptq_query(5)
    one = 4096                          # a fake, aligned, non-NULL "device pointer": nothing is launched
    h = ctypes.c_void_p()
    assert lib.gptq_layer_prepare(ctypes.byref(h), one, one, one, None, None, None, None, None, None, K, N, 5, 128, None, 0, None) == -1
    assert lib.gptq_layer_prepare(ctypes.byref(h), one, one, one, None, None, None, None, None, None, K + 8, N, 4, 128, None, 0, None) == -2
    assert lib.gptq_layer_prepare(ctypes.byref(h), None, one, one, None, None, None, None, None, None, K, N, 4, 128, None, 0, None) == -4
    assert lib.gptq_layer_prepare(ctypes.byref(h), one + 2, one, one, None, None, None, None, None, None, K, N, 4, 128, None, 0, None) == -3
    assert lib.gptq_layer_prepare(ctypes.byref(h), one, one, one, None, None, one, None, None, None, K, N, 4, 128, None, 0, None) == -4   # pair without scales_up
    assert lib.gptq_layer_prepare(ctypes.byref(h), one, one, one, None, None, None, None, None, None, K, N, 4, 128, None, 0, None) == 0 and h.value
    try:
        assert lib.gptq_layer_kind(h) == 0
        st, nb, p16 = ctypes.c_void_p(1), ctypes.c_size_t(1), ctypes.c_void_p(1)
        assert lib.gptq_layer_stripe_image(h, ctypes.byref(st), ctypes.byref(nb), ctypes.byref(p16)) == 0 and not st.value and nb.value == 0 and not p16.value
        assert lib.gptq_layer_scratch_bytes(h, 1) == 0 and lib.gptq_layer_scratch_bytes(h, 64) == 0
        assert lib.gptq_layer_scratch_bytes(h, 65) == lib.gptq_prefill_workspace_bytes(65, K, N, 1)                  # no image: dense route from 65 rows
        ws = lib.gptq_layer_workspace_bytes()
        assert lib.gptq_layer_forward(h, one, K, one, N, 0, None, 0, None, 0, None) == 0                              # empty batch
        assert lib.gptq_layer_forward(h, one, K, one, N, -1, one, ws, None, 0, None) == -2
        assert lib.gptq_layer_forward(h, one, K - 8, one, N, 1, one, ws, None, 0, None) == -2                        # ldx < K
        assert lib.gptq_layer_forward(h, None, K, one, N, 1, one, ws, None, 0, None) == -4
        assert lib.gptq_layer_forward(h, one + 2, K, one, N, 1, one, ws, None, 0, None) == -3
        assert lib.gptq_layer_forward(h, one, K, one, N, 1, None, 0, None, 0, None) == -5                            # no workspace
        assert lib.gptq_layer_forward(h, one, K, one, N, 1, one, ws - 1, None, 0, None) == -5
    finally:
        lib.gptq_layer_destroy(h)
    assert lib.gptq_layer_forward(None, one, K, one, N, 1, one, 1 << 30, None, 0, None) == -4 and lib.gptq_layer_kind(None) == -4


def test_p2p_allreduce_protocol_simulation():
    """csrc/p2p.hip's epoch / parity protocol as a host-side model: P ranks run a sequence of all-reduces, a random
    scheduler advances ONE atomic action of ONE rank at a time (a slot write at one peer, a flag write at one peer, one poll, one slot
    read), i.e. every interleaving the device could produce under sequential consistency -- including a rank that races one whole
    collective ahead of a slow peer.  Claims checked: every rank reads exactly the partials of ITS epoch (two slot sets by epoch
    parity suffice: nobody can be two collectives ahead), every sum is the rank-ordered sum, and no rank ever blocks for ever."""
    import random
    for world in (2, 3, 8):
        for seed in range(6):
            rnd = random.Random(1000 * world + seed)
            calls = 7
            slots = [[[None] * world for _ in range(2)] for _ in range(world)]     # slots[peer][parity][source rank]
            flags = [[[0] * world for _ in range(2)] for _ in range(world)]        # flags[peer][parity][source rank] = epoch
            part = lambda r, e: (r + 1) * 1000 + e                                  # rank r's partial in its e-th collective
            # per rank: program counter over (epoch, phase, index); phases 0 push, 1 flag, 2 poll, 3 sum
            st = [dict(e=1, ph=0, i=0, acc=[], sums=[]) for _ in range(world)]
            steps = 0
            while any(s['e'] <= calls for s in st):
                steps += 1
                assert steps < 200000, 'protocol model dead-locked'
                r = rnd.randrange(world)
                if rnd.random() < 0.3:
                    r = 0                                                           # rank 0 is fast: it tries to run ahead
                s = st[r]
                if s['e'] > calls:
                    continue
                e, par = s['e'], s['e'] & 1
                if s['ph'] == 0:
                    slots[s['i']][par][r] = (e, part(r, e))                         # write-through store into peer i's slot set
                    s['i'] += 1
                elif s['ph'] == 1:
                    flags[s['i']][par][r] = e                                       # release store of the flag at peer i
                    s['i'] += 1
                elif s['ph'] == 2:
                    if flags[r][par][s['i']] != e:                                  # relaxed poll of ONE source rank's flag: not yet -> try again later
                        continue
                    s['i'] += 1
                else:
                    got = slots[r][par][s['i']]
                    assert got == (e, part(s['i'], e)), ('rank %d read a slot of another epoch' % r, got, e)
                    s['acc'].append(got[1])
                    s['i'] += 1
                if s['i'] == world:
                    s['i'] = 0
                    s['ph'] += 1
                    if s['ph'] == 4:
                        s['sums'].append(sum(s['acc']))
                        s['acc'], s['ph'], s['e'] = [], 0, e + 1
            for r in range(world):
                assert st[r]['sums'] == [sum(part(q, e) for q in range(world)) for e in range(1, calls + 1)]


def test_layer_route_table_is_host_logic():
    """gptq_layer_route_for_shape: the M -> kernel table of gptq_layer_forward (include/gptq_mi355x.h) as a host-only query -- what the
    reference's Autotuner decides per (M, N, K) at run time (quant/custom_autotune.py:76-102), readable and pinned here for the LLaMA-7B
    shapes of BASELINE.json (4-bit, g128)."""
    lib = _native.lib()
    DEC, TILES, SGEMM, OWN, LIBR, CKPT = 1, 2, 3, 4, 5, 6

    def route(M, K, N, bits=4, gs=128, nsets=1, kind=0, image=1):
        return lib.gptq_layer_route_for_shape(M, K, N, bits, gs, nsets, kind, image)
    for K, N, ns in [(4096, 4096, 1), (4096, 12288, 1), (11008, 4096, 1), (4096, 11008, 2)]:
        assert route(1, K, N, nsets=ns) == DEC and route(4, K, N, nsets=ns) == DEC
        # (round 6: 9 .. 16 rows of the gate | up pair stay in the decode launch -- sixteen A rows of its 16x16x16 inner product; single sets: 16-row tiles)
        assert route(16, K, N, nsets=ns) == (DEC if ns == 2 else TILES) and route(17, K, N, nsets=ns) == TILES and route(128, K, N, nsets=ns) == TILES
        assert route(129, K, N, nsets=ns) == SGEMM and route(640, K, N, nsets=ns) == SGEMM        # a prompt: weights stay packed ...
        # ... while the image route's 128 x 128 tiles fit the chip at once (512) or the batch is too short to pay for a dequantise pass; above, the
        # dense route is the faster own kernel since gemm8 balances its tiles over the XCDs (round 5, profiles/r5e_gemm8_tile/)
        wide = N * ns > 4096
        assert route(2048, K, N, nsets=ns) == (OWN if wide else SGEMM)
        assert route(768, K, N, nsets=ns) == (OWN if wide and ns == 1 else SGEMM) and route(1152, K, N, nsets=ns) == (OWN if wide and ns == 1 else SGEMM)
        assert route(1280, K, N, nsets=ns) == (OWN if wide else SGEMM)
        assert route(2049, K, N, nsets=ns) == OWN and route(4095, K, N, nsets=ns) == OWN          # above: dequantise per call + the own tile GEMM
        assert route(65536, K, N, nsets=ns) == OWN                                                # BASELINE config 3
        for M in (1, 5, 64, 129, 1025, 1536, 2048, 3072, 4095, 65536):
            assert route(M, K, N, nsets=ns) != LIBR and route(M, K, N, nsets=ns, image=0) != LIBR  # the library is off the default route (K % 128 == 0)
    # round 6: 5 .. 8 rows stay in the decode launch wherever its 16x16x16 inner product serves the layer (groups spanning a row block, not 2-bit);
    # before: row groups only while one round of workgroups covers N
    assert route(8, 4096, 4096) == DEC and route(8, 4096, 12288) == DEC and route(9, 4096, 12288) == TILES and route(8, 4096, 11008, nsets=2) == DEC
    assert route(8, 4096, 12288, gs=32) == TILES and route(8, 4096, 12288, bits=2) == TILES and route(8, 8192, 24576) == TILES
    assert route(8, 11008, 4096) == DEC and route(9, 11008, 4096) == TILES     # round 5: eight rows of x in two K halves (stripe_gemv2p_kernel); round 4: TILES
    assert route(8, 16384, 4096) == TILES                                       # ... up to K = 12288
    assert route(300, 4000 // 128 * 128 + 32, 4096, image=0) == LIBR             # K % 128 != 0: no own dense kernel -> the library
    # act-order: regular (group-sorted image + gather) like trivial; irregular: no image
    assert route(1, 4096, 4096, kind=1) == DEC and route(64, 4096, 4096, kind=1) == TILES and route(300, 4096, 4096, kind=1) == SGEMM
    assert route(1, 4096, 4096, kind=2) == CKPT and route(64, 4096, 4096, kind=2) == CKPT and route(65, 4096, 4096, kind=2) == OWN
    # groups smaller than a row block have tiles (prescale mode) but no fused GEMM; 2-bit likewise; no image at all
    assert route(100, 4096, 4096, gs=32) == TILES and route(300, 4096, 4096, gs=32) == OWN
    assert route(300, 4096, 4096, bits=2) == OWN and route(300, 4096, 4096, bits=8) == SGEMM and route(300, 4096, 4096, bits=3, gs=4096) == SGEMM
    assert route(1, 4096, 4096, image=0) == CKPT and route(64, 4096, 4096, image=0) == CKPT and route(1000, 4096, 4096, image=0) == OWN and route(4096, 4096, 4096, image=0) == OWN
    # the switches move the table
    prev = lib.gptq_set_stripe_gemm_max_rows(0)
    assert prev == 2048 and route(300, 4096, 4096) == OWN
    lib.gptq_set_stripe_gemm_max_rows(prev)
    prev = lib.gptq_set_prefill_route(0)
    assert route(3000, 4096, 4096) == LIBR
    assert route(65536, 4096, 4096) == LIBR
    lib.gptq_set_prefill_route(prev)
    assert route(0, 4096, 4096) == -2 and route(1, 4096, 4096, bits=5) == -1 and lib.gptq_layer_route_for(None, 1) == -4


def test_bench_gpus_n_spawns_n_ranks_and_refuses_a_wrong_world_size():
    """bench.py --gpus N without a launcher re-executes itself under torch.distributed.run with N ranks (dry run: the command only);
    with a launcher whose WORLD_SIZE differs from --gpus it refuses to print a line for the wrong N."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4', '--steps', '2'], env=dict(env, GPTQ_BENCH_SPAWN_DRY='1'),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    cmd = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])['spawn']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-4:] == ['--gpus', '4', '--steps', '2'] and 'torch.distributed.run' in cmd
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], env=dict(env, WORLD_SIZE='4', RANK='0'),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and 'refusing' in out.stderr


def test_checkpoint_source_reads_only_the_slices_it_is_asked_for(tmp_path):
    """round 5, shard at load (quant/tp_decode.py): a rank's rows / columns out of a reference-format checkpoint -- a dict of tensors or a
    .safetensors file -- are the slices of the full tensors, and the byte counter says how little was pulled"""
    import torch
    from safetensors.torch import save_file
    from quant.tp_decode import CheckpointSource
    g = torch.Generator().manual_seed(0)
    sd = {'a.qweight': torch.randint(-2**31, 2**31 - 1, (64, 96), dtype=torch.int32, generator=g), 'a.scales': torch.rand((4, 96), generator=g).half(),
          'a.bias': torch.rand(96, generator=g).half()}
    path = str(tmp_path / 'ckpt.safetensors')
    save_file(sd, path)
    for src in (CheckpointSource(sd), CheckpointSource(path)):
        assert src.has('a.qweight') and not src.has('b.qweight')
        assert torch.equal(src.get('a.qweight', rows=(16, 48)), sd['a.qweight'][16:48])
        assert torch.equal(src.get('a.qweight', cols=(32, 64)), sd['a.qweight'][:, 32:64]) and src.get('a.qweight', cols=(32, 64)).is_contiguous()
        assert torch.equal(src.get('a.scales', rows=(1, 3), cols=(0, 32)), sd['a.scales'][1:3, 0:32])
        assert torch.equal(src.get('a.bias'), sd['a.bias'])
        assert src.bytes_read == 32 * 96 * 4 + 2 * 64 * 32 * 4 + 2 * 32 * 2 + 96 * 2


def test_bench_refuses_a_pmc_pass_of_another_build(tmp_path):
    """roofline.traffic comes from a separate rocprofv3 --pmc pass; since round 5 tools/pmc_traffic.py stamps the pass with a hash of the kernel
    sources and bench.py refuses one taken on other sources"""
    import json
    import bench
    good, bad, old = tmp_path / 'good.json', tmp_path / 'bad.json', tmp_path / 'old.json'
    good.write_text(json.dumps({'hbm_bytes_per_launch_avg': 123, 'csrc_sha16': bench.csrc_sha16()}))
    bad.write_text(json.dumps({'hbm_bytes_per_launch_avg': 123, 'csrc_sha16': '0' * 16}))
    old.write_text(json.dumps({'hbm_bytes_per_launch_avg': 456}))
    v, src = bench.pmc_traffic(str(good))
    assert v == 123 and 'same kernel sources' in src
    v, src = bench.pmc_traffic(str(bad))
    assert v is None and 'refused' in src
    v, src = bench.pmc_traffic(str(old))
    assert v == 456 and 'unstamped' in src
    assert len(bench.csrc_sha16()) == 16
