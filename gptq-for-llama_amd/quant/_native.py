"""ctypes binding of libgptq_mi355x.so (C ABI: include/gptq_mi355x.h).

There is NO CPU fallback and no second backend: if the shared library is missing or a tensor
is not on a ROCm device the call raises.  PyTorch is only the owner of device memory and
streams here; every launch goes to ``torch.cuda.current_stream()`` of the input's device, so the
ops are asynchronous and hipGraph-capturable like the reference's Triton launches
(reference quant/quant_linear.py:263-269).
"""
import ctypes
import os
import threading

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, 'lib', 'libgptq_mi355x.so')
CSRC_DIR = os.path.join(_PKG, 'csrc')

_lib = None
_lock = threading.Lock()

c_void_p, c_int, c_int64, c_size_t, c_float = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                ctypes.c_size_t, ctypes.c_float)

# name -> argtypes; mirrors include/gptq_mi355x.h one to one
_SIGNATURES = {
    'gptq_query': [c_int],
    'gptq_set_gemv_variant': [c_int],
    'gptq_set_split_k': [c_int],
    'gptq_set_debug_buffer': [c_void_p],
    'gptq_debug_dirty_lds': [ctypes.c_uint32, c_void_p],
    'gptq_set_gemm_kernel': [c_int],
    'gptq_set_prefill_route': [c_int],
    'gptq_set_stripe_mm_pass_rows': [c_int],
    'gptq_set_stripe_gemm_max_rows': [c_int],
    'gptq_prefill_route_for': [c_int, c_int, c_int, c_int, c_int],
    'gptq_set_library_enabled': [c_int],
    'gptq_set_gemm8_mfma': [c_int],
    'gptq_set_gemm8_tile': [c_int],
    'gptq_prefill_plan_count': [],
    'gptq_matmul248_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                           c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p],
    'gptq_matmul248_partial_f32': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'gptq_gemv_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                      c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p],
    'gptq_skinny_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                        c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p],
    'gptq_gemm_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                      c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'gptq_fused_mlp_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                           c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int,
                           c_void_p, c_size_t, c_void_p],
    'gptq_transpose_matmul248_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'gptq_rmsnorm_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p],
    'gptq_dense_matvec_f16': [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_void_p],
    'gptq_rope_f16': [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p],
    'gptq_pack_f32': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                      c_void_p, c_void_p, c_void_p],
    'gptq_g_idx_is_trivial': [c_void_p, c_int, c_int, c_void_p, c_void_p],
    'gptq_dequant_f16': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    'gptq_dequant_ld_f16': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p],
    'gptq_silu_mul_f16': [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p],
    'gptq_prefill_workspace_bytes': [c_int, c_int, c_int, c_int],
    'gptq_prefill_matmul_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                c_int, c_void_p, c_size_t, c_void_p],
    'gptq_prefill_transpose_matmul248_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                             c_int, c_void_p, c_size_t, c_void_p],
    'gptq_prefill_fused_mlp_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p],
    'gptq_act_order_repack': [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    'gptq_matmul248_sorted_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                  c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p],
    'gptq_fused_mlp_sorted_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p],
    'gptq_rmsnorm_sorted_f16': [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p],
    'gptq_rmsnorm_matmul248_f16': [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                   c_int, c_int, c_int, c_void_p, c_size_t, c_void_p],
    'gptq_rmsnorm_fused_mlp_f16': [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p],
    'gptq_decode_rope_kv_f16': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p],
    'gptq_decode_attn_workspace_bytes': [c_int, c_int, c_int],
    'gptq_rope_table_f32': [c_void_p, c_int, c_int, c_float, c_void_p],
    'gptq_decode_attn_fused_table_f16': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int,
                                         c_float, c_float, c_void_p, c_void_p],
    'gptq_decode_attn_fused_f16': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int,
                                   c_float, c_float, c_void_p],
    'gptq_decode_attn_f16': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int,
                             c_float, c_void_p],
    'gptq_solver_block_f32': [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int64,
                              c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p],
    'gptq_stripe_bytes': [c_int, c_int, c_int, c_int, c_int],
    'gptq_stripe_repack': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int,
                           c_void_p],
    'gptq_stripe_matvec_f16': [c_void_p, c_int64, c_void_p, c_size_t, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int,
                               c_void_p, c_float, c_void_p, c_void_p],
    'gptq_p2p_buffer_bytes': [c_int, c_int],
    'gptq_p2p_create': [c_int, c_int, c_void_p, c_void_p],
    'gptq_p2p_open': [c_void_p, c_void_p],
    'gptq_p2p_close': [c_void_p, c_int],
    'gptq_p2p_status': [c_void_p, c_int, c_int, c_void_p],
    'gptq_p2p_allreduce_f32': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    'gptq_p2p_allreduce_silu_mul_f32': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    'gptq_stripe_matmul_f16': [c_void_p, c_int64, c_void_p, c_size_t, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                               c_size_t, c_void_p],
    'gptq_layer_inspect': [c_void_p, c_int, c_int, c_void_p],
    'gptq_layer_image_bytes': [c_int, c_int, c_int, c_int, c_int, c_int],
    'gptq_layer_prepare': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                           c_int, c_void_p, c_size_t, c_void_p],
    'gptq_layer_destroy': [c_void_p],
    'gptq_layer_kind': [c_void_p],
    'gptq_layer_stripe_image': [c_void_p, c_void_p, c_void_p, c_void_p],
    'gptq_layer_release_checkpoint': [c_void_p],
    'gptq_layer_unpack_checkpoint': [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    'gptq_layer_route_for_shape': [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int],
    'gptq_layer_route_for': [c_void_p, c_int],
    'gptq_layer_workspace_bytes': [],
    'gptq_layer_scratch_bytes': [c_void_p, c_int],
    'gptq_layer_fallback_scratch_bytes': [c_void_p, c_int],
    'gptq_layer_forward': [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p],
    'gptq_set_progress_counter': [c_void_p],
    'gptq_stripe_matmul_partial_f32': [c_void_p, c_int64, c_void_p, c_size_t, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'gptq_stripe_matvec_partial_f32': [c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    # round 5: the batched decode engine
    'gptq_dense_matmat_f16': [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_float, c_void_p],
    'gptq_add_rows_f16': [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p],
    'gptq_decode_attn_batch_workspace_bytes': [c_int, c_int, c_int, c_int],
    'gptq_decode_attn_batch_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_int, c_int, c_int, c_int,
                                   c_float, c_float, c_void_p, c_void_p, c_void_p],
    'gptq_layer_inverse_perm': [c_void_p, c_void_p],
    'gptq_stripe_matvec_perm_out_f16': [c_void_p, c_int64, c_void_p, c_size_t, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p, c_float, c_void_p, c_void_p, c_void_p],
    # round 6: streaming decode attention whose splits are merged by o_proj's decode kernel
    'gptq_decode_attn_splits': [c_int, c_int, c_int, c_int],
    'gptq_decode_attn_split_f16': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_float, c_float,
                                   c_void_p, c_int, c_void_p],
    'gptq_layer_decode_attn_supported': [c_void_p, c_int, c_int, c_int],
    'gptq_layer_decode_attn_f16': [c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64,
                                   c_void_p],
    'gptq_layer_decode_scratch_bytes': [c_void_p, c_int],
    'gptq_layer_decode_f16': [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_float, c_void_p, c_int64, c_void_p, c_size_t,
                              c_void_p, c_size_t, c_void_p],
    'gptq_layer_decode_next_norm_f16': [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_float, c_void_p, c_int64, c_void_p, c_float, c_void_p,
                                        c_int64, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p],
}


EXPORTS = sorted(list(_SIGNATURES) + ['gptq_strerror'])


class NativeLibraryMissing(RuntimeError):
    pass


def lib():
    """Load libgptq_mi355x.so once.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise NativeLibraryMissing(
                    'libgptq_mi355x.so not found at %s -- build it with `make -C %s` (or '
                    '__graft_entry__.build()); there is no CPU/Triton fallback.' % (LIB_PATH, CSRC_DIR))
            L = ctypes.CDLL(LIB_PATH)
            for name, args in _SIGNATURES.items():
                fn = getattr(L, name)
                fn.argtypes = args
                fn.restype = c_int
            L.gptq_set_debug_buffer.restype = c_void_p
            L.gptq_decode_attn_workspace_bytes.restype = c_size_t
            L.gptq_stripe_bytes.restype = c_size_t
            L.gptq_p2p_buffer_bytes.restype = c_size_t
            L.gptq_prefill_workspace_bytes.restype = c_size_t
            L.gptq_layer_image_bytes.restype = c_size_t
            L.gptq_layer_workspace_bytes.restype = c_size_t
            L.gptq_layer_scratch_bytes.restype = c_size_t
            L.gptq_layer_fallback_scratch_bytes.restype = c_size_t
            L.gptq_layer_decode_scratch_bytes.restype = c_size_t
            L.gptq_decode_attn_batch_workspace_bytes.restype = c_size_t
            L.gptq_layer_destroy.restype = None
            L.gptq_strerror.argtypes = [c_int]
            L.gptq_strerror.restype = ctypes.c_char_p
            _lib = L
    return _lib


def check(rc, what):
    if rc == 0:
        return
    msg = lib().gptq_strerror(rc).decode()
    if rc == -1:
        raise NotImplementedError(msg)          # reference quant_linear.py:309
    if rc == -7:
        raise RuntimeError(msg)                 # reference triton_norm.py:60
    raise RuntimeError('%s failed (%d): %s' % (what, rc, msg))


def require_device(t, what):
    if not t.is_cuda:
        raise RuntimeError('%s: tensor is on %s -- the MI355X HIP path has no CPU fallback' % (what, t.device))


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


class _NoGuard:
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """``with on_device(x.device):`` -- torch.cuda.device(device) only when it is NOT already the current device (kernels are launched on the
    current device: the guard is needed for a tensor that lives elsewhere, and costs 3-5 us of host time per call where it is not)."""
    if not isinstance(device, (torch.device, int)):
        device = torch.device(device)               # 'cuda:0' style strings (bench.py, tools)
    idx = device if isinstance(device, int) else device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(idx)


def stream_ptr(device):
    """raw handle of the CURRENT stream of `device` (asked per call: `with torch.cuda.stream(...)` must be honoured).  The private torch entry
    returns the handle without building a torch.cuda.Stream object: the module chain asks ~11 times per decoder layer, and
    torch.cuda.current_stream() cost 2-4 us each (tools/profile_eager_host.py)."""
    if not isinstance(device, (torch.device, int)):
        device = torch.device(device)               # 'cuda:0' style strings (bench.py, tools)
    if _raw_stream is not None:
        idx = device.index if isinstance(device, torch.device) else device
        try:
            return _raw_stream(torch.cuda.current_device() if idx is None else idx)
        except TypeError:                           # a torch version with another private signature: the public (slower) way
            pass
    return torch.cuda.current_stream(device).cuda_stream


_workspaces = {}


def workspace(device, stream=None):
    """Zero-initialised split-K scratch of the rowwave / stream kernels (they restore it to zero), one buffer per
    (device, stream): the kernels use its words as accumulators between the first and the last K slice of a launch, so two
    launches that may overlap -- different streams, or a captured graph replaying next to eager work -- must not share it.
    The stripe16 decode kernels need no workspace at all.  Buffers live as long as the process (a captured hipGraph keeps
    pointing at the one of its capture stream)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if stream is None:
        stream = stream_ptr(idx)
    key = (device.type, idx, int(stream))
    ws = _workspaces.get(key)
    if ws is None:
        nbytes = lib().gptq_query(3)
        with torch.cuda.device(idx):
            ws = torch.zeros(nbytes, dtype=torch.uint8, device=torch.device('cuda', idx))
        _workspaces[key] = ws
    return ws


_layer_workspaces = {}


def layer_workspace(device, stream=None):
    """The persistent workspace of gptq_layer_forward, one per (device, stream): [split-K words, zero-initialised and left zero by
    every call][scratch of the 16-row MFMA tiles].  Launches that may overlap (different streams; a captured graph replaying next
    to eager work) must not share it, hence the stream in the key; buffers live as long as the process (a captured hipGraph keeps
    pointing at the one of its capture stream)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if stream is None:
        stream = stream_ptr(idx)
    key = (device.type, idx, int(stream))
    ws = _layer_workspaces.get(key)
    if ws is None:
        L = lib()
        with torch.cuda.device(idx):
            ws = torch.empty(L.gptq_layer_workspace_bytes(), dtype=torch.uint8, device=torch.device('cuda', idx))
            ws[:L.gptq_query(3)].zero_()
        _layer_workspaces[key] = ws
    return ws


_mm_workspaces = {}


def mm_workspace(device, stream=None):
    """Scratch of gptq_stripe_matmul_f16 (fp32 partial tiles of the K slices; no state survives a launch), one per (device, stream)
    so that launches that may overlap never share it.  Separate from ``workspace``: the split-K words of the rowwave kernels
    must stay zero, these tiles are overwritten freely."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if stream is None:
        stream = stream_ptr(idx)
    key = (device.type, idx, int(stream))
    ws = _mm_workspaces.get(key)
    if ws is None:
        with torch.cuda.device(idx):
            ws = torch.empty(lib().gptq_query(5), dtype=torch.uint8, device=torch.device('cuda', idx))
        _mm_workspaces[key] = ws
    return ws


def ptr(t):
    return None if t is None else t.data_ptr()
