"""Prepared layers: the Python face of ``gptq_layer_prepare`` / ``gptq_layer_forward`` (include/gptq_mi355x.h).

The reference has ONE call site per operator -- ``matmul248(input, qweight, scales, qzeros, g_idx, bits, maxq)``
(quant/quant_linear.py:263-269) and ``fusedmatmul_248`` behind ``QuantLlamaMLP`` (quant/fused_mlp.py:203-218) -- and so has this
package since round 3: the whole M -> kernel table lives in the C library (csrc/capi.hip, "Prepared layers"); this module only
owns the objects the handle borrows -- the image tensor (stripe16 copy, group-sorted rows and permutation of an act-order layer)
-- and finds the handle again when the SAME checkpoint tensors come back.

Derived state is kept in ONE registry keyed by the ``qweight`` tensor OBJECT (a ``WeakTensorKeyDictionary``: no attributes are hung on
tensors, the entry dies with the tensor) and validated by the (version counter, address, shape) of every buffer of the layer:
``load_state_dict`` copies in place (version bump), ``make_quant_attn`` reassigns buffers (new objects), ``.to(device)`` makes new
tensors -- each of them gets a fresh handle, as SURVEY 8(b) "Ownership" demands.
"""
import ctypes
import os

import torch
from torch.utils.weak import WeakTensorKeyDictionary   # keys compared by identity (tensor == tensor is elementwise)

from . import _native

# test / A-B hooks (the dispatch itself is in C): GPTQ_STRIPE=0 prepares layers without an image (checkpoint-layout kernels only)
USE_IMAGE = os.environ.get('GPTQ_STRIPE', '1') != '0'


# bumped whenever a module takes its checkpoint buffers back (restore_checkpoint of a QuantLinear / QuantLlamaMLP): the decode-engine hook keeps
# the value it built its engines at and rebuilds (and releases again) when it moved -- a restored module is otherwise invisible to the cheap
# per-token signature, which looks at one layer only
RESTORE_EPOCH = [0]


def _ver(t):
    """version counter of a tensor; inference-mode tensors have none (they are immutable outside inference mode)."""
    try:
        return t._version
    except Exception:
        return -1


def _sig(t):
    return None if t is None else (_ver(t), t.data_ptr(), tuple(t.shape), t.dtype)


class PreparedLayer:
    """owner of one ``gptq_layer_t`` and of the image buffer it points into.  ``sets``: ((qweight, scales, qzeros, g_idx), ...) of one
    layer or of a gate/up pair; the tensors must be int32 / fp16 / int32 / int32, contiguous, on one ROCm device."""

    def __init__(self, sets, bias, bits, groupsize, K, N, use_image=True, sort=True):
        lib = _native.lib()
        self.lib = lib
        self.bits, self.groupsize, self.K, self.N, self.nsets = bits, groupsize, K, N, len(sets)
        dev = sets[0][0].device
        self.device = dev
        # everything the handle borrows except the registry key (sets[0][0]: a strong reference to the key would keep the entry alive
        # for ever); the caller of forward() holds that tensor anyway
        self._keep = [t for s in sets for t in s[1:]] + [s[0] for s in sets[1:]]
        self._bias = bias        # read by the handle for as long as it lives (also after release())
        self._gidx = [s[3] for s in sets if s[3] is not None]   # ... and so is g_idx of an act-order layer (K ints: kept, not released)
        ptr = _native.ptr
        s0, s1 = sets[0], (sets[1] if len(sets) > 1 else (None, None, None, None))
        stream = _native.stream_ptr(dev)
        with torch.cuda.device(dev):
            kinds = []
            for (_, _, _, gi) in sets:
                k = 0 if gi is None else lib.gptq_layer_inspect(gi.data_ptr(), K, groupsize, stream)
                if k < 0:
                    _native.check(k, 'gptq_layer_inspect')
                kinds.append(k)
            kind = kinds[0] if len(set(kinds)) == 1 else 2
            # sort=False (GPTQ_ACT_ORDER_SORT=0, tests): an act-order layer gets no group-sorted image -> the generic g_idx kernels serve it
            nbytes = lib.gptq_layer_image_bytes(K, N, bits, groupsize, self.nsets, kind) if (use_image and (sort or kind != 1)) else 0
            self.image = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
            h = ctypes.c_void_p()
            rc = lib.gptq_layer_prepare(ctypes.byref(h), s0[0].data_ptr(), s0[1].data_ptr(), s0[2].data_ptr(), ptr(s0[3]), ptr(bias), ptr(s1[0]), ptr(s1[1]),
                                        ptr(s1[2]), ptr(s1[3]), K, N, bits, groupsize, ptr(self.image), nbytes, stream)
        _native.check(rc, 'gptq_layer_prepare')
        self.handle = h
        self.released = False
        self.kind = lib.gptq_layer_kind(h)
        st, nb, p16 = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_void_p()
        lib.gptq_layer_stripe_image(h, ctypes.byref(st), ctypes.byref(nb), ctypes.byref(p16))
        # views into the image for the engines that drive the stripe kernels themselves (fused RMSNorm / residual epilogues)
        self.stripe = self.image[:nb.value] if (st.value and self.image is not None) else None
        self.perm16 = None
        if p16.value and self.image is not None:
            off = p16.value - self.image.data_ptr()
            self.perm16 = self.image[off:off + 2 * K].view(torch.int16)
        # regular act-order layer: original k -> sorted position (what a PRODUCER of this layer's input stores through, so that the layer needs no gather)
        self.invperm32 = None
        inv = ctypes.c_void_p()
        if lib.gptq_layer_inverse_perm(h, ctypes.byref(inv)) == 0 and inv.value and self.image is not None:
            off = inv.value - self.image.data_ptr()
            self.invperm32 = self.image[off:off + 4 * K].view(torch.int32)

    def __del__(self):
        h, self.handle = getattr(self, 'handle', None), None
        if h:
            try:
                self.lib.gptq_layer_destroy(h)
            except Exception:   # interpreter shutdown
                pass

    def release(self):
        """memory mode (gptq_layer_release_checkpoint): the handle stops reading qweight / scales / qzeros -- the stripe16 image (of the
        group-sorted rows + both permutations for a regular act-order layer, round 4) is a bijection of them -- and this object lets go of its
        references, so the caller can free them; g_idx stays borrowed.  False (nothing changed) for layers that need the checkpoint layout:
        irregular act-order, no image."""
        if self.lib.gptq_layer_release_checkpoint(self.handle) != 0:
            return False
        self._keep = []          # incl. a converted copy of the key tensor prepared() may have parked here; the bias lives in self._bias
        self.released = True
        return True

    def unpack(self, which=0):
        """(qweight, scales, qzeros) of weight set ``which`` reproduced bit-exactly from the image (gptq_layer_unpack_checkpoint)"""
        K, N, bits = self.K, self.N, self.bits
        G = 1 if self.groupsize >= K else K // self.groupsize
        with torch.cuda.device(self.device):
            qw = torch.empty((K // 32 * bits, N), dtype=torch.int32, device=self.device)
            sc = torch.empty((G, N), dtype=torch.float16, device=self.device)
            qz = torch.empty((G, N // 32 * bits), dtype=torch.int32, device=self.device)
            rc = self.lib.gptq_layer_unpack_checkpoint(self.handle, which, qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), _native.stream_ptr(self.device))
        _native.check(rc, 'gptq_layer_unpack_checkpoint')
        return qw, sc, qz

    def forward(self, x, out):
        """out[M, N] = layer(x[M, K]) on the current stream of x's device.  x: fp16, unit column stride, 16-byte aligned rows."""
        lib = self.lib
        M = x.shape[0]
        stream = _native.stream_ptr(x.device)            # ONE query per call (the workspace is keyed by it)
        ws = _native.layer_workspace(x.device, stream)
        need = lib.gptq_layer_scratch_bytes(self.handle, M)
        scratch = torch.empty(need, dtype=torch.uint8, device=x.device) if need else None     # caching allocator: the next layer reuses it
        rc = lib.gptq_layer_forward(self.handle, x.data_ptr(), x.stride(0) if M > 1 else self.K, out.data_ptr(), out.stride(0) if M > 1 else self.N, M,
                                    ws.data_ptr(), ws.numel(), _native.ptr(scratch), need, stream)
        if rc == -5:
            # GPTQ_E_WORKSPACE: a kernel of the fast route declined at launch (LDS opt-in, a strided x beyond its address range ...) and the fall-back
            # wants scratch the fast route did not ask for (the rebuilt checkpoint layout of a released layer, the dense route's workspace): once
            # more with everything
            need = lib.gptq_layer_fallback_scratch_bytes(self.handle, M)
            scratch = torch.empty(need, dtype=torch.uint8, device=x.device)
            rc = lib.gptq_layer_forward(self.handle, x.data_ptr(), x.stride(0) if M > 1 else self.K, out.data_ptr(), out.stride(0) if M > 1 else self.N, M,
                                        ws.data_ptr(), ws.numel(), scratch.data_ptr(), need, stream)
        _native.check(rc, 'gptq_layer_forward')
        return out


_LAYERS = WeakTensorKeyDictionary()     # qweight tensor object -> (signature, PreparedLayer)


def _int32c(t):
    return t if (t.dtype == torch.int32 and t.is_contiguous()) else t.to(torch.int32).contiguous()


def _f16c(t):
    return t if (t.dtype == torch.float16 and t.is_contiguous()) else t.half().contiguous()


def prepared(sets, bias, bits, groupsize, K, N, sort=True):
    """the PreparedLayer of these checkpoint tensors: ``sets`` = ((qweight, scales, qzeros, g_idx-or-None), ...) AS THE CALLER HOLDS
    THEM (the signature is taken from the caller's objects; dtype / layout conversions happen once, on a miss, and the converted
    copies live in the PreparedLayer).  Built on first use, rebuilt when any buffer changed identity or content version."""
    key_t = sets[0][0]
    sig = (tuple(_sig(t) for s in sets for t in s), _sig(bias), bits, groupsize, USE_IMAGE, bool(sort))
    hit = _LAYERS.get(key_t)
    if hit is not None and hit[0] == sig:
        return hit[1]
    conv = tuple((_int32c(qw), _f16c(sc), _int32c(qz), None if gi is None else _int32c(gi[:K])) for (qw, sc, qz, gi) in sets)
    pl = PreparedLayer(conv, None if bias is None else _f16c(bias), bits, groupsize, K, N, use_image=USE_IMAGE, sort=sort)
    if conv[0][0] is not key_t:
        pl._keep.append(conv[0][0])       # a converted copy of the key tensor is a different object: safe to hold
    _LAYERS[key_t] = (sig, pl)
    return pl
