"""Graph-speed decode behind the reference's own call sites.

``llama_inference.py:119-127`` calls ``model.generate(...)`` and ``llama.py:385-438`` (``benchmark``) calls
``model(input_ids[:, i:i+1], past_key_values=...)``: one token per forward through the swapped modules
(``QuantLlamaAttention`` / ``QuantLlamaMLP`` / the HIP RMSNorm -- fused_attn.py:117-161, fused_mlp.py:203-218).
Eagerly that is ~330 small launches per token and the host cannot keep up.  ``make_quant_attn``
therefore installs this hook on the ``LlamaForCausalLM`` it is given: a forward with ONE new token, batch 1 and a
KV cache is answered by ``quant.decode.DecodeEngine`` -- the same arithmetic as the module chain, issued as 5 launches
per layer and replayed as one hipGraph -- and everything else (prefill, batches, training, beams, output_attentions ...)
goes to the original forward untouched.  The caller keeps its API: HF ``generate`` with its own sampling, stopping
criteria and cache object.

Cache protocol: the engine owns a static K/V cache.  The first decode step after a prefill copies the caller's cache
into it; later steps only advance the engine (HF ``generate`` carries position_ids / attention_mask itself and never
looks into the cache between steps), so WHILE a run is tracked the caller's cache object lags behind by the tokens only
the engine holds.  It is brought up to date
  * when ``model.generate`` returns (the hook wraps it: the cache a caller gets back is complete),
  * on ``flush_decode_engine(model)`` (for hand-written loops such as ``llama.py:385-438`` that keep the cache), and
  * before any call that has to go the eager way: the engine's tokens are appended up to the position the incoming call starts
    at -- its ``cache_position`` / ``position_ids`` when it carries them, else everything (a bare call derives its positions
    from the cache length, so the cache must be complete).  A caller that starts again at the lagging length (it never saw the
    engine's tokens and re-sends them, e.g. a second ``generate(full_ids, past_key_values=cache)``) gets NOTHING appended: the eager
    forward recomputes those tokens itself, appending them as well would duplicate K/V entries.
Eligibility is decided once per sequence, on the first untracked step (one host sync): batch 1, one new token, an
all-ones 2-D attention mask of exactly the cache length + 1 (left padding -> eager) and, when given, position_ids /
cache_position equal to the cache length.  ``GPTQ_DECODE_ENGINE=0`` disables the hook.
"""
import os
import types
import weakref

import torch

ENABLED = os.environ.get('GPTQ_DECODE_ENGINE', '1') != '0'
# Memory mode is the DEFAULT since round 4: when the engine is first built every eligible module (trivial g_idx, 2 / 4 / 8 bits, an image)
# frees qweight / scales / qzeros and runs on its stripe16 image alone -- ONE copy of the packed weights, the footprint the reference
# publishes (README.md:23-29).  state_dict() / load_state_dict() / .to() / copy.deepcopy see the original tensors (reproduced bit-exactly
# from the image; a move or copy restores them first).  GPTQ_RELEASE_CHECKPOINT=0 keeps both copies (e.g. to read module.qweight).
RELEASE_CHECKPOINT = os.environ.get('GPTQ_RELEASE_CHECKPOINT', '1') != '0'


class _State:
    __slots__ = ('engine', 'sig', 'cache_ref', 'hf_len', 'pos', 'steps')

    def __init__(self):
        self.engine, self.sig, self.cache_ref, self.hf_len, self.pos, self.steps = None, None, None, 0, 0, 0


def _signature(model):
    """cheap identity of the weights the engine captured pointers of (rebuilt when the model moved or was reloaded)."""
    from .quant_linear import _ver
    l0 = model.model.layers[0]
    w = l0.self_attn.qkv_proj.qweight
    return (w.device, w.data_ptr(), _ver(w), model.lm_head.weight.data_ptr(), len(model.model.layers))


def _eligible_model(model):
    from .fused_attn import QuantLlamaAttention
    from .fused_mlp import QuantLlamaMLP
    try:
        layers = model.model.layers
        if not len(layers):
            return False
        for layer in layers:
            if type(layer.self_attn) is not QuantLlamaAttention or type(layer.mlp) is not QuantLlamaMLP:
                return False
        a = layers[0].self_attn
        return a.head_dim == 128 and a.qkv_proj.qweight.is_cuda and model.lm_head.weight.is_cuda and not model.training
    except AttributeError:
        return False


def _cache_len(cache):
    try:
        return int(cache.get_seq_length())
    except Exception:
        return None


def _sync_back(st, upto=None):
    """append the tokens only the engine holds -- positions [hf_len, min(upto, pos)) -- to the caller's cache."""
    cache = st.cache_ref() if st.cache_ref is not None else None
    if cache is None or st.pos <= st.hf_len or st.engine is None:
        return
    eng, a, b = st.engine, st.hf_len, st.pos if upto is None else min(int(upto), st.pos)
    if b <= a:
        return
    for li in range(len(eng.layers)):
        k = eng.kc[li, a:b].view(b - a, eng.heads, eng.head_dim).transpose(0, 1).unsqueeze(0)
        v = eng.vc[li, a:b].view(b - a, eng.heads, eng.head_dim).transpose(0, 1).unsqueeze(0)
        cache.update(k.contiguous(), v.contiguous(), li)
    st.hf_len = b


def _sync_in(st, cache, T):
    """copy the caller's cache (T tokens) into the engine's static cache."""
    from .decode import _cache_layer_kv
    eng = st.engine
    if T:
        for li in range(len(eng.layers)):
            k, v = _cache_layer_kv(cache, li)
            eng.kc[li, :T].copy_(k[0].transpose(0, 1).reshape(T, -1))
            eng.vc[li, :T].copy_(v[0].transpose(0, 1).reshape(T, -1))
    eng.pos.fill_(T)
    st.cache_ref, st.hf_len, st.pos = weakref.ref(cache), T, T


def _start_position(position_ids, kw):
    """first position of the incoming tokens when the call says so (cache_position wins), else None.  One host sync."""
    for t in (kw.get('cache_position'), position_ids):
        if torch.is_tensor(t) and t.numel():
            return int(t.reshape(-1)[0])
    return None


def _engine_forward(model, st, input_ids, cache, attention_mask, position_ids, kw):
    """one token through the DecodeEngine, or None when this call has to go the eager way."""
    from .decode import DecodeEngine
    sig = _signature(model)
    if st.sig != sig or st.engine is None:
        if not _eligible_model(model):      # (walks all layers: only when the weights changed identity, not once per token)
            return None
    elif model.training:
        return None
    if st.engine is None or st.sig != sig:
        t_max = int(min(max(getattr(model.config, 'max_position_embeddings', 2048), 256), 8192))
        if RELEASE_CHECKPOINT:      # memory mode: from the first decode step on ONE copy of the packed weights (quant.release_checkpoint)
            from . import release_checkpoint
            release_checkpoint(model)
            sig = _signature(model)     # the placeholders are new tensor objects
        st.engine = DecodeEngine(model, t_max=t_max).capture()
        st.sig, st.cache_ref = sig, None
    eng = st.engine
    T = _cache_len(cache)
    if T is None:
        return None
    tracked = st.cache_ref is not None and st.cache_ref() is cache and T == st.hf_len
    pos = st.pos if tracked else T
    if attention_mask is not None and (attention_mask.dim() != 2 or attention_mask.shape[0] != 1 or attention_mask.shape[-1] != pos + 1):
        return None          # a mask that is not "everything so far" (4-D masks, other lengths): eager
    if pos + 1 > eng.t_max:
        return None
    if not tracked:
        # once per sequence (host sync): the mask must be ALL ones -- a left-padded prompt has a correctly shaped mask with zeros
        # and shifted positions, the engine would attend to the pads -- and explicit positions must continue the cache
        if attention_mask is not None and not bool(attention_mask.all()):
            return None
        start = _start_position(position_ids, kw)
        if start is not None and start != T:
            return None
        _sync_in(st, cache, T)
    logits = eng.decode(input_ids.reshape(1))
    st.pos += 1
    st.steps += 1
    from transformers.modeling_outputs import CausalLMOutputWithPast
    return CausalLMOutputWithPast(loss=None, logits=logits.view(1, 1, -1).clone(), past_key_values=cache)


def install_decode_engine(model):
    """wrap ``model.forward`` (instance level).  Idempotent; returns the model."""
    if not ENABLED or getattr(model, '_gptq_engine_state', None) is not None:
        return model
    if not (hasattr(model, 'model') and hasattr(model.model, 'layers') and hasattr(model, 'lm_head')):
        return model          # not a *ForCausalLM (e.g. make_quant_attn on a bare decoder stack): nothing to route
    orig_forward = model.forward
    st = _State()

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, **kw):
        fast = (ENABLED and not getattr(self, '_gptq_engine_disabled', False) and input_ids is not None and inputs_embeds is None
                and labels is None and use_cache is not False and past_key_values is not None and hasattr(past_key_values, 'update')
                and input_ids.dim() == 2 and input_ids.shape[0] == 1 and input_ids.shape[1] == 1 and input_ids.is_cuda
                and not torch.is_grad_enabled() and not kw.get('output_attentions') and not kw.get('output_hidden_states')
                and not torch.cuda.is_current_stream_capturing())
        if fast:
            out = _engine_forward(self, st, input_ids, past_key_values, attention_mask, position_ids, kw)
            if out is not None:
                return out if kw.get('return_dict', True) is not False else (out.logits, out.past_key_values)
        if st.cache_ref is not None and st.cache_ref() is past_key_values and st.pos > st.hf_len:
            # the eager path continues the tracked cache: it must hold the engine's tokens up to where THIS call starts (see the
            # module docstring: explicit positions decide, a bare call needs the complete cache)
            _sync_back(st, _start_position(position_ids, kw))
        elif st.cache_ref is not None:
            _sync_back(st)    # another cache object takes over: leave the tracked one complete
        st.cache_ref = None
        return orig_forward(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                            past_key_values=past_key_values, inputs_embeds=inputs_embeds, labels=labels, use_cache=use_cache, **kw)

    model._gptq_engine_state = st
    model._gptq_orig_forward = orig_forward
    model.forward = types.MethodType(forward, model)
    if hasattr(model, 'generate'):
        orig_generate = model.generate

        def generate(self, *args, **kwargs):
            try:
                return orig_generate(*args, **kwargs)
            finally:
                flush_decode_engine(self)     # the cache handed back to the caller holds every generated token

        model._gptq_orig_generate = orig_generate
        model.generate = types.MethodType(generate, model)
    return model


def flush_decode_engine(model):
    """append the tokens only the engine holds to the cache it tracks and stop tracking it (idempotent).  Hand-written decode loops
    that keep their cache object call this before they look into it; ``model.generate`` does it on return."""
    st = getattr(model, '_gptq_engine_state', None)
    if st is not None and st.cache_ref is not None:
        _sync_back(st)
        st.cache_ref = None


def engine_steps(model):
    """number of forwards the engine answered so far (tests / bench)."""
    st = getattr(model, '_gptq_engine_state', None)
    return st.steps if st is not None else 0
