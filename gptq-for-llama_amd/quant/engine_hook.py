"""Graph-speed decode behind the reference's own call sites.

``llama_inference.py:119-127`` calls ``model.generate(...)`` and ``llama.py:385-438`` (``benchmark``) calls
``model(input_ids[:, i:i+1], past_key_values=...)``: one token per forward through the swapped modules
(``QuantLlamaAttention`` / ``QuantLlamaMLP`` / the HIP RMSNorm -- fused_attn.py:117-161, fused_mlp.py:203-218).
Eagerly that is ~330 small launches per token and the host cannot keep up.  ``make_quant_attn``
therefore installs this hook on the ``LlamaForCausalLM`` it is given: a forward with ONE new token, batch 1 and a
KV cache is answered by ``quant.decode.DecodeEngine`` -- the same arithmetic as the module chain, issued as 5 launches
per layer and replayed as one hipGraph -- and everything else (prefill, training, output_attentions ...) goes to the
original forward untouched.  Beam search goes the eager way too (round 6): HF permutes the rows of the cache IN PLACE after
every step (``cache.reorder_cache(beam_idx)``), which the engine's static per-row K/V cache would not follow -- ``generate(num_beams > 1)``
switches the hook off for the call, and a tracked cache whose ``reorder_cache`` is called by anyone is completed, released and never
tracked again.  The caller keeps its API: HF ``generate`` with its own sampling, stopping
criteria and cache object.  Round 6: the one call HF's loop only slows down -- plain GREEDY ``generate`` of one prompt with nothing between the steps --
runs the engine's self-feeding greedy graph instead (``_greedy_fast`` below: the same tokens, no host round trip per token).

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
Eligibility is decided once per sequence, on the first untracked step (one host sync): one new token per row, at most
``decode.MAX_BATCH`` rows, a 2-D attention mask of exactly the cache length + 1 whose zeros -- if any -- are a LEFT-padding prefix
of each row, and, when given, position_ids / cache_position that continue the rows.  ``GPTQ_DECODE_ENGINE=0`` disables the hook.

Round 5 -- batches and left padding.  ``model.generate`` on several prompts sends ``[B, 1]`` steps with a left-padded mask and per-row
``position_ids`` (= tokens of the row so far).  The engine keeps every row as an UNPADDED sequence: row b of its K/V cache holds the
``T - npad_b`` real tokens of the caller's cache (the pads the prefill computed K/V for are masked out for ever, so they are simply not
copied), its position is ``T - npad_b`` -- what HF's position_ids say -- and the attention launch runs over exactly that many entries.
Going back (``_sync_back``) the engine-only tokens of row b are appended at the caller's uniform index, where HF expects them.  One engine
(graph, static K/V cache) per batch size, at most ``MAX_ENGINES`` alive.
"""
import os
import types
import weakref

import torch

from .decode import MAX_BATCH

ENABLED = os.environ.get('GPTQ_DECODE_ENGINE', '1') != '0'
# Memory mode is the DEFAULT since round 4: when the engine is first built every eligible module (trivial g_idx, 2 / 4 / 8 bits, an image)
# frees qweight / scales / qzeros and runs on its stripe16 image alone -- ONE copy of the packed weights, the footprint the reference
# publishes (README.md:23-29).  state_dict() / load_state_dict() / .to() / copy.deepcopy see the original tensors (reproduced bit-exactly
# from the image; a move or copy restores them first).  GPTQ_RELEASE_CHECKPOINT=0 keeps both copies (e.g. to read module.qweight).
RELEASE_CHECKPOINT = os.environ.get('GPTQ_RELEASE_CHECKPOINT', '1') != '0'


MAX_ENGINES = 4       # engines (one per batch size) kept alive per model; the least recently used one goes first
# bytes of static K / V cache ONE engine may hold (layers x rows x t_max x hidden x 2 x fp16): t_max is sized from the request and grows on
# demand up to min(max_position_embeddings, 8192, what this budget allows) -- a 7B engine of 16 rows at 4096 tokens would be 34 GB
CACHE_BUDGET = int(float(os.environ.get('GPTQ_ENGINE_CACHE_GB', '16')) * (1 << 30))


class _State:
    __slots__ = ('engine', 'engines', 'declined', 'sig', 'cache_ref', 'hf_len', 'pos', 'steps', 'npad', 'released', 'len_hint', 'reordered')

    def __init__(self):
        self.engine, self.sig, self.cache_ref, self.hf_len, self.pos, self.steps = None, None, None, 0, 0, 0
        self.len_hint = 0         # tokens the running generate() call may reach (prompt + max_new_tokens), 0 = unknown
        self.reordered = None     # weakref of a cache whose rows its caller permutes (beam search): the engine stays away from it
        self.engines = {}         # batch size -> DecodeEngine (insertion order = recency)
        self.declined = set()     # batch sizes the model has no engine route for
        self.npad = None          # per row of the tracked batch: pads of the caller's left-padded cache (host ints)
        self.released = False


def _signature(model):
    """cheap identity of the weights the engine captured pointers of (rebuilt when the model moved or was reloaded)."""
    from .quant_linear import _ver
    from .layer import RESTORE_EPOCH
    l0 = model.model.layers[0]
    w = l0.self_attn.qkv_proj.qweight
    return (w.device, w.data_ptr(), _ver(w), model.lm_head.weight.data_ptr(), len(model.model.layers), RESTORE_EPOCH[0])


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
    """append the tokens only the engine holds -- positions [hf_len, min(upto, pos)) of the caller's (uniform, padded) index -- to the
    caller's cache; row r of the engine stores them npad[r] entries earlier (it keeps the row without its pads)."""
    cache = st.cache_ref() if st.cache_ref is not None else None
    if cache is None or st.pos <= st.hf_len or st.engine is None:
        return
    eng, a, b = st.engine, st.hf_len, st.pos if upto is None else min(int(upto), st.pos)
    if b <= a:
        return
    npad = st.npad if st.npad is not None else [0] * eng.batch
    for li in range(len(eng.layers)):
        k = torch.stack([eng.kcb[li, r, a - n:b - n].view(b - a, eng.heads, eng.head_dim).transpose(0, 1) for r, n in enumerate(npad)])
        v = torch.stack([eng.vcb[li, r, a - n:b - n].view(b - a, eng.heads, eng.head_dim).transpose(0, 1) for r, n in enumerate(npad)])
        cache.update(k.contiguous(), v.contiguous(), li)
    st.hf_len = b


def _watch_reorder(st, cache):
    """beam search permutes the rows of the cache in place after every step (HF: cache.reorder_cache(beam_idx)); the engine's rows would keep
    attending to another beam's history.  A tracked cache therefore gets its reorder_cache wrapped: the engine's tokens are appended first (in
    the order the caller knows), the cache is released, and the engine never tracks this object again (copying every row back in per step
    would cost more than the eager step)."""
    if getattr(cache, '_gptq_reorder_watched', False) or not hasattr(cache, 'reorder_cache'):
        return
    orig = cache.reorder_cache

    def reorder_cache(*a, **k):
        if st.cache_ref is not None and st.cache_ref() is cache:
            _sync_back(st)
            st.cache_ref = None
        st.reordered = weakref.ref(cache)
        return orig(*a, **k)
    try:
        cache.reorder_cache = reorder_cache
        cache._gptq_reorder_watched = True
    except Exception:         # (a cache type that refuses attributes: generate()'s num_beams check still covers HF's own beam search)
        pass


def _sync_in(st, cache, T, npad=None):
    """copy the caller's cache (T entries per row, the first npad[r] of row r being pads) into the engine's static cache."""
    from .decode import _cache_layer_kv
    eng = st.engine
    _watch_reorder(st, cache)
    npad = [0] * eng.batch if npad is None else npad
    if T:
        for li in range(len(eng.layers)):
            k, v = _cache_layer_kv(cache, li)
            for r, n in enumerate(npad):
                if T - n > 0:
                    eng.kcb[li, r, :T - n].copy_(k[r, :, n:T].transpose(0, 1).reshape(T - n, -1))
                    eng.vcb[li, r, :T - n].copy_(v[r, :, n:T].transpose(0, 1).reshape(T - n, -1))
    eng.pos.copy_(torch.tensor([T - n for n in npad], dtype=torch.int64), non_blocking=False)
    st.cache_ref, st.hf_len, st.pos, st.npad = weakref.ref(cache), T, T, (npad if any(npad) else None)


def _start_index(st, position_ids, kw):
    """index in the caller's cache at which the incoming tokens start, when the call says so (cache_position wins; position_ids of row 0
    are that row's TOKEN count, the cache index lies its left pads further), else None.  One host sync."""
    cp = kw.get('cache_position')
    if torch.is_tensor(cp) and cp.numel():
        return int(cp.reshape(-1)[0])
    if torch.is_tensor(position_ids) and position_ids.numel():
        return int(position_ids.reshape(-1)[0]) + (st.npad[0] if st.npad is not None else 0)
    return None


def _left_pads(attention_mask, T):
    """pads per row of a [B, T + 1] mask whose zeros form a left prefix of every row (what a tokenizer with padding_side='left' and HF
    generate produce), or None when the mask is anything else (holes, right padding, a fully masked row).  One host sync."""
    m = attention_mask != 0
    npad = (~m).sum(1)
    form = m == (torch.arange(T + 1, device=m.device)[None, :] >= npad[:, None])
    flat = torch.cat([npad, form.all().reshape(1).to(npad.dtype)]).tolist()
    if not flat[-1] or max(flat[:-1]) > T:
        return None
    return [int(n) for n in flat[:-1]]


def _t_max_cap(model, B):
    """longest static cache an engine of B rows may hold: the model's positions, 8192, and the byte budget"""
    cfg = model.config
    per_token = len(model.model.layers) * B * cfg.hidden_size * 2 * 2            # K and V, fp16, all layers, all rows
    by_budget = max(256, CACHE_BUDGET // per_token // 128 * 128)
    return int(min(max(getattr(cfg, 'max_position_embeddings', 2048), 256), 8192, by_budget))


def _engine_for(model, st, B, sig, need=0):
    """the engine of this batch size (built and captured on first use), or None when the model has no route for it.  need: tokens the
    engine's cache must hold NOW (the row length + 1); the cache is sized for max(need, the running generate()'s length hint) rounded up to 256,
    at least 512, at most _t_max_cap -- and rebuilt larger when a sequence outgrows it (grow on demand)."""
    from .decode import DecodeEngine
    if st.sig != sig:
        st.engines.clear()
        st.declined.clear()
        st.engine, st.cache_ref, st.released = None, None, False
        if not _eligible_model(model):      # (walks all layers: only when the weights changed identity, not once per token)
            return None, sig
    elif model.training:
        return None, sig
    if B in st.declined:
        return None, sig
    cap = _t_max_cap(model, B)
    if need > cap:
        return None, sig
    eng = st.engines.get(B)
    grown = 0
    if eng is not None and need > eng.t_max:      # the sequence outgrew this engine's cache: one twice as long takes over (the caller re-syncs)
        grown = 2 * eng.t_max
        if st.engine is eng:
            _sync_back(st)
            st.engine, st.cache_ref = None, None
        del st.engines[B]
        eng = None
        torch.cuda.empty_cache()
    if eng is None:
        if RELEASE_CHECKPOINT and not st.released:      # memory mode: from the first decode step on ONE copy of the packed weights
            from . import release_checkpoint
            release_checkpoint(model)
            st.released = True
            sig = _signature(model)     # the placeholders are new tensor objects
        want = max(need + 256, st.len_hint, grown, 512)
        t_max = int(min(cap, -(-want // 256) * 256))

        def evict_others():
            for old in [k for k in st.engines if k != B]:
                if st.engines[old] is st.engine:
                    _sync_back(st)
                    st.engine, st.cache_ref = None, None
                del st.engines[old]
            torch.cuda.empty_cache()
        try:
            try:
                eng = DecodeEngine(model, t_max=t_max, batch=B).capture()
            except torch.cuda.OutOfMemoryError:
                #  the eager path would have worked: drop the other engines, try once more, else this batch size goes eager
                eng = None
                evict_others()
                eng = DecodeEngine(model, t_max=t_max, batch=B).capture()
        except (NotImplementedError, torch.cuda.OutOfMemoryError):
            eng = None
            torch.cuda.empty_cache()
            st.declined.add(B)
            st.sig = sig
            return None, sig
        while len(st.engines) >= MAX_ENGINES:
            old = next(iter(st.engines))
            if st.engines[old] is st.engine:
                _sync_back(st)
                st.engine, st.cache_ref = None, None
            del st.engines[old]
        st.sig = sig
    else:
        del st.engines[B]               # re-insert: most recently used last
    st.engines[B] = eng
    return eng, sig


def _engine_forward(model, st, input_ids, cache, attention_mask, position_ids, kw):
    """one token per row through the DecodeEngine of this batch size, or None when this call has to go the eager way."""
    B = input_ids.shape[0]
    if st.reordered is not None and st.reordered() is cache:
        return None          # its caller permutes the rows in place (beam search): eager
    T = _cache_len(cache)
    if T is None:
        return None
    was = st.engine
    tracked = was is not None and st.cache_ref is not None and st.cache_ref() is cache and T == st.hf_len and was.batch == B
    pos = st.pos if tracked else T
    eng, sig = _engine_for(model, st, B, _signature(model), need=pos + 1)
    if eng is None:
        return None
    if tracked and st.engine is not eng:         # a rebuilt, larger engine: the tracked cache was completed and released -- start from it again
        tracked = False
        T = _cache_len(cache)
        pos = T
    if attention_mask is not None and (attention_mask.dim() != 2 or attention_mask.shape[0] != B or attention_mask.shape[-1] != pos + 1):
        return None          # a mask that is not "everything so far" (4-D masks, other lengths): eager
    if pos + 1 > eng.t_max:
        return None
    if not tracked:
        # once per sequence (host sync): zeros in the mask must be left padding, and explicit positions must continue the rows
        npad = None
        if attention_mask is not None:
            npad = _left_pads(attention_mask, T)
            if npad is None:
                return None
        cp = kw.get('cache_position')
        if torch.is_tensor(cp) and cp.numel() and int(cp.reshape(-1)[0]) != T:
            return None
        if torch.is_tensor(position_ids) and position_ids.numel():
            want = torch.tensor([T - n for n in (npad or [0] * B)], device=position_ids.device, dtype=position_ids.dtype)
            got = position_ids.reshape(position_ids.shape[0], -1)[:, 0]
            if got.numel() not in (1, B) or not bool((got == want).all()):
                return None
        if st.cache_ref is not None:
            _sync_back(st)   # another sequence was being tracked: leave its cache complete before the engine moves on
        st.engine = eng
        _sync_in(st, cache, T, npad)
    logits = eng.decode(input_ids.reshape(B))
    st.pos += 1
    st.steps += 1
    from transformers.modeling_outputs import CausalLMOutputWithPast
    return CausalLMOutputWithPast(loss=None, logits=logits.view(B, 1, -1).clone(), past_key_values=cache)


# ---- greedy generate without a host round trip per token (round 6) ----
# model.generate(ids[B <= 16, T], do_sample=False, max_new_tokens=N) (left-padded mask allowed) with nothing that looks at the scores between the steps is what engine_generate
# (quant/decode.py) does on the device: the prompt and the first token through HF's own generate (one new token), then ONE hipGraph replay per token whose argmax feeds the next
# replay, the host looking at the stream every 16 tokens only (EOS).  HF's loop costs ~170 us of host work per token on top of the same
# replays (bench.py: drop_in_generate 784 tok/s against 920-937 for the engine under the reference's protocol).  The tokens are the ones HF's
# loop picks: both take the argmax of the logits the SAME engine step writes.  SAMPLING (llama_inference.py:119-127's own call) the same way: HF's warpers and
# torch.multinomial captured behind the step -- the same draws for a seed, the generator left where HF's loop leaves it (GPTQ_SAMPLE_FAST=0: off).  Anything this
# path does not reproduce to the letter --
# warpers other than temperature / top-k / top-p, processors, criteria, streamers, score outputs, more than 16 rows, masks with holes -- takes HF's loop as before.  Batches: a row that has its EOS gets the pad
# token from then on, the call ends when every row has one (HF's own rule); the rows never see each other, so the stream is cut and padded afterwards.  GPTQ_GREEDY_FAST=0: off.
GREEDY_FAST = os.environ.get('GPTQ_GREEDY_FAST', '1') != '0'
SAMPLE_FAST = os.environ.get('GPTQ_SAMPLE_FAST', '1') != '0'
_GREEDY_KW = {'input_ids', 'inputs', 'do_sample', 'max_new_tokens', 'max_length', 'min_length', 'min_new_tokens', 'eos_token_id', 'pad_token_id',
              'attention_mask', 'use_cache', 'num_beams', 'temperature', 'top_p', 'top_k'}
# generation_config fields that must sit at their neutral value (name, neutral values)
_GREEDY_NEUTRAL = (('num_return_sequences', (None, 1)), ('repetition_penalty', (None, 1.0)), ('no_repeat_ngram_size', (None, 0)),
                   ('encoder_no_repeat_ngram_size', (None, 0)), ('bad_words_ids', (None,)), ('force_words_ids', (None,)), ('renormalize_logits', (None, False)),
                   ('constraints', (None,)), ('forced_bos_token_id', (None,)), ('forced_eos_token_id', (None,)), ('remove_invalid_values', (None, False)),
                   ('exponential_decay_length_penalty', (None,)), ('suppress_tokens', (None,)), ('begin_suppress_tokens', (None,)), ('sequence_bias', (None,)),
                   ('guidance_scale', (None, 1, 1.0)), ('penalty_alpha', (None,)), ('output_scores', (None, False)), ('output_logits', (None, False)),
                   ('output_attentions', (None, False)), ('output_hidden_states', (None, False)), ('return_dict_in_generate', (None, False)),
                   ('stop_strings', (None,)), ('prompt_lookup_num_tokens', (None,)), ('max_time', (None,)), ('cache_implementation', (None,)),
                   ('num_beam_groups', (None, 1)), ('dola_layers', (None,)), ('assistant_early_exit', (None,)), ('watermarking_config', (None,)),
                   ('token_healing', (None, False)), ('low_memory', (None, False)), ('compile_config', (None,)))


def _greedy_fast(model, st, orig_generate, args, kwargs):
    """the sequences [B, T + n] model.generate would return, or None when this call is not the plain greedy case (HF's loop takes it)."""
    if not (ENABLED and GREEDY_FAST) or model.training or getattr(model, '_gptq_engine_disabled', False):
        return None
    if len(args) > 1 or not set(kwargs) <= _GREEDY_KW:
        return None
    if getattr(model, 'generation_config', None) is None:
        return None
    # the generation config of THIS call as HF resolves it (kwargs over the model's config over the library's defaults -- an unset field of
    # model.generation_config reads None, and e.g. top_k = 50 only appears here)
    try:
        gc, _ = model._prepare_generation_config(None, **{k: v for k, v in kwargs.items() if k not in ('input_ids', 'inputs')})
    except Exception:
        return None
    get = lambda name: getattr(gc, name, None)
    if int(get('num_beams') or 1) != 1 or get('use_cache') is False:
        return None
    for name, neutral in _GREEDY_NEUTRAL:
        if getattr(gc, name, None) not in neutral:
            return None
    sample = bool(get('do_sample'))
    warpers, wkey = [], None
    if sample:
        # HF's warper chain for this call (generation/utils.py _get_logits_processor, the do_sample branch, in its order): temperature, top-k, top-p; any
        # other warper configured -> HF's loop.  GPTQ_SAMPLE_FAST=0: sampling always takes HF's loop.
        if not SAMPLE_FAST or not hasattr(torch.Generator, 'get_offset') or not hasattr(torch.Generator, 'set_offset'):
            return None                   # (the generator's offset is read and set below: torch >= 2.2)
        for name, neutral in (('min_p', (None,)), ('typical_p', (None, 1.0)), ('epsilon_cutoff', (None, 0.0)), ('eta_cutoff', (None, 0.0)), ('top_h', (None,))):
            if getattr(gc, name, None) not in neutral:
                return None
        from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
        temp, top_k, top_p = get('temperature'), get('top_k'), get('top_p')
        try:
            if temp is not None and temp != 1.0:
                warpers.append(TemperatureLogitsWarper(temp))
            if top_k is not None and top_k != 0:
                warpers.append(TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=1))
            if top_p is not None and top_p < 1.0:
                warpers.append(TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=1))
        except (ValueError, TypeError):
            return None                   # (values HF itself rejects: let it say so)
        wkey = (temp, top_k, top_p)
    ids = args[0] if args else kwargs.get('input_ids', kwargs.get('inputs'))
    dev = next(model.parameters()).device
    if not torch.is_tensor(ids) or ids.dim() != 2 or not 1 <= ids.shape[0] <= MAX_BATCH or ids.shape[1] < 2 or ids.dtype != torch.int64 or ids.device != dev or dev.type != 'cuda':
        return None
    B, T = int(ids.shape[0]), int(ids.shape[1])
    mask = kwargs.get('attention_mask')
    npad = None
    if mask is not None:                 # all ones, or zeros as a LEFT-padding prefix of each row (one host sync)
        if not torch.is_tensor(mask) or tuple(mask.shape) != (B, T) or mask.device != dev:
            return None
        npad = _left_pads(mask, T - 1)
        if npad is None:
            return None
    new = get('max_new_tokens')          # (wins over max_length, as in HF)
    if new is None:
        new = int(get('max_length') or 0) - T
    new = int(new)
    if new < 1:
        return None
    eos = get('eos_token_id')
    eos = None if eos is None else [int(e) for e in (eos if isinstance(eos, (list, tuple)) else [eos])]
    pad = get('pad_token_id')
    if mask is None and pad is not None and bool((ids == int(pad)).any()):
        return None                       # HF derives a mask from the pad tokens it finds in the prompt: its loop
    if B > 1 and eos and pad is None:
        pad = eos[0]                      # (HF's own default, with a warning)
    # min_length / min_new_tokens only ever change a step whose argmax is an EOS token (HF sets those logits to -inf until then): the stream is
    # generated without the mask, and if an EOS shows up before the minimum the call is handed to HF's loop after all (nothing was returned yet)
    min_new = max(int(get('min_new_tokens') or 0), int(get('min_length') or 0) - T, 0) if eos else 0
    eng, _ = _engine_for(model, st, B, _signature(model), need=T + new + 1)
    if eng is None or eng.batch != B or T + new + 1 > eng.t_max:
        return None
    if st.cache_ref is not None:          # a sequence some caller steps by hand: its cache is completed before the engine moves on
        _sync_back(st)
    st.engine, st.cache_ref = None, None
    rng0 = torch.cuda.get_rng_state(dev) if sample else None          # (a call handed back to HF's loop must draw what it would have drawn)
    with torch.no_grad():
        # the prompt and the FIRST token are HF's own: its generate for one new token (its prefill inputs, its processors on that step)
        kw1 = {k: v for k, v in kwargs.items() if k not in ('input_ids', 'inputs', 'max_new_tokens', 'max_length', 'min_length', 'min_new_tokens')}
        if min_new >= 1:
            kw1['min_new_tokens'] = 1         # (all of the caller's minimum that this one step can see; the full value would only draw HF's "unfeasible" warning)
        out = orig_generate(ids, max_new_tokens=1, return_dict_in_generate=True, **kw1)
        cache = getattr(out, 'past_key_values', None)
        if cache is None or tuple(out.sequences.shape) != (B, T + 1) or _cache_len(cache) != T:
            if sample:
                torch.cuda.set_rng_state(rng0, dev)
            return None
        first = out.sequences[:, T].clone()
        st.engine = eng
        _sync_in(st, cache, T, npad)          # the rows' K / V behind their pads, per-row positions
        st.engine, st.cache_ref = None, None  # (nobody tracks this cache: it is dropped below)
        del out, cache
        if sample:
            graph = eng.sample_graphs.get(wkey) or eng.capture_sample_rows(wkey, warpers)
            gen_state = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
            off0 = gen_state.get_offset()
        else:
            if eng.greedy_rows_graph is None:
                eng.capture_greedy_rows()
            graph = eng.greedy_rows_graph
        eng.stepc.zero_()
        eng.ids.copy_(first)
        eos_t = torch.tensor(eos, device=dev, dtype=torch.int64) if eos else None
        done = 1                              # tokens per row so far: `first`, then rows 0 .. done - 2 of stream_rows
        stream = lambda: torch.cat([first[None, :], eng.stream_rows[:done - 1]], 0)           # [done, B]
        all_hit = lambda: eos_t is not None and bool(torch.isin(stream(), eos_t).any(0).all())
        while done < new and not all_hit():
            burst = min(16, new - done)                       # the host looks at the stream every 16 tokens only
            for _ in range(burst):
                graph.replay()
            done += burst
            st.steps += burst
        gen = stream().t().contiguous()                       # [B, done]
        length = done
        if eos_t is not None:
            hit = torch.isin(gen, eos_t)
            anyhit = hit.any(1)
            firsthit = torch.where(anyhit, hit.int().argmax(1), torch.full_like(anyhit, done, dtype=torch.int64))          # per row: index of its first EOS (done: none)
            info = torch.cat([firsthit, anyhit.to(torch.int64)]).tolist()
            fh, ah = info[:B], info[B:]
            if any(a and f < min_new for f, a in zip(fh, ah)):
                if sample:
                    torch.cuda.set_rng_state(rng0, dev)
                return None                                   # HF's loop would have masked that EOS: it takes the call
            length = max((f + 1 if a else done) for f, a in zip(fh, ah))          # HF stops when every row has its EOS
            if any(ah) and B > 1:                             # behind a row's EOS HF writes the pad token while the other rows go on
                col = torch.arange(done, device=dev)[None, :]
                gen = torch.where(col > firsthit[:, None], torch.full_like(gen, int(pad)), gen)
        gen = gen[:, :length]
        if sample and done > 1:
            # the generator stands where HF's loop would have left it: `length - 1` draws behind the first token's (the bursts may have drawn more)
            per_step = (gen_state.get_offset() - off0) // (done - 1)
            gen_state.set_offset(off0 + (length - 1) * per_step)
    return torch.cat([ids, gen.to(ids.dtype)], 1)


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
                and input_ids.dim() == 2 and 1 <= input_ids.shape[0] <= MAX_BATCH and input_ids.shape[1] == 1 and input_ids.is_cuda
                and not torch.is_grad_enabled() and not kw.get('output_attentions') and not kw.get('output_hidden_states')
                and not torch.cuda.is_current_stream_capturing())
        if fast:
            out = _engine_forward(self, st, input_ids, past_key_values, attention_mask, position_ids, kw)
            if out is not None:
                return out if kw.get('return_dict', True) is not False else (out.logits, out.past_key_values)
        if st.cache_ref is not None and st.cache_ref() is past_key_values and st.pos > st.hf_len:
            # the eager path continues the tracked cache: it must hold the engine's tokens up to where THIS call starts (see the
            # module docstring: explicit positions decide, a bare call needs the complete cache)
            _sync_back(st, _start_index(st, position_ids, kw))
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
            gc = kwargs.get('generation_config') or getattr(self, 'generation_config', None)
            pick = lambda name, default: kwargs[name] if kwargs.get(name) is not None else (getattr(gc, name, None) if gc is not None else None) or default
            beams = int(pick('num_beams', 1) or 1)
            # how far this call can get: the engine sizes (or grows) its static K / V cache for it instead of for max_position_embeddings
            ids = args[0] if args and torch.is_tensor(args[0]) else kwargs.get('input_ids', kwargs.get('inputs'))
            plen = int(ids.shape[-1]) if torch.is_tensor(ids) and ids.dim() == 2 else 0
            new = pick('max_new_tokens', 0)
            st.len_hint = plen + int(new) + 1 if new else int(pick('max_length', 0) or 0) + 1
            was_off = getattr(self, '_gptq_engine_disabled', False)
            if beams > 1:                     # beams: the cache is permuted in place after every step -- eager (see the module docstring)
                self._gptq_engine_disabled = True
            try:
                if beams == 1:
                    fast = _greedy_fast(self, st, orig_generate, args, kwargs)
                    if fast is not None:
                        return fast
                return orig_generate(*args, **kwargs)
            finally:
                self._gptq_engine_disabled = was_off
                st.len_hint = 0
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


def drop_decode_engines(model):
    """flush, then free every engine the hook built (graphs, static K/V caches); the next one-token forward builds what it needs again"""
    st = getattr(model, '_gptq_engine_state', None)
    if st is None:
        return
    flush_decode_engine(model)
    st.engines.clear()
    st.declined.clear()
    st.engine, st.sig, st.cache_ref, st.npad = None, None, None, None


def engine_steps(model):
    """number of forwards the engine answered so far (tests / bench)."""
    st = getattr(model, '_gptq_engine_state', None)
    return st.steps if st is not None else 0
