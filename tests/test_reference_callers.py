"""The acceptance callers of the drop-in boundary (BASELINE north_star: "so llama_inference.py runs unmodified").

CPU leg (``-m "not gpu"``, needs the read-only upstream tree, skipped where it is absent): the reference's OWN
``llama_inference.load_quant`` (/root/reference/llama_inference.py:27-72) and ``llama.load_quant``
(/root/reference/llama.py:279-325) are imported unmodified with this package answering ``import quant`` / ``import gptq``
(``utils`` is the reference's own; ``texttable`` / ``toml`` are absent from the image and stubbed), and run on a tiny
config + a checkpoint saved in the reference's state_dict format: module surgery, checkpoint keys, buffer contents.
The warm-ups need a GPU (the reference moves buffers with ``.cuda()``, quant_linear.py:409), so the CPU leg passes
``warmup_autotune=False``.

GPU leg (``-m gpu``): the default load path, warm-ups included.  /root/reference does not exist on the GPU box, so
``load_quant_flow`` below restates the reference function line by line (each step cited) on top of the package; when
the tree IS present the real function is used instead.  Then ``model.to(DEV)`` and ``model.generate`` as
llama_inference.py:117-127 does, and the two warm-ups once more on their own with the buffer placement the reference
leaves behind (``autotune_warmup_fused`` ends with ``fused2cpu``, fused_mlp.py:283-286).
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

import quant
from quant import decode as D

REF = '/root/reference'
TINY = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
            vocab_size=320, max_position_embeddings=256, rms_norm_eps=1e-6)
WBITS, GROUPSIZE = 4, 128


def _tiny_config_dir(tmp_path):
    from transformers import LlamaConfig
    cfg = dict(D.LLAMA_7B)
    cfg.update(TINY)
    d = tmp_path / 'tiny-llama'
    d.mkdir()
    LlamaConfig(**cfg).save_pretrained(str(d))
    return str(d)


def _tiny_checkpoint(tmp_path, config_dir, seed=0):
    """state_dict of a model whose decoder linears are QuantLinear (before the fusions) -- the checkpoint format
    llama.py:490-500 saves: the reference's keys, the reference's shapes."""
    from transformers import LlamaConfig, LlamaForCausalLM
    config = LlamaConfig.from_pretrained(config_dir)
    model = LlamaForCausalLM(config).half().eval()
    layers = D.find_layers(model)
    layers.pop('lm_head', None)
    quant.make_quant_linear(model, layers, WBITS, GROUPSIZE)
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, m in model.named_modules():
            if isinstance(m, quant.QuantLinear):
                m.qweight.copy_(torch.randint(-2**31, 2**31 - 1, m.qweight.shape, dtype=torch.int32, generator=gen))
                m.qzeros.copy_(torch.randint(-2**31, 2**31 - 1, m.qzeros.shape, dtype=torch.int32, generator=gen))
                m.scales.copy_((torch.rand(m.scales.shape, generator=gen) * 0.01 + 0.001).half())
    path = str(tmp_path / 'tiny-4bit-128g.pt')
    sd = model.state_dict()
    torch.save(sd, path)
    return path, sd


def _import_reference(name):
    """import /root/reference/<name>.py unmodified with this package shadowing `quant` and `gptq`."""
    for stub in ('texttable', 'toml'):
        if stub not in sys.modules:
            try:
                __import__(stub)
            except ImportError:
                m = types.ModuleType(stub)
                if stub == 'texttable':
                    m.Texttable = type('Texttable', (), {})
                sys.modules[stub] = m
    import gptq as product_gptq          # gptq-for-llama_amd/gptq.py (conftest puts the package first on sys.path)
    assert 'gptq-for-llama_amd' in os.path.abspath(product_gptq.__file__)
    assert 'gptq-for-llama_amd' in os.path.abspath(quant.__file__)
    if REF not in sys.path:
        sys.path.append(REF)             # behind the package: only `utils`, `llama`, `llama_inference` resolve there
    mod = __import__(name)
    assert os.path.abspath(mod.__file__).startswith(REF)
    assert mod.quant is quant
    return mod


def _restore_torch_defaults(saved):
    torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_, torch.nn.init.normal_ = saved
    torch.set_default_dtype(torch.float)


def _check_loaded_model(model, sd):
    from quant.fused_attn import QuantLlamaAttention
    from quant.fused_mlp import QuantLlamaMLP
    from quant.triton_norm import TritonLlamaRMSNorm
    for layer in model.model.layers:
        assert type(layer.self_attn) is QuantLlamaAttention
        assert type(layer.self_attn.qkv_proj) is quant.QuantLinear and type(layer.self_attn.o_proj) is quant.QuantLinear
        assert type(layer.mlp) is QuantLlamaMLP and type(layer.mlp.down_proj) is quant.QuantLinear
        assert type(layer.input_layernorm) is TritonLlamaRMSNorm and type(layer.post_attention_layernorm) is TritonLlamaRMSNorm
    assert type(model.model.norm) is TritonLlamaRMSNorm
    assert type(model.lm_head) is torch.nn.Linear                     # llama_inference.py:46-48: lm_head stays dense
    # checkpoint contents survived load_state_dict + the fusions (qkv = cat along N, fused_attn.py:177-188)
    a0 = model.model.layers[0].self_attn
    pre = 'model.layers.0.self_attn.'
    for buf, dim in (('qweight', 1), ('qzeros', 1), ('scales', 1)):
        want = torch.cat([sd[pre + p + '_proj.' + buf] for p in 'qkv'], dim=dim)
        assert torch.equal(getattr(a0.qkv_proj, buf).cpu(), want)
    m0 = model.model.layers[0].mlp
    assert torch.equal(m0.gate_proj_qweight.cpu(), sd['model.layers.0.mlp.gate_proj.qweight'])
    assert torch.equal(m0.up_proj_scales.cpu(), sd['model.layers.0.mlp.up_proj.scales'])
    assert torch.equal(m0.down_proj.qzeros.cpu(), sd['model.layers.0.mlp.down_proj.qzeros'])
    assert model.seqlen == 2048


needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason='the upstream tree is not present on this machine')


@needs_reference
@pytest.mark.parametrize('module', ['llama_inference', 'llama'])
def test_reference_load_quant_runs_unmodified_on_the_package(tmp_path, module, capsys):
    saved = (torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_, torch.nn.init.normal_)
    try:
        cfg_dir = _tiny_config_dir(tmp_path)
        ckpt, sd = _tiny_checkpoint(tmp_path, cfg_dir)
        ref = _import_reference(module)
        model = ref.load_quant(cfg_dir, ckpt, WBITS, GROUPSIZE, warmup_autotune=False)
    finally:
        _restore_torch_defaults(saved)
    out = capsys.readouterr().out
    assert 'Loading model ...' in out and 'Done.' in out
    _check_loaded_model(model, sd)
    # find_layers(model, [quant.QuantLinear]) is an exact type match (utils/modelutils.py:8; llama.py:269)
    q = ref.find_layers(model, [quant.QuantLinear])
    assert len(q) == 3 * TINY['num_hidden_layers']                    # qkv_proj, o_proj, down_proj per layer


def load_quant_flow(model, checkpoint, wbits, groupsize=-1, fused_mlp=True, eval=True, warmup_autotune=True):
    """llama_inference.py:27-72 restated on the package (the tree is absent on the GPU box)."""
    from transformers import LlamaConfig, LlamaForCausalLM
    config = LlamaConfig.from_pretrained(model)                        # :29
    torch.set_default_dtype(torch.half)                                # :38-41
    try:
        model = LlamaForCausalLM(config)
    finally:
        torch.set_default_dtype(torch.float)                           # :42
    if eval:
        model = model.eval()                                           # :43-44
    layers = D.find_layers(model)                                      # :45
    for name in ['lm_head']:
        if name in layers:
            del layers[name]                                           # :46-48
    quant.make_quant_linear(model, layers, wbits, groupsize)           # :49
    print('Loading model ...')
    model.load_state_dict(torch.load(checkpoint), strict=False)        # :58-60
    if eval:
        quant.make_quant_attn(model)                                   # :61-63
        quant.make_quant_norm(model)
        if fused_mlp:
            quant.make_fused_mlp(model)                                # :64-65
    if warmup_autotune:
        quant.autotune_warmup_linear(model, transpose=not (eval))      # :66-67
        if eval and fused_mlp:
            quant.autotune_warmup_fused(model)                         # :68-69
    model.seqlen = 2048
    print('Done.')
    return model


@pytest.mark.gpu
def test_default_load_path_with_warmups_then_generate(tmp_path, capsys):
    """load_quant with warmup_autotune=True (the default, llama_inference.py:27), model.to(DEV), generate (:117-127)."""
    saved = (torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_, torch.nn.init.normal_)
    DEV = torch.device('cuda:0')
    try:
        cfg_dir = _tiny_config_dir(tmp_path)
        ckpt, sd = _tiny_checkpoint(tmp_path, cfg_dir)
        load = _import_reference('llama_inference').load_quant if os.path.isdir(REF) else load_quant_flow
        model = load(cfg_dir, ckpt, WBITS, GROUPSIZE)
    finally:
        _restore_torch_defaults(saved)
    out = capsys.readouterr().out
    assert 'unique KN Linear values' in out and 'unique fused mlp KN values' in out and 'Done.' in out
    _check_loaded_model(model, sd)
    # what the reference leaves behind: module buffers where they were (CPU), fused MLP buffers back on the CPU (fused2cpu)
    assert model.model.layers[0].self_attn.qkv_proj.qweight.device.type == 'cpu'
    assert model.model.layers[0].mlp.gate_proj_qweight.device.type == 'cpu'
    model.to(DEV)                                                      # llama_inference.py:117
    assert model.model.layers[1].mlp.up_proj_qzeros.device.type == 'cuda'
    ids = torch.randint(0, TINY['vocab_size'], (1, 9), generator=torch.Generator().manual_seed(1)).to(DEV)
    with torch.no_grad():
        gen = model.generate(ids, do_sample=True, min_length=10, max_length=24, top_p=0.95, temperature=0.8)
        greedy = model.generate(ids, do_sample=False, max_new_tokens=12, min_new_tokens=12)     # (a random model may pick EOS at once)
        greedy2 = model.generate(ids, do_sample=False, max_new_tokens=12, min_new_tokens=12)
    assert gen.shape[0] == 1 and 10 <= gen.shape[1] <= 24 and torch.equal(gen[:, :9], ids)
    assert greedy.shape == (1, 21) and torch.equal(greedy, greedy2)
    assert int(greedy.max()) < TINY['vocab_size']


# llama_inference.py:75-128 restated (the tree is absent on the GPU box): the script's own argument names, its load / to(DEV) / tokenizer / generate /
# decode / print sequence -- run as a PROCESS below.  `load_quant` = this file's restatement of :27-72 (or the reference's own function where the tree is).
MAIN_FLOW = r"""
import argparse, os, sys
import torch
sys.path.insert(0, {tests!r}); sys.path.insert(0, {pkg!r}); sys.path.insert(0, {root!r})
import test_reference_callers as T
from transformers import AutoTokenizer
DEV = torch.device('cuda:0')                                            # utils/modelutils.py:4
parser = argparse.ArgumentParser()                                      # llama_inference.py:78-103
parser.add_argument('model', type=str)
parser.add_argument('--wbits', type=int, default=16, choices=[2, 3, 4, 8, 16])
parser.add_argument('--groupsize', type=int, default=-1)
parser.add_argument('--load', type=str, default='')
parser.add_argument('--text', type=str)
parser.add_argument('--min_length', type=int, default=10)
parser.add_argument('--max_length', type=int, default=50)
parser.add_argument('--top_p', type=float, default=0.95)
parser.add_argument('--temperature', type=float, default=0.8)
parser.add_argument('--device', type=int, default=-1)
parser.add_argument('--fused_mlp', action='store_true')
parser.add_argument('--no_fused_mlp', dest='fused_mlp', action='store_false')
parser.set_defaults(fused_mlp=True)
args = parser.parse_args()
load_quant = T._import_reference('llama_inference').load_quant if os.path.isdir(T.REF) else T.load_quant_flow
model = load_quant(args.model, args.load, args.wbits, args.groupsize, fused_mlp=args.fused_mlp)     # :110-111
model.to(DEV)                                                           # :116
tokenizer = AutoTokenizer.from_pretrained(args.model, use_fast=False)   # :117
input_ids = tokenizer.encode(args.text, return_tensors="pt").to(DEV)    # :118
torch.manual_seed(int(os.environ.get('SEED', '0')))
with torch.no_grad():                                                   # :120-128
    generated_ids = model.generate(input_ids, do_sample=True, min_length=args.min_length, max_length=args.max_length, top_p=args.top_p,
                                   temperature=args.temperature)
print('IDS', [el.item() for el in generated_ids[0]])
print('TEXT', repr(tokenizer.decode([el.item() for el in generated_ids[0]])))
import quant
from quant import engine_hook
print('NATIVE', quant._native.lib() is not None)
print('ENGINE_STEPS', engine_hook.engine_steps(model))
"""


def _tiny_tokenizer(config_dir):
    """a BPE tokenizer of TINY['vocab_size'] entries at most, saved next to the config (AutoTokenizer.from_pretrained(args.model), llama_inference.py:117)"""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.BPE(unk_token='<unk>'))
    tok.pre_tokenizer = pre_tokenizers.Metaspace()
    tok.decoder = decoders.Metaspace()
    corpus = ['this is a llama . the quick brown fox jumps over the lazy dog', 'hello world again and again', 'a b c d e f g h i j k l m n o p q r s t u v w x y z'] * 20
    tok.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=TINY['vocab_size'], special_tokens=['<unk>', '<s>', '</s>']))
    PreTrainedTokenizerFast(tokenizer_object=tok, unk_token='<unk>', bos_token='<s>', eos_token='</s>').save_pretrained(config_dir)


def test_tiny_tokenizer_round_trip(tmp_path):
    """(CPU) the tokenizer leg of the process test below: AutoTokenizer on the model directory, encode / decode"""
    from transformers import AutoTokenizer
    cfg_dir = _tiny_config_dir(tmp_path)
    _tiny_tokenizer(cfg_dir)
    tok = AutoTokenizer.from_pretrained(cfg_dir, use_fast=False)
    ids = tok.encode('this is a llama', return_tensors='pt')
    assert ids.shape[0] == 1 and ids.shape[1] >= 2 and int(ids.max()) < TINY['vocab_size']
    assert tok.decode([el.item() for el in ids[0]]).strip() == 'this is a llama'


@pytest.mark.gpu
def test_inference_main_as_a_process(tmp_path):
    """llama_inference.py's `__main__` (:75-128) as a PROCESS with the script's own command line: load_quant with the warm-ups, model.to(DEV), the
    tokenizer of the model directory, sampling generate with the script's arguments, decode, print.  Two runs with one seed print the same text (the
    decode engine behind model.generate is deterministic; sampling draws from torch's seeded generator), another seed still yields a valid text."""
    import subprocess
    saved = (torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_, torch.nn.init.normal_)
    try:
        cfg_dir = _tiny_config_dir(tmp_path)
        ckpt, _ = _tiny_checkpoint(tmp_path, cfg_dir)
    finally:
        _restore_torch_defaults(saved)
    _tiny_tokenizer(cfg_dir)
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    script = tmp_path / 'llama_inference_main.py'
    script.write_text(MAIN_FLOW.format(tests=here, pkg=os.path.join(root, 'gptq-for-llama_amd'), root=root))
    cmd = [sys.executable, str(script), cfg_dir, '--wbits', str(WBITS), '--groupsize', str(GROUPSIZE), '--load', ckpt, '--text', 'this is a llama',
           '--min_length', '12', '--max_length', '32']

    def run(seed):
        env = dict(os.environ, SEED=str(seed))
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        out = r.stdout
        assert 'Loading model ...' in out and 'Done.' in out and 'NATIVE True' in out
        ids = eval([l for l in out.splitlines() if l.startswith('IDS ')][0][4:])
        text = eval([l for l in out.splitlines() if l.startswith('TEXT ')][0][5:])
        assert 12 <= len(ids) <= 32 and max(ids) < TINY['vocab_size'] and isinstance(text, str)
        steps = int([l for l in out.splitlines() if l.startswith('ENGINE_STEPS ')][0].split()[1])
        assert steps >= len(ids) - 6          # the decode engine answered the one-token forwards of generate (the prompt is one prefill call)
        return ids, text
    a, ta = run(0)
    b, tb = run(0)
    assert a == b and ta == tb
    assert ta.lstrip('<s>').strip().startswith('this is a llama')
    run(1)


@pytest.mark.gpu
@pytest.mark.parametrize('transpose', [False, True])
def test_autotune_warmup_linear_walks_every_m(transpose, capsys):
    """quant_linear.py:393-423: every unique (K, N) at M = 1 .. 2048, forward (and the transposed product when
    transpose=True -- the training path, llama.py `eval=False`); buffers of the modules are not moved."""
    model = D.build_random_llama('cpu', bits=WBITS, groupsize=GROUPSIZE, seed=3, fused=True, **TINY)
    quant.autotune_warmup_linear(model, transpose=transpose)
    torch.cuda.synchronize()
    out = capsys.readouterr().out
    assert 'Found 3 unique KN Linear values.' in out and 'Warming up autotune cache ...' in out   # qkv 256x768, o 256x256, down 512x256
    for _, m in model.named_modules():
        if isinstance(m, quant.QuantLinear):
            assert m.qweight.device.type == 'cpu' and m.scales.device.type == 'cpu'


@pytest.mark.gpu
def test_autotune_warmup_fused_round_trip():
    """fused_mlp.py:256-288: fused2cuda on every QuantLlamaMLP, M = 1 .. 2048 through the fused kernel, then fused2cpu;
    a forward afterwards needs model.to(DEV) again, exactly like upstream."""
    from quant.fused_mlp import QuantLlamaMLP
    model = D.build_random_llama('cpu', bits=WBITS, groupsize=GROUPSIZE, seed=4, fused=True, **TINY)
    quant.autotune_warmup_fused(model)
    torch.cuda.synchronize()
    mlps = [m for _, m in model.named_modules() if isinstance(m, QuantLlamaMLP)]
    assert len(mlps) == TINY['num_hidden_layers']
    # upstream quirk kept: fused2cuda() runs on EVERY QuantLlamaMLP (fused_mlp.py:271), fused2cpu() only on the first
    # module of each unique (K, N) (:283-286) -- the others stay on the GPU until model.to(DEV) evens it out
    for i, m in enumerate(mlps):
        for n in QuantLlamaMLP._FUSED:
            assert getattr(m, n).device.type == ('cpu' if i == 0 else 'cuda')
    model.to('cuda:0')
    x = torch.randn((1, 1, TINY['hidden_size']), device='cuda:0').half()
    y = mlps[0](x)
    assert y.shape == x.shape and bool(torch.isfinite(y.float()).all())
