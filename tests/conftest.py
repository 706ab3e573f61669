"""pytest config: registers the `gpu` marker and puts the product package, the oracle and the
repo root on sys.path.  `-m "not gpu"` runs on CPU only; `-m gpu` needs one MI355X."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'gptq-for-llama_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    # a fresh checkout has no built artefacts (they are git-ignored): build them once, exactly as __graft_entry__.build()
    # does (hipcc cross-compiles gfx950 without a GPU).  The product itself never builds or falls back on its own.
    lib = os.path.join(PKG, 'lib', 'libgptq_mi355x.so')
    ora = os.path.join(ROOT, 'oracle', 'libgptq_oracle.so')
    if not (os.path.exists(lib) and os.path.exists(ora)):
        import subprocess
        subprocess.check_call(['make', '-C', os.path.join(PKG, 'csrc'), '-j', str(max(1, os.cpu_count() or 1))])
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])


def pytest_collection_modifyitems(config, items):
    """A gpu-marked test must never silently pass on a box without a GPU."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
