import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def load_package():
    """Import mpeg-pcc-tmc13_amd/ under the alias mpeg_pcc_tmc13_amd."""
    name = "mpeg_pcc_tmc13_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg_dir = os.path.join(ROOT, "mpeg-pcc-tmc13_amd")
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(pkg_dir, "__init__.py"),
        submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libtmc3_ref.so (compiled reference)")


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session", autouse=True)
def _uploads_from_pinned_memory():
    """GPU tier, OPT-IN (GPCC_TEST_PIN_UPLOADS=1).  Round 4 made this the default after ONE silent abort of the tier at a
    `tensor.cpu()` that was never reproduced; it changes what the tests do (torch.from_numpy then copies instead of
    sharing memory) and would hide an out-of-bounds device write just as well as the pageable-pin hazard it was meant
    for.  Since round 5 the tier runs WITHOUT it and with the library's guard bands armed instead
    (GPCC_GUARD=1, csrc/gpcc_attr_mi355.hip: canaries around every device allocation of the library and between the
    sub-allocations of its arena, checked at every synchronisation); profiles/r05_gpu_tier_guarded.txt records those
    runs.  The switch stays for bisecting, should the abort ever come back."""
    if os.environ.get("GPCC_TEST_PIN_UPLOADS", "0") != "1":
        yield
        return
    try:
        import torch
    except ImportError:
        yield
        return
    if not torch.cuda.is_available():
        yield
        return
    orig = torch.from_numpy

    def pinned(a):
        t = orig(a)
        return t.pin_memory() if t.numel() else t
    torch.from_numpy = pinned
    try:
        yield
    finally:
        torch.from_numpy = orig
