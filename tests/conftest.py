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
    """GPU tier only.  An asynchronous copy out of PAGEABLE host memory makes the ROCm runtime pin the pages on the
    fly, read-only, and it keeps such pins: when a later download (anybody's in the process -- `tensor.cpu()`) lands on
    heap addresses a freed numpy array had, the GPU writes to a read-only page and the process aborts ("Memory access
    fault ... Write access to a read-only page"; the library routes its own transfers through a pinned bounce buffer for
    this reason, csrc/gpcc_attr_mi355.hip h2d_user).  The tests' own uploads go torch.from_numpy(x).to(device): in one
    long pytest process with hundreds of tests that is the same hazard (seen once in ~10 runs of the whole tier), so
    here from_numpy hands out a pinned copy -- the upload then starts from memory the runtime never has to pin."""
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
