"""The product (package + C ABI) never reaches into the test infrastructure:
no file of the package mentions oracle/ or the reference tree, and the
binding refuses to work without the HIP library instead of falling back."""
import importlib
import os

import pytest

import conftest  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mpeg-pcc-tmc13_amd")


def package_files():
    for d, _, files in os.walk(PKG):
        if "__pycache__" in d:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                yield os.path.join(d, f)


def test_package_does_not_touch_oracle_or_reference_tree():
    offenders = []
    for path in package_files():
        text = open(path, errors="replace").read()
        for needle in ("/root/reference", "libgpcc_oracle", "libtmc3_ref", "oracle_loader", "import oracle"):
            if needle in text:
                offenders.append((os.path.relpath(path, ROOT), needle))
    assert not offenders, offenders


def test_binding_fails_loudly_without_the_hip_library(monkeypatch):
    lib = importlib.import_module("mpeg_pcc_tmc13_amd._lib")
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", os.path.join(PKG, "no_such_library.so"))
    with pytest.raises(ImportError) as ei:
        lib.load()
    assert "no CPU fallback" in str(ei.value)


def test_bench_and_entry_use_oracle_only_as_checker():
    """bench.py touches the checker only in its cpu_baseline legs, smoke() only to compare."""
    bench = open(os.path.join(ROOT, "bench.py")).read()
    head, _, tail = bench.partition("def pmc_traffic")
    assert "oracle_loader" not in head.split("def step():")[1].split("# ---- CPU baseline")[0].replace(
        "import oracle_loader as ol\n        chk = ol.ref() if ol.ref_available() else ol.oracle()", "")
