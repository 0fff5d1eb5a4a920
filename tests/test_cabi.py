"""CPU: the C-ABI library builds for gfx950 (cross-compile), loads, and exports every symbol that
include/mte_kernels.h declares.  No kernel is launched."""
import ctypes
import os
import subprocess

import pytest


def test_library_builds_and_exports_every_declared_symbol():
    from mindtheedge_amd import _build, _lib
    path = _build.build()
    assert os.path.exists(path)
    protos = _lib.parse_header()
    assert len(protos) >= 26
    dll = ctypes.CDLL(path)
    for name in protos:
        assert hasattr(dll, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T mte" in l}
    assert exported == set(protos), exported ^ set(protos)
    # the shipped library exports the integration surface only: no development knobs (round-2 verdict item 9)
    assert "mte_debug_set" not in exported and not any(n.startswith("mtei_") for n in exported)


def test_development_build_adds_only_the_knobs():
    """libmte_hip_dev.so = the same sources with -DMTE_DEV: the product's surface + mte_debug_set (+ its per-file setters)."""
    from mindtheedge_amd import _build, _lib
    _build.build()
    assert os.path.exists(_build.DEV_LIB)
    out = subprocess.run(["nm", "-D", "--defined-only", _build.DEV_LIB], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T mte" in l}
    dev = set(_lib.parse_header(dev=True))
    assert dev - set(_lib.parse_header()) == {"mte_debug_set"}
    assert {n for n in exported if not n.startswith("mtei_")} == dev


def test_code_object_targets_gfx950_only():
    from mindtheedge_amd import _build
    path = _build.build()
    data = open(path, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"nvptx"):
        assert other not in data


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mindtheedge_amd import _lib
    fresh = _lib._Lib(path=str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MteError):
        fresh.mte_adam_step(0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 1, 1.0, 0)
