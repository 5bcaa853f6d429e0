"""CPU suite: the C-ABI library loads without a GPU and exports every symbol that
include/cnf2hip.h declares; with no device it refuses to create a context (no fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from cnf2freq_amd import capi
    return capi.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cnf2hip.h")).read()
    return sorted(set(re.findall(r"\b(cnf2_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    from cnf2freq_amd import capi
    decl = declared_symbols()
    assert decl, "no declarations parsed"
    for name in decl:
        assert hasattr(lib, name), "missing export: " + name
    assert sorted(capi.SYMBOLS) == decl


def test_host_library_exports_its_header(lib):
    """libcnf2host.so (the host side of a run, include/cnf2host.h) loads without a GPU and exports what it declares."""
    from cnf2freq_amd import host
    text = open(os.path.join(ROOT, "include", "cnf2host.h")).read()
    decl = sorted(set(re.findall(r"\b(cnf2h_[a-z0-9_]+)\s*\(", text)))
    L = host.load()
    for name in decl:
        assert hasattr(L, name), "missing export: " + name
    assert sorted(host.SYMBOLS) == decl


def test_no_device_is_an_error_not_a_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cnf2freq_amd import capi
    assert lib.cnf2_device_count() == 0
    with pytest.raises(capi.Cnf2Error):
        capi.Context(0)
