"""The C-ABI library loads on a machine without a GPU and exports every symbol include/caphn.h
declares; the ctypes binding covers exactly that set.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import REPO, PKG

HEADER = os.path.join(REPO, "include", "caphn.h")
LIB = os.path.join(PKG, "caphn", "libcaphn.so")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(caphn_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-j4"], check=True)
    return ctypes.CDLL(LIB)


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_covers_header(lib):
    from caphn import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    loaded = _lib.load()
    assert loaded.caphn_abi_version() == 1          # host-only call


def test_argument_validation_without_gpu(lib):
    """Entry points validate before touching the device: bad arguments return CAPHN_EINVAL."""
    from caphn import _lib
    L = _lib.load()
    assert L.caphn_gemm_f32(0, 1, 0, 4, 4, None, 4, None, 4, None, 4, None, None, 0, 0, 1, None) == -1
    assert L.caphn_colsum_f32(0, 3, None, 3, None, None, None) == -1
    assert L.caphn_decoder_workspace_bytes(None) == 0
    d = _lib.DecoderDims(128, 20, 49, 2048, 200, 200, 200, 9684)
    n = L.caphn_decoder_workspace_bytes(ctypes.byref(d))
    assert 50e6 < n < 400e6                         # saved activations of one step at the canonical size
    assert L.caphn_sumsq_blocks(10) == 1 and L.caphn_sumsq_blocks(8193) == 2
    hp = _lib.AdamHParams(1e-3, 0.9, 0.999, 1e-8, 0)
    assert L.caphn_adam_dense_f32(4, None, None, None, None, None, ctypes.byref(hp), None) == -1


def test_product_path_never_imports_oracle():
    """Nothing under the package may import oracle/ (the oracle is test infrastructure)."""
    bad = []
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "caphn_oracle" in txt:
                    bad.append(os.path.join(root, f))
    assert not bad, bad
