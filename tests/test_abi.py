"""The C-ABI library loads without a GPU and exports every symbol include/uds_hip.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from gnn_uds_amd import _lib
from gnn_uds_amd.graph import DrainageGraph, synthetic_drainage_network

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'uds_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(uds_[a-z_0-9]+)\s*\(', text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert _lib.load().uds_abi_version() == _lib.ABI_VERSION == 22


def test_argument_errors_are_reported_not_thrown():
    lib = _lib.load()
    h = ctypes.c_void_p()
    rowptr = np.array([0, 2, 1], dtype=np.int32)           # decreasing rowptr
    col = np.array([0, 1], dtype=np.int32)
    rc = lib.uds_csr_create(rowptr.ctypes.data, col.ctypes.data, 2, 2, 1, ctypes.byref(h))
    assert rc == -22 and b'rowptr' in lib.uds_last_error() and not h.value
    rowptr = np.array([0, 1, 2], dtype=np.int32)
    col = np.array([0, 5], dtype=np.int32)                 # column out of range
    rc = lib.uds_csr_create(rowptr.ctypes.data, col.ctypes.data, 2, 2, 2, ctypes.byref(h))
    assert rc == -22 and b'col[1]=5' in lib.uds_last_error()
    assert lib.uds_dense_act(None, 4, None, 0, 1, None, None, 4, 1, None, None, None, None, None, None) == -22
    assert lib.uds_gat_workspace_floats(10, 3, 8) == 300


@pytest.mark.skipif(torch.cuda.is_available(), reason='CPU-only behaviour')
def test_no_cpu_fallback():
    g = DrainageGraph.from_edges(synthetic_drainage_network(50, 60, 0))
    with pytest.raises(_lib.UdsError):
        _lib.CsrHandle(g.adj)                              # no device -> loud failure, not a CPU path
    with pytest.raises(_lib.UdsError, match='no CPU fallback'):
        _lib.dense_act(torch.zeros(2, 4), torch.zeros(4, 4))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'gnn_uds_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
