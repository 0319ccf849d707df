"""CPU-only checks of the drop-in boundary: the library builds, loads, and exports every symbol that
include/f2cnn_hip.h declares; the ctypes table mirrors the header; contexts fail loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from f2cnn_amd import _lib, build


def header_symbols():
    text = open(os.path.join(ROOT, "include", "f2cnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(f2_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_for_gfx950():
    path = build.build_library()
    assert os.path.exists(path) and path.endswith("libf2cnn_hip.so")


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(build.build_library())
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/f2cnn_hip.h but not exported"


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.F2Error):
        _lib.Context(0)
    from f2cnn_amd.gammatone import filters
    co = filters.make_erb_filters(16000, filters.centre_freqs(16000, 8, 100))
    with pytest.raises(_lib.F2Error):
        filters.erb_filterbank(np.zeros(100, np.int16), co)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "f2cnn_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "f2cnn_oracle" not in src and "import oracle" not in src, f
