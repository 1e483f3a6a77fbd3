"""CPU: the C-ABI library builds, loads and exports every symbol include/aread_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from aread_amd import _lib
    return _lib


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "aread_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aread_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/aread_hip.h but not exported"
    assert sorted(built.exported_symbols()) == names, "python binding table and header disagree"


def test_version_and_error_string(built):
    lib = built.lib()
    assert lib.aread_version() >= 100
    assert isinstance(lib.aread_last_error(), bytes)


def test_argument_errors_do_not_need_a_gpu(built):
    lib = built.lib()
    lay = built.PlanLayout()
    assert lib.aread_plan_layout_get(8192, 25, lay) == 0
    assert lay.max_rows % 64 == 0 and lay.max_rows >= 8192 and lay.words > lay.max_rows
    assert lib.aread_plan_layout_get(0, 25, lay) != 0
    assert b"bad B" in lib.aread_last_error()
    assert lib.aread_plan_layout_get(10, 1000, lay) != 0


def test_product_path_refuses_cpu_tensors(built):
    import torch
    import aread_amd
    mh = {"multi_hot_flag": [False] * 3, "itemid_idx": 0, "seq_maxlen": 5, "method": None}
    emb = aread_amd.FeaturesEmbedding([5, 6, 7], 8, mh)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        emb(torch.zeros((2, 3), dtype=torch.int32))
    with pytest.raises(ValueError):
        aread_amd.FeaturesEmbedding([5, 6, 7], 8, dict(mh, method="max"))
