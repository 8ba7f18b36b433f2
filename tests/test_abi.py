"""The C-ABI library builds for gfx950, loads, and exports every symbol include/agx.h declares (no compute: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from aligngraph_amd import build as B
    B.build()
    import aligngraph_amd as A
    return A


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, "include", "agx.h")).read()
    declared = set(re.findall(r"\b(agx_[a-z_]+)\s*\(", hdr))
    assert declared == set(lib.EXPORTS)
    L = lib.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_struct_sizes_match_c_layout(lib):
    assert ctypes.sizeof(lib.Hit) == 32 and ctypes.sizeof(lib.Run) == 12 and ctypes.sizeof(lib.ContiMer) == 20 and ctypes.sizeof(lib.Params) == 24


def test_no_gpu_is_a_loud_error_not_a_fallback(lib):
    if lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(lib.AgxError) as e:
        lib.Unit()
    assert e.value.code == lib.AGX_E_NOGPU
    with pytest.raises(lib.AgxError) as e:
        lib.run_unit("/nonexistent", 0)
    assert e.value.code == lib.AGX_E_NOGPU


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "aligngraph_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in text and "agx_oracle" not in text and "hostsim" not in text.replace("tests/hostsim", "").replace("(tests/hostsim)", ""), f
