"""CPU: the C-ABI shared library builds, loads, and exports every symbol include/cyclediff.h declares
(no compute calls without a GPU); the product refuses to run without a HIP device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "cyclediff.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cd_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from cycle_diffusion_amd import _ffi
    lib = _ffi.load_library()
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), s
    # the ctypes signature table covers exactly the header
    assert sorted(_ffi.SIGNATURES.keys()) == syms
    assert lib.cd_version() >= 100


def test_struct_layouts_match_the_header():
    import ctypes as C
    from cycle_diffusion_amd import _ffi
    assert C.sizeof(_ffi.StepCoef) == 32 and _ffi.STEP_COEF_DTYPE.itemsize == 32
    assert C.sizeof(_ffi.NetDesc) == 4 * (6 + 1 + 8 + 1 + 8 + 2 + 3 + 3 + 3 + 8)


def test_no_cpu_fallback():
    import torch
    import cycle_diffusion_amd as cda
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception) as e:
        cda.Engine("cuda:0")
    assert "no CPU fallback" in str(e.value)
    # and the C entry point itself fails loudly
    import ctypes as C
    from cycle_diffusion_amd import _ffi
    lib = _ffi.load_library()
    h = C.c_void_p()
    assert lib.cd_engine_create(None, C.c_size_t(1 << 28), C.byref(h)) != 0
    assert b"no CPU fallback" in lib.cd_last_error()
