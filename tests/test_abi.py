"""CPU checks of the drop-in boundary: the library loads, exports every symbol the public
header declares, and refuses to compute without a GPU (no silent fallback)."""
import os
import re
import ctypes as C
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "sqmc_gpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sqmc_gpu_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import sqmc_amd
    sqmc_amd.build_library()
    L = C.CDLL(os.path.join(ROOT, "sqmc_amd", "libsqmc_gpu.so"))
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n


def test_python_binding_lists_the_same_symbols():
    from sqmc_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()


def test_fortran_module_binds_the_same_symbols():
    src = open(os.path.join(ROOT, "sqmc_amd", "fortran", "sqmc_gpu_mod.f90")).read().lower()
    bound = set(re.findall(r"name\s*=\s*'(sqmc_gpu_\w+)'", src))
    missing = [n for n in _declared() if n not in bound]
    assert not missing, missing


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import sqmc_amd
    with pytest.raises(sqmc_amd.SqmcGpuError):
        sqmc_amd.SpmvPlan(np.array([1]), np.array([1]), np.array([1.0]))
    with pytest.raises(sqmc_amd.SqmcGpuError):
        sqmc_amd.set_device(0)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sqmc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".f90", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dirpath, f)
