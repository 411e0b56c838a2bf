"""The C-ABI library loads without a GPU and exports exactly what include/reloc.h declares; compute
entry points fail loudly (no CPU fallback) when no device is usable."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "reloc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(reloc_[a-z0-9_]+)\s*\(", txt)) - {"reloc_ctx"})


def test_header_binding_and_library_agree():
    from nclt_slam_project_amd import _native as N
    declared = _declared()
    assert len(declared) >= 35
    assert sorted(N.SIGNATURES) == declared, "include/reloc.h and _native.SIGNATURES differ"
    assert os.path.exists(N.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(N.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"libreloc_hip.so lacks {missing}"
    N.load(strict=True)


def test_no_cpu_fallback_without_device():
    from nclt_slam_project_amd import _native as N
    from nclt_slam_project_amd.engine import Engine
    lib = N.load()
    if lib.reloc_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(N.RelocError, match="no HIP device"):
        Engine()
    import nclt_slam_project_amd.cv2_shim as shim
    shim._default = None
    with pytest.raises(shim.error):
        shim.cvtColor(np.zeros((8, 8, 3), np.uint8), shim.COLOR_BGR2GRAY)


def test_product_never_imports_the_oracle():
    """the shipped path must not route through the CPU oracle in any form"""
    pkg = os.path.join(ROOT, "nclt-slam-project_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "libreloc_oracle", "orc_"):
                    assert needle not in txt, f"{f} mentions {needle}"
