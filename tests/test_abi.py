"""The C-ABI library loads and exports every symbol include/gat_mi355.h declares; without a GPU it
fails loudly instead of falling back (CPU-only checks, no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "gat_mi355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gat_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from gat_amd import _lib
    so = _lib.LIB_PATH
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), "libgat_mi355.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names
    assert b"gfx950" in ctypes.c_char_p(ctypes.cast(L.gat_version, ctypes.c_void_p).value and _lib.lib().gat_version()).value


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    from gat_amd import _lib
    with pytest.raises(_lib.GatError) as e:
        _lib.Context(0)
    assert "no HIP device" in str(e.value) or "no CPU path" in str(e.value)


def test_product_does_not_import_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gat_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "gat_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f
