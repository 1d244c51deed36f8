"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every
symbol include/komb_accel.h declares, and fails loudly without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "komb_accel.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(komb_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(built):
    import komb_amd
    lib = ctypes.CDLL(komb_amd._lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in komb_accel.h but not exported"
    # and the Python binding covers exactly the header
    assert sorted(komb_amd._lib.SIGNATURES) == names


def test_abi_version(built):
    import komb_amd
    assert komb_amd._lib.load().komb_abi_version() == 7


def test_stats_struct_matches_header(built):
    """ctypes mirror of komb_stats has the fields of the header, in order."""
    import komb_amd
    text = open(os.path.join(ROOT, "include", "komb_accel.h")).read()
    body = re.search(r"typedef struct komb_stats \{(.*?)\} komb_stats;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        typ, names = decl.split(None, 1)
        fields += [n.strip() for n in names.split(",")]
    assert fields == [n for n, _ in komb_amd._lib.KombStats._fields_]


def test_generator_deterministic(built):
    import komb_amd
    a = komb_amd.gen_hug_edges(5000, 12000, 2.6, 42)
    os.environ["OMP_NUM_THREADS"] = "1"
    b = komb_amd.gen_hug_edges(5000, 12000, 2.6, 42)
    c = komb_amd.gen_hug_edges(5000, 12000, 2.6, 43)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.min() >= 0 and a.max() < 5000
    with pytest.raises(ValueError):
        komb_amd.gen_hug_edges(1, 10, 2.6, 1)


def test_no_device_fails_loudly(built):
    """No silent CPU fallback: without a GPU every compute entry point errors."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import komb_amd
    g = komb_amd.KombAccel()
    with pytest.raises(komb_amd.KombError) as e:
        g.from_edges(3, [[0, 1]])
    assert e.value.code == komb_amd._lib.KOMB_ERR_DEVICE
    for call in (g.core_run, g.truss_run, lambda: g.get_anomaly_score([1], [1])):
        with pytest.raises(komb_amd.KombError):
            call()


def test_product_never_imports_oracle():
    """The product tree must not reference oracle/ in any form."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "komb_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in text.lower().replace("no cpu fallback", ""), os.path.join(dirpath, fn)
