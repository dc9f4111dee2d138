"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports exactly what
include/lpslam_hip.h declares; without a GPU the product fails loudly instead of falling back."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "lpslam_hip.h")).read()
    return sorted(set(re.findall(r"\b(lpslam_hip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(hiplib):
    lib = hiplib.load()
    declared = _declared()
    assert len(declared) >= 40
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(hiplib.SYMBOLS) == declared


def test_code_object_targets_gfx950(hiplib):
    data = open(hiplib.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"gfx942" not in data and b"sm_" not in data


def test_no_cpu_fallback_without_device(hiplib):
    if hiplib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(hiplib.LpslamHipError) as e:
        hiplib.Context(640, 480)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_product_does_not_reference_the_oracle():
    bad = []
    for d in ("lpslam_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", ".inc")):
                    t = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle|oracle/|liblpslam_oracle|ora_[a-z]+\(", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_keypoint_layout_matches_cv_keypoint(hiplib):
    assert hiplib.KP_DTYPE.itemsize == 28 and hiplib.KP_DTYPE.names == ("x", "y", "size", "angle", "response", "octave", "class_id")
    assert hiplib.BA_OBS_DTYPE.itemsize == 40


def test_reference_header_client_links():
    """tests/golden/abi_symbols.txt holds the mangled names a client compiled against the REFERENCE's own interface headers
    (src/Interface/LpSlamManager.h:17-121, LpSlamConfiguration.h) leaves undefined -- made by tools/make_abi_symbols.py in the
    build container, where the same client is also linked and loaded against the product library.  Every one of them must be a
    defined dynamic symbol of lpslam_amd/liblpslam.so: that is what "existing clients relink unchanged" means."""
    import subprocess
    from lpslam_amd import _build
    lib = _build.host_library()
    want = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "abi_symbols.txt")) if l.strip()]
    assert len(want) == 37 and sum("LpSlamManager" in s for s in want) == 36
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True)
    have = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    missing = [s for s in want if s not in have]
    assert not missing, missing
