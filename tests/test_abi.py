"""
The C-ABI library loads on a machine without a GPU and exports every symbol that
include/mqslam.h declares; the ctypes table in _lib.py covers the same set; compute calls
fail loudly (RuntimeError) instead of falling back when no device is present.
"""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mqslam.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mqs_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_declares_symbols():
    syms = header_symbols()
    assert "mqs_triangulate_iterative_ls" in syms and "mqs_linear_LS_triangulation" in syms
    assert len(syms) >= 20


def test_library_exports_every_declared_symbol(mqs):
    assert mqs.loaded, "libmqslam_hip.so failed to load: %r" % (mqs._lib.load_error,)
    out = subprocess.check_output(["nm", "-D", "--defined-only", mqs._lib.LIB_PATH]).decode()
    exported = set(re.findall(r"\bT (mqs_[A-Za-z0-9_]+)", out))
    missing = [s for s in header_symbols() if s not in exported]
    assert not missing, "declared in mqslam.h but not exported: %s" % missing


def test_ctypes_table_matches_header(mqs):
    assert sorted(mqs._lib.SIGNATURES) == header_symbols()


def test_version_and_error_strings(mqs):
    lib = mqs._lib.lib()
    assert b"gfx950" in lib.mqs_version()
    assert isinstance(lib.mqs_last_error(), bytes)


def test_no_cpu_fallback_without_device(mqs):
    if mqs._lib.lib().mqs_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        mqs.triangulation.linear_LS_triangulation(np.zeros((3, 2)), np.eye(4), np.zeros((3, 2)), np.eye(4))


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "multiple-quadrotor-slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "liboracle" not in text, f


def test_static_evidence_is_stamped_with_the_sources_it_was_taken_from():
    """profiles/kernel_flops.json and pmc_traffic.json carry the digest of the kernel sources they describe
    (tools/evidence_stamp.py); bench.py drops a record whose digest no longer matches.  This test keeps the committed evidence
    in step with the committed kernels: re-emit (tools/emit_kernel_flops.py, tools/emit_pmc_traffic.py) when it fails."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import evidence_stamp
    kf = json.load(open(os.path.join(ROOT, "profiles", "kernel_flops.json")))
    assert evidence_stamp.is_current(kf["ba_linearize_kernel<4>"]["source"]), "profiles/kernel_flops.json is stale"
    pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for fam in ("ba", "tri"):
        assert evidence_stamp.is_current(pm["sources"][fam]), "profiles/pmc_traffic.json (%s) is stale" % fam
    # and a changed source is noticed
    stale = dict(kf["ba_linearize_kernel<4>"]["source"], sha256_16="0" * 16)
    assert not evidence_stamp.is_current(stale)
