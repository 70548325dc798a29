"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/ovgpu.h declares, refuses to run without a device (no CPU fallback), and its host-side helpers agree
with scipy."""
import ctypes as C
import os
import re

import pytest
from scipy.stats import chi2 as sp_chi2

from open_vins_amd import capi

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            txt = open(os.path.join(inc, fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(ovgpu_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ovgpu.h but not exported by libovgpu.so"
    # and the ctypes mirror knows all of them
    assert set(capi.declare(lib)) == set(declared)


def test_struct_layouts_match_header():
    # sizes the C compiler produces for the PODs of include/ovgpu.h (checked against a compiled probe in build())
    assert C.sizeof(capi.Options) == 2 * 8 + 4 * 4 + 9 * 8 + 4 * 4
    assert C.sizeof(capi.StateView) == 4 * 4 + 9 * 8
    assert C.sizeof(capi.FeaturesView) == 2 * 4 + 5 * 8
    assert C.sizeof(capi.UpdateStats) == 6 * 4 + 6 * 4


def test_default_options_are_the_reference_defaults():
    lib = capi.load()
    o = capi.Options()
    lib.ovgpu_default_options(C.byref(o))
    assert (o.chi2_multipler, o.sigma_pix) == (5.0, 1.0)  # UpdaterOptions.h:35-38
    assert (o.triangulate_1d, o.refine_features, o.max_runs) == (0, 1, 5)  # FeatureInitializerOptions.h:36-42
    assert (o.init_lamda, o.max_lamda, o.min_dx, o.min_dcost, o.lam_mult) == (1e-3, 1e10, 1e-6, 1e-6, 10.0)
    assert (o.min_dist, o.max_dist, o.max_baseline, o.max_cond_number) == (0.10, 60.0, 40.0, 10000.0)


def test_chi2_quantile_host_helper():
    lib = capi.load()
    for k in [1, 2, 3, 10, 57, 117, 237, 397, 499, 650, 1000]:
        assert lib.ovgpu_chi2_quantile_95(k) == pytest.approx(sp_chi2.ppf(0.95, k), rel=1e-11)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = capi.load()
    ctx = C.c_void_p()
    o = capi.default_options()
    rc = lib.ovgpu_create(C.byref(o), 0, C.byref(ctx))
    assert rc == capi.ERR_NO_DEVICE
    assert not ctx.value
    assert b"no CPU fallback" in lib.ovgpu_last_error()


def test_product_package_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "open_vins_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "pyoracle" not in txt and "ov_oracle" not in txt and "libov_oracle" not in txt, f"{fn} references the oracle"
