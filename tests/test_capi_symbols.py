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


def test_struct_layouts_match_header(tmp_path):
    """sizeof / offsetof of every field of the PODs of include/ovgpu.h as gcc lays them out, against the ctypes mirror."""
    import subprocess
    structs = {"ovgpu_options": capi.Options, "ovgpu_state_view": capi.StateView, "ovgpu_features_view": capi.FeaturesView,
               "ovgpu_landmarks_view": capi.LandmarksView, "ovgpu_update_stats": capi.UpdateStats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ovgpu.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
    assert C.sizeof(capi.Options) == 2 * 8 + 4 * 4 + 9 * 8 + 4 * 4 + 10 * 4 + 8 + 2 * 4


def test_enum_values_match_header(tmp_path):
    """The route / status / representation codes of the ctypes mirror are the header's enumerators as gcc evaluates them."""
    import subprocess
    names = {"OVGPU_COMPRESS_GRAM": capi.COMPRESS_GRAM, "OVGPU_COMPRESS_TSQR": capi.COMPRESS_TSQR,
             "OVGPU_COMPRESS_PCHOLQR": capi.COMPRESS_PCHOLQR, "OVGPU_FEAT_USED": capi.FEAT_USED, "OVGPU_FEAT_CHI2_REJECTED": capi.FEAT_CHI2_REJECTED}
    lines = ['#include <stdio.h>', '#include "ovgpu.h"', 'int main(void) {'] + [f'  printf("{n} %d\\n", (int){n});' for n in names] + ['  return 0;', '}']
    src = tmp_path / "enums.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "enums"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for n, v in names.items():
        assert int(got[n]) == v, n


def test_default_options_are_the_reference_defaults():
    lib = capi.load()
    o = capi.Options()
    lib.ovgpu_default_options(C.byref(o))
    assert (o.chi2_multipler, o.sigma_pix) == (5.0, 1.0)  # UpdaterOptions.h:35-38
    assert (o.triangulate_1d, o.refine_features, o.max_runs) == (0, 1, 5)  # FeatureInitializerOptions.h:36-42
    assert (o.init_lamda, o.max_lamda, o.min_dx, o.min_dcost, o.lam_mult) == (1e-3, 1e10, 1e-6, 1e-6, 10.0)
    assert (o.min_dist, o.max_dist, o.max_baseline, o.max_cond_number) == (0.10, 60.0, 40.0, 10000.0)
    # library switches: zero = default route (whitened Gram matrix, prior block factored on the side stream, timing on)
    assert (o.compress_route, o.gram_no_whiten, o.no_prior_overlap, o.tsqr_workers, o.tsqr_no_pipeline, o.tsqr_overlap, o.gate_always_factor, o.no_timing) == (0,) * 8


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
                # ... nor the oracle-backed test double of the C ABI (tests/fake_ovgpu) or the reference builds of oracle/_ref
                assert "fake_ovgpu" not in txt and "ovgpu_fake" not in txt and "libov_ref" not in txt and "libov_dropin" not in txt, f"{fn} references test infrastructure"
    # and bench.py / __graft_entry__.smoke() load the product library only (the drop-in and fake builds are made by build(), never loaded there)
    for fn in ("bench.py",):
        txt = open(os.path.join(ROOT, fn)).read()
        assert "fake_ovgpu" not in txt and "ovgpu_fake" not in txt and "libov_dropin" not in txt, fn
