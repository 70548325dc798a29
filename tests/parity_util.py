"""Helpers shared by the GPU parity tests."""
import numpy as np

from open_vins_amd import capi

GATE_MARGIN = 1e-8  # = the chi2 tolerance of assert_chi2's callers; measured agreement of the statistic is ~1e-12


def oracle_with_the_same_gate_verdicts(oracle, opts, v, tri, ref, out):
    """Accept / reject sets must be identical.  The one excuse: a feature whose chi2 is within GATE_MARGIN = 1e-8 (relative: the
    tolerance the suite holds the chi2 statistic itself to; SURVEY 8(c) allows 1e-9 of excuse on top of a 1e-9 statistic) of its threshold
    may be gated differently by two float64 evaluations.  The comparison is then NOT skipped: the oracle is run again with the GPU's
    verdict imposed on exactly those features (ORACLE_FORCE_ACCEPT / _REJECT of ov_oracle.h), so dx and P' are always compared."""
    diff = np.nonzero(out["feat_status"] != ref["feat_status"])[0]
    if len(diff) == 0:
        return ref
    st = np.array(tri["status"], dtype=np.int32)
    for f in diff:
        margin = abs(ref["chi2"][f] / ref["chi2_thresh"][f] - 1.0)
        assert margin < GATE_MARGIN, f"feature {f}: status {out['feat_status'][f]} vs {ref['feat_status'][f]} (gate margin {margin})"
        assert {int(out["feat_status"][f]), int(ref["feat_status"][f])} == {capi.FEAT_USED, capi.FEAT_CHI2_REJECTED}
        st[f] = -1 if out["feat_status"][f] == capi.FEAT_USED else -2
    print(f"{len(diff)} feature(s) within {GATE_MARGIN:g} of the gate decided differently: oracle re-run with the GPU's verdicts")
    return oracle.msckf_update(opts, v, want_compressed="H_comp" in ref, given=dict(tri, status=st))


def assert_chi2(out, ref, rtol, strict=False):
    """The gate statistic against the oracle's.  With the library's default (ovgpu_options::gate_always_factor = 0) a feature whose
    residual bound |r'|^2 / sigma^2 is under its threshold is accepted without its gate matrix being formed, and its chi2 output
    is that BOUND: not below the reference's statistic, not above the threshold, on an accepted feature — and at most
    stats.n_gate_bound features may differ from the oracle at all.  strict: every statistic is the reference's (gate_always_factor = 1).
    Returns the number of features that reported a bound."""
    gate = np.isfinite(ref["chi2"])
    o, r = out["chi2"][gate], ref["chi2"][gate]
    close = np.abs(o - r) <= rtol * np.abs(r)
    if strict:
        np.testing.assert_allclose(o, r, rtol=rtol)
        assert out["stats"].get("n_gate_bound", 0) == 0
        return 0
    b = ~close
    if b.any():
        thr, st = out["chi2_thresh"][gate][b], out["feat_status"][gate][b]
        assert (o[b] >= r[b] * (1 - 1e-9)).all(), "a reported bound lies below the reference's chi2"
        assert (o[b] <= thr).all() and (st == capi.FEAT_USED).all(), "a bound above the threshold was used to accept"
    assert b.sum() <= out["stats"]["n_gate_bound"], (int(b.sum()), out["stats"]["n_gate_bound"])
    return int(b.sum())
