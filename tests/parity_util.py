"""Helpers shared by the GPU parity tests."""
import numpy as np

from open_vins_amd import capi


def oracle_with_the_same_gate_verdicts(oracle, opts, v, tri, ref, out):
    """Accept / reject sets must be identical.  The one excuse: a feature whose chi2 is within 1e-6 (relative) of its threshold may
    be gated differently by two float64 evaluations.  The comparison is then NOT skipped: the oracle is run again with the GPU's
    verdict imposed on exactly those features (ORACLE_FORCE_ACCEPT / _REJECT of ov_oracle.h), so dx and P' are always compared."""
    diff = np.nonzero(out["feat_status"] != ref["feat_status"])[0]
    if len(diff) == 0:
        return ref
    st = np.array(tri["status"], dtype=np.int32)
    for f in diff:
        margin = abs(ref["chi2"][f] / ref["chi2_thresh"][f] - 1.0)
        assert margin < 1e-6, f"feature {f}: status {out['feat_status'][f]} vs {ref['feat_status'][f]} (gate margin {margin})"
        assert {int(out["feat_status"][f]), int(ref["feat_status"][f])} == {capi.FEAT_USED, capi.FEAT_CHI2_REJECTED}
        st[f] = -1 if out["feat_status"][f] == capi.FEAT_USED else -2
    print(f"{len(diff)} feature(s) within 1e-6 of the gate decided differently: oracle re-run with the GPU's verdicts")
    return oracle.msckf_update(opts, v, want_compressed="H_comp" in ref, given=dict(tri, status=st))
