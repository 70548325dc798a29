"""Numerical prototype (round 5, NEGATIVE result): an upper bound of the gate statistic from a COLUMN SUBSET of the
whitened projected rows (Woodbury on k of the 208 columns).  CPU only (numpy + the oracle: a developer tool).  The bound only decides
once ~96 of 208 columns are in: the gate matrix is not low-rank in the prior's coordinates."""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from open_vins_amd import capi, synth
from oracle import pyoracle
prob = synth.make_problem(3, F=300)
opts = capi.default_options(chi2_multipler=1.0)
views = capi.Views(prob)
cols = pyoracle.column_map(opts, views)
D = len(cols)
tri = pyoracle.triangulate(opts, views)
idx = np.array(cols)
L = np.linalg.cholesky(prob.P[np.ix_(idx, idx)])
res = {k: [] for k in (8, 16, 32, 48, 64, 96)}
true = []; thr = []
for f in range(prob.F):
    if tri["status"][f] != 0: continue
    Hf, Hx, r = pyoracle.feature_jacobian(opts, views, f, tri["p_FinG"][f])
    _, Hp, rp = pyoracle.nullspace_project(Hf, Hx, r)
    Y = Hp @ L
    n = Y.shape[0]
    S = Y @ Y.T + np.eye(n)
    c = rp @ np.linalg.solve(S, rp)
    true.append(c); thr.append(pyoracle.chi2_quantile_95(n))
    # choose columns greedily by explained residual energy: order by |Y^T r| / norm? simple: column norms
    score = np.abs(Y.T @ rp)
    order = np.argsort(-score)
    for k in res:
        Yc = Y[:, order[:k]]
        M = np.eye(k) + Yc.T @ Yc
        v = Yc.T @ rp
        res[k].append(rp @ rp - v @ np.linalg.solve(M, v))
true = np.array(true); thr = np.array(thr)
acc = true <= thr
print("features", len(true), "accepted", acc.sum())
for k in res:
    b = np.array(res[k])
    print(k, "bound<=thr among accepted: %.3f" % np.mean(b[acc] <= thr[acc]), " median bound/true %.2f" % np.median(b[acc] / true[acc]))
