import numpy as np, sys, time
sys.path.insert(0, '.')
from cloud_map_evaluation_amd.engine import Engine
from oracle import ref
import oracle

def sheet(n, angle_deg, jitter, seed):
    rng = np.random.default_rng(seed)
    side = np.sqrt(n / 2500.0)
    uv = rng.uniform(0, side, (n, 2))
    w = rng.uniform(-jitter, jitter, n) if jitter > 0 else np.zeros(n)
    a = np.deg2rad(angle_deg)
    # a rotation that tilts the sheet against ALL three axes: about x by a, then about z by a, then about y by a/2
    def R(ax, t):
        c, s = np.cos(t), np.sin(t)
        M = np.eye(3); i, j = [(1, 2), (0, 2), (0, 1)][ax]
        M[i, i] = c; M[j, j] = c; M[i, j] = -s; M[j, i] = s
        return M
    Rm = R(1, a / 2) @ R(2, a) @ R(0, a)
    p = np.stack([uv[:, 0], uv[:, 1], w], 1) @ Rm.T + np.array([3.0, -2.0, 1.5])
    return np.ascontiguousarray(p)

with Engine(0) as e:
    for ang in (17.0, 30.0, 45.0):
        for jit in (1e-3, 1e-5, 1e-6, 0.0):
            pts = sheet(100_000, ang, jit, int(ang * 1000 + jit * 1e7))
            e.upload(0, pts, cell_size=0.1)
            for variant, mink in ((2, 10), (0, 5)):
                t0 = time.time()
                rmean, rent, rval = ref.mme(variant, pts, 0.1)
                mean, ent, val, nv, s = e.mme(0, 0.1, mink)
                val = val.astype(bool)
                det = np.where(rval, np.exp(2 * rent) / (2 * np.pi * np.e), 0.0)
                safe = rval & (det > 3.5e-22)
                both = rval & val
                err = np.abs(ent - rent)[both]
                oent = oracle.mme(pts, 0.1, mink)[1]
                oerr = np.abs(oent - rent)[rval & (oent != 0)]
                print(f"ang {ang} jit {jit:g} var {variant}: ref valid {rval.sum()} dev valid {val.sum()} flag mismatches {np.sum(rval != val)} "
                      f"(in safe set {np.sum((rval != val) & safe)}; safe {safe.sum()}) max|dH| dev-ref {err.max() if len(err) else 0:.3e} "
                      f"(safe {np.abs(ent - rent)[safe & val].max() if (safe & val).any() else 0:.3e}) oracle-ref {oerr.max() if len(oerr) else 0:.3e} "
                      f"|H| ~ {np.abs(rent[rval]).mean() if rval.any() else 0:.2f} mean dev {mean!r} ref {rmean!r}  [{time.time()-t0:.1f}s]")
