import numpy as np, sys
sys.path.insert(0, '.')
from cloud_map_evaluation_amd import synth
n = 2_000_000
est, gt = synth.scan_pair(n, density=2500.0, seed=100)
def hilbert_order(cloud, h=0.1, depth=1):
    o = np.floor(cloud.min(0) / h) * h
    fine = np.floor((cloud - o) / (h / 2**depth)).astype(np.int64)
    bits = int(np.ceil(np.log2(fine.max() + 1)))
    X = [fine[:, 0].copy(), fine[:, 1].copy(), fine[:, 2].copy()]
    M = 1 << (bits - 1); Q = M
    while Q > 1:
        P = Q - 1
        for i in range(3):
            m = (X[i] & Q) != 0
            X[0] = np.where(m, X[0] ^ P, X[0]); t = np.where(~m, (X[0] ^ X[i]) & P, 0); X[0] ^= t; X[i] ^= t
        Q >>= 1
    for i in range(1, 3): X[i] ^= X[i - 1]
    t = np.zeros_like(X[0]); Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t); Q >>= 1
    for i in range(3): X[i] ^= t
    key = np.zeros(len(cloud), dtype=np.uint64)
    for b in range(bits - 1, -1, -1):
        for i in range(3): key = (key << np.uint64(1)) | ((X[i] >> b) & 1).astype(np.uint64)
    return np.argsort(key, kind="stable")
for name, c in (("gt", gt.numpy()), ("est", est.numpy())):
    order = hilbert_order(c)
    p = c[order]
    for vs in (3.0, 2.0):
        v = np.floor(p / vs).astype(np.int64)
        k = (v[:, 0] << 42) ^ (v[:, 1] << 21) ^ v[:, 2]
        head = np.ones(len(k), bool); head[1:] = k[1:] != k[:-1]
        rows = len(k) // 64
        hr = head[:rows * 64].reshape(rows, 64).copy(); hr[:, 0] = True
        runs = hr.sum(1)
        print(name, "vs", vs, "rows", rows, "runs/row %.3f" % runs.mean(), "rows with 1/2/3+/10+ runs: %.3f %.3f %.3f %.4f" % ((runs == 1).mean(), (runs == 2).mean(), (runs >= 3).mean(), (runs >= 10).mean()), "appended", int(np.maximum(runs - 2, 0).sum()))
        big = np.flatnonzero(runs >= 10)[:3]
        for r in big:
            seg = p[r * 64:(r + 1) * 64]
            print("   row", r, "extent", (seg.max(0) - seg.min(0)).round(3), "z range", seg[:, 2].min().round(3), seg[:, 2].max().round(3))
