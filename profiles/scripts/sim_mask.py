"""Model of a per-lane accept mask in k_mme3: per tile of T streamed candidates, trips of the fp64 accumulation today (candidates that ANY
lane of the wave accepts) against the largest per-lane accepted count (what a walk over each lane's own mask would take)."""
import numpy as np, sys
sys.path.insert(0, '.')
from cloud_map_evaluation_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dens = float(sys.argv[2]) if len(sys.argv) > 2 else 2500.0
r = 0.1
est, gt = synth.multisession_pair(n, 3, density=dens, seed=100)
for name, cloud in (("gt", gt.numpy()), ("est", est.numpy())):
    h = 0.1
    o = np.floor(cloud.min(0) / h) * h
    fine = np.floor((cloud - o) / (h / 2)).astype(np.int64)
    bits = int(np.ceil(np.log2(fine.max() + 1)))
    X = [fine[:, 0].copy(), fine[:, 1].copy(), fine[:, 2].copy()]
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(3):
            m = (X[i] & Q) != 0
            X[0] = np.where(m, X[0] ^ P, X[0])
            t = np.where(~m, (X[0] ^ X[i]) & P, 0)
            X[0] ^= t
            X[i] ^= t
        Q >>= 1
    for i in range(1, 3):
        X[i] ^= X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for i in range(3):
        X[i] ^= t
    key = np.zeros(len(cloud), dtype=np.uint64)
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            key = (key << np.uint64(1)) | ((X[i] >> b) & 1).astype(np.uint64)
    order = np.argsort(key, kind="stable")
    pts = cloud[order]
    cell = (fine[order] >> 1)
    ck = (cell[:, 0] << 42) | (cell[:, 1] << 21) | cell[:, 2]
    # runs of equal cell in sorted order (a cell is one contiguous run in Hilbert order)
    starts = np.flatnonzero(np.r_[True, ck[1:] != ck[:-1]])
    ends = np.r_[starts[1:], len(ck)]
    run_of = {int(c): (int(s), int(e)) for c, s, e in zip(ck[starts], starts, ends)}
    offs = [(dx << 42) + (dy << 21) + dz for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)]
    rng = np.random.default_rng(0)
    nw = len(cloud) // 64
    sel = rng.choice(nw, min(nw, 1500), replace=False)
    res = {32: [0, 0, 0], 64: [0, 0, 0], 128: [0, 0, 0], 256: [0, 0, 0], 512: [0, 0, 0], 100000: [0, 0, 0]}
    cand_tot = pairs_tot = 0
    for w in sel:
        q = pts[w * 64:(w + 1) * 64]
        cs = set()
        for c in set(ck[w * 64:(w + 1) * 64].tolist()):
            for d in offs:
                if c + d in run_of: cs.add(c + d)
        runs = sorted(run_of[c] for c in cs)
        merged = []
        for s, e in runs:
            if merged and merged[-1][1] == s: merged[-1][1] = e
            else: merged.append([s, e])
        idx = np.concatenate([np.arange(s, e) for s, e in merged])
        c_all = pts[idx]
        d2 = ((q[:, None, :] - c_all[None, :, :]) ** 2).sum(-1)
        acc_all = d2 < r * r
        pairs_tot += int(acc_all.sum())
        for T in res:
            for b in range(0, len(idx), T):
                acc = acc_all[:, b:b + T]
                res[T][0] += int(acc.any(0).sum())
                res[T][1] += int(acc.sum(1).max())
                res[T][2] += acc.shape[1]
    for T, (a, m, c) in res.items():
        print(name, "tile", T, "| candidates/wave", round(c / len(sel), 1), "| trips today (any lane accepts)", round(a / len(sel), 1),
              "| with masks (max lane per tile)", round(m / len(sel), 1), "| ratio", round(m / a, 3))
    print(name, "accepted pairs per wave", round(pairs_tot / len(sel), 1), "lane efficiency today", round(pairs_tot / (64 * res[32][0]), 3))
