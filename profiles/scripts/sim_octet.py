"""Model: candidates per lane if every octet (8 curve-consecutive queries) of a wave streamed only ITS OWN cell box grown by one,
against the wave-wide stream of k_mme3 (cells adjacent to any lane).  Hilbert order via Skilling's transpose on the cell grid + Morton below."""
import numpy as np, sys, itertools
sys.path.insert(0, '.')
from cloud_map_evaluation_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dens = float(sys.argv[2]) if len(sys.argv) > 2 else 2500.0
est, gt = synth.multisession_pair(n, 3, density=dens, seed=100)
for name, cloud in (("gt", gt.numpy()), ("est", est.numpy())):
    h = 0.1
    o = np.floor(cloud.min(0) / h) * h
    fine = np.floor((cloud - o) / (h / 2)).astype(np.int64)   # one level below the cell, as the bench's sort (depth 1)
    bits = int(np.ceil(np.log2(fine.max() + 1)))
    # Hilbert index (Skilling), vectorised
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
    cell = (fine[order] >> 1)
    ck = (cell[:, 0] << 42) | (cell[:, 1] << 21) | cell[:, 2]
    uniq, cnt = np.unique(ck, return_counts=True)
    pop = dict(zip(uniq.tolist(), cnt.tolist()))
    offs = [(dx << 42) + (dy << 21) + dz for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)]
    rng = np.random.default_rng(0)
    nw = len(cloud) // 64
    sel = rng.choice(nw, min(nw, 4000), replace=False)
    tot_wave, tot_oct_max, tot_oct_mean, tot_q4_max, tot_h_max, per_lane = [], [], [], [], [], []
    for w in sel:
        cw = ck[w * 64:(w + 1) * 64]
        def stream(cells):
            s = set()
            for c in set(cells.tolist()):
                for d in offs:
                    s.add(c + d)
            return sum(pop.get(c, 0) for c in s)
        def box_stream(cells):  # bounding box grown by one (what a per-group table would stream without the cull)
            cs = np.array(list(set(cells.tolist())))
            x, y, z = cs >> 42, (cs >> 21) & ((1 << 21) - 1), cs & ((1 << 21) - 1)
            tot = 0
            for ix in range(x.min() - 1, x.max() + 2):
                for iy in range(y.min() - 1, y.max() + 2):
                    for iz in range(z.min() - 1, z.max() + 2):
                        tot += pop.get((int(ix) << 42) | (int(iy) << 21) | int(iz), 0)
            return tot
        tot_wave.append(stream(cw))
        o8 = [stream(cw[i:i + 8]) for i in range(0, 64, 8)]
        tot_oct_max.append(max(o8)); tot_oct_mean.append(np.mean(o8))
        q4 = [stream(cw[i:i + 16]) for i in range(0, 64, 16)]
        tot_q4_max.append(max(q4))
        h2 = [stream(cw[i:i + 32]) for i in range(0, 64, 32)]
        tot_h_max.append(max(h2))
        per_lane.append(np.mean([stream(cw[i:i + 1]) for i in range(0, 64, 8)]))
    print(name, "n", len(cloud), "waves sampled", len(sel), "| wave-wide stream (adjacency cull, one round assumed)", round(np.mean(tot_wave), 1),
          "| halves max", round(np.mean(tot_h_max), 1), "| quarters max", round(np.mean(tot_q4_max), 1), "| octets max", round(np.mean(tot_oct_max), 1),
          "mean", round(np.mean(tot_oct_mean), 1), "| single lane", round(np.mean(per_lane), 1))
