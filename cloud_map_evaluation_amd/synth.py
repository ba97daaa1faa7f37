"""Seeded synthetic map pairs for tests and bench.py (no datasets or network are available).

The perturbations restate the reference's own (unused-by-process()) noise generators as recipes:
addGaussianNoise (map_eval.cpp:1745-1755), addNonUniformDensity (:1757-1784), addSparseOutliers (:1786-1806).
Everything is fp64, generated with torch on the requested device (CPU for tests, cuda for the bench) and
jittered off-lattice / off-plane so that inlier counts and MME validity do not sit on rounding boundaries
(SURVEY.md §7 "hard parts").

Workloads (restated as surface-density-driven scenes so that nn_radius finds neighbours):
  cube_pair     C1  100 k points on a 4 m cube + noised copy.
  campus_pair   C2-C3  ground + boxes + poles at a target surface density (points / m^2); est = perturbed GT points.
  scan_pair     C2-C4  same scene, est = an INDEPENDENT scan of it, both clouds exactly n points (the bench default).
  multisession_pair  C4  est = union of three independent scans with their own drifts.
  tunnel_pair   C5  degenerate geometry (tunnel + flat field + staircase) for the AWD eigen-clamp.
"""
from __future__ import annotations

import math

import torch

F64 = torch.float64


def _gen(seed: int, device) -> torch.Generator:
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def _rand(n, g, device, lo=0.0, hi=1.0):
    return torch.rand(n, generator=g, device=device, dtype=F64) * (hi - lo) + lo


def _randn(shape, g, device, std=1.0):
    return torch.randn(shape, generator=g, device=device, dtype=F64) * std


def cube_pair(n: int = 100_000, seed: int = 42, noise_std: float = 0.02, device="cpu"):
    """C1: GT on the surface of a 4 m cube spanning [0.25, 4.25]^3 (+-1 mm off-plane jitter), est = GT + N(0, s^2)."""
    g = _gen(seed, device)
    face = torch.randint(0, 6, (n,), generator=g, device=device)
    u = _rand(n, g, device, 0.25, 4.25)
    v = _rand(n, g, device, 0.25, 4.25)
    jit = _rand(n, g, device, -1e-3, 1e-3)
    w = torch.where(face % 2 == 0, torch.full_like(u, 0.25), torch.full_like(u, 4.25)) + jit
    axis = face // 2
    x = torch.where(axis == 0, w, u)
    y = torch.where(axis == 1, w, torch.where(axis == 0, u, v))
    z = torch.where(axis == 2, w, v)
    gt = torch.stack([x, y, z], dim=1).contiguous()
    g2 = _gen(seed + 1, device)
    est = (gt + _randn((n, 3), g2, device, noise_std)).contiguous()
    return est, gt


def _box_surface(n, g, device, cx, cy, w, d, h, yaw):
    """n points on the 4 walls + roof of a w x d x h box centred at (cx, cy), rotated by yaw about z."""
    a_wx, a_wy, a_roof = w * h, d * h, w * d
    tot = 2 * a_wx + 2 * a_wy + a_roof
    r = _rand(n, g, device) * tot
    s = _rand(n, g, device)
    t = _rand(n, g, device)
    x = torch.empty(n, dtype=F64, device=device)
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    b0, b1, b2, b3 = a_wx, 2 * a_wx, 2 * a_wx + a_wy, 2 * a_wx + 2 * a_wy
    m0 = r < b0
    m1 = (r >= b0) & (r < b1)
    m2 = (r >= b1) & (r < b2)
    m3 = (r >= b2) & (r < b3)
    m4 = r >= b3
    x = torch.where(m0 | m1 | m4, (s - 0.5) * w, torch.where(m2, torch.full_like(s, -0.5 * w), torch.full_like(s, 0.5 * w)))
    y = torch.where(m2 | m3, (s - 0.5) * d,
                    torch.where(m0, torch.full_like(s, -0.5 * d), torch.where(m1, torch.full_like(s, 0.5 * d), (t - 0.5) * d)))
    z = torch.where(m4, torch.full_like(s, h), t * h)
    c, sn = math.cos(yaw), math.sin(yaw)
    return torch.stack([cx + c * x - sn * y, cy + sn * x + c * y, z], dim=1)


def campus_scene(n: int, density: float = 2500.0, seed: int = 100, device="cpu", origin=(0.0, 0.0, 0.0), sample_seed=None):
    """Ground + rotated boxes + poles, surface-sampled at ~`density` points/m^2, 2 mm off-surface jitter.

    The scene extent follows from n / density (50 % ground, 40 % buildings, 10 % poles), so the local
    neighbourhood statistics (k in nn_radius) do not change with n.  `seed` fixes the GEOMETRY (buildings, poles);
    `sample_seed` (default: seed) the points drawn on it: two calls with the same seed and n / density ratio but different
    sample_seed are two independent scans of the same place.
    """
    g = _gen(seed if sample_seed is None else sample_seed, device)
    n_ground = n // 2
    n_pole = n // 10
    n_box = n - n_ground - n_pole
    L = math.sqrt(max(n_ground, 1) / density)
    parts = []
    gx = _rand(n_ground, g, device, 0.0, L)
    gy = _rand(n_ground, g, device, 0.0, L)
    gz = 0.05 * torch.sin(gx * 0.7) * torch.cos(gy * 0.45)
    parts.append(torch.stack([gx, gy, gz], dim=1))
    # buildings: each ~ 400 m^2 of surface
    host_rng = torch.Generator().manual_seed(int(seed) + 7)
    area_left = n_box / density
    left = n_box
    while left > 0:
        w = 4.0 + 8.0 * torch.rand(1, generator=host_rng).item()
        d = 4.0 + 8.0 * torch.rand(1, generator=host_rng).item()
        h = 3.0 + 6.0 * torch.rand(1, generator=host_rng).item()
        yaw = math.pi * torch.rand(1, generator=host_rng).item()
        cx = L * torch.rand(1, generator=host_rng).item()
        cy = L * torch.rand(1, generator=host_rng).item()
        a = 2 * w * h + 2 * d * h + w * d
        cnt = min(left, max(1, int(round(a * density))))
        if area_left - a < 0.2 * a:
            cnt = left
        parts.append(_box_surface(cnt, g, device, cx, cy, w, d, h, yaw))
        left -= cnt
        area_left -= a
    # poles: vertical cylinders r = 0.15..0.4 m, height 3..7 m
    left = n_pole
    while left > 0:
        rad = 0.15 + 0.25 * torch.rand(1, generator=host_rng).item()
        hgt = 3.0 + 4.0 * torch.rand(1, generator=host_rng).item()
        cx = L * torch.rand(1, generator=host_rng).item()
        cy = L * torch.rand(1, generator=host_rng).item()
        a = 2 * math.pi * rad * hgt
        cnt = min(left, max(1, int(round(a * density))))
        th = _rand(cnt, g, device, 0.0, 2 * math.pi)
        zz = _rand(cnt, g, device, 0.0, hgt)
        parts.append(torch.stack([cx + rad * torch.cos(th), cy + rad * torch.sin(th), zz], dim=1))
        left -= cnt
    pts = torch.cat(parts, dim=0)
    pts = pts + _randn(pts.shape, g, device, 2e-3)
    perm = torch.randperm(pts.shape[0], generator=g, device=device)  # map clouds are not spatially ordered
    pts = pts[perm]
    o = torch.tensor(origin, dtype=F64, device=device)
    return (pts + o).contiguous()


def perturb(gt: torch.Tensor, seed: int = 101, noise_std: float = 0.02, drift: float = 0.05,
            outlier_ratio: float = 0.001, outlier_std: float = 5.0, keep_sparse: float = 0.7,
            region_size: float = 10.0, phase: float = 0.0):
    """est = thinned(GT) + smooth drift (<= `drift` m) + N(0, noise_std^2) + sparse outliers."""
    device = gt.device
    g = _gen(seed, device)
    n = gt.shape[0]
    # region-wise thinning (addNonUniformDensity recipe): regions on a checkerboard keep `keep_sparse`
    cell = torch.floor(gt[:, :2] / region_size).to(torch.int64)
    sparse = ((cell[:, 0] + cell[:, 1]) % 2) == 0
    keep = (_rand(n, g, device) < torch.where(sparse, keep_sparse, 1.0))
    p = gt[keep]
    m = p.shape[0]
    # low-frequency drift field
    ph = 0.013
    dx = drift * torch.sin(p[:, 1] * ph + 0.3 + phase) * torch.cos(p[:, 2] * 0.05)
    dy = drift * torch.sin(p[:, 0] * ph * 1.3 + 1.1 + 2.0 * phase)
    dz = 0.5 * drift * torch.cos(p[:, 0] * ph * 0.7 + p[:, 1] * ph * 0.9 + 0.5 * phase)
    p = p + torch.stack([dx, dy, dz], dim=1)
    p = p + _randn((m, 3), g, device, noise_std)
    out = _rand(m, g, device) < outlier_ratio
    p = p + out.to(F64)[:, None] * _randn((m, 3), g, device, outlier_std)
    return p.contiguous()


def campus_pair(n: int, density: float = 2500.0, seed: int = 100, device="cpu", origin=(0.0, 0.0, 0.0), **kw):
    gt = campus_scene(n, density, seed, device, origin)
    est = perturb(gt, seed + 1, **kw)
    return est, gt


def _scan(n_out: int, n_ref: int, density: float, seed: int, sample_seed: int, device, origin, oversample: float, **kw):
    """One independent scan of the `seed` scene (the geometry of campus_scene(n_ref, density, seed)), perturbed, cut to
    exactly n_out points (the perturbation thins ~15 %, hence the oversampling)."""
    n_raw = int(n_out * oversample)
    raw = campus_scene(n_raw, density * n_raw / n_ref, seed, device, origin, sample_seed=sample_seed)
    est = perturb(raw, sample_seed + 1, **kw)
    del raw
    if est.shape[0] < n_out:
        raise ValueError("oversampling factor too small for the requested point count")
    return est[:n_out].contiguous()  # (campus_scene shuffles, so a prefix is a uniform subsample)


def scan_pair(n: int, density: float = 2500.0, seed: int = 100, device="cpu", origin=(0.0, 0.0, 0.0), n_est=None, **kw):
    """Map pair with EQUAL sizes (BASELINE.json: "50 M vs 50 M"): the ground truth is campus_scene(n), the estimated map an
    independent scan of the same scene (its own sample points, not noisy copies of the ground-truth points) with drift,
    noise, outliers and region-wise thinning, cut to exactly n_est (default n) points."""
    gt = campus_scene(n, density, seed, device, origin)
    est = _scan(n if n_est is None else n_est, n, density, seed, seed + 1000, device, origin, 1.5, **kw)
    return est, gt


def multisession_pair(n: int, sessions: int = 3, density: float = 2500.0, seed: int = 100, device="cpu", origin=(0.0, 0.0, 0.0)):
    """C4 ("MS-Dataset multi-session map vs GT"): the estimated map is the union of `sessions` independent
    scans of the scene, each with its own drift field (amplitude and phase), noise level and thinning pattern; n points
    in total, shuffled."""
    gt = campus_scene(n, density, seed, device, origin)
    parts = []
    left = n
    for k in range(sessions):
        m = left // (sessions - k)
        left -= m
        p = _scan(m, n, density, seed, seed + 2000 + 17 * k, device, origin, 1.5,
                  noise_std=0.015 + 0.005 * k, drift=0.03 + 0.02 * k, region_size=8.0 + 3.0 * k, phase=0.9 * k)
        parts.append(p)
    est = torch.cat(parts, 0)
    del parts
    g = _gen(seed + 2999, device)
    est = est[torch.randperm(est.shape[0], generator=g, device=device)].contiguous()
    return est, gt


def tunnel_scene(n: int, density: float = 2500.0, seed: int = 300, device="cpu", n_ref=None, sample_seed=None):
    """Tunnel (cylinder r = 3 m) + flat field + staircase.  The extents follow from n_ref (default n) points at `density`;
    n points are sampled on them (a different sample_seed = an independent scan of the same geometry)."""
    n_ref = n if n_ref is None else n_ref
    g = _gen(seed if sample_seed is None else sample_seed, device)
    n_t = n // 2
    n_f = n // 3
    n_s = n - n_t - n_f
    r_t, r_f = (n_ref // 2), (n_ref // 3)
    r_s = n_ref - r_t - r_f
    Lt = r_t / density / (2 * math.pi * 3.0)
    th = _rand(n_t, g, device, 0.0, 2 * math.pi)
    xt = _rand(n_t, g, device, 0.0, Lt)
    tun = torch.stack([xt, 3.0 * torch.cos(th), 3.0 + 3.0 * torch.sin(th)], dim=1)
    Lf = math.sqrt(r_f / density)
    fld = torch.stack([_rand(n_f, g, device, 0.0, Lf), _rand(n_f, g, device, 10.0, 10.0 + Lf),
                       torch.zeros(n_f, dtype=F64, device=device)], dim=1)
    # staircase: steps 0.3 m deep, 0.17 m high, 2 m wide
    Ls = r_s / density / 2.0
    sx = _rand(n_s, g, device, 0.0, Ls)
    step = torch.floor(sx / 0.3)
    stairs = torch.stack([sx, _rand(n_s, g, device, -12.0, -10.0), step * 0.17], dim=1)
    gt = torch.cat([tun, fld, stairs], dim=0)
    del tun, fld, stairs, th, xt, sx, step
    gt = gt + _randn(gt.shape, g, device, 1e-3)
    return gt[torch.randperm(gt.shape[0], generator=g, device=device)].contiguous()


def tunnel_pair(n: int, density: float = 2500.0, seed: int = 300, device="cpu", equal_sizes: bool = False):
    """C5: tunnel + flat field + staircase: near-rank-1/2 voxel covariances.  est = the perturbed ground truth (thinned to
    ~85 %), or with equal_sizes (BASELINE.json: "100 M-pt dense map pair") an independent, oversampled scan of the same
    geometry, perturbed and cut to exactly n points."""
    gt = tunnel_scene(n, density, seed, device)
    if not equal_sizes:
        return perturb(gt, seed + 1, noise_std=0.01, drift=0.03, outlier_ratio=0.0), gt
    n_raw = int(n * 1.25)
    raw = tunnel_scene(n_raw, density, seed, device, n_ref=n, sample_seed=seed + 1000)
    est = perturb(raw, seed + 1, noise_std=0.01, drift=0.03, outlier_ratio=0.0)
    del raw
    if est.shape[0] < n:
        raise ValueError("oversampling factor too small for the requested point count")
    return est[:n].contiguous(), gt
