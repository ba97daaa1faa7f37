"""Point-to-point ICP (reference map_eval.cpp:1369-1371 with registration_methods: 0): the
device does the correspondence search + the Kabsch sums, the host the 3x3 solve.  Checked against a CPU loop built from the
oracle's KD-tree (same gate d2 < max^2, same convergence rule) and against a known rigid motion."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "cloud_map_evaluation_amd", "host", "map_eval")
TRUNC = (0.2, 0.1, 0.08, 0.05, 0.01)


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from cloud_map_evaluation_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


def _rigid(rx, ry, rz, t):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = t
    return T


def _pair(n=60_000, seed=3):
    from cloud_map_evaluation_amd import synth

    est, gt = synth.campus_pair(n, seed=seed)
    return est.numpy(), gt.numpy()


def _cpu_icp(est, gt, max_d, max_iteration=30):
    """The same loop on the CPU: oracle KD-tree for the correspondences, numpy for the sums."""
    import oracle
    from cloud_map_evaluation_amd.icp import kabsch_update

    def evaluate(src):
        idx, d2 = oracle.nn1(gt, src, threads=0)
        m = d2 < max_d * max_d
        n = int(m.sum())
        return m, idx, n / len(src), (float(np.sqrt(d2[m].sum() / n)) if n else 0.0), n

    src = est.copy()
    total = np.eye(4)
    m, idx, fit, rmse, n = evaluate(src)
    it = 0
    for it in range(1, max_iteration + 1):
        if n < 3:
            break
        p, q = src[m], gt[idx[m]]
        upd = kabsch_update(n, p.sum(0), q.sum(0), p.T @ q)
        total = upd @ total
        src = oracle.transform(src, upd)
        pf, pr = fit, rmse
        m, idx, fit, rmse, n = evaluate(src)
        if abs(pf - fit) < 1e-6 and abs(pr - rmse) < 1e-6:
            break
    return dict(transformation=total, fitness=fit, inlier_rmse=rmse, n_corr=n, iterations=it, cloud=src)


def test_icp_sums_match_numpy(eng):
    est, gt = _pair()
    est = est + np.array([0.03, -0.02, 0.01])
    eng.upload(0, est)
    eng.upload(1, gt)
    idx, d2 = eng.nn1(0, 1)
    for max_d in (0.05, 0.5, 1e6):
        s = eng.icp_p2p_sums(0, max_d)
        m = d2 < max_d * max_d
        o = np.array(list(s.origin))
        p, q = est[m] - o, gt[idx[m]] - o
        assert s.n_corr == int(m.sum()) and s.n_source == len(est)
        np.testing.assert_allclose(list(s.sum_p), p.sum(0), rtol=1e-10, atol=1e-7)
        np.testing.assert_allclose(list(s.sum_q), q.sum(0), rtol=1e-10, atol=1e-7)
        np.testing.assert_allclose(np.array(list(s.sum_pq)).reshape(3, 3), p.T @ q, rtol=1e-10, atol=1e-5)
        np.testing.assert_allclose(s.sum_d2, d2[m].sum(), rtol=1e-12)
    s0 = eng.icp_p2p_sums(0, 1e-9)  # nothing passes
    assert s0.n_corr == 0 and s0.sum_d2 == 0.0
    from cloud_map_evaluation_amd.engine import MapEvalError

    with pytest.raises(MapEvalError):  # Open3D: "Invalid max_correspondence_distance" for <= 0
        eng.icp_p2p_sums(0, 0.0)


def test_icp_recovers_a_known_rigid_motion(eng):
    """Noise-free copy moved by a small rigid motion: ICP must bring it back onto the target."""
    _, gt = _pair(40_000, seed=5)
    T = _rigid(0.004, -0.003, 0.006, [0.04, -0.03, 0.02])
    c = gt.mean(0)
    A = np.eye(4)
    A[:3, :3] = T[:3, :3]
    A[:3, 3] = c + T[:3, 3] - T[:3, :3] @ c  # rotate about the centroid so the motion stays small everywhere
    import oracle

    est = oracle.transform(gt[::2].copy(), A)
    eng.upload(0, est)
    eng.upload(1, gt)
    r = eng.performICPRegistration(1.0)
    # the default criteria stop once fitness and RMSE change by < 1e-6 per iteration, not at machine precision
    np.testing.assert_allclose(r["transformation"] @ A, np.eye(4), atol=1e-3)
    assert r["fitness"] == 1.0 and r["inlier_rmse"] < 1e-3
    np.testing.assert_allclose(eng.download(0), gt[::2], atol=5e-3)


def test_icp_matches_cpu_loop(eng):
    est, gt = _pair(60_000, seed=9)
    A = _rigid(0.002, 0.001, -0.003, [0.02, 0.01, -0.015])
    import oracle

    est = oracle.transform(est, A)
    ref = _cpu_icp(est, gt, 0.5)
    eng.upload(0, est)
    eng.upload(1, gt)
    r = eng.performICPRegistration(0.5)
    assert r["iterations"] == ref["iterations"] and r["n_corr"] == ref["n_corr"]
    assert r["fitness"] == ref["fitness"]
    np.testing.assert_allclose(r["transformation"], ref["transformation"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(r["inlier_rmse"], ref["inlier_rmse"], rtol=1e-9)
    np.testing.assert_allclose(eng.download(0), ref["cloud"], rtol=0, atol=1e-9)
    assert r["fitness"] > 0.9


def test_icp_with_too_few_correspondences_stops(eng):
    est = np.array([[100.0, 100.0, 100.0], [101.0, 100.0, 100.0], [100.0, 101.0, 100.0], [100.0, 100.0, 101.0]])
    gt = np.random.default_rng(0).uniform(0, 1, (1000, 3))
    eng.upload(0, est)
    eng.upload(1, gt)
    r = eng.performICPRegistration(0.5)
    assert r["n_corr"] == 0 and r["fitness"] == 0.0 and r["iterations"] == 1
    np.testing.assert_array_equal(r["transformation"], np.eye(4))


def _write_pcd(path, pts):
    n = len(pts)
    hdr = (f"# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 8 8 8\nTYPE F F F\nCOUNT 1 1 1\nWIDTH {n}\nHEIGHT 1\n"
           f"VIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(np.ascontiguousarray(pts, dtype="<f8").tobytes())


def test_host_icp_path(tmp_path):
    """evaluate_using_initial: false + registration_methods: 0 through the host executable (map_eval.cpp:60-63,
    :191-237, :1147-1202): "Aligned cloud" / "Aligned results" lines and metrics on ICP's final correspondence set."""
    import oracle
    from cloud_map_evaluation_amd import synth

    est, gt = synth.cube_pair(50_000, seed=8)
    est, gt = est.numpy(), gt.numpy()
    A = _rigid(0.001, -0.002, 0.0015, [0.01, -0.008, 0.006])
    est = oracle.transform(est, A)
    est_dir = tmp_path / "est"
    est_dir.mkdir()
    _write_pcd(est_dir / "map.pcd", est)
    _write_pcd(tmp_path / "gt.pcd", gt)
    cfg = tmp_path / "config.yaml"
    cfg.write_text(f"""registration_methods: 0
icp_max_distance: 0.3
accuracy_level: [0.2, 0.1, 0.08, 0.05, 0.01]
initial_matrix:
  - [1.0, 0.0, 0.0, 0.0]
  - [0.0, 1.0, 0.0, 0.0]
  - [0.0, 0.0, 1.0, 0.0]
  - [0.0, 0.0, 0.0, 1.0]
estimate_map_path: {est_dir}
gt_map_path: {tmp_path / 'gt.pcd'}
scene_name: cube_icp
save_immediate_result: true
evaluate_mme: false
evaluate_gt_mme: false
nn_radius: 0.1
evaluate_using_initial: false
evaluate_noise_gt: false
vmd_voxel_size: 0.5
downsample_size: 0.0
use_visualization: false
enable_debug: true
""")
    r = subprocess.run([EXE, str(cfg)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    txt = open(est_dir / "map_results" / "map_results.txt").read()
    ref = _cpu_icp(est, gt, 0.3)
    m = re.search(r"Aligned cloud:\s+((?:[-\d.e+]+\s+){16})", txt)
    assert m, txt
    Th = np.array([float(v) for v in m.group(1).split()]).reshape(4, 4)
    np.testing.assert_allclose(Th, ref["transformation"], atol=2e-5)  # 5 decimals in the file
    m = re.search(r"Aligned results: ([\d.]+) (\d+)", txt)
    assert int(m.group(2)) == ref["n_corr"]
    np.testing.assert_allclose(float(m.group(1)), ref["fitness"], atol=6e-6)
    o_eg = oracle.reg_stats(ref["cloud"], gt, 0.3, 1, TRUNC)
    o_ge = oracle.reg_stats(gt, ref["cloud"], 0.3, 1, TRUNC)
    vals = {k: [float(v) for v in re.search(rf"^{re.escape(k)}: (.*)$", txt, flags=re.M).group(1).split()]
            for k in ("RMSE/AC", "Comp", "FULL CD")}
    np.testing.assert_allclose(vals["RMSE/AC"], o_eg.rmse, rtol=1e-7)
    np.testing.assert_allclose(vals["Comp"], o_eg.fitness, rtol=0, atol=2e-15)
    assert np.all(np.isfinite(o_ge.rmse))  # gt -> est direction feeds cd_vec only (not written: map_eval.cpp:449 is commented out)
    np.testing.assert_allclose(vals["FULL CD"][0], oracle.chamfer(ref["cloud"], gt), atol=6e-6)


def _write_pcd_normals(path, pts, nrm):
    n = len(pts)
    hdr = (f"# .PCD v0.7\nVERSION 0.7\nFIELDS x y z normal_x normal_y normal_z\nSIZE 8 8 8 8 8 8\nTYPE F F F F F F\n"
           f"COUNT 1 1 1 1 1 1\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(np.ascontiguousarray(np.hstack([pts, nrm]), dtype="<f8").tobytes())


def _host_cfg(tmp_path, est_dir, method, max_d=0.3):
    cfg = tmp_path / "config.yaml"
    cfg.write_text(f"""registration_methods: {method}
icp_max_distance: {max_d}
accuracy_level: [0.2, 0.1, 0.08, 0.05, 0.01]
initial_matrix:
  - [1.0, 0.0, 0.0, 0.0]
  - [0.0, 1.0, 0.0, 0.0]
  - [0.0, 0.0, 1.0, 0.0]
  - [0.0, 0.0, 0.0, 1.0]
estimate_map_path: {est_dir}
gt_map_path: {tmp_path / 'gt.pcd'}
scene_name: x
save_immediate_result: true
evaluate_mme: false
evaluate_gt_mme: false
nn_radius: 0.1
evaluate_using_initial: false
evaluate_noise_gt: false
vmd_voxel_size: 0.5
downsample_size: 0.0
use_visualization: false
enable_debug: false
""")
    return cfg


@pytest.mark.parametrize("method", [1, 2])
def test_host_point_to_plane_and_generalized_icp(tmp_path, method):
    """registration_methods 1 / 2 through the host executable (map_eval.cpp:1373-1384): the shipped configs use 2.
    Checked against the oracle's restatement of Open3D's loops (same normals for point-to-plane: they travel in the PCD)."""
    import oracle
    from cloud_map_evaluation_amd import synth

    est, gt = synth.campus_pair(40_000, seed=12)
    est, gt = est.numpy(), gt.numpy()
    est = oracle.transform(est, _rigid(0.002, -0.001, 0.003, [0.02, -0.015, 0.01]))
    est_dir = tmp_path / "est"
    est_dir.mkdir()
    _write_pcd(est_dir / "map.pcd", est)
    n_gt = oracle.estimate_normals_knn(gt, 20)
    if method == 1:
        _write_pcd_normals(tmp_path / "gt.pcd", gt, n_gt)
    else:
        _write_pcd(tmp_path / "gt.pcd", gt)
    r = subprocess.run([EXE, str(_host_cfg(tmp_path, est_dir, method, 0.5))], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    txt = open(est_dir / "map_results" / "map_results.txt").read()
    ref = oracle.registration_icp(method, est, gt, 0.5, tgt_normals=n_gt)
    m = re.search(r"Aligned cloud:\s+((?:[-\d.e+]+\s+){16})", txt)
    Th = np.array([float(v) for v in m.group(1).split()]).reshape(4, 4)
    np.testing.assert_allclose(Th, ref["transformation"], atol=2e-5)  # 5 decimals in the file
    m = re.search(r"Aligned results: ([\d.]+) (\d+)", txt)
    assert int(m.group(2)) == ref["n_corr"]
    o_eg = oracle.reg_stats(ref["cloud"], gt, 0.5, 1, TRUNC)
    vals = {k: [float(v) for v in re.search(rf"^{re.escape(k)}: (.*)$", txt, flags=re.M).group(1).split()]
            for k in ("RMSE/AC", "Comp", "FULL CD")}
    np.testing.assert_allclose(vals["RMSE/AC"], o_eg.rmse, rtol=1e-6)
    np.testing.assert_allclose(vals["Comp"], o_eg.fitness, rtol=0, atol=2e-15)
    np.testing.assert_allclose(vals["FULL CD"][0], oracle.chamfer(ref["cloud"], gt), atol=6e-6)


def test_host_point_to_plane_needs_target_normals(tmp_path):
    """Open3D refuses point-to-plane ICP on a target without normals; so does the host (no silent estimate)."""
    est_dir = tmp_path / "est"
    est_dir.mkdir()
    pts = np.random.default_rng(1).uniform(0, 1, (500, 3))
    _write_pcd(est_dir / "map.pcd", pts)
    _write_pcd(tmp_path / "gt.pcd", pts)
    r = subprocess.run([EXE, str(_host_cfg(tmp_path, est_dir, 1))], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "normals" in (r.stdout + r.stderr)
    r = subprocess.run([EXE, str(_host_cfg(tmp_path, est_dir, 3))], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "Invalid registration type" in (r.stdout + r.stderr)
