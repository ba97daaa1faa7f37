"""End-to-end drop-in check on the GPU: the host executable reads the reference's config format and PCD files, runs the
whole suite through the C ABI, and writes map_results.txt / voxel_errors.txt / voxel_wasserstein_cdf.txt whose values
match the CPU oracle (BASELINE config 0: 100 k-point cube map vs noised copy, all metrics)."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "cloud_map_evaluation_amd", "host", "map_eval")
TRUNC = (0.2, 0.1, 0.08, 0.05, 0.01)


def _write_pcd(path, pts):
    n = len(pts)
    hdr = (f"# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 8 8 8\nTYPE F F F\nCOUNT 1 1 1\nWIDTH {n}\nHEIGHT 1\n"
           f"VIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(np.ascontiguousarray(pts, dtype="<f8").tobytes())


def _cfg(est_dir, gt_path, T, mme=True, gt_mme=True, strict=False, downsample=0.0):
    rows = "\n".join("  - [" + ", ".join(repr(float(v)) for v in T[i]) + "]" for i in range(4))
    return f"""registration_methods: 2
icp_max_distance: 1.0
accuracy_level: [0.2, 0.1, 0.08, 0.05, 0.01]
initial_matrix:
{rows}
estimate_map_path: {est_dir}
gt_map_path: {gt_path}
scene_name: cube_c1
save_immediate_result: true
evaluate_mme: {'true' if mme else 'false'}
use_tbb_mme: true
evaluate_gt_mme: {'true' if gt_mme else 'false'}
nn_radius: 0.1
evaluate_using_initial: true
evaluate_noise_gt: false
vmd_voxel_size: 0.5
downsample_size: {downsample}
use_visualization: false
enable_debug: true
strict_reference: {'true' if strict else 'false'}
"""


def _parse_results(path):
    txt = open(path).read()
    out = {}
    for key in ("RMSE/AC", "Comp", "FULL CD", "VMD", "SCS", "MME"):
        m = re.search(rf"^{re.escape(key)}: (.*)$", txt, flags=re.M)
        if m:
            out[key] = [float(v) for v in m.group(1).split()]
    m = re.search(r"point count: (\d+) / (\d+)", txt)
    out["counts"] = (int(m.group(1)), int(m.group(2)))
    return out, txt


@pytest.mark.parametrize("identity", [True, False])
def test_host_run_matches_oracle(tmp_path, identity):
    import oracle
    from cloud_map_evaluation_amd import synth

    assert os.path.exists(EXE), "build the host first (__graft_entry__.build())"
    est, gt = synth.cube_pair(100_000, seed=42)
    est, gt = est.numpy(), gt.numpy()
    T = np.eye(4)
    if not identity:
        th = 0.002
        T[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
        T[:3, 3] = [0.004, -0.002, 0.001]
    est_dir = tmp_path / "est"
    est_dir.mkdir()
    _write_pcd(est_dir / "map.pcd", est)
    _write_pcd(tmp_path / "gt.pcd", gt)
    cfg = tmp_path / "config.yaml"
    cfg.write_text(_cfg(est_dir, tmp_path / "gt.pcd", T))
    r = subprocess.run([EXE, str(cfg)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res, txt = _parse_results(est_dir / "map_results" / "map_results.txt")
    assert res["counts"] == (100_000, 100_000)

    est_t = oracle.transform(est, T)
    o = oracle.reg_stats(est_t, gt, 1.0, 0, TRUNC)
    np.testing.assert_allclose(res["RMSE/AC"], o.rmse, rtol=0, atol=2e-15)     # printed with 15 decimals
    np.testing.assert_allclose(res["Comp"], o.fitness, rtol=0, atol=2e-15)     # inlier counts / N: exact
    np.testing.assert_allclose(res["FULL CD"][0], oracle.chamfer(est_t, gt), atol=6e-6)  # 5 decimals
    # MME is computed on the map as loaded (before the transform), like the reference (map_eval.cpp:56 vs :1206)
    np.testing.assert_allclose(res["MME"][0], oracle.mme(est, 0.1, 10)[0], atol=6e-6)
    np.testing.assert_allclose(res["MME"][1], oracle.mme(gt, 0.1, 5)[0], atol=6e-6)
    ov = oracle.awd_scs(oracle.VoxelMap(gt, 0.5), oracle.VoxelMap(est_t, 0.5))
    np.testing.assert_allclose(res["VMD"][0], ov["awd"], atol=6e-6)
    np.testing.assert_allclose(res["SCS"][0], ov["scs"], atol=6e-6)
    for line in ("Time load-MME-mesh-ICP-Metric-AC-FCD:", "VMD Time voxelization-WD-CDF-SCS:", "AC+MME Time:", "CD+MME Time:",
                 "AWD+SCS Time:", "Ground Truth Path:", "Evaluation Map Path:"):
        assert line in txt
    # voxel_errors.txt: 27 columns, same voxels and W as the oracle (6 significant digits in the file)
    rows = np.loadtxt(est_dir / "map_results" / "voxel_errors.txt")
    assert rows.shape == ov["rows"].shape
    np.testing.assert_allclose(rows, ov["rows"], rtol=2e-5, atol=1e-12)
    cdf = np.loadtxt(est_dir / "map_results" / "voxel_wasserstein_cdf.txt")
    np.testing.assert_allclose(cdf[:, 0], ov["w_sorted"], rtol=2e-5)
    np.testing.assert_allclose(cdf[:, 1], (np.arange(len(cdf)) + 1) / len(cdf), rtol=1e-5)


def test_strict_reference_reproduces_zero_full_cd(tmp_path):
    """The reference never calls computeChamferDistance on the initial-matrix path: FULL CD prints 0.00000."""
    from cloud_map_evaluation_amd import synth

    est, gt = synth.cube_pair(20_000, seed=1)
    est_dir = tmp_path / "est"
    est_dir.mkdir()
    _write_pcd(est_dir / "map.pcd", est.numpy())
    _write_pcd(tmp_path / "gt.pcd", gt.numpy())
    cfg = tmp_path / "config.yaml"
    cfg.write_text(_cfg(est_dir, tmp_path / "gt.pcd", np.eye(4), mme=False, gt_mme=False, strict=True))
    r = subprocess.run([EXE, str(cfg)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res, _ = _parse_results(est_dir / "map_results" / "map_results.txt")
    assert res["FULL CD"] == [0.0] and "MME" not in res


def test_host_run_with_voxel_downsample(tmp_path):
    """downsample_size > 0: the host down-samples both clouds on the device (map_eval.cpp:38-39) before any metric."""
    import oracle
    from cloud_map_evaluation_amd import synth

    est, gt = synth.cube_pair(100_000, seed=4)
    est, gt = est.numpy(), gt.numpy()
    est_dir = tmp_path / "est"
    est_dir.mkdir()
    _write_pcd(est_dir / "map.pcd", est)
    _write_pcd(tmp_path / "gt.pcd", gt)
    cfg = tmp_path / "config.yaml"
    cfg.write_text(_cfg(est_dir, tmp_path / "gt.pcd", np.eye(4), mme=True, gt_mme=False, downsample=0.05))
    r = subprocess.run([EXE, str(cfg)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res, _ = _parse_results(est_dir / "map_results" / "map_results.txt")
    e_ds, g_ds = oracle.voxel_downsample(est, 0.05), oracle.voxel_downsample(gt, 0.05)
    assert res["counts"] == (len(e_ds), len(g_ds))
    o = oracle.reg_stats(e_ds, g_ds, 1.0, 0, TRUNC)
    np.testing.assert_allclose(res["RMSE/AC"], o.rmse, rtol=0, atol=2e-15)
    np.testing.assert_allclose(res["Comp"], o.fitness, rtol=0, atol=2e-15)
    np.testing.assert_allclose(res["MME"][0], oracle.mme(e_ds, 0.1, 10)[0], atol=6e-6)
