"""bench.py's host-side pieces that need no GPU: the workload table, the kernel-source fingerprint that gates the committed
rocprofv3 traffic figure, the full-size-tree CPU baseline's arithmetic (on a small pair), the scene generators' contracts."""
import json
import os
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_workloads_and_defaults(monkeypatch):
    import bench

    monkeypatch.setattr("sys.argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.workload == "c4_multisession" and a.points == 50_000_000 and a.voxel == 3.0 and a.nn_radius == 0.1
    assert a.cpu_baseline == "full"
    monkeypatch.setattr("sys.argv", ["bench.py", "--workload", "c5_tunnel", "--cpu-sample", "0", "--points", "1000"])
    a = bench.parse()
    assert a.voxel == 2.0 and a.points == 1000 and a.cpu_baseline == "off"  # config_geode.yaml:60; --cpu-sample 0 kept as an alias
    assert set(bench.WORKLOADS) == {"c4_multisession", "c4_dense", "campus", "c3_20m", "c5_tunnel"}


def test_committed_traffic_figure_belongs_to_the_committed_kernels():
    """profiles/traffic.json is quoted by bench.py only when its fingerprint equals the kernel sources': a profile that
    predates a kernel change reads as null, it cannot go stale silently.  The committed pair must match."""
    import bench

    sha = bench.kernel_source_sha()
    assert len(sha) == 12 and sha == bench.kernel_source_sha()
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert {"mme", "nn_grid", "_kernel_source_sha", "_workload", "_points"} <= set(tj)
    assert tj["_kernel_source_sha"] == sha, "profiles/traffic.json was collected at other kernel sources: re-run profiles/run_profile.sh"
    assert tj["_workload"] == "c4_multisession" and tj["_points"] == 50_000_000
    assert tj["mme"] > 33.0 * 50e6  # at least the algorithmic bytes of a 50 M-query launch


def test_scan_generators_give_equal_sizes_and_independent_samples():
    from cloud_map_evaluation_amd import synth

    est, gt = synth.scan_pair(60_000, density=2500.0, seed=5)
    assert est.shape == gt.shape == (60_000, 3)
    # an independent scan: no estimated point is a noisy copy of a ground-truth point at the same index
    assert float((est - gt).norm(dim=1).median()) > 0.5
    e2, g2 = synth.scan_pair(60_000, density=2500.0, seed=5)
    assert np.array_equal(e2.numpy(), est.numpy()) and np.array_equal(g2.numpy(), gt.numpy())  # seeded
    em, gm = synth.multisession_pair(60_000, 3, density=2500.0, seed=5)
    assert em.shape == gm.shape == (60_000, 3) and np.array_equal(gm.numpy(), gt.numpy())
    # same scene: the clouds overlap (median nearest distance of a few centimetres)
    import oracle

    d2 = oracle.nn1(gt.numpy(), em.numpy())[1]
    assert np.sqrt(np.median(d2)) < 0.1


def test_cpu_baseline_full_arithmetic_on_a_small_pair():
    import bench
    from cloud_map_evaluation_amd import synth

    est, gt = synth.scan_pair(40_000, density=2500.0, seed=9)
    P = types.SimpleNamespace(nn_radius_=0.1, vmd_voxel_size_=1.0)
    r = bench.cpu_baseline_full(est.numpy(), gt.numpy(), P, True, frac=0.05)
    s = r["seconds"]
    assert r["kind"] == "port" and r["cores"] >= 1 and r["unit"] == "Mpts/s"
    ref = (3 * s["build_gt_serial"] + 3 * s["build_est_serial"] + s["mme_est_par"] + s["mme_gt_serial"] + s["nn_est_gt_serial"]
           + s["nn_gt_est_serial"] + s["nn_est_gt_par"] + s["nn_gt_est_par"] + s["voxel_serial"] + s["awd_scs"])
    np.testing.assert_allclose(r["extrapolated_suite_seconds"], ref, rtol=0.02)  # (the dict is rounded to ms)
    np.testing.assert_allclose(r["value"], 0.08 / r["extrapolated_suite_seconds"], rtol=1e-9)
    assert r["all_parallel"]["value"] >= 0.5 * r["value"]  # all-parallel is never much slower than the reference's structure
