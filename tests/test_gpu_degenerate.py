"""Degenerate geometry (BASELINE.json config 4): a tunnel, a flat field and a staircase give voxel
covariances of rank ~1-2, so the eigen-clamp (1e-6) and the Cholesky factors of computeWassersteinDistanceGaussian
(voxel_calculator.cpp:115-140) are exercised where they matter; plus exactly coplanar / collinear / duplicated points for
the MME determinant gate (map_eval.cpp:1692)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from cloud_map_evaluation_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("vs", [1.0, 2.0])
def test_tunnel_scene_awd_scs_parity(eng, vs):
    import oracle
    from cloud_map_evaluation_amd import synth

    est, gt = synth.tunnel_pair(400_000, density=2500.0, seed=300)
    est, gt = est.numpy(), gt.numpy()
    eng.upload(0, est)
    eng.upload(1, gt)
    res = eng.calculateVMD(vs)
    ores = oracle.awd_scs(oracle.VoxelMap(gt, vs), oracle.VoxelMap(est, vs))
    assert res["counts"] == ores["counts"] and res["rows"].shape == ores["rows"].shape and len(res["rows"]) > 30
    assert np.array_equal(res["rows"][:, :6], ores["rows"][:, :6])
    assert np.array_equal(res["rows"][:, 10:12], ores["rows"][:, 10:12])  # populations: exact
    # near-singular covariances: the smallest eigenvalue sits on the clamp for many voxels
    sig = ores["rows"][:, 12:21].reshape(-1, 3, 3)
    lam = np.linalg.eigvalsh(0.5 * (sig + sig.transpose(0, 2, 1)))
    assert (lam[:, 0] < 1e-6).mean() > 0.2
    np.testing.assert_allclose(res["rows"][:, 9], ores["rows"][:, 9], rtol=1e-7)
    np.testing.assert_allclose(res["w_sorted"], ores["w_sorted"], rtol=1e-7)
    np.testing.assert_allclose(res["awd"], ores["awd"], rtol=1e-8)
    np.testing.assert_allclose(res["scs"], ores["scs"], rtol=1e-8)


def test_tunnel_scene_mme_and_stats_parity(eng):
    import oracle
    from cloud_map_evaluation_amd import synth

    est, gt = synth.tunnel_pair(150_000, density=2500.0, seed=301)
    est, gt = est.numpy(), gt.numpy()
    eng.upload(0, est, cell_size=0.1)
    eng.upload(1, gt, cell_size=0.1)
    for slot, cloud, k in ((0, est, 10), (1, gt, 5)):
        mean, ent, valid, nv, s = eng.mme(slot, 0.1, k)
        omean, oent, ovalid, onv, osum = oracle.mme(cloud, 0.1, k)
        assert nv == onv and np.array_equal(valid.astype(bool), ovalid.astype(bool))
        np.testing.assert_allclose(ent, oent, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(mean, omean, rtol=1e-9)
    idx, d2 = eng.nn1(0, 1)
    oidx, od2 = oracle.nn1(gt, est)
    assert np.array_equal(d2, od2) and np.array_equal(idx, oidx)


def test_exactly_degenerate_neighbourhoods_follow_the_determinant_gate(eng):
    """Coplanar grid, collinear points and a pile of duplicates: det is ~0 or exactly 0 -> log gives -inf / nan, the
    finite gate (map_eval.cpp:1692) decides; the device must take the same decisions as the CPU path."""
    import oracle

    g = np.arange(40) * 0.02
    plane = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
    plane = np.concatenate([plane, np.zeros((len(plane), 1))], 1)            # z == 0 exactly
    line = np.stack([np.arange(400) * 0.005 + 5.0, np.full(400, 1.0), np.full(400, 2.0)], 1)
    dup = np.tile(np.array([[9.0, 9.0, 9.0]]), (50, 1))
    cloud = np.concatenate([plane, line, dup]).astype(np.float64)
    eng.upload(0, cloud, cell_size=0.1)
    mean, ent, valid, nv, s = eng.mme(0, 0.1, 10)
    omean, oent, ovalid, onv, osum = oracle.mme(cloud, 0.1, 10)
    assert nv == onv and np.array_equal(valid.astype(bool), ovalid.astype(bool))
    assert np.array_equal(ent == 0.0, oent == 0.0)
    m = ovalid.astype(bool)
    np.testing.assert_allclose(ent[m], oent[m], rtol=1e-6, atol=1e-9)
