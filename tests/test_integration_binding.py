"""The reference-side binding of INTEGRATION.md section B is CODE, not prose, and it runs inside the reference's OWN
MapEval::process() (map_eval.cpp:4-104): oracle/ref_build/apply_binding.py extracts the four fenced blocks verbatim and applies
them to /root/reference/map_eval/src/map_eval.{h,cpp} at build time (temporary directory; nothing of the reference is committed),
`make -C oracle/ref_build patched` compiles that over the functional stand-in headers and links libmapeval_hip.so.
CPU: the blocks use only the public ABI; the patched reference builds, loads, and — without a GPU — fails loudly through the
reference's own error path (process() returns -1).
GPU: patched process() vs the unpatched reference on the same PCD files: map_results.txt metric lines, voxel_errors.txt, the CDF."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocks():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    out = {}
    for name in ("include", "members", "destructor", "process"):
        m = re.search(r"<!-- binding:%s -->\s*```cpp\n(.*?)```" % name, md, re.S)
        assert m, f"INTEGRATION.md lost its binding:{name} block"
        out[name] = m.group(1)
    return out


def _write_pcd(path, p):
    with open(path, "wb") as f:
        f.write((f"# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 8 8 8\nTYPE F F F\nCOUNT 1 1 1\nWIDTH {len(p)}\nHEIGHT 1\n"
                 f"VIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(p)}\nDATA binary\n").encode())
        f.write(np.ascontiguousarray(p, np.float64).tobytes())


def _patched():
    import __graft_entry__ as g
    from oracle import ref

    if not ref.patched_available():
        pytest.skip("neither oracle/_ref/libmapeval_ref_patched.so nor the reference sources are here")
    g.build()  # libmapeval_hip.so first: the patched reference links it
    return ref


def test_binding_blocks_use_only_the_public_abi():
    b = _blocks()
    assert '#include "mapeval_hip.h"' in b["include"] and "me_ctx *gpu_" in b["members"] and "me_destroy(gpu_)" in b["destructor"]
    called = set(re.findall(r"\b(me_[a-z0-9_]+)\(", "\n".join(
        l.split("//")[0] for l in b["process"].splitlines() if not l.lstrip().startswith("//"))))
    called.discard("me_ok")
    header = open(os.path.join(ROOT, "include", "mapeval_hip.h")).read()
    # round 5: the initial-matrix path is ONE call (me_run_suite_from) + fetches of what it left on the device
    assert {"me_create", "me_run_suite_from", "me_mme_fetch", "me_nn_fetch", "me_upload_cloud", "me_mme", "me_awd_scs"} <= called
    for f in called:
        assert re.search(r"\b%s\(" % f, header), f"{f} is not declared in include/mapeval_hip.h"


def test_patched_reference_builds_and_fails_loudly_without_a_gpu(tmp_path):
    import torch

    from cloud_map_evaluation_amd import synth

    ref = _patched()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    est, gt = synth.cube_pair(3_000, seed=6)
    _write_pcd(tmp_path / "global_pcd_lidar.pcd", est.numpy())
    _write_pcd(tmp_path / "gt.pcd", gt.numpy())
    cfg = ref.config(nn_radius=0.2, vmd_voxel_size=0.5, downsample_size=0.05, save_immediate_result=True)
    r = ref.process_patched(cfg, tmp_path, tmp_path / "gt.pcd")
    assert r["rc"] == -1  # me_create found no device: process() returns -1 as for any load failure (map_eval.cpp:15-35); no CPU path


@pytest.mark.gpu
def test_patched_process_equals_the_unpatched_reference(tmp_path):
    """C1 (100 k-point cube pair) through MapEval::process() twice: the reference as it is, and the reference with its
    three hot-path calls replaced by the library.  Same PCDs, same Param."""
    from cloud_map_evaluation_amd import synth

    ref = _patched()
    est, gt = synth.cube_pair(100_000, seed=42)
    est, gt = est.numpy(), gt.numpy()
    out = {}
    for which in ("reference", "patched"):
        d = tmp_path / which
        d.mkdir()
        _write_pcd(d / "global_pcd_lidar.pcd", est)
        _write_pcd(d / "gt.pcd", gt)
        cfg = ref.config(nn_radius=0.2, vmd_voxel_size=0.5, downsample_size=0.01, save_immediate_result=True, evaluate_gt_mme=True)
        out[which] = (ref.process if which == "reference" else ref.process_patched)(cfg, d, d / "gt.pcd")
        assert out[which]["rc"] == 0
    a, b = out["reference"], out["patched"]
    assert (a["n_est"], a["n_gt"]) == (b["n_est"], b["n_gt"])
    # est -> gt statistics: counts bit for bit, the rest to 1e-12 (summation order)
    assert np.array_equal(a["est_gt"]["number"], b["est_gt"]["number"])
    for row in ("mean", "rmse", "fitness", "sigma"):
        np.testing.assert_allclose(b["est_gt"][row], a["est_gt"][row], rtol=1e-12)
    np.testing.assert_allclose([b["mme_est"], b["mme_gt"]], [a["mme_est"], a["mme_gt"]], rtol=1e-10)
    np.testing.assert_allclose([b["vmd"], b["scs"]], [a["vmd"], a["scs"]], rtol=1e-10)

    def lines(txt):
        return {l.split(":")[0]: l.split(":", 1)[1].split() for l in txt.splitlines() if ":" in l}

    la, lb = lines(a["files"]["map_results.txt"]), lines(b["files"]["map_results.txt"])
    for key in ("Estimated-Ground Truth point count", "RMSE/AC", "Comp", "VMD", "SCS", "MME"):
        assert la[key] == lb[key], (key, la[key], lb[key])  # the printed digits (15 / 5 decimals) are the same
    # FULL CD: the reference never computes it on the initial-matrix path (prints 0); the binding fills it in
    assert float(la["FULL CD"][0]) == 0.0 and float(lb["FULL CD"][0]) > 0.0
    # voxel files: same voxels, the reference in hash-map order, the library in ascending voxel order
    va, vb = a["files"]["voxel_errors.txt"], b["files"]["voxel_errors.txt"]
    assert va.shape == vb.shape and va.shape[0] > 50
    va, vb = va[np.lexsort(va[:, :3].T[::-1])], vb[np.lexsort(vb[:, :3].T[::-1])]
    np.testing.assert_allclose(vb, va, rtol=1e-5, atol=1e-9)  # (6 significant digits in the file)
    np.testing.assert_allclose(b["files"]["voxel_wasserstein_cdf.txt"], a["files"]["voxel_wasserstein_cdf.txt"], rtol=1e-5)
    for name in ("map_entropy.pcd", "gt_entropy.pcd", "raw_rendered_dis_map.pcd", "inlier_rendered_dis_map.pcd"):
        assert os.path.getsize(tmp_path / "patched" / "map_results" / name) > 0
