"""Host executable (cloud_map_evaluation_amd/host/map_eval), CPU-only checks: the reference's YAML key set is accepted with
the same required/optional split (map_eval_main.cpp:120-208), PCD/PLY readers handle every encoding and strip NaN/inf
(map_eval.cpp:6), and without a GPU the run fails loudly (exit code != 0, no CPU metric path)."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "cloud_map_evaluation_amd", "host", "map_eval")

CONFIG = """# same keys as map_eval/config/config.yaml
registration_methods: 2
icp_max_distance: 1.0      # trailing comment
accuracy_level: [0.2, 0.1, 0.08, 0.05, 0.01]
initial_matrix:
  - [0.5, -0.5, 0.0, 1.25]
  - [0.5, 0.5, 0.0, -2.0]
  - [0.0, 0.0, 1.0, 0.125]
  - [0.0, 0.0, 0.0, 1.0]
estimate_map_path: {est}
gt_map_path: {gt}
scene_name: unit_test
save_immediate_result: true
evaluate_mme: true
use_tbb_mme: true
evaluate_gt_mme: false
nn_radius: 0.1
evaluate_using_initial: true
evaluate_noise_gt: false
vmd_voxel_size: 3.0
downsample_size: 0.0
use_visualization: false
enable_debug: false
"""


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.dirname(EXE), "-s"])
    return EXE


def run(exe, *args):
    return subprocess.run([exe, *args], capture_output=True, text=True, timeout=120)


def test_parse_config_same_keys_as_reference(exe, tmp_path):
    cfg = tmp_path / "config.yaml"
    cfg.write_text(CONFIG.format(est=str(tmp_path / "est"), gt=str(tmp_path / "gt.pcd")))
    r = run(exe, "--parse-config", str(cfg))
    assert r.returncode == 0, r.stderr
    p = json.loads(r.stdout)
    assert p["registration_methods"] == 2 and p["icp_max_distance"] == 1.0
    assert p["accuracy_level"] == [0.2, 0.1, 0.08, 0.05, 0.01]
    assert p["initial_matrix"] == [0.5, -0.5, 0.0, 1.25, 0.5, 0.5, 0.0, -2.0, 0.0, 0.0, 1.0, 0.125, 0.0, 0.0, 0.0, 1.0]
    assert p["evaluate_mme"] is True and p["evaluate_gt_mme"] is False and p["evaluate_using_initial"] is True
    assert p["estimate_map_path"].endswith("/est/")              # '/' appended (map_eval_main.cpp:165-167)
    assert p["result_path"].endswith("/est/map_results/")        # (:170)
    assert p["pcd_file_name"] == "map.pcd"                       # default (map_eval.h:65)
    assert p["downsample_size"] == 0.0 and p["vmd_voxel_size"] == 3.0 and p["nn_radius"] == 0.1


@pytest.mark.parametrize("missing", ["registration_methods", "icp_max_distance", "save_immediate_result", "evaluate_mme",
                                     "evaluate_gt_mme", "evaluate_using_initial", "nn_radius", "vmd_voxel_size", "downsample_size",
                                     "estimate_map_path", "gt_map_path", "scene_name", "enable_debug"])
def test_missing_required_key_is_an_error(exe, tmp_path, missing):
    text = "\n".join(l for l in CONFIG.format(est="/a", gt="/b.pcd").splitlines() if not l.startswith(missing + ":"))
    cfg = tmp_path / "config.yaml"
    cfg.write_text(text)
    r = run(exe, "--parse-config", str(cfg))
    assert r.returncode != 0 and "Failed to load configuration" in r.stderr and missing in r.stderr


def test_optional_keys_and_the_shipped_reference_configs(exe, tmp_path):
    cfg = tmp_path / "config.yaml"
    cfg.write_text(CONFIG.format(est="/a", gt="/b.ply") + "pcd_file_name: final_map_lidar.pcd\ngpu_device: 3\nstrict_reference: true\n")
    p = json.loads(run(exe, "--parse-config", str(cfg)).stdout)
    assert p["pcd_file_name"] == "final_map_lidar.pcd" and p["gpu_device"] == 3 and p["strict_reference"] is True
    ref_dir = "/root/reference/map_eval/config"
    if os.path.isdir(ref_dir):  # only where the reference tree is mounted
        for name in ("config.yaml", "config_building_day.yaml", "config_corridor.yaml", "config_geode.yaml"):
            r = run(exe, "--parse-config", os.path.join(ref_dir, name))
            assert r.returncode == 0, (name, r.stderr)
            assert len(json.loads(r.stdout)["accuracy_level"]) == 5


def _lzf_literal(data: bytes) -> bytes:
    """A valid LZF stream made of literal runs only, except one hand-made back-reference."""
    out = bytearray()
    i = 0
    while i < len(data):
        n = min(32, len(data) - i)
        out.append(n - 1)
        out += data[i:i + n]
        i += n
    return bytes(out)


def _write_pcd(path, pts, kind, dtype="f4"):
    n = len(pts)
    size = 4 if dtype == "f4" else 8
    hdr = (f"# .PCD v0.7\nVERSION 0.7\nFIELDS x y z intensity\nSIZE {size} {size} {size} 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
           f"WIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA {kind}\n")
    inten = np.arange(n, dtype="f4")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if kind == "ascii":
            for p, it in zip(pts, inten):
                f.write((" ".join(repr(float(v)) if np.isfinite(v) else "nan" for v in p) + f" {it}\n").encode())
        elif kind == "binary":
            rec = np.zeros(n, dtype=[("x", dtype), ("y", dtype), ("z", dtype), ("i", "f4")])
            rec["x"], rec["y"], rec["z"], rec["i"] = pts[:, 0], pts[:, 1], pts[:, 2], inten
            f.write(rec.tobytes())
        else:  # binary_compressed: SoA payload, LZF
            raw = b"".join(pts[:, d].astype(dtype).tobytes() for d in range(3)) + inten.tobytes()
            comp = _lzf_literal(raw)
            f.write(struct.pack("<II", len(comp), len(raw)) + comp)


@pytest.mark.parametrize("kind,dtype", [("ascii", "f4"), ("binary", "f4"), ("binary", "f8"), ("binary_compressed", "f4")])
def test_pcd_reader_all_encodings_and_nan_removal(exe, tmp_path, kind, dtype):
    rng = np.random.default_rng(0)
    pts = rng.uniform(-100, 100, (1000, 3)).astype(dtype).astype(np.float64)
    pts[5, 1] = np.nan
    pts[77, 0] = np.inf
    path = tmp_path / f"c_{kind}_{dtype}.pcd"
    _write_pcd(path, pts, kind, dtype)
    r = run(exe, "--cloud-info", str(path))
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout)
    good = pts[np.isfinite(pts).all(1)]
    assert info["points"] == len(good) == 998
    np.testing.assert_allclose(info["sum"], good.sum(0), rtol=1e-9, atol=1e-6)


def test_lzf_back_reference(exe, tmp_path):
    # payload "abcabcabc..." encoded as literal "abc" + one long back-reference (exercises the overlap copy)
    n = 64
    raw = (np.tile(np.array([1.5, -2.25, 3.0], dtype="f4"), n)).tobytes()  # x,y,z blocks are all identical floats
    pts = np.zeros((n, 3))
    # build SoA payload: x block (n floats 1.5 ...)?  simpler: all three fields hold the repeating 12-byte pattern
    payload = raw  # 3 * n floats = x block | y block | z block
    comp = bytearray([11]) + payload[:12]  # literal run of 12 bytes
    remaining = len(payload) - 12
    while remaining > 0:
        ln = min(remaining, 264)  # max match length 7 + 255 + 2
        if ln < 3:
            comp += bytearray([ln - 1]) + payload[len(payload) - remaining:len(payload) - remaining + ln]
        elif ln - 2 < 7:
            comp += bytearray([((ln - 2) << 5) | 0, 11])  # offset 12 -> (0 << 8) + 11 + 1
        else:
            comp += bytearray([(7 << 5) | 0, ln - 2 - 7, 11])
        remaining -= ln
    hdr = f"VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH {n}\nHEIGHT 1\nPOINTS {n}\nDATA binary_compressed\n"
    path = tmp_path / "lzf.pcd"
    path.write_bytes(hdr.encode() + struct.pack("<II", len(comp), len(payload)) + bytes(comp))
    info = json.loads(run(exe, "--cloud-info", str(path)).stdout)
    vals = np.frombuffer(payload, dtype="f4").reshape(3, n)  # SoA
    assert info["points"] == n
    np.testing.assert_allclose(info["sum"], vals.sum(1), rtol=1e-12)


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian"])
def test_ply_reader(exe, tmp_path, fmt):
    rng = np.random.default_rng(1)
    pts = rng.uniform(-5, 5, (500, 3))
    path = tmp_path / f"c_{fmt}.ply"
    hdr = (f"ply\nformat {fmt} 1.0\nelement vertex {len(pts)}\nproperty double x\nproperty double y\nproperty double z\n"
           "property uchar red\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if fmt == "ascii":
            for p in pts:
                f.write((" ".join(repr(float(v)) for v in p) + " 7\n").encode())
        else:
            rec = np.zeros(len(pts), dtype=[("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("r", "u1")])
            rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
            f.write(rec.tobytes())
    info = json.loads(run(exe, "--cloud-info", str(path)).stdout)
    assert info["points"] == 500
    np.testing.assert_allclose(info["sum"], pts.sum(0), rtol=1e-12)


def test_full_run_without_gpu_fails_loudly(exe, tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    est_dir = tmp_path / "est"
    est_dir.mkdir()
    pts = np.random.default_rng(0).uniform(0, 1, (200, 3))
    _write_pcd(est_dir / "map.pcd", pts, "binary", "f8")
    _write_pcd(tmp_path / "gt.pcd", pts, "binary", "f8")
    cfg = tmp_path / "config.yaml"
    cfg.write_text(CONFIG.format(est=str(est_dir), gt=str(tmp_path / "gt.pcd")))
    r = run(exe, str(cfg))
    assert r.returncode != 0
    assert "GPU engine unavailable" in r.stderr and "no CPU fallback" in r.stderr
    # the header of map_results.txt is still written (append mode, map_eval.h:168-185)
    txt = (est_dir / "map_results" / "map_results.txt").read_text()
    assert "unit_test =====================" in txt and "Estimated-Ground Truth point count: 200 / 200" in txt


@pytest.mark.parametrize("kind", ["ascii", "binary", "binary_compressed"])
def test_pcd_reader_keeps_normals_with_their_points(exe, tmp_path, kind):
    """normal_x / normal_y / normal_z travel with the points (point-to-plane ICP needs them on the target): a row whose
    coordinates are not finite is dropped together with its normal; a file without the fields yields no normals."""
    rng = np.random.default_rng(3)
    n = 500
    pts = rng.uniform(-50, 50, (n, 3)).astype("f4").astype(np.float64)
    nrm = rng.normal(size=(n, 3)).astype("f4").astype(np.float64)
    pts[7, 2] = np.nan
    hdr = (f"# .PCD v0.7\nVERSION 0.7\nFIELDS x y z normal_x normal_y normal_z curvature\nSIZE 4 4 4 4 4 4 4\n"
           f"TYPE F F F F F F F\nCOUNT 1 1 1 1 1 1 1\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA {kind}\n")
    cols = np.hstack([pts, nrm, np.zeros((n, 1))]).astype("f4")
    path = tmp_path / f"n_{kind}.pcd"
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if kind == "ascii":
            for row in cols:
                f.write((" ".join("nan" if not np.isfinite(v) else repr(float(v)) for v in row) + "\n").encode())
        elif kind == "binary":
            f.write(cols.tobytes())
        else:
            raw = b"".join(cols[:, c].tobytes() for c in range(7))
            comp = _lzf_literal(raw)
            f.write(struct.pack("<II", len(comp), len(raw)) + comp)
    info = json.loads(run(exe, "--cloud-info", str(path)).stdout)
    keep = np.isfinite(pts).all(1)
    assert info["points"] == info["normals"] == keep.sum() == n - 1
    np.testing.assert_allclose(info["normal_sum"], nrm[keep].sum(0), rtol=1e-9, atol=1e-6)
    plain = tmp_path / "plain.pcd"
    _write_pcd(plain, pts[keep], "binary")
    assert json.loads(run(exe, "--cloud-info", str(plain)).stdout)["normals"] == 0


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian"])
def test_ply_reader_keeps_normals_with_their_points(exe, tmp_path, fmt):
    rng = np.random.default_rng(5)
    pts = rng.uniform(-5, 5, (300, 3))
    nrm = rng.normal(size=(300, 3)).astype("f4").astype(np.float64)
    pts[11, 0] = np.nan
    path = tmp_path / f"n_{fmt}.ply"
    hdr = (f"ply\nformat {fmt} 1.0\nelement vertex {len(pts)}\nproperty double x\nproperty double y\nproperty double z\n"
           "property float nx\nproperty float ny\nproperty float nz\nend_header\n")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if fmt == "ascii":
            for p, q in zip(pts, nrm):
                f.write((" ".join("nan" if not np.isfinite(v) else repr(float(v)) for v in list(p) + list(q)) + "\n").encode())
        else:
            rec = np.zeros(len(pts), dtype=[("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")])
            rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
            rec["nx"], rec["ny"], rec["nz"] = nrm[:, 0], nrm[:, 1], nrm[:, 2]
            f.write(rec.tobytes())
    info = json.loads(run(exe, "--cloud-info", str(path)).stdout)
    keep = np.isfinite(pts).all(1)
    assert info["points"] == info["normals"] == 299
    np.testing.assert_allclose(info["normal_sum"], nrm[keep].sum(0), rtol=1e-9, atol=1e-6)
