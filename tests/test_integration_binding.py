"""The reference-side binding of INTEGRATION.md section B is CODE, not prose: its three fenced blocks are extracted verbatim,
compiled inside a cut-down MapEval (tests/integration/harness.cpp: the reference's member names and types, map_eval.h:60-116,
:322-353, over stand-in Open3D / Eigen headers — neither library is installed here) and linked against libmapeval_hip.so.
CPU: it compiles, links and, without a GPU, fails loudly through the reference's own error path (process() returns -1).
GPU: the binary's scalars equal those of the Python face on the same clouds."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "integration")


def _blocks():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    out = {}
    for name in ("include", "members", "process"):
        m = re.search(r"<!-- binding:%s -->\s*```cpp\n(.*?)```" % name, md, re.S)
        assert m, f"INTEGRATION.md lost its binding:{name} block"
        out[name] = m.group(1)
    return out


@pytest.fixture(scope="module")
def binary(tmp_path_factory):
    import __graft_entry__ as g

    g.build()
    d = tmp_path_factory.mktemp("binding")
    b = _blocks()
    for name in ("members", "process"):
        open(d / f"binding_{name}.inc", "w").write(b[name])
    # the include block goes where map_eval.h has its includes: in front of the harness
    open(d / "unit.cpp", "w").write(b["include"] + '#include "harness.cpp"\n')
    exe = str(d / "binding_check")
    lib = os.path.join(ROOT, "cloud_map_evaluation_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
           f"-I{d}", f"-I{SRC}", f"-I{SRC}/stub", f"-I{ROOT}/include", str(d / "unit.cpp"), "-o", exe, f"-L{lib}", "-lmapeval_hip",
           f"-Wl,-rpath,{lib}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, "the binding of INTEGRATION.md section B does not compile / link:\n" + r.stderr[-4000:]
    return exe


def test_binding_blocks_use_only_the_public_abi():
    b = _blocks()
    assert '#include "mapeval_hip.h"' in b["include"] and "me_ctx *gpu_" in b["members"]
    called = set(re.findall(r"^\s*(?:[\w:<>\s\*&=]*?=\s*)?(me_\w+)\(", "\n".join(
        l for l in b["process"].splitlines() if not l.lstrip().startswith("//")), re.M))
    header = open(os.path.join(ROOT, "include", "mapeval_hip.h")).read()
    assert {"me_create", "me_upload_cloud", "me_mme", "me_nn1", "me_nn_stats", "me_awd_scs"} <= called
    for f in called:
        assert re.search(r"\b%s\(" % f, header), f"{f} is not declared in include/mapeval_hip.h"


def test_binding_compiles_links_and_fails_loudly_without_a_gpu(binary):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([binary, "2000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "ERROR:" in r.stderr and "RESULT" not in r.stdout  # process() == -1, as map_eval.cpp:15-35


@pytest.mark.gpu
def test_binding_runs_and_matches_the_python_face(binary, tmp_path):
    from cloud_map_evaluation_amd.engine import Engine, Param

    n = 20000
    dump = str(tmp_path / "clouds.bin")
    r = subprocess.run([binary, str(n), dump], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    vals = [float(x) for x in re.search(r"RESULT (.*)", r.stdout).group(1).split()]
    raw = np.fromfile(dump, dtype=np.float64).reshape(2, n, 3)
    T = np.eye(4)
    T[0, 3], T[1, 3] = 0.004, -0.003
    P = Param(icp_max_distance_=1.0, nn_radius_=0.1, vmd_voxel_size_=0.5, initial_matrix_=T)
    with Engine(0) as eng:
        eng.upload(0, raw[0], T=T, cell_size=0.1)
        eng.upload(1, raw[1], cell_size=0.1)
        out = eng.run_suite(P)
    exp = [out.est_gt.rmse[0], out.est_gt.fitness[0], out.gt_est.rmse[0], out.full_chamfer, out.mme_est, out.mme_gt, out.awd,
           out.scs, out.n_w_voxels]
    assert vals[8] == exp[8] > 0
    np.testing.assert_array_equal(vals[:8], exp[:8])  # same library, same calls: bit for bit
