#!/usr/bin/env python
"""bench.py — BASELINE.json metric: Mpts/s for the full metric suite (AC/COM/CD + MME + AWD/SCS) on a 50 M vs 50 M map pair.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by torch.distributed.run,
one rank per GPU over RCCL.  One "step" = one pass of the whole hot path over the synthetic pair, starting from the
two clouds RESIDENT IN HBM (raw, unsorted fp64 AoS) and ending with every scalar on the host: space-filling-curve sort + index
build, both 1-NN passes + AC/COM/CD statistics, est-MME (+ GT-MME), voxel Gaussians, AWD, CDF sort, SCS.
`value` = (N_est + N_gt) / step time of that span.  SURVEY.md 8(d) defines the span from HOST memory; the same steps timed from
pinned host buffers (PCIe included) are reported next to it as `h2d_inclusive` — never as `value`.

Workloads (--workload; all synthetic, seeded, equal-size clouds):
  c4_multisession (default)  50 M vs 50 M, est = union of three independent scans with their own drifts     (configs[3])
  c4_dense                   the same pair sampled at the reference's default density (downsample_size 0.01 -> 10^4 pts/m^2)
  campus                     50 M vs 50 M, est = one independent scan (drift + noise + outliers + thinning)
  c3_20m                     20 M vs 20 M, full suite                                                          (configs[2])
  c5_tunnel                  100 M-point ground truth: tunnel + flat field + staircase, vmd_voxel_size 2.0     (configs[4])
--points / --density / --nn-radius / --voxel override the workload's values.

Multi-GPU (N > 1, strong scaling of the same pair): every rank holds 1/N of each cloud (as if each had read its part of
the files); one all-to-all moves every point to the rank that owns its slab or needs it as halo, then each rank sorts,
indexes and searches only that (cloud_map_evaluation_amd/dist.py::suite_step_dist).

Extra objects on the JSON line: "roofline" (dominant kernel, HIP-event timed inside this process), "h2d_inclusive", and
"cpu_baseline" (the CPU oracle = port of the reference's CPU path, on this box's host cores: full-size KD-trees, 1 % query
subsample, best of 3 — BASELINE.md section 3).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

# HIP runtime setting, before the runtime starts: kernel arguments in device memory.  The step is ~500 launches; this takes
# 0.2-0.3 ms off it at 1 and at 8 ranks (profiles/README.md, measurement knobs).  An explicit HIP_FORCE_DEV_KERNARG=0 wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VECTOR_PEAK_TFLOPS = 157.3 / 2.0  # fp64 vector FMA at half the guide's FP32 vector rate (157.3 TFLOP/s spec): 78.6
# algorithmic (compulsory) bytes per unit of work
BYTES_PER_NN_QUERY = 60.0   # 24 B query + 24 B reference point + 12 B result (M = N)
BYTES_PER_MME_QUERY = 33.0  # 24 B point + 8 B entropy + 1 B valid
OVERLAP = True  # --no-overlap switches the second lane off (the per-kernel timing pass always runs without it)
PY_DRIVER = False  # --py-driver: the single-GPU step driven from Python (dist.suite_step + a Python thread) instead of the one C call

WORKLOADS = {
    "c4_multisession": dict(points=50_000_000, density=2500.0, radius=0.1, voxel=3.0,
                            what="multisession_pair: GT campus scene, est = union of 3 independent scans with their own drifts"),
    "c4_dense": dict(points=50_000_000, density=10_000.0, radius=0.1, voxel=3.0,
                     what="multisession_pair at the reference's default density: downsample_size 0.01 (config.yaml:66) = 10^4 pts/m^2 of "
                          "surface, k ~ 300 neighbours inside nn_radius 0.1"),
    "campus": dict(points=50_000_000, density=2500.0, radius=0.1, voxel=3.0,
                   what="scan_pair: GT campus scene, est = one independent scan (drift, noise, outliers, thinning)"),
    "c3_20m": dict(points=20_000_000, density=2500.0, radius=0.1, voxel=3.0, what="scan_pair at 20 M"),
    "c5_tunnel": dict(points=100_000_000, density=2500.0, radius=0.1, voxel=2.0,
                      what="tunnel_pair: tunnel + flat field + staircase (degenerate voxel covariances), est = an independent scan, 100 M + 100 M"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c4_multisession")
    ap.add_argument("--points", type=int, default=0, help="points per cloud (0 = the workload's size)")
    ap.add_argument("--density", type=float, default=0.0, help="surface density, points / m^2 (0 = the workload's)")
    ap.add_argument("--nn-radius", type=float, default=0.0)
    ap.add_argument("--voxel", type=float, default=0.0)
    ap.add_argument("--no-gt-mme", action="store_true")
    ap.add_argument("--cpu-baseline", choices=("full", "sample", "off"), default="full",
                    help="full: full-size KD-trees + 1 %% query subsample (BASELINE.md 3); sample: a 2 M-point pair; off")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="(compat) 0 = --cpu-baseline off")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive timing")
    ap.add_argument("--settle-max", type=int, default=12, help="N = 1: at most this many untimed settling steps before the warm-up (0 = none)")
    ap.add_argument("--no-overlap", action="store_true", help="single lane: every stage back to back on one stream")
    ap.add_argument("--py-driver", action="store_true", help="N = 1: drive the step from Python (round 4's bench) instead of me_run_suite_from")
    a = ap.parse_args()
    w = WORKLOADS[a.workload]
    a.points = a.points or w["points"]
    a.density = a.density or w["density"]
    a.nn_radius = a.nn_radius or w["radius"]
    a.voxel = a.voxel or w["voxel"]
    if a.cpu_sample == 0:
        a.cpu_baseline = "off"
    return a


def make_pair(args, device):
    from cloud_map_evaluation_amd import synth

    if args.workload in ("c4_multisession", "c4_dense"):
        return synth.multisession_pair(args.points, 3, density=args.density, seed=100, device=device)
    if args.workload == "c5_tunnel":
        return synth.tunnel_pair(args.points, density=args.density, seed=300, device=device, equal_sizes=True)
    return synth.scan_pair(args.points, density=args.density, seed=100, device=device)


def kernel_source_sha() -> str:
    """Fingerprint of the kernel sources: committed rocprofv3 figures are only quoted when they were taken at this code."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "cloud_map_evaluation_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def suite_step(eng, dist, world, est_d, gt_d, P, evaluate_gt_mme, comm_dev=None):
    """One full pass; returns the scalars.  All ranks run it (cloud_map_evaluation_amd/dist.py)."""
    import torch

    from cloud_map_evaluation_amd import dist as medist

    dev = torch.device("cuda", torch.cuda.current_device())
    if world > 1:
        # est_d / gt_d are this rank's 1/N of the clouds: slabs + one all-to-all halo exchange
        return medist.suite_step_dist(eng, dist, comm_dev or dev, est_d, gt_d, P, dist.get_rank(), world, evaluate_gt_mme, halo=1.0,
                                      overlap=OVERLAP)  # (the per-kernel timing pass and --no-overlap run ONE lane here too)
    # single GPU: ONE call through the C ABI (me_run_suite_from) — what a C++ host makes from its one thread.  The HBM-bound stages
    # (index of the ground truth, both voxel tables) and one 1-NN direction run on the library's internal second lane under the
    # VALU-bound MME / 1-NN kernels (csrc/me_suite.hip).  PY_DRIVER: the same schedule driven from Python (dist._Lane: the round-4
    # bench), kept as a cross-check of the numbers.
    if PY_DRIVER:
        return medist.suite_step(eng, None, dev, est_d, gt_d, P, evaluate_gt_mme, overlap=OVERLAP)
    from cloud_map_evaluation_amd.engine import Engine

    return Engine.suite_dict(eng.run_suite_from(est_d, gt_d, P, overlap=OVERLAP))


def _best_of(fn, reps=3):
    best = float("inf")
    out = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        best = min(best, time.perf_counter() - t0)
    return best, out


def cpu_baseline_full(est, gt, P, evaluate_gt_mme, frac=0.01):
    """BASELINE.md section 3.  The oracle (port of the reference's CPU path) on THIS workload's clouds: KD-trees over the FULL
    clouds, a seeded `frac` subsample of the queries of every per-point loop, each loop best of 3, extrapolated to all
    queries.  Two totals over the same span as the GPU step (clouds in host memory -> scalars):
      value         the reference's own parallel structure: serial SetGeometry, 6 tree builds (MME est, MME gt, AC/COM x2,
                    CD x2: map_eval.cpp:1619,1449,1214,1227,1401-1402), TBB-like parallel est-MME (:1717), SERIAL gt-MME
                    (:1451), SERIAL AC/COM loops (:1215,:1228), OpenMP CD loops (:1411,:1420), serial voxel build
                    (voxel_calculator.cpp:25);
      all_parallel  every loop and every tree build on all cores (the fair upper end).
    Tree builds and the voxel build are timed once (a 50 M-point serial build takes tens of seconds)."""
    import numpy as np

    import oracle

    cores = os.cpu_count() or 1
    n_e, n_g = len(est), len(gt)
    rng = np.random.default_rng(7)
    sel_e = np.sort(rng.choice(n_e, max(1, int(n_e * frac)), replace=False))
    sel_g = np.sort(rng.choice(n_g, max(1, int(n_g * frac)), replace=False))
    t = {}
    t0 = time.perf_counter(); te = oracle.KDTree(est, 1); t["build_est_serial"] = time.perf_counter() - t0
    t0 = time.perf_counter(); tg = oracle.KDTree(gt, 1); t["build_gt_serial"] = time.perf_counter() - t0
    t0 = time.perf_counter(); tp = oracle.KDTree(est, 0); t["build_est_parallel"] = time.perf_counter() - t0
    tp.close()
    t0 = time.perf_counter(); tp = oracle.KDTree(gt, 0); t["build_gt_parallel"] = time.perf_counter() - t0
    tp.close()
    sc_e, sc_g = n_e / len(sel_e), n_g / len(sel_g)
    r = P.nn_radius_
    # per-point loops on the subsample (seconds for the subsample; scaled below)
    t["mme_est_par"] = _best_of(lambda: te.mme_points(sel_e, r, 10, threads=0))[0] * sc_e
    t["mme_gt_par"] = _best_of(lambda: tg.mme_points(sel_g, r, 5, threads=0))[0] * sc_g
    t["mme_gt_serial"] = _best_of(lambda: tg.mme_points(sel_g[::8], r, 5, threads=1), 1)[0] * sc_g * 8
    qe, qg = est[sel_e], gt[sel_g]
    t["nn_est_gt_par"] = _best_of(lambda: tg.nn1(qe, threads=0))[0] * sc_e
    t["nn_gt_est_par"] = _best_of(lambda: te.nn1(qg, threads=0))[0] * sc_g
    t["nn_est_gt_serial"] = _best_of(lambda: tg.nn1(qe[::4], threads=1), 1)[0] * sc_e * 4
    t["nn_gt_est_serial"] = _best_of(lambda: te.nn1(qg[::4], threads=1), 1)[0] * sc_g * 4
    te.close()
    tg.close()
    # voxel build: serial hash insert + Welford per point; a prefix of the (shuffled) cloud touches the same voxels, so the
    # per-point cost is that of the whole cloud
    m = max(1, int(n_g * 0.04))
    t0 = time.perf_counter()
    vg, ve = oracle.VoxelMap(gt[:m], P.vmd_voxel_size_), oracle.VoxelMap(est[:m], P.vmd_voxel_size_)
    t["voxel_serial"] = (time.perf_counter() - t0) * (n_g + n_e) / (2.0 * m)
    t0 = time.perf_counter(); oracle.awd_scs(vg, ve); t["awd_scs"] = time.perf_counter() - t0
    gt_mme = 1.0 if evaluate_gt_mme else 0.0
    ref = ((3 if evaluate_gt_mme else 2) * t["build_gt_serial"] + 3 * t["build_est_serial"]  # est: MME, AC, CD; gt: (MME), AC, CD
           + t["mme_est_par"] + gt_mme * t["mme_gt_serial"]
           + t["nn_est_gt_serial"] + t["nn_gt_est_serial"]      # AC / COM loops (serial in the reference)
           + t["nn_est_gt_par"] + t["nn_gt_est_par"]            # CD loops (OpenMP)
           + t["voxel_serial"] + t["awd_scs"])
    par = ((3 if evaluate_gt_mme else 2) * t["build_gt_parallel"] + 3 * t["build_est_parallel"]
           + t["mme_est_par"] + gt_mme * t["mme_gt_par"] + 2 * (t["nn_est_gt_par"] + t["nn_gt_est_par"])
           + t["voxel_serial"] + t["awd_scs"])  # (the voxel hash build has no parallel form in the reference)
    n = n_e + n_g
    return {"value": n / 1e6 / ref, "unit": "Mpts/s", "cores": cores, "kind": "port",
            "sample": f"this workload's clouds ({n_g} + {n_e} pts): full-size KD-trees, {frac:.0%} seeded query subsample of every "
                      f"per-point loop (best of 3, extrapolated x{sc_e:.0f}), tree builds timed once, voxel build on a 4 % prefix; "
                      "reference parallel structure (serial builds x6, serial AC/COM and GT-MME loops, parallel est-MME and CD)",
            "extrapolated_suite_seconds": ref,
            "all_parallel": {"value": n / 1e6 / par, "unit": "Mpts/s", "extrapolated_suite_seconds": par,
                             "note": "every tree build and every per-point loop on all cores"},
            "seconds": {k: round(v, 3) for k, v in t.items()}}


def cpu_baseline_reference(args, P, evaluate_gt_mme, n=1_000_000):
    """The REFERENCE'S OWN code (oracle/_ref: its map_eval.cpp + voxel_calculator.cpp compiled over stand-in headers, see
    oracle/ref_build/) running the body of MapEval::process() — computeMME (TBB est loop, serial GT loop),
    calculateMetricsWithInitialMatrix (two serial loops), calculateVMD — on a bounded pair of the same generator, wall clock on
    this box's host cores.  The trees of a 1 M-point cloud are 5-6 levels shallower than the workload's: per-point cost is
    UNDER-stated, the figure is an upper bound of what the reference would reach on the full pair."""
    from oracle import ref
    from cloud_map_evaluation_amd import synth

    if not ref.available():
        return None
    est, gt = synth.scan_pair(n, density=args.density, seed=100)
    est, gt = est.numpy(), gt.numpy()
    cfg = ref.config(trunc=P.trunc_dist_, icp_max_distance=P.icp_max_distance_, nn_radius=P.nn_radius_, vmd_voxel_size=P.vmd_voxel_size_,
                     evaluate_gt_mme=evaluate_gt_mme)
    import contextlib

    t0 = time.perf_counter()
    with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):
        fd = os.dup(1)
        os.dup2(dn.fileno(), 1)  # the reference prints its progress from C++
        try:
            r = ref.suite_initial(est, gt, cfg)
        finally:
            os.dup2(fd, 1)
            os.close(fd)
    dt = time.perf_counter() - t0
    return {"value": 2 * n / 1e6 / dt, "unit": "Mpts/s", "cores": os.cpu_count() or 1, "kind": "reference",
            "sample": f"oracle/_ref (the reference's own sources) on scan_pair {n} + {n} pts: computeMME + calculateMetricsWithInitialMatrix + "
                      f"calculateVMD, {dt:.1f} s wall; serial loops as the reference has them, its TBB / OpenMP loops on all cores",
            "mme_est": r["mme_est"], "vmd": r["vmd"]}


def cpu_baseline_sample(args, P, evaluate_gt_mme, n=2_000_000):
    """Quick variant: the oracle end to end on a small pair of the same generator (trees are NOT full size)."""
    import oracle
    from cloud_map_evaluation_amd import synth

    est, gt = synth.scan_pair(n, density=args.density, seed=100)
    est, gt = est.numpy(), gt.numpy()
    t0 = time.perf_counter()
    oracle.mme(est, P.nn_radius_, 10, mode=2, threads=0)
    if evaluate_gt_mme:
        oracle.mme(gt, P.nn_radius_, 5, mode=0, threads=1)
    oracle.reg_stats(est, gt, P.icp_max_distance_, 0, P.trunc_dist_, threads=1)
    oracle.reg_stats(gt, est, P.icp_max_distance_, 0, P.trunc_dist_, threads=1)
    oracle.chamfer(est, gt, threads=0)
    oracle.awd_scs(oracle.VoxelMap(gt, P.vmd_voxel_size_), oracle.VoxelMap(est, P.vmd_voxel_size_))
    dt = time.perf_counter() - t0
    return {"value": 2 * n / 1e6 / dt, "unit": "Mpts/s", "cores": os.cpu_count() or 1, "kind": "port",
            "sample": f"scan_pair {n} + {n} pts end to end (small trees), reference parallel structure, {dt:.1f} s wall"}


def main():
    global OVERLAP, PY_DRIVER
    args = parse()
    OVERLAP = not args.no_overlap
    PY_DRIVER = args.py_driver
    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # (test hooks, tests/test_gpu_dist.py: ME_BENCH_BACKEND=gloo + ME_BENCH_SINGLE_DEVICE=1 run the N > 1 path with several ranks
    #  on ONE GPU — RCCL refuses two ranks on a device; the driver's runs use neither)
    backend = os.environ.get("ME_BENCH_BACKEND", "nccl")
    if os.environ.get("ME_BENCH_SINGLE_DEVICE", "0") == "1":
        local_rank = 0
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")

    from cloud_map_evaluation_amd.engine import Engine, Param

    evaluate_gt_mme = not args.no_gt_mme
    P = Param(icp_max_distance_=1.0, nn_radius_=args.nn_radius, vmd_voxel_size_=args.voxel,
              evaluate_gt_mme_=evaluate_gt_mme)
    # synthetic, seeded, generated directly in HBM (identical on every rank)
    est_d, gt_d = make_pair(args, dev)
    n_e, n_g = est_d.shape[0], gt_d.shape[0]
    if world > 1:
        # distributed input: rank r holds the r-th of `world` contiguous pieces of each (shuffled) cloud, as if it had read
        # its part of the files; the rest is dropped before the timed region
        from cloud_map_evaluation_amd.dist import shard_range

        b, e = shard_range(n_e, rank, world)
        est_d = est_d[b:e].clone()
        b, e = shard_range(n_g, rank, world)
        gt_d = gt_d[b:e].clone()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()

    # stream priorities: the library's default (main lane first) at every N — since the distributed step keeps its statistics and
    # voxel collectives under the other lane's MME kernels, "MME lane first" no longer helps there (10.8 vs 10.8-11.2 ms emulated
    # at 8 ranks, profiles/README.md)
    # (the clouds stay resident and unchanged for the whole run: the engine reads them where they lie — ME_FLAG_BORROW_DEVICE_INPUT)
    eng = Engine(local_rank, borrow_device_input=True)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(est, gt, steps, warmup, per_step=None):
        res = None
        for _ in range(warmup):
            res = suite_step(eng, dist, world, est, gt, P, evaluate_gt_mme, comm_dev)
        sync()
        t0 = time.perf_counter()
        tp = t0
        for _ in range(steps):
            res = suite_step(eng, dist, world, est, gt, P, evaluate_gt_mme, comm_dev)
            if per_step is not None:  # (a step returns with its scalars on the host: the call IS synchronous, no extra sync is added)
                tn = time.perf_counter()
                per_step.append((tn - tp) * 1e3)
                tp = tn
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / max(1, steps) * 1e3, res

    # Settling (N = 1; untimed, BEFORE the contract's warm-up + timed region): a fresh process on a fresh box
    # needs more than a handful of steps before the step time is flat — first-touch of ~10 GB of library buffers, code objects, and
    # the clocks coming up under load.  Steps are repeated until two consecutive ones agree within 1 % (at most `--settle-max`), so that
    # the timed block measures the steady state whatever W the caller passes; how many it took is on the line.
    settle_ms = []
    if world == 1 and args.settle_max > 0:
        timed(est_d, gt_d, 0, 1)
        while len(settle_ms) < args.settle_max:
            one = []
            timed(est_d, gt_d, 1, 0, one)
            settle_ms.append(one[0])
            if len(settle_ms) >= 3 and abs(settle_ms[-1] - settle_ms[-2]) <= 0.01 * settle_ms[-1] and abs(settle_ms[-2] - settle_ms[-3]) <= 0.01 * settle_ms[-2]:
                break

    step_ms = []
    ms_per_step, res = timed(est_d, gt_d, args.steps, args.warmup, step_ms)
    value = (n_e + n_g) / 1e6 / (ms_per_step / 1e3)

    # cross-checks of the headline (N = 1): the same schedule driven from Python, and the engine without ME_FLAG_BORROW_DEVICE_INPUT
    # (the default of me_create: the upload copies the resident cloud first).  INTERLEAVED (A / B / C / A / B / C ..., one step each,
    # round 6): three blocks one after the other compared three moments of the box (clocks, temperature) as much as three drivers.
    xcheck = {}
    if world == 1 and not args.no_roofline:
        k = max(3, min(8, args.steps))
        eng_copy = Engine(local_rank, borrow_device_input=False)
        eng_main = eng
        variants = [("headline_driver", eng_main, PY_DRIVER), ("other_driver", eng_main, not PY_DRIVER), ("copying_upload", eng_copy, PY_DRIVER)]
        per = {name: [] for name, _, _ in variants}
        last = {}
        py_saved = PY_DRIVER
        for rnd in range(k + 1):  # (round 0: untimed — the copying engine's first allocations, the other driver's first call)
            for name, e, py in variants:
                eng, PY_DRIVER = e, py
                one = []
                _, r = timed(est_d, gt_d, 1, 0, one)
                last[name] = r
                if rnd > 0:
                    per[name].append(one[0])
        eng, PY_DRIVER = eng_main, py_saved
        eng_copy.close()

        def same(a, b):
            return bool(a["cd"] == b["cd"] and a["mme_valid"] == b["mme_valid"] and a["awd"] == b["awd"] and a["mme_est"] == b["mme_est"])

        def stat(name):
            v = per[name]
            return {"ms_per_step": sum(v) / len(v), "median_ms": sorted(v)[len(v) // 2], "step_ms": [round(x, 3) for x in v], "steps": len(v),
                    "same_results": same(last[name], res)}
        xcheck["schedule"] = "interleaved: one step of each variant per round, %d rounds after one untimed round" % k
        xcheck["headline_driver"] = stat("headline_driver")
        xcheck["headline_driver"]["driver"] = "python (dist.suite_step)" if PY_DRIVER else "C ABI (me_run_suite_from)"
        xcheck["other_driver"] = stat("other_driver")
        xcheck["other_driver"]["driver"] = "python (dist.suite_step)" if not PY_DRIVER else "C ABI (me_run_suite_from)"
        xcheck["copying_upload"] = stat("copying_upload")
        xcheck["other_driver_vs_headline_driver"] = xcheck["other_driver"]["ms_per_step"] / xcheck["headline_driver"]["ms_per_step"]

    w = WORKLOADS[args.workload]
    line = {
        "metric": "Mpts/sec full metric suite (CD+MME+AWD) on 50M-pt pair",
        "value": value, "unit": "Mpts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "step_ms": [round(x, 3) for x in step_ms],
        "settling": {"untimed_steps_before_warmup": len(settle_ms), "ms": [round(x, 2) for x in settle_ms],
                     "rule": "N = 1: untimed steps before the W warm-up steps until two consecutive steps agree within 1 % (at most --settle-max)"},
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {w['what']}; GT={n_g} est={n_e} pts @ {args.density:g} pts/m^2 (seeded): AC/COM/CD + "
                               f"est-MME{'+GT-MME' if evaluate_gt_mme else ''} (r={args.nn_radius}) + voxel Gaussians/AWD/CDF/SCS "
                               f"(voxel={args.voxel}), clouds resident in HBM, index build included",
                   "n_est": n_e, "n_gt": n_g, "nn_radius": args.nn_radius, "vmd_voxel_size": args.voxel, "density": args.density,
                   "parallelism": ("single GPU" if world == 1 else
                                   f"distributed input (1/{world} of each cloud per rank), spatial slabs x{world} along the longest axis "
                                   "(+1 m halo) filled by one all-to-all halo exchange; cross-rank 1-NN resolve (all-gather + "
                                   "min-reduce), all-reduced partial sums, all-gathered voxel partials merged on the device (RCCL)")},
        "driver": ("python: dist.suite_step + a Python thread on me_twin" if (world == 1 and PY_DRIVER) else
                   "C ABI: one me_run_suite_from call per step (ME_SUITE_DEVICE_INPUT%s), the second lane is a thread inside the library"
                   % ("|ME_SUITE_OVERLAP" if OVERLAP else "")) if world == 1 else "python: dist.suite_step_dist over torch.distributed (RCCL)",
        "engine_flags": {"ME_FLAG_BORROW_DEVICE_INPUT": True,
                         "note": "the resident clouds are read where they lie; without the flag an upload copies them first (48 B/pt more "
                                 "traffic per cloud: see `copying_upload` for that step time)"},
        "results": {"AC": [float(x) for x in res["ac"]], "COM": [float(x) for x in res["com"]], "CD": float(res["cd"]),
                    "MME_est": float(res["mme_est"]), "MME_gt": float(res["mme_gt"]), "AWD": float(res["awd"]),
                    "SCS": float(res["scs"]), "W_voxels": int(res["n_w"]), "MME_valid": res["mme_valid"]},
    }

    if xcheck:
        line["cross_checks"] = xcheck

    # ---- roofline of the dominant kernel: HIP events on the library's own stream, one extra (untimed) step ----
    if not args.no_roofline:
        eng.timers_enable(True)
        eng.timers_reset()
        OVERLAP = False  # kernels timed one at a time
        suite_step(eng, dist, world, est_d, gt_d, P, evaluate_gt_mme, comm_dev)
        OVERLAP = not args.no_overlap
        fam = {}
        for name in ("nn_grid", "nn_grid2", "nn1", "nn_far", "mme", "sort", "morton", "gather", "cells", "octree", "nn_stats", "slab_filter", "voxel",
                     "w2", "scs", "halo_pack"):
            ms, cnt = eng.timer(name)
            if cnt:
                fam[name] = (ms, cnt)
        mme_pairs = eng.timer("mme_pairs")[1]
        nn_fallback = eng.timer("nn_fallback_queries")[1]
        nn_total = eng.timer("nn_queries")[1]
        eng.timers_enable(False)
        if rank == 0 and fam:
            dom = max(fam, key=lambda k: fam[k][0])
            ms, cnt = fam[dom]
            avg_ms = ms / cnt
            shard = 1.0 / world
            if dom == "mme":
                units = (n_e + (n_g if evaluate_gt_mme else 0)) * shard / cnt
                alg_bytes = BYTES_PER_MME_QUERY * units
            elif dom in ("nn1", "nn_grid"):
                units = (n_e + n_g) * shard / cnt
                alg_bytes = BYTES_PER_NN_QUERY * units
            else:
                units = (n_e + n_g) / cnt
                alg_bytes = 24.0 * units
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            # HBM bytes per launch from the rocprofv3 PMC passes (profiles/run_profile.sh, separate --pmc runs of this very
            # command): quoted only when they were collected at THIS kernel source and workload, otherwise null
            traffic, traffic_x2, traffic_src = None, None, None
            sha = kernel_source_sha()
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if tj.get("_kernel_source_sha") == sha and tj.get("_workload") == args.workload and tj.get("_points") == args.points:
                        traffic = tj.get(dom)
                        traffic_x2 = tj.get(dom + "_fetch_x2")
                        traffic_src = {"file": "profiles/traffic.json", "kernel_source_sha": sha, "collected": tj.get("_tag")}
                except Exception:
                    traffic = None
            line["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_corrected": traffic_x2,
                                "traffic_note": "bytes per launch: raw FETCH_SIZE + WRITE_SIZE, and with the guide's gfx950 FETCH x2 correction",
                                "traffic_source": traffic_src,
                                "avg_launch_ms": avg_ms, "units_per_launch": units, "algorithmic_bytes_per_launch": alg_bytes,
                                "kernel_ms_per_step": {k: v[0] for k, v in fam.items()},
                                # the second hot kernel, same definition (60 B per 1-NN query)
                                "nn_grid": ({"avg_launch_ms": fam["nn_grid"][0] / fam["nn_grid"][1],
                                             "achieved": BYTES_PER_NN_QUERY * (n_e + n_g) * shard / fam["nn_grid"][1] / (fam["nn_grid"][0] / fam["nn_grid"][1] * 1e-3) / 1e9,
                                             "frac": BYTES_PER_NN_QUERY * (n_e + n_g) * shard / fam["nn_grid"][1] / (fam["nn_grid"][0] / fam["nn_grid"][1] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "traffic": (json.load(open(tpath)).get("nn_grid") if (os.path.exists(tpath) and traffic is not None) else None)}
                                            if "nn_grid" in fam else None),
                                # The roofline that BINDS the dominant kernel: k_mme3 is bound by fp64 vector issue, not
                                # by HBM.  Useful work = accepted (query, neighbour) pairs x 9 fp64 lane-instructions (3 v_add_f64 + 6
                                # v_fma_f64: sum u, sum u u^T) = 15 flop per pair; peak = the fp64 vector rate, half of the guide's FP32
                                # vector figure (157.3 TFLOP/s / 2 = 78.6: 16 lanes x 2 flop per SIMD and clock, 1024 SIMDs, 2.4 GHz).
                                # The rest of the launch is the SIMT cost of 64 queries sharing one candidate stream: every accepted
                                # trip is executed by the whole wavefront (lane efficiency), plus the FP32 pre-test of every candidate.
                                "valu": ({"kernel": "mme", "accepted_pairs_per_step": int(mme_pairs), "flop_per_pair": 15,
                                          "achieved": 15.0 * mme_pairs / (fam["mme"][0] * 1e-3) / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS,
                                          "unit": "TFLOP/s", "frac": 15.0 * mme_pairs / (fam["mme"][0] * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                          "pairs_per_query": mme_pairs / max(1.0, (n_e + (n_g if evaluate_gt_mme else 0)) * shard),
                                          "ps_per_pair": fam["mme"][0] * 1e-3 / max(1, mme_pairs) * 1e12}
                                         if ("mme" in fam and mme_pairs) else None),
                                "nn_fallback_fraction": (nn_fallback / nn_total) if nn_total else None,
                                "queries_per_s": {"nn": (n_e + n_g) * shard / ((fam.get("nn1", (0, 0))[0] + fam.get("nn_far", (0, 0))[0] + fam.get("nn_grid", (0, 0))[0] + fam.get("nn_grid2", (0, 0))[0]) * 1e-3)
                                                  if ("nn1" in fam or "nn_grid" in fam) else None,
                                                  "mme": (n_e + (n_g if evaluate_gt_mme else 0)) * shard / (fam["mme"][0] * 1e-3)
                                                  if "mme" in fam else None}}

    # ---- the same step from pinned HOST buffers (SURVEY.md 8d span: H2D included) ----
    est_h = gt_h = None
    need_host = world == 1 and rank == 0 and (not args.no_h2d or args.cpu_baseline == "full")
    if need_host:
        est_h, gt_h = est_d.cpu(), gt_d.cpu()
    if world == 1 and not args.no_h2d:
        est_p, gt_p = est_h.pin_memory(), gt_h.pin_memory()
        h_ms, h_res = timed(est_p, gt_p, max(2, min(3, args.steps)), 1)
        # the two spans side by side (small items): SURVEY 8(d) words the span from HOST memory, the task contract
        # fixes `value` to the HBM-resident one.  first_cloud_idle_ms: nothing can be computed on a cloud before its last byte has
        # arrived, so the GPU idles for the map's copy (measured alone, pinned -> device)
        up = torch.empty_like(est_d) if est_d is not None else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            up.copy_(est_p, non_blocking=True)
        torch.cuda.synchronize()
        first_idle = (time.perf_counter() - t0) / 3 * 1e3
        del up
        line["hbm_resident"] = {"ms_per_step": ms_per_step, "value": value, "unit": "Mpts/s",
                                "note": "clouds resident in HBM when the timed region starts (= `value`, the contract's span)"}
        line["h2d_inclusive"] = {"ms_per_step": h_ms, "value": (n_e + n_g) / 1e6 / (h_ms / 1e3), "unit": "Mpts/s",
                                 "first_cloud_idle_ms": first_idle, "pcie_gb_s": 24 * n_e / (first_idle * 1e-3) / 1e9,
                                 "note": "same step with both clouds starting in pinned host memory (2 x 24 B/pt over PCIe Gen5; the "
                                         "ground truth crosses the link under the map's MME kernel) and every scalar back on the host",
                                 "bytes_h2d": 24 * (n_e + n_g),
                                 "same_results": bool(h_res["mme_valid"] == res["mme_valid"] and h_res["cd"] == res["cd"])}
        del est_p, gt_p

    if rank == 0 and world == 1 and args.cpu_baseline != "off":
        del est_d, gt_d
        torch.cuda.empty_cache()
        if args.cpu_baseline == "full":
            line["cpu_baseline"] = cpu_baseline_full(est_h.numpy(), gt_h.numpy(), P, evaluate_gt_mme)
        else:
            line["cpu_baseline"] = cpu_baseline_sample(args, P, evaluate_gt_mme)
        try:  # the reference's own code on a bounded sample, next to the port's full-size extrapolation
            line["cpu_baseline"]["reference_run"] = cpu_baseline_reference(args, P, evaluate_gt_mme)
        except Exception as e:  # (a checker, never the product: its absence or failure does not fail the bench)
            line["cpu_baseline"]["reference_run"] = {"error": str(e)[:200]}
        if args.cpu_baseline == "full":
            try:  # ... and on a 5 M + 5 M pair WITHOUT the GT-MME (its serial loop, map_eval.cpp:1451, is what makes 50 M impractical): the
                #     reference's scaling with tree depth
                r5 = cpu_baseline_reference(args, P, False, n=5_000_000)
                if r5 is not None:
                    r5["note"] = "evaluate_gt_mme: false (the serial GT-MME loop left out); same code, 5x the points of reference_run"
                line["cpu_baseline"]["reference_run_5m"] = r5
            except Exception as e:
                line["cpu_baseline"]["reference_run_5m"] = {"error": str(e)[:200]}

    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
