import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from cloud_map_evaluation_amd import synth
from cloud_map_evaluation_amd.engine import Engine, Param
dev = torch.device('cuda', 0)
est, gt = synth.multisession_pair(50_000_000, 3, density=2500.0, seed=100, device=dev)
P = Param(icp_max_distance_=1.0, nn_radius_=0.1, vmd_voxel_size_=3.0)
est_p, gt_p = est.cpu().pin_memory(), gt.cpu().pin_memory()
names = ["upload+index est", "nn est->gt", "wait lane (nn gt->est)", "sigma", "mme est", "mme gt", "voxel/awd", "total"]
with Engine(0, borrow_device_input=True) as e:
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = e.run_suite_from(est_p, gt_p, P, overlap=True)
        dt = (time.perf_counter() - t0) * 1e3
        print(f"host input: {dt:.2f} ms", {n: round(o.stage_ms[i], 2) for i, n in enumerate(names)})
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = e.run_suite_from(est, gt, P, overlap=True)
        dt = (time.perf_counter() - t0) * 1e3
        print(f"device input: {dt:.2f} ms", {n: round(o.stage_ms[i], 2) for i, n in enumerate(names)})
