# per-phase wall times of the single-GPU two-lane step (main thread's view), 50 M + 50 M pair
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from cloud_map_evaluation_amd import synth, dist as medist
from cloud_map_evaluation_amd.engine import Engine, Param
dev = torch.device('cuda', 0)
est, gt = synth.multisession_pair(50_000_000, 3, density=2500.0, seed=100, device=dev)
P = Param(icp_max_distance_=1.0, nn_radius_=0.1, vmd_voxel_size_=3.0)
marks = []
orig = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); marks.append((name + str(a[:2] if name in ('mme','nn1','upload','nn_partial_sums','nn_sigma_sums') else ''), (time.perf_counter() - t0) * 1e3, t0)); return r
    setattr(obj, name, g)
with Engine(0) as eng:
    for _ in range(2):
        medist.suite_step(eng, None, dev, est, gt, P, True, overlap=True)
    for n in ('upload', 'mme', 'nn1', 'nn_partial_sums', 'nn_sigma_sums', 'calculateVMD', 'size'):
        wrap(eng, n)
    for rep in range(3):
        marks.clear()
        torch.cuda.synchronize(); T0 = time.perf_counter()
        # lane waits are inside suite_step; measure them by wrapping the lane class methods
        import cloud_map_evaluation_amd.dist as D
        for nm in ('wait_gt', 'join'):
            if not hasattr(D._Lane, '_orig_' + nm):
                setattr(D._Lane, '_orig_' + nm, getattr(D._Lane, nm))
                def mk(nm):
                    def h(self, *a, **k):
                        t0 = time.perf_counter(); r = getattr(D._Lane, '_orig_' + nm)(self, *a, **k); marks.append(('lane.' + nm, (time.perf_counter() - t0) * 1e3, t0)); return r
                    return h
                setattr(D._Lane, nm, mk(nm))
        medist.suite_step(eng, None, dev, est, gt, P, True, overlap=True)
        torch.cuda.synchronize(); total = (time.perf_counter() - T0) * 1e3
        print('step %.2f ms: ' % total + ' | '.join('%s@%.1f=%.2f' % (n, (t0 - T0) * 1e3, d) for n, d, t0 in marks))
