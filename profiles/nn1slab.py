import sys, torch
sys.path.insert(0,'.')
from cloud_map_evaluation_amd import synth, dist as medist
from cloud_map_evaluation_amd.engine import Engine
dev=torch.device('cuda',0)
n=int(sys.argv[1]); world=int(sys.argv[2])
est,gt=synth.scan_pair(n, density=2500.0, seed=100, device=dev)
K=('nn_fallback_queries','nn_queries','nn1_opened','nn1_scans','nn1_points','nn1_max_opened','nn1_far')
with Engine(0) as eng:
    for rank in ([None] + list(range(world))):
        if rank is None:
            eng.set_slab(-1)
        else:
            axis, lo, hi = medist.slab_bounds(gt.cpu(), rank, world)
            eng.set_slab(axis, lo, hi, 1.0)
        eng.upload(0, est, cell_size=0.1); eng.upload(1, gt, cell_size=0.1)
        for q,r in ((0,1),(1,0)):
            eng.timers_enable(True); eng.timers_reset()
            eng.nn1(q,r,fetch=False)
            print('rank',rank,'dir',q,r,'held',eng.size(q),eng.size(r), {k: eng.timer(k)[1] for k in K}, 'nn1 ms', round(eng.timer('nn1')[0],3), 'grid ms', round(eng.timer('nn_grid')[0],3), 'unresolved', eng.nn_unresolved_count(q) if rank is not None else '-')
    # the cross-rank pass as the emulation runs it: the rank's own open queries, bounded by their own result, against its own slab
    import time
    eng.set_slab(*medist.slab_bounds(gt.cpu(), 0, world)[:1], *medist.slab_bounds(gt.cpu(), 0, world)[1:], 1.0)
    eng.upload(0, est, cell_size=0.1); eng.upload(1, gt, cell_size=0.1)
    eng.nn1(0,1,fetch=False)
    q = eng.nn_unresolved(0, with_d2=True)
    print('open queries', q.shape, 'bounds: median %.3f max %.3f' % (float(q[:,3].sqrt().median()), float(q[:,3].sqrt().max())))
    for bounded in (True, False):
        eng.timers_enable(True); eng.timers_reset()
        torch.cuda.synchronize(); t0=time.perf_counter()
        ans = eng.nn_points(1, q[:, :3].contiguous(), bound=q[:, 3].contiguous() if bounded else None)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)*1e3
        print('nn_points bounded=%s' % bounded, 'wall %.2f ms' % dt, {k: eng.timer(k)[1] for k in K}, 'nn1 ms', round(eng.timer('nn1')[0],3))
