"""Developer tool (GPU): where the pipelined host-boundary loop spends its time -- per half-batch and step, events after the
glue (prediction), after the solve and after the download kernel; plain two-stream loop next to it."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch, numpy as np
from omgtools import workloads
from omgtools.batch import StreamedP2P
dev = torch.device('cuda', 0)
problem, P = workloads.holonomic_p2p(1024)
opts = dict(tol=1e-3, max_iter=300)
for mode in ('plain', 'plain', 'plain_noev', 'download', 'mapped_p+download', 'plain', 'download', 'plain_noev'):
    rh = StreamedP2P(problem, P, n_streams=2, device=dev, options=opts)
    n = 512
    tpl = rh.tpl
    xh = [torch.empty((n, tpl.n_var), dtype=torch.float64).pin_memory() for _ in rh.parts]
    sh = [torch.empty(n, dtype=torch.int32).pin_memory() for _ in rh.parts]
    ih = [torch.empty(n, dtype=torch.int32).pin_memory() for _ in rh.parts]
    rh.solve_cold(bends=())
    for _ in range(3):
        rh.step()
    rh.synchronize()
    if mode.startswith('mapped'):
        for m in rh.parts:
            ph = torch.empty(m.p.shape, dtype=torch.float64).pin_memory(); ph.copy_(m.p); m.p = ph
        torch.cuda.synchronize()
    K = 40
    ev = [[[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in rh.parts] for _ in range(K)]
    base = torch.cuda.Event(enable_timing=True); base.record()
    bench.quiet_host()          # (a generation-2 collection inside the loop costs a millisecond per step on average)
    t0 = time.perf_counter()
    for k in range(K):
        for kp, (m, st) in enumerate(zip(rh.parts, rh.streams)):
            with torch.cuda.stream(st):
                if mode == 'plain_noev':
                    m.step()
                    continue
                ev[k][kp][0].record()
                m.step(before_solve=lambda mm, e=ev[k][kp][1]: e.record())
                ev[k][kp][2].record()
                if mode == 'download_memcpy':
                    xh[kp].copy_(m.x, non_blocking=True); sh[kp].copy_(m.status, non_blocking=True); ih[kp].copy_(m.iters, non_blocking=True)
                elif not mode.startswith('plain'):
                    m.solver.transfer([(xh[kp], m.x), (sh[kp], m.status), (ih[kp], m.iters)])
                ev[k][kp][3].record()
    rh.synchronize()
    wall = (time.perf_counter() - t0) / K * 1e3
    import gc; gc.enable()
    if mode == 'plain_noev':
        print('%-20s %.3f ms/step wall' % (mode, wall), flush=True); rh.close(); continue
    seg = np.array([[[ev[k][kp][i].elapsed_time(ev[k][kp][i + 1]) for i in range(3)] for kp in range(2)] for k in range(K)])
    print('%-20s %.3f ms/step wall;  per half-step (median ms): glue %.3f  solve %.3f  download %.3f' % (mode, wall, np.median(seg[:, :, 0]), np.median(seg[:, :, 1]), np.median(seg[:, :, 2])), flush=True)
    rh.close()
