import sys, time, torch, numpy as np
sys.path.insert(0, 'omg-tools_amd')
from omgtools import workloads
from omgtools.batch import BatchP2P, StreamedP2P
dev = torch.device('cuda', 0)
problem, P = workloads.holonomic_p2p(1024)
opts = dict(tol=1e-3, max_iter=300)
def run(make, stepper, label):
    res = []
    for rep in range(3):
        m = make()
        m.solve_cold(bends=())
        for _ in range(5): stepper(m)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): stepper(m)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res.append(1024 * 20 / dt)
        x = (m.x if not isinstance(m, StreamedP2P) else m.x).cpu().numpy().copy()
        (m.close() if isinstance(m, StreamedP2P) else m.solver.close())
    print(label, ' '.join('%.0f' % v for v in res), flush=True)
    return x
def roll1_streamed(m):
    for part, st in zip(m.parts, m.streams):
        with torch.cuda.stream(st):
            part.rollout(1)
xa = run(lambda: BatchP2P(problem, P, ops='hip', device=dev, options=opts), lambda m: m.step(), 'one handle, step()        ')
xb = run(lambda: BatchP2P(problem, P, ops='hip', device=dev, options=opts), lambda m: m.rollout(1), 'one handle, rollout(1)    ')
print('same bits', np.array_equal(xa, xb))
xc = run(lambda: StreamedP2P(problem, P, n_streams=3, device=dev, slots=512, options=opts), lambda m: m.step(), 'three streams, step()     ')
xd = run(lambda: StreamedP2P(problem, P, n_streams=3, device=dev, slots=512, options=opts), roll1_streamed, 'three streams, rollout(1) ')
print('same bits', np.array_equal(xc, xd), np.array_equal(xa, xc))
