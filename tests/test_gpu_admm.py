"""HIP ADMM kernels (centre, z/lambda/residual update, neighbour exchange) and the
device-resident BatchADMM iteration against the numpy ops backend (which is itself
pinned to the reference's formulas in tests/test_admm_cpu.py)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_batch_admm_matches_numpy_backend():
    import torch
    from test_admm_cpu import _scenario
    from admm_numpy_ops import NumpyAdmmOps
    from omgtools.admm import BatchADMM, HipAdmmOps
    from omgtools.backend import BatchSolver
    tpl, lay, P = _scenario(8)
    dev = torch.device('cuda', 0)
    solver = BatchSolver(tpl, 8, options=dict(tol=1e-6, max_iter=200, warm_z_cap=0.0, max_soc=0))
    gpu = BatchADMM(lay, P['nbr'], HipAdmmOps(solver, tpl, lay, P['p'], P['x0'], dev), rho=1.0)
    cpu_ops = NumpyAdmmOps(tpl, lay, P['p'], P['x0'])
    cpu = BatchADMM(lay, P['nbr'], cpu_ops, rho=1.0)
    gpu.initialize()
    cpu.initialize()
    for it in range(4):
        t_rel = 0.1 * it                      # exercises the forward/backward first-knot transforms
        st_g, res_g = gpu.iterate(t_rel)
        st_c, res_c = cpu.iterate(t_rel)
        assert np.all(st_g.cpu().numpy() == 0) and np.all(st_c == 0)
        assert np.allclose(res_g, res_c, rtol=1e-6, atol=1e-9)
        ops = gpu.ops
        lo = lay.x_spl
        assert np.abs(ops.x.cpu().numpy()[:, lo:lo + lay.ns] - cpu_ops.x[:, lo:lo + lay.ns]).max() < 1e-6
        assert np.abs(ops.z_ij.cpu().numpy() - cpu_ops.z_ij).max() < 1e-6
        assert np.abs(ops.p.cpu().numpy() - cpu_ops.p).max() < 1e-6
    assert gpu.residuals[-1][0] < gpu.residuals[0][0]
    # the centre step rode on the x-updates (omgx_batch_set_center: no launch of its own): what the solve kernel's epilogue
    # left in x_i is, bit for bit, what the stand-alone launch writes for the same x and p
    ops = gpu.ops
    assert ops.center_fused
    l0 = ops.launches
    ops.solve()
    fused = ops.center(lay).clone()
    assert ops.launches == l0 + 1
    ops._xi_fresh = False
    alone = ops.center(lay)
    assert ops.launches == l0 + 2
    assert torch.equal(fused, alone)
    solver.close()


def test_nesterov_acceleration_matches_numpy_backend():
    """`problems/admm.py:510-554` on the device (branch-free, alpha and the reset decision never visit the host)
    against the reference's statements in the numpy backend; no host copy inside the loop."""
    import torch
    from test_admm_cpu import _scenario
    from admm_numpy_ops import NumpyAdmmOps
    from omgtools.admm import BatchADMM, HipAdmmOps
    from omgtools.backend import BatchSolver
    tpl, lay, P = _scenario(8)
    dev = torch.device('cuda', 0)
    for kw in (dict(nesterov_acceleration=True), dict(nesterov_acceleration=True, nesterov_reset=True, eta=0.9),
               dict(nesterov_acceleration=True, AMA=True)):
        solver = BatchSolver(tpl, 8, options=dict(tol=1e-6, max_iter=200, warm_z_cap=0.0, max_soc=0))
        gpu = BatchADMM(lay, P['nbr'], HipAdmmOps(solver, tpl, lay, P['p'], P['x0'], dev), rho=1.0, **kw)
        cpu_ops = NumpyAdmmOps(tpl, lay, P['p'], P['x0'])
        cpu = BatchADMM(lay, P['nbr'], cpu_ops, rho=1.0, **kw)
        gpu.initialize()
        cpu.initialize()
        for it in range(5):
            gpu.iterate(0.0, sync=False)
            cpu.iterate(0.0)
        assert np.allclose(np.array(gpu.residuals), np.array(cpu.residuals), rtol=1e-5, atol=1e-8), kw
        assert abs(float(gpu.ops.alpha) - cpu_ops.alpha) < 1e-12, kw
        assert np.abs(gpu.ops.z_ij.cpu().numpy() - cpu_ops.z_ij).max() < 1e-6, kw
        assert np.abs(gpu.ops.l_ij.cpu().numpy() - cpu_ops.l_ij).max() < 1e-6, kw
        assert np.abs(gpu.ops.p.cpu().numpy() - cpu_ops.p).max() < 1e-6, kw
        solver.close()


def test_shift_rows_on_consensus_state():
    import ctypes as C
    import torch
    from test_admm_cpu import _scenario
    from omgtools.backend import BatchSolver
    from omgtools.splines import shiftoverknot_T
    tpl, lay, P = _scenario(4, perturb=False)
    solver = BatchSolver(tpl, 4)
    rng = np.random.default_rng(1)
    z = rng.normal(size=(4, 2, lay.ns))
    zd = torch.as_tensor(z, device='cuda:0')
    T = shiftoverknot_T(lay.basis)
    entries = np.array([[k * lay.ns, lay.L, lay.n_dim, 0] for k in range(2)], dtype=np.int32)
    lib = solver.lib
    lib.omgx_shift_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.c_void_p, C.c_int32]
    Tm = np.ascontiguousarray(T.reshape(-1))
    rc = lib.omgx_shift_rows(solver._h, zd.data_ptr(), 2 * lay.ns, 4, None, entries.ctypes.data, 2,
                             Tm.ctypes.data, Tm.size)
    assert rc == 0
    want = np.einsum('ij,bkdj->bkdi', T, z.reshape(4, 2, lay.n_dim, lay.L)).reshape(4, 2, lay.ns)
    assert np.abs(zd.cpu().numpy() - want).max() < 1e-13
    solver.close()


def test_full_interconnection_fused_form_on_the_device():
    """`FullConsensusADMM` (interconnection 'full': one fleet sum per iteration) on the HIP ops -- x-update and centre by the
    library, the consensus arithmetic on device tensors -- against the numpy backend."""
    import torch
    from test_admm_cpu import _full_scenario
    from admm_numpy_ops import NumpyAdmmOps
    from omgtools.admm import FullConsensusADMM, HipAdmmOps
    from omgtools.backend import BatchSolver
    n = 8
    tpl, lay, P = _full_scenario(n, 'circular')
    dev = torch.device('cuda', 0)
    solver = BatchSolver(tpl, n, options=dict(tol=1e-6, max_iter=200, warm_z_cap=0.0, max_soc=0))
    gpu = FullConsensusADMM(lay, HipAdmmOps(solver, tpl, lay, P['p'], P['x0'], dev), n, rho=1.0)
    cpu_ops = NumpyAdmmOps(tpl, lay, P['p'], P['x0'])
    cpu = FullConsensusADMM(lay, cpu_ops, n, rho=1.0)
    gpu.initialize()
    cpu.initialize()
    for it in range(4):
        st_g, st_c = gpu.iterate(0.0), cpu.iterate(0.0)
        assert np.all(st_g.cpu().numpy() == 0) and np.all(st_c == 0)
        lo = lay.x_spl
        assert np.abs(gpu.ops.x.cpu().numpy()[:, lo:lo + lay.ns] - cpu_ops.x[:, lo:lo + lay.ns]).max() < 1e-6
        assert np.abs(gpu.c.cpu().numpy() - cpu.c).max() < 1e-6
    assert np.allclose(np.array(gpu.residuals), np.array(cpu.residuals), rtol=1e-5, atol=1e-8)
    solver.close()
