"""`examples/formation_holonomic.py:22-57` through `Simulator` with `FormationPoint2point` on the HIP
path: x-updates by `omgx_batch_solve`, z / lambda / residuals / exchange by the `omgx_admm_*` kernels,
knot-crossing shifts by `omgx_shift_rows`."""
import pytest

pytestmark = pytest.mark.gpu


def test_formation_holonomic_example_hip():
    from test_formation_cpu import formation_example, check_formation_run
    out = formation_example('hip')
    check_formation_run(*out)
    assert out[0].ops.solver.workspace()['mode'] in (0, 1, 4)       # (4: two agents per CU, the Jacobian values in a slab)
