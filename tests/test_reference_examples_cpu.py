"""Drop-in check against the reference's OWN example scripts: each file under
`/root/reference/examples/` listed below is executed VERBATIM (`from omgtools import *` resolves
to this package; plotting is a no-op layer) with the oracle host port standing in for the HIP
solver (CPU tier; tests/port_solver.py).  This is the reference's `tests/test_examples.py`
(run every example, expect it to complete) plus an assertion on the outcome: every vehicle ends
at its target.  Skipped where `/root/reference` does not exist (the GPU box); the GPU tier runs
the same scenarios from inlined copies of the scripts (tests/test_gpu_simulator.py)."""
import os

import numpy as np
import pytest

EXAMPLES = '/root/reference/examples'

# (script, goal tolerance)
SCRIPTS = [('p2p_holonomic', 1e-2), ('tutorial_example', 1e-2), ('p2p_holonomic_octroom', 1e-2),
           ('p2p_holonomic_disturbances', 5e-2), ('p2p_holonomic_interveh_avoidance', 1e-2),
           ('annoying_obstacle', 1e-2), ('p2p_quadrotor', 2e-2)]


@pytest.fixture(autouse=True)
def port_backend(monkeypatch):
    import omgtools.backend as be
    import port_solver
    monkeypatch.setattr(be, 'create_nlp', port_solver.create_nlp)


@pytest.mark.parametrize('name,tol', SCRIPTS)
def test_reference_example_runs_unchanged(name, tol):
    path = os.path.join(EXAMPLES, name + '.py')
    if not os.path.exists(path):
        pytest.skip('reference checkout not present')
    g = {'__name__': '__main__'}
    exec(compile(open(path).read(), path, 'exec'), g)
    vehicles = g.get('vehicles') or [g['vehicle']]
    if not isinstance(vehicles, (list, tuple)):
        vehicles = [vehicles]
    for veh in vehicles:
        goal = np.asarray(veh.poseT, dtype=float)[:veh.n_dim]
        end = veh.signals['state'][:veh.n_dim, -1]
        assert np.linalg.norm(end - goal) < tol, (name, end, goal)
