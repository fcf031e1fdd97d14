import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def cfg2_small():
    """8 agents of the config-2 workload (Holonomic, K=11, 3 circular obstacles)."""
    from omgtools.scenarios import holonomic_p2p
    import omgtools.backend as be
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)     # template only, no device
    try:
        problem, P = holonomic_p2p(8)
    finally:
        be.create_nlp = saved
    return problem, P
