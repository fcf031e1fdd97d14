"""Writes the problem bundles of bench.py's workloads (`omg-tools_amd/omgtools/data/*.npz`, read by `omgtools.workloads`):
every configuration is built ONCE here through the front end (`omgtools.scenarios`), its NLP template and the few facts
the receding-horizon / consensus loops read off the problem objects are stored, and the benchmark path never imports
the front-end modules again.  `tests/test_workload_bundles.py` rebuilds them and compares array for array; the config-2
template is also compared with the one the reference's own modules produce on `omgx_shim`
(tests/test_reference_shim_cpu.py).  Run from the repository root:  python tools/generate_workload_bundles.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
import numpy as np
import omgtools.backend as be
from omgtools import scenarios, workloads as wl

FLEET_SIZES = (512, 64)      # bench.py --workload formation / rendezvous: BASELINE configs[3] (512 agents) and its 64-agent shard size


def main():
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)       # templates only: no solver object, no device
    os.makedirs(wl.DATA_DIR, exist_ok=True)
    for name, fn in (('holonomic_p2p_k11_o3', scenarios.holonomic_p2p), ('quadrotor_p2p_k13_o5', scenarios.quadrotor_p2p),
                     ('holonomic3d_p2p_k15_o10', scenarios.holonomic3d_p2p)):
        problem, P = fn(1)
        path = wl.save_bundle(os.path.join(wl.DATA_DIR, name + '.npz'), problem.father.template, wl.describe(problem))
        print(path, os.path.getsize(path), 'bytes')
    for n in FLEET_SIZES:
        for name, fn, kind in (('formation_holonomic_k10_%d' % n, scenarios.formation_holonomic, 'formation'),
                               ('rendezvous_holonomic_k10_%d' % n, scenarios.rendezvous_holonomic, 'rendezvous')):
            problem, updater, father, lay, P = fn(n)
            meta = wl.describe(problem, father, extra=dict(updater_label=updater.label, layout_class=kind,
                                                           obstacles=scenarios._obstacle_facts(problem.environment)))
            path = wl.save_bundle(os.path.join(wl.DATA_DIR, name + '.npz'), father.template, meta)
            print(path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
