#!/bin/bash
# the GPU tier and the smoke test, as the driver runs them at round end
python -m pytest tests -x -q -m gpu 2>&1 | tail -n 4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
