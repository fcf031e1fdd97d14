"""Test helper: the independent solver of the oracle (scipy SLSQP on the restated NLP), see oracle/slsqp_numpy.py."""
from oracle.slsqp_numpy import solve_slsqp  # noqa: F401
