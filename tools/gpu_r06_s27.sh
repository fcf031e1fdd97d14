#!/bin/bash
python -m pytest tests/test_gpu_solver.py -q -m gpu 2>&1 | tail -n 15
