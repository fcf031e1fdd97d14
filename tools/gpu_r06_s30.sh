#!/bin/bash
python -m pytest tests/test_gpu_perf_guard.py -q -m gpu 2>&1 | tail -n 15
