#!/bin/bash
timeout 1500 python -m pytest tests/test_multirank_gpu.py tests/test_hip_parity.py -x -q -k "multirank or two_process or four_process or pipelined or non_default_feature or pipeline" 2>&1 | tail -6
