#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_lm_gpu.py -q -m gpu 2>&1 | tail -40
