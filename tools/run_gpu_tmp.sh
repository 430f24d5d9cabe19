#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 200 python -m pytest tests/test_gpu_model.py -q -x -k "one_launch" 2>&1 | tail -12
