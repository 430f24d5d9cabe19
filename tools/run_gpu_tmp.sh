#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_elementwise.py -q -x -k "fusion" 2>&1 | tail -25
