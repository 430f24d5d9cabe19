cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python tools/gpu_probe.py nobar 2>&1 | grep -v amdgpu
