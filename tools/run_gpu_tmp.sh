cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/ab.sh "A=1" "DYK_SCHED_PRIO=1" "DYK_STREAMS=5" "DYK_STREAMS=5 DYK_SCHED_PRIO=1"
