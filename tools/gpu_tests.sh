#!/bin/bash
# full GPU suite into gpurun_out/<dir>/pytest.log  (usage: tools/gpu_tests.sh <dir> [pytest args])
d=gpurun_out/$1; shift
mkdir -p $d
timeout -k 10 1000 python -m pytest tests -m gpu -q -x "$@" > $d/pytest.log 2>&1
rc=$?
tail -15 $d/pytest.log
exit $rc
