#!/bin/bash
# usage: tools/r4_run.sh <out> <pytest -k expr or ""> [extra commands file]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ -n "${2:-}" ]; then
  timeout -k 10 700 python3 -m pytest tests -m gpu -x -q -k "$2" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/pytest.log
  [ $rc -ne 0 ] && exit $rc
fi
if [ -n "${3:-}" ]; then
  bash $3 $OUT
fi
