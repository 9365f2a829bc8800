#!/bin/bash
# round-4 baseline: GPU tests, headline bench, raw-frame benches + their kernel-trace stats
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 200 python3 $B > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc=$?"
timeout -k 10 200 python3 $B --envs 256 --raw-frames --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_256envs_raw.json 2> $OUT/b256raw.err; echo "256raw rc=$?"
timeout -k 10 200 python3 $B --envs 32 --raw-frames --no-cpu-baseline > $OUT/bench_32envs_raw.json 2> $OUT/b32raw.err; echo "32raw rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_raw256 -- python3 $B --envs 256 --raw-frames --steps 40 --warmup 8 --windows 1 --no-cpu-baseline > $OUT/stats_raw256.json 2> $OUT/stats_raw256.err; echo "stats rc=$?"
ls $OUT
