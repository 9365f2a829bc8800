OUT=$1
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 tools/probe_magnitudes.py > $OUT/magnitudes.txt 2>&1; echo "probe rc=$?"
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 200 python3 $B --envs 256 --raw-frames --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_256envs_raw.json 2> $OUT/b256raw.err; echo "256raw rc=$?"
timeout -k 10 200 python3 $B --envs 32 --raw-frames --no-cpu-baseline > $OUT/bench_32envs_raw.json 2> $OUT/b32raw.err; echo "32raw rc=$?"
timeout -k 10 200 python3 $B --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bdef.err; echo "default rc=$?"
