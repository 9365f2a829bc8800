OUT=$1
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 200 python3 $B --envs 128 --tmax 20 --game seaquest --no-cpu-baseline --steps 40 --warmup 8 > $OUT/bench_sq_ahead.json 2> $OUT/bsq.err; echo "sq rc=$?"
PAAC_MT_AHEAD=0 timeout -k 10 200 python3 $B --envs 128 --tmax 20 --game seaquest --no-cpu-baseline --steps 40 --warmup 8 > $OUT/bench_sq_noahead.json 2> $OUT/bsqn.err; echo "sqn rc=$?"
timeout -k 10 200 python3 $B --envs 256 --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_256_ahead.json 2> $OUT/b256.err; echo "256 rc=$?"
PAAC_MT_AHEAD=0 timeout -k 10 200 python3 $B --envs 256 --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_256_noahead.json 2> $OUT/b256n.err; echo "256n rc=$?"
timeout -k 10 200 python3 $B --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bdef.err; echo "default rc=$?"
timeout -k 10 200 python3 $B --arch NIPS --game pong --envs 32 --no-cpu-baseline > $OUT/bench_nips_pong32.json 2> $OUT/bn32.err; echo "nips32 rc=$?"
timeout -k 10 200 python3 $B --arch NIPS --game pong --envs 8 --no-cpu-baseline > $OUT/bench_nips_pong8.json 2> $OUT/bn8.err; echo "nips8 rc=$?"
