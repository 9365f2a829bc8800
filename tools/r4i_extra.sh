OUT=$1
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 200 python3 $B --envs 128 --tmax 20 --game seaquest --no-cpu-baseline --steps 40 --warmup 8 > $OUT/bench_sq_fold.json 2> $OUT/bsq.err; echo "sq rc=$?"
PAAC_HEADS_IN_SAMPLER=0 timeout -k 10 200 python3 $B --envs 128 --tmax 20 --game seaquest --no-cpu-baseline --steps 40 --warmup 8 > $OUT/bench_sq_nofold.json 2> $OUT/bsqn.err; echo "sqn rc=$?"
timeout -k 10 200 python3 $B --envs 256 --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_256_fold.json 2> $OUT/b256.err; echo "256 rc=$?"
PAAC_HEADS_IN_SAMPLER=0 timeout -k 10 200 python3 $B --envs 256 --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_256_nofold.json 2> $OUT/b256n.err; echo "256n rc=$?"
timeout -k 10 200 python3 $B --envs 256 --raw-frames --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_256_raw.json 2> $OUT/b256r.err; echo "256raw rc=$?"
timeout -k 10 200 python3 $B --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bdef.err; echo "default rc=$?"
