OUT=$1
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
for i in 1 2; do
timeout -k 10 200 python3 $B --no-cpu-baseline > $OUT/bench_reuse_$i.json 2> $OUT/bre.err; echo "reuse rc=$?"
PAAC_REUSE_ACTING=0 timeout -k 10 200 python3 $B --no-cpu-baseline > $OUT/bench_noreuse_$i.json 2> $OUT/bnre.err; echo "noreuse rc=$?"
done
timeout -k 10 200 python3 $B --envs 256 --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_256_reuse.json 2> $OUT/b256.err; echo "256 rc=$?"
PAAC_REUSE_ACTING=0 timeout -k 10 200 python3 $B --envs 256 --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_256_noreuse.json 2> $OUT/b256n.err; echo "256n rc=$?"
timeout -k 10 200 python3 $B --envs 128 --tmax 20 --game seaquest --no-cpu-baseline --steps 40 --warmup 8 > $OUT/bench_sq_reuse.json 2> $OUT/bsq.err; echo "sq rc=$?"
PAAC_REUSE_ACTING=0 timeout -k 10 200 python3 $B --envs 128 --tmax 20 --game seaquest --no-cpu-baseline --steps 40 --warmup 8 > $OUT/bench_sq_noreuse.json 2> $OUT/bsqn.err; echo "sqn rc=$?"
cd $GRAFT_REPO_ROOT
PAAC_HIP_LIB=$GRAFT_REPO_ROOT/paac_amd/libpaac_hip_stamps.so PROBE_N=128 PROBE_A=18 PROBE_MULTI=1 timeout -k 10 120 python3 tools/probe_sampler.py > $OUT/probe_sampler_128x18.txt 2>&1; echo "probe rc=$?"
PAAC_HIP_LIB=$GRAFT_REPO_ROOT/paac_amd/libpaac_hip_stamps.so PROBE_N=32 PROBE_A=4 timeout -k 10 120 python3 tools/probe_sampler.py > $OUT/probe_sampler_32x4.txt 2>&1; echo "probe rc=$?"
