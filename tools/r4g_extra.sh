OUT=$1
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/paac_amd/libpaac_hip_stamps.so
PAAC_HIP_LIB=$L PROBE_N=128 PROBE_A=18 PROBE_MULTI=1 PROBE_ACT=1 timeout -k 10 120 python3 tools/probe_sampler.py > $OUT/probe_act_128x18.txt 2>&1; echo "probe rc=$?"
PAAC_HIP_LIB=$L PROBE_N=256 PROBE_A=4 PROBE_MULTI=1 PROBE_ACT=1 timeout -k 10 120 python3 tools/probe_sampler.py > $OUT/probe_act_256x4.txt 2>&1; echo "probe rc=$?"
PAAC_HIP_LIB=$L PROBE_N=32 PROBE_A=4 PROBE_ACT=1 timeout -k 10 120 python3 tools/probe_sampler.py > $OUT/probe_act_32x4.txt 2>&1; echo "probe rc=$?"
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 200 python3 $B --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bdef.err; echo "default rc=$?"
PAAC_MT_AHEAD=0 timeout -k 10 200 python3 $B --no-cpu-baseline > $OUT/bench_default_noahead.json 2> $OUT/bdefn.err; echo "default noahead rc=$?"
timeout -k 10 200 python3 $B --no-cpu-baseline > $OUT/bench_default2.json 2> $OUT/bdef.err; echo "default rc=$?"
timeout -k 10 200 python3 $B --envs 128 --tmax 20 --game seaquest --no-cpu-baseline --steps 40 --warmup 8 > $OUT/bench_sq.json 2> $OUT/bsq.err; echo "sq rc=$?"
timeout -k 10 200 python3 $B --envs 256 --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_256.json 2> $OUT/b256.err; echo "256 rc=$?"
timeout -k 10 200 python3 $B --arch NIPS --game pong --envs 32 --no-cpu-baseline > $OUT/bench_nips_pong32.json 2> $OUT/bn32.err; echo "nips32 rc=$?"
