OUT=$1
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/paac_amd/libpaac_hip_stamps.so
PAAC_HIP_LIB=$L PROBE_N=128 PROBE_A=18 PROBE_MULTI=1 PROBE_ACT=1 timeout -k 10 120 python3 tools/probe_sampler.py > $OUT/probe_act_128x18.txt 2>&1; echo "probe rc=$?"
PAAC_HIP_LIB=$L PROBE_N=128 PROBE_A=18 PROBE_MULTI=1 PROBE_ACT=1 PROBE_MANAGED=0 timeout -k 10 120 python3 tools/probe_sampler.py > $OUT/probe_act_128x18_unmanaged.txt 2>&1; echo "probe rc=$?"
PAAC_HIP_LIB=$L PROBE_N=256 PROBE_A=4 PROBE_MULTI=1 PROBE_ACT=1 timeout -k 10 120 python3 tools/probe_sampler.py > $OUT/probe_act_256x4.txt 2>&1; echo "probe rc=$?"
PAAC_HIP_LIB=$L PROBE_N=256 PROBE_A=4 PROBE_MULTI=1 PROBE_ACT=1 PROBE_MANAGED=0 timeout -k 10 120 python3 tools/probe_sampler.py > $OUT/probe_act_256x4_unmanaged.txt 2>&1; echo "probe rc=$?"
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
PAAC_TOWER=0 timeout -k 10 200 python3 $B --arch NIPS --game pong --envs 32 --no-cpu-baseline > $OUT/bench_nips_pong32_notower.json 2> $OUT/bn32.err; echo "nips32 notower rc=$?"
PAAC_TOWER=0 timeout -k 10 200 python3 $B --arch NIPS --game pong --envs 8 --no-cpu-baseline > $OUT/bench_nips_pong8_notower.json 2> $OUT/bn8.err; echo "nips8 notower rc=$?"
