OUT=$1
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/paac_amd/libpaac_hip_stamps.so
PAAC_HEADS_IN_SAMPLER=0 PAAC_HIP_LIB=$L PROBE_N=128 PROBE_A=18 PROBE_MULTI=1 PROBE_ACT=1 timeout -k 10 120 python3 tools/probe_sampler.py > $OUT/probe_act_128x18.txt 2>&1; echo "probe rc=$?"
PAAC_HEADS_IN_SAMPLER=0 PAAC_HIP_LIB=$L PROBE_N=256 PROBE_A=4 PROBE_MULTI=1 PROBE_ACT=1 timeout -k 10 120 python3 tools/probe_sampler.py > $OUT/probe_act_256x4.txt 2>&1; echo "probe rc=$?"
tail -3 $OUT/probe_act_128x18.txt; tail -3 $OUT/probe_act_256x4.txt
