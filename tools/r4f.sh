#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_misc.py tests/test_learner_gpu.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/tests.log
for rep in 1 2; do
for v in old new; do
  lib=$PWD/paac_amd/libpaac_hip.so; [ $v = old ] && lib=$PWD/.wt_old/paac_amd/libpaac_hip.so
  PAAC_HIP_LIB=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline > $out/b_$v.json 2>$out/b_$v.err
  python -c "import json; d=json.loads(open('$out/b_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
  PAAC_HIP_LIB=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline --game qbert > $out/q_$v.json 2>$out/q_$v.err
  python -c "import json; d=json.loads(open('$out/q_$v.json').read().strip().splitlines()[-1]); print('qbert $v', d['value'], d['ms_per_step'])"
done
done
PAAC_HIP_LIB=$PWD/paac_amd/libpaac_hip_stamps.so PROBE_ACT=1 timeout -k 10 120 python tools/probe_sampler.py 2>&1 | tail -11
