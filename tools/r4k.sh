#!/bin/bash
# tower K-split variants A/B
out=gpurun_out/$1; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_hip_network.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/tests.log
PAAC_HIP_LIB=$PWD/paac_amd/libpaac_hip_stamps.so timeout -k 10 120 python tools/probe_tower_stamps.py > $out/stamps.txt 2>&1; tail -2 $out/stamps.txt
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["ms_per_step"], [(k["kernel"],k["batch"],k["avg_us"]) for k in d.get("kernels",[]) if "tower" in k["kernel"]])
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2; do
for v in ks0 def ks4; do
  lib=$PWD/paac_amd/libpaac_hip_$v.so; [ $v = def ] && lib=$PWD/paac_amd/libpaac_hip.so
  PAAC_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $out/b32_$v.json 2>$out/b32_$v.err; show $out/b32_$v.json "32 $v"
  [ $rep = 2 ] && continue
  PAAC_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --envs 128 --tmax 20 --game seaquest > $out/b128_$v.json 2>$out/b128_$v.err; show $out/b128_$v.json "128x20 $v"
  PAAC_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --envs 256 > $out/b256_$v.json 2>$out/b256_$v.err; show $out/b256_$v.json "256 $v"
  PAAC_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --envs 64 > $out/b64_$v.json 2>$out/b64_$v.err; show $out/b64_$v.json "64 $v"
done
done
