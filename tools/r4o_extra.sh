OUT=$1
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 100 python3 $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/d_driver.json 2>> $OUT/err.txt; echo "driver-style rc=$?"
timeout -k 10 100 python3 $B --no-cpu-baseline > $OUT/d_default.json 2>> $OUT/err.txt; echo "default rc=$?"
