OUT=$1
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 100 python3 $B --sampler philox --no-cpu-baseline > $OUT/d_philox.json 2>> $OUT/err.txt; echo "philox rc=$?"
PAAC_REUSE_ACTING=0 timeout -k 10 100 python3 $B --sampler philox --no-cpu-baseline > $OUT/d_philox_noreuse.json 2>> $OUT/err.txt; echo "philox noreuse rc=$?"
timeout -k 10 100 python3 $B --sampler philox --envs 256 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/d_philox256.json 2>> $OUT/err.txt; echo "philox256 rc=$?"
PAAC_REUSE_ACTING=0 timeout -k 10 100 python3 $B --sampler philox --envs 256 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/d_philox256_noreuse.json 2>> $OUT/err.txt; echo "philox256 noreuse rc=$?"
