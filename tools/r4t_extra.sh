OUT=$1
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 100 python3 $B --arch NIPS --game pong --envs 32 --no-cpu-baseline > $OUT/nips_pair.json 2>> $OUT/err.txt; echo "pair rc=$?"
PAAC_WGRAD_PAIR=0 timeout -k 10 100 python3 $B --arch NIPS --game pong --envs 32 --no-cpu-baseline > $OUT/nips_nopair.json 2>> $OUT/err.txt; echo "nopair rc=$?"
timeout -k 10 100 python3 $B --arch NIPS --game pong --envs 32 --no-cpu-baseline > $OUT/nips_pair2.json 2>> $OUT/err.txt; echo "pair rc=$?"
