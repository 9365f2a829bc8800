OUT=$1
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
for i in 1 2; do
timeout -k 10 100 python3 $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/d_new_$i.json 2>> $OUT/err.txt; echo "new rc=$?"
PAAC_SHORT_GRAPH_FIRST=0 PAAC_SPIN_SYNC=0 timeout -k 10 100 python3 $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/d_old_$i.json 2>> $OUT/err.txt; echo "old rc=$?"
PAAC_SHORT_GRAPH_FIRST=0 timeout -k 10 100 python3 $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/d_spin_$i.json 2>> $OUT/err.txt; echo "spin rc=$?"
PAAC_SPIN_SYNC=0 timeout -k 10 100 python3 $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/d_short_$i.json 2>> $OUT/err.txt; echo "short rc=$?"
done
timeout -k 10 100 python3 $B --gpus 1 --steps 20 --warmup 5 > $OUT/d_full.json 2>> $OUT/err.txt; echo "full rc=$?"
