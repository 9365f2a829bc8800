OUT=$1
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
for i in 1 2; do
timeout -k 10 200 python3 $B --host-envs --no-cpu-baseline --no-roofline --steps 40 --windows 1 > $OUT/host_graph_$i.json 2>> $OUT/err.txt; echo "graph rc=$?"
PAAC_HOST_GRAPH=0 timeout -k 10 200 python3 $B --host-envs --no-cpu-baseline --no-roofline --steps 40 --windows 1 > $OUT/host_eager_$i.json 2>> $OUT/err.txt; echo "eager rc=$?"
done
