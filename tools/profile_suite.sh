#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats, PMC traffic passes (separate, as rocprofv3 requires), bench lines.
# usage: tools/profile_suite.sh <out dir under gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
python3 $B > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default done"; tail -c 300 $OUT/bench_default.json | head -c 300; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B --steps 100 --warmup 20 --no-cpu-baseline > $OUT/stats.json 2> $OUT/stats.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $B --steps 12 --warmup 4 --windows 1 --no-cpu-baseline --no-roofline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $B --steps 12 --warmup 4 --windows 1 --no-cpu-baseline --no-roofline > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "pmc write done"
# MFMA utilisation (north_star): matrix-pipe busy cycles per kernel against the kernel's own clock cycles, its own pass
rocprofv3 -L > $OUT/counters_available.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $B --steps 12 --warmup 4 --windows 1 --no-cpu-baseline --no-roofline > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err
echo "pmc mfma done"
python3 $B --envs 256 --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_256envs.json 2> $OUT/b256.err
python3 $B --envs 128 --tmax 20 --game seaquest --no-cpu-baseline --steps 40 --warmup 8 > $OUT/bench_seaquest_128envs_tmax20.json 2> $OUT/b128.err
python3 $B --game qbert --no-cpu-baseline > $OUT/bench_qbert_32envs.json 2> $OUT/bq.err
python3 $B --sampler philox --no-cpu-baseline > $OUT/bench_philox.json 2> $OUT/bp.err
python3 $B --host-envs --no-cpu-baseline --no-roofline > $OUT/bench_host_envs.json 2> $OUT/bh.err
echo "benches done"
ls $OUT
