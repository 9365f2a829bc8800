#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats, PMC traffic passes (separate, as rocprofv3 requires), bench lines.
# usage: tools/profile_suite.sh <out dir under gpurun_out> [part]     part: all (default) | headline | raw | lines
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
PART=${2:-all}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
if [ $PART = all ] || [ $PART = headline ]; then
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
fi
if [ $PART = all ] || [ $PART = raw ]; then
# BASELINE configs[2]'s "large-batch preprocess/HBM path": 256 environments, raw 210x160 screen pairs + GPU max/resize/stack
RAW="--envs 256 --raw-frames --no-cpu-baseline"
python3 $B $RAW --steps 100 --warmup 10 > $OUT/bench_256envs_raw.json 2> $OUT/b256raw.err
python3 $B --envs 32 --raw-frames --no-cpu-baseline > $OUT/bench_32envs_raw.json 2> $OUT/b32raw.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_raw256 -- python3 $B $RAW --steps 40 --warmup 8 --windows 1 > $OUT/stats_raw256.json 2> $OUT/stats_raw256.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_raw256 -- python3 $B $RAW --steps 8 --warmup 2 --windows 1 --no-roofline > $OUT/pmc_fetch_raw256.json 2> $OUT/pmc_fetch_raw256.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_raw256 -- python3 $B $RAW --steps 8 --warmup 2 --windows 1 --no-roofline > $OUT/pmc_write_raw256.json 2> $OUT/pmc_write_raw256.err
echo "raw done"
fi
if [ $PART = all ] || [ $PART = lines ]; then
python3 $B --envs 256 --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_256envs.json 2> $OUT/b256.err
python3 $B --envs 128 --tmax 20 --game seaquest --no-cpu-baseline --steps 40 --warmup 8 > $OUT/bench_seaquest_128envs_tmax20.json 2> $OUT/b128.err
python3 $B --game qbert --no-cpu-baseline > $OUT/bench_qbert_32envs.json 2> $OUT/bq.err
python3 $B --sampler philox --no-cpu-baseline > $OUT/bench_philox.json 2> $OUT/bp.err
# the reference's default architecture (train.py:93) at its README.md:36 setting and at BASELINE configs[0]'s shape
python3 $B --arch NIPS --game pong --envs 32 --no-cpu-baseline > $OUT/bench_nips_pong_32envs.json 2> $OUT/bn32.err
python3 $B --arch NIPS --game pong --envs 8 --no-cpu-baseline > $OUT/bench_nips_pong_8envs.json 2> $OUT/bn8.err
python3 $B --host-envs --no-cpu-baseline --no-roofline > $OUT/bench_host_envs.json 2> $OUT/bh.err
echo "benches done"
fi
ls $OUT
