# Redo only the rocprofv3 --kernel-trace --stats pass of tools/measure.sh for one mode (same output names).  gpurun -- 'bash tools/restat.sh fp16 r06_fp16'
PREC=${1:-fp16}; TAG=${2:-meas_$PREC}
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --precision $PREC --steps 2 --warmup 0 --no-cpu-baseline --no-other-modes --no-other-configs > $OUT/bench_under_rocprof_$PREC.json 2> $OUT/bench_under_rocprof_$PREC.err
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$PREC.csv
head -8 $OUT/kernel_stats_$PREC.csv
