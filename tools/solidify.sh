# Round-end measurement pass on the GPU box: tests, smoke, bench line, per-op table, conv phase timing, rocprofv3 kernel stats.
#   gpurun -- 'bash tools/solidify.sh'      (results under gpurun_out/, copied into profiles/ afterwards)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
if [ "$1" != "prof" ]; then
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_r1_final.json 2> gpurun_out/bench_r1_final.err; tail -c 600 gpurun_out/bench_r1_final.json
timeout 300 python tools/profile_ops.py > gpurun_out/ops_final2.txt 2>&1
timeout 200 python tools/conv_phases.py > gpurun_out/conv_phases_final.txt 2>&1
fi
cd /tmp && rm -rf /tmp/prof_stats && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.err
find /tmp/prof_stats -name "*.csv" | head
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r01_kernel_stats.csv
