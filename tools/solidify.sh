# Round-end measurement pass on the GPU box: tests, smoke, then tools/measure.sh for the benchmarked mode (bench line, per-op
# table, rocprofv3 kernel stats of the same command, PMC traffic).   gpurun -- 'bash tools/solidify.sh [fp16] [tag]'
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/measure.sh ${1:-fp16} ${2:-final}
