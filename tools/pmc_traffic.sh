# HBM traffic per launch of one unet64 forward: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (MI355X_MICROARCH.md), joined with the library's op list.  gpurun -- 'bash tools/pmc_traffic.sh'
set -x
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/tools/profile_ops.py --reps 1 > /tmp/pmc_$c.log 2>&1
  find /tmp/pmc_$c -name "*counter_collection.csv" | head -2
done
NKERNELS=305 python $GRAFT_REPO_ROOT/tools/pmc_per_op.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic_per_op.csv
head -3 $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic_per_op.csv
