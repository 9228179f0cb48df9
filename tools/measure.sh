# Measurement pass on the GPU box for ONE precision mode (default fp16, the benchmarked mode):
#   gpurun -- 'bash tools/measure.sh [fp16|bf16|fp32] [tag]'
# Writes under gpurun_out/<tag>/: the bench line, the per-op table (live hipEvents), the rocprofv3 --kernel-trace --stats
# summary of the same bench command (labelled by mode: the bench runs ONLY that mode, --no-other-modes), and the per-launch
# HBM traffic from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md: never combined with other traces).
PREC=${1:-fp16}
TAG=${2:-meas_$PREC}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py --precision $PREC > $OUT/bench_$PREC.json 2> $OUT/bench_$PREC.err; tail -c 400 $OUT/bench_$PREC.json
timeout 300 python tools/profile_ops.py --precision $PREC > $OUT/ops_unet64_$PREC.txt 2>&1
cd /tmp && rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --precision $PREC --steps 2 --warmup 0 --no-cpu-baseline --no-other-modes --no-other-configs > $OUT/bench_under_rocprof_$PREC.json 2> $OUT/bench_under_rocprof_$PREC.err
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$PREC.csv
head -5 $OUT/kernel_stats_$PREC.csv
if [ "$3" != "nopmc" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/tools/profile_ops.py --precision $PREC --reps 1 > /tmp/pmc_$c.log 2>&1
done
NKERNELS=auto python $GRAFT_REPO_ROOT/tools/pmc_per_op.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $OUT/pmc_traffic_per_op_unet64_$PREC.csv
head -3 $OUT/pmc_traffic_per_op_unet64_$PREC.csv
# issue-side counters per op (MFMA busy cycles, VALU / SALU / LDS instruction counts, LDS bank conflicts, wait cycles), two more passes
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/pmc_s$i
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_s$i -o pmc -- python $GRAFT_REPO_ROOT/tools/profile_ops.py --precision $PREC --reps 1 > /tmp/pmc_s$i.log 2>&1 || tail -3 /tmp/pmc_s$i.log
done
NKERNELS=auto python $GRAFT_REPO_ROOT/tools/pmc_per_op.py /tmp/pmc_s1 /tmp/pmc_s2 > $OUT/pmc_per_op_unet64_$PREC.csv
head -3 $OUT/pmc_per_op_unet64_$PREC.csv
# stamp: the library these counters were collected on (bench.py replays profiles/latest_pmc_* only for the same build id)
python - > $OUT/pmc_stamp_unet64_$PREC.json <<PY
import json, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from vq_voice_swap_amd import _native
print(json.dumps({"library": _native.lib().vqvs_version().decode(), "precision": "$PREC", "workload": "unet64 forward, 64 clips x 64000 samples",
                  "collected_by": "tools/measure.sh: rocprofv3 --pmc, one counter set per pass"}))
PY
cat $OUT/pmc_stamp_unet64_$PREC.json
fi
