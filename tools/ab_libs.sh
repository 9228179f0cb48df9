# Same-box A/B of two builds of the library on the headline workload: alternating runs of bench.py (no CPU baseline, no other modes).
#   gpurun -- 'bash tools/ab_libs.sh <libA.so> <libB.so> [rounds]'      (a path of "-" = the in-tree library)
A=${1:--}; B=${2:--}; R=${3:-3}
cd $GRAFT_REPO_ROOT
for i in $(seq $R); do
  for L in "$A" "$B"; do
    if [ "$L" = "-" ]; then unset VQVS_LIB_PATH; else export VQVS_LIB_PATH=$GRAFT_REPO_ROOT/$L; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-modes 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L', d['value'], 'clips/s  conv', r['conv_ms_per_forward'], 'ms  frac', r['frac'], ' gn', d['kernel_ms_per_forward'].get('gn_prepare'))"
  done
done
