# A/B of library variants on ONE box: per-op tables of the unet64 forward for each libvqvs_<tag>.so given, interleaved with the
# product library so box drift shows.   gpurun -- 'bash tools/ab_ops.sh <outdir> <tag> [<tag> ...]'
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tools/profile_ops.py --precision fp16 > $OUT/ops_hip.txt 2>&1
for t in "$@"; do
  VQVS_LIB_PATH=vq_voice_swap_amd/libvqvs_$t.so timeout 300 python tools/profile_ops.py --precision fp16 > $OUT/ops_$t.txt 2>&1
done
timeout 300 python tools/profile_ops.py --precision fp16 > $OUT/ops_hip2.txt 2>&1
for f in $OUT/ops_*.txt; do echo $f; grep "^  conv" $f; done
