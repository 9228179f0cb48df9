# per-op times of selected convolutions for ablation builds libvqvs_x<bits>.so (VQVS_WS_EXP), at two launch widths
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for g in 256 128; do
  for t in hip "$@"; do
    VQVS_WS_GRID=$g VQVS_LIB_PATH=vq_voice_swap_amd/libvqvs_$t.so timeout 300 python tools/profile_ops.py --precision fp16 --reps 3 > $OUT/ops_${t}_g$g.txt 2>&1
  done
done
python - <<PY
import glob, re, os
out = "$OUT"
ops = [4, 6, 32, 34, 100, 234, 262, 266, 268]
rows = {}
for f in sorted(glob.glob(out + "/ops_*_g*.txt")):
    tag = os.path.basename(f)[4:-4]
    d = {}
    for l in open(f):
        t = l.split()
        if len(t) > 5 and t[0].isdigit() and t[1] == "conv": d[int(t[0])] = (" ".join(t[2:-4]), float(t[-4]))
    rows[tag] = d
print("%-14s" % "variant", " ".join("%9d" % o for o in ops))
first = next(iter(rows.values()))
print("%-14s" % "", " ".join("%9s" % first[o][0][:9] for o in ops))
for tag, d in rows.items():
    print("%-14s" % tag, " ".join("%9.3f" % d[o][1] if o in d else "        -" for o in ops))
PY
