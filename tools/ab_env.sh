# Same-box A/B of an environment switch of the library (read once per process): per-op tables and short bench runs, alternating.
#   gpurun -- 'bash tools/ab_env.sh <outdir> VQVS_WS_PST 0 1 [rounds]'
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; VAR=$2; A=$3; B=$4; R=${5:-2}
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for i in $(seq $R); do
  for v in $A $B; do
    env $VAR=$v timeout 300 python tools/profile_ops.py --precision fp16 > $OUT/ops_${VAR}${v}_$i.txt 2>&1
    grep "^  conv" $OUT/ops_${VAR}${v}_$i.txt | sed "s/^/$VAR=$v round $i: /"
  done
done
for i in $(seq $R); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-modes --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$VAR=$v', d['value'], 'clips/s  conv', r['conv_ms_per_forward'], 'ms  frac', r['frac'])"
  done
done
python - <<PY
import glob, os
out = "$OUT"
rows = {}
for f in sorted(glob.glob(out + "/ops_*.txt")):
    tag = os.path.basename(f)[4:-4]
    d = {}
    for l in open(f):
        t = l.split()
        if len(t) > 5 and t[0].isdigit() and t[1] == "conv": d[int(t[0])] = (" ".join(t[2:-4]), float(t[-4]))
    rows[tag] = d
ops = [4, 6, 12, 28, 30, 32, 34, 66, 68, 100, 126, 186, 234, 236, 250, 262, 266, 268]
first = next(iter(rows.values()))
print("%-18s" % "variant", " ".join("%7d" % o for o in ops))
for tag, d in rows.items():
    print("%-18s" % tag, " ".join("%7.3f" % d[o][1] if o in d else "      -" for o in ops))
for o in ops: print(o, first[o][0])
PY
