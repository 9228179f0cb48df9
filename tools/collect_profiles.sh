# Copy one tools/measure.sh pass (gpurun_out/<round>_<mode>/) into profiles/: round-tagged bench line, per-op table, rocprofv3 kernel
# stats, and the stamped latest_pmc_* summaries bench.py replays.     bash tools/collect_profiles.sh r05
R=${1:-r05}
cd "$(dirname "$0")/.."
for p in fp16 fp32 bf16; do
  d=gpurun_out/${R}_$p
  [ -d $d ] || continue
  cp $d/pmc_stamp_unet64_$p.json profiles/latest_pmc_stamp_unet64_$p.json
  cp $d/pmc_traffic_per_op_unet64_$p.csv profiles/latest_pmc_traffic_per_op_unet64_$p.csv
  cp $d/pmc_per_op_unet64_$p.csv profiles/latest_pmc_per_op_unet64_$p.csv
  cp $d/kernel_stats_$p.csv profiles/${R}_kernel_stats_unet64_$p.csv
  grep -v "amdgpu.ids\|RuntimeWarning\|L = lib()" $d/ops_unet64_$p.txt > profiles/${R}_ops_unet64_$p.txt
  cp $d/bench_$p.json profiles/${R}_bench_unet64_$p.json
done
[ -f gpurun_out/${R}_configs_1gpu.json ] && cp gpurun_out/${R}_configs_1gpu.json profiles/${R}_configs_1gpu.json
[ -f gpurun_out/parity_margins.jsonl ] && cp gpurun_out/parity_margins.jsonl profiles/${R}_parity_margins.jsonl
[ -f gpurun_out/${R}_ws_phases.txt ] && cp gpurun_out/${R}_ws_phases.txt profiles/${R}_ws_phases.txt
python3 - <<PY
import json, csv
for prec in ("fp16", "fp32"):
    try:
        d = json.load(open(f"profiles/${R}_bench_unet64_{prec}.json"))
    except OSError:
        continue
    r = d["roofline"]
    rows = list(csv.DictReader(open(f"profiles/${R}_kernel_stats_unet64_{prec}.csv")))
    conv = [x for x in rows if "conv_ws_kernel" in x["Name"] or "conv_mfma_kernel" in x["Name"]]
    tot, calls = sum(float(x["TotalDurationNs"]) for x in conv), sum(int(x["Calls"]) for x in conv)
    tr = [x for x in csv.DictReader(open(f"profiles/latest_pmc_traffic_per_op_unet64_{prec}.csv")) if x["kernel"] in ("conv_ws_kernel", "conv_mfma_kernel")]
    gb = (sum(float(x["FETCH_SIZE"]) for x in tr) * 2 + sum(float(x["WRITE_SIZE"]) for x in tr)) * 1024 / 1e9
    alg = r["algorithmic_bytes_per_forward"] / 1e9
    print(f"{prec}: {d['value']} clips/s, {d['ms_per_step']} ms/step; live {r['avg_launch_us']} us/launch frac {r['frac']} (bracketed {r['frac_bracketed']}); "
          f"rocprof {tot / calls / 1e3:.1f} us/launch = {tot / calls * 130 / 1e6:.2f} ms/forward = frac {alg / (tot / calls * 130 / 1e9) / 8000:.4f}; "
          f"traffic {gb:.1f} GB = {gb / alg:.3f} x; library {r['library']}; others {[(o['dtype'], o['value']) for o in d.get('other_modes') or []]}")
    print("   cpu", d.get("cpu_baseline"))
PY
