"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel name: sum of each counter,
dispatch count.  Usage: python tools/pmc_summary.py <dir> [<dir> ...]"""
import csv, glob, os, sys, collections

def short(name):
    n = name
    for key in ("conv_ws_kernel", "conv_mfma_kernel", "gn_prepare_kernel", "in_conv_kernel", "out_conv_kernel", "film_kernel", "time_embed_kernel",
                "ddpm_step_kernel", "ddpm_x0sum_kernel", "randn_kernel"):
        if key in n:
            tail = ""
            if key == "conv_mfma_kernel":
                import re
                m = re.search(r"conv_mfma_kernelI(\w+?)Lb([01])ELi(\d+)ELi(\d+)E", n)
                if m:
                    tail = f"<{'f32' if m.group(1)=='f' else 'bf16'},x3={m.group(2)},WN={m.group(3)},HALO={m.group(4)}>"
            return key + tail
    return n[:60]

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k].add((f, row["Dispatch_Id"]))
counters = sorted({c for v in agg.values() for c in v})
print("kernel,dispatches," + ",".join(counters))
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", agg[k].get("GRBM_GUI_ACTIVE", 0))):
    print(f"{k},{len(cnt[k])}," + ",".join(f"{agg[k].get(c, 0):.0f}" for c in counters))
