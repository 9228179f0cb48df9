"""Join rocprofv3 --pmc per-dispatch counters with the library's op list (dispatch order inside one forward
== op order).  Usage: python tools/pmc_per_op.py <pmc_dir> [...]  -> table for the LAST forward in the trace."""
import csv, glob, os, sys, collections

# launches per forward: NKERNELS, or (NKERNELS=auto) all matching dispatches of the trace divided by the NFWD forwards it holds
# (tools/profile_ops.py --reps 1 runs one warm-up and one profiled forward).  The schedule has more ENTRIES than launches since
# round 4 (a gn_prepare entry whose convolution builds the rows itself launches nothing), so the count comes from the trace.
NK_ENV = os.environ.get("NKERNELS", "auto")
NFWD = int(os.environ.get("NFWD", "2"))
ours = ("conv_ws_kernel", "conv_mfma_kernel", "gn_prepare_kernel", "in_conv_kernel", "out_conv_kernel", "film_kernel", "time_embed_kernel", "xform_kernel")
per_pass = []
for d in sys.argv[1:]:
    rows = collections.OrderedDict()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if not any(k in row["Kernel_Name"] for k in ours):
                continue
            did = int(row["Dispatch_Id"])
            r = rows.setdefault(did, {"name": row["Kernel_Name"], "grid": row.get("Grid_Size", ""), "wg": row.get("Workgroup_Size", ""),
                                      "lds": row.get("LDS_Block_Size", ""), "vgpr": row.get("VGPR_Count", "")})
            r[row["Counter_Name"]] = float(row["Counter_Value"])
    NK = len(rows) // NFWD if NK_ENV == "auto" else int(NK_ENV)
    ids = sorted(rows)[-NK:]
    per_pass.append([rows[i] for i in ids])
n = min(len(p) for p in per_pass)
merged = []
for i in range(n):
    m = {}
    for p in per_pass:
        m.update(p[i])
    merged.append(m)
cols = sorted({k for m in merged for k in m if k not in ("name", "grid", "wg", "lds", "vgpr")})
print("op,kernel,grid,lds,vgpr," + ",".join(cols))
for i, m in enumerate(merged):
    nm = m["name"]
    short = next((k for k in ours if k in nm), nm[:20])
    print(f"{i},{short},{m['grid']},{m['lds']},{m['vgpr']}," + ",".join(f"{m.get(c, 0):.0f}" for c in cols))
