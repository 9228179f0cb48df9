"""Instruction-class histogram of the kernels in a gfx950 assembly dump (hipcc -S --offload-device-only).

    python tools/asm_stats.py conv.s [substring-filter]
Counts are static (whole kernel body), which is what the K-loop-dominated kernels here are compared on.
"""
import collections
import re
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
stats = {}
meta = {}
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        stats[cur] = collections.Counter()
        continue
    if cur is None:
        continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        cur = None
        continue
    t = line.strip().split()
    if not t or t[0].startswith((";", ".", "//")):
        continue
    op = t[0]
    c = stats[cur]
    if op.startswith("v_mfma"):
        c["mfma"] += 1
    elif op.startswith("v_pk_"):
        c["valu_pk"] += 1
    elif op.startswith(("v_cvt", "v_fma_mix")):
        c["valu_cvt/mix"] += 1
    elif op.startswith(("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt")):
        c["valu_trans"] += 1
    elif op.startswith("v_"):
        c["valu"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        c["vmem"] += 1
    elif op.startswith("s_waitcnt"):
        c["waitcnt"] += 1
    elif op.startswith("s_barrier"):
        c["barrier"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
for m in re.finditer(r"\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+)", open(path).read(), re.S):
    meta[m.group(1)] = int(m.group(2))
txt = open(path).read()
for name, c in stats.items():
    if flt not in name or not c:
        continue
    short = re.sub(r"^_ZN4vqvs12_GLOBAL__N_1\d+", "", name)[:70]
    mm = re.search(re.escape(name) + r".*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", txt, re.S)
    keys = ["mfma", "valu", "valu_pk", "valu_cvt/mix", "valu_trans", "salu", "lds", "vmem", "waitcnt", "barrier"]
    print(f"{short:72s} " + " ".join(f"{k}={c[k]}" for k in keys))
