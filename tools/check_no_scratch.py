"""Build gate for csrc/conv_ws.hip (run by the Makefile): the producers' asynchronous loads sit in registers behind a hand-counted
`s_waitcnt vmcnt(N)`; that is only sound while the compiler emits no memory traffic of its own inside the loop, i.e. while no
instantiation of the kernel has a private segment.  Parses hipcc's -Rpass-analysis=kernel-resource-usage remarks and fails when
any kernel whose mangled name contains the filter reports scratch or spilled registers.

    python tools/check_no_scratch.py build/conv_ws.resource.txt conv_ws_kernel
"""
import re
import sys

path, flt = sys.argv[1], sys.argv[2]
cur, bad, seen = None, [], 0
vals = {}
for line in open(path):
    m = re.search(r"remark:\s+Function Name:\s+(\S+)", line)
    if m:
        cur = m.group(1)
        vals[cur] = {}
        continue
    m = re.search(r"remark:\s+(ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|VGPRs|Dynamic Stack):\s+(\S+)", line)
    if m and cur:
        vals[cur][m.group(1)] = m.group(2)
for name, v in vals.items():
    if flt not in name:
        continue
    seen += 1
    scratch = int(v.get("ScratchSize [bytes/lane]", "0"))
    spill = int(v.get("VGPRs Spill", "0"))
    dyn = v.get("Dynamic Stack", "False") != "False"
    print(f"  {name[-40:]:40s} VGPRs {v.get('VGPRs', '?'):>4s}  scratch {scratch}  spilled VGPRs {spill}")
    if scratch or spill or dyn:
        bad.append(name)
if seen == 0:
    sys.exit(f"check_no_scratch: no kernel matching '{flt}' in {path}")
if bad:
    sys.exit("check_no_scratch: scratch / spills in a kernel with hand-counted vmcnt waits:\n  " + "\n  ".join(bad))
