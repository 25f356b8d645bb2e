#!/bin/bash
# Register / LDS / spill figures of every kernel of one .hip file (compiles it for gfx950, nothing is kept).
#   tools/kernel_regs.sh kafka_topic_analyzer_amd/csrc/kta_alive.hip [extra hipcc flags]
set -u
cd "$(dirname "$0")/.."
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I kafka_topic_analyzer_amd/csrc "$@" -c "$src" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, subprocess, sys
cur = None
def flush():
    if cur: print(cur)
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        flush()
        name = t.split(":", 1)[1].strip()
        try: name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except Exception: pass
        name = re.sub(r"\(.*", "", name).replace("kta::(anonymous namespace)::", "").replace("void ", "")
        cur = name[:56].ljust(58)
    elif cur:
        for k, s in (("VGPRs:", "vgpr"), ("AGPRs:", "agpr"), ("ScratchSize", "scratch"), ("Occupancy", "occ"), ("LDS Size", "lds"), ("SGPRs Spill", "sspill"), ("VGPRs Spill", "vspill"), ("TotalSGPRs", "sgpr")):
            if t.startswith(k): cur += " %s %s" % (s, t.rsplit(":", 1)[1].strip())
flush()
'
