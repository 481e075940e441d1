# Register / LDS / spill summary of every kernel in one HIP source:  bash tools/kres.sh robustvideomatting_amd/csrc/mbconv_fused.hip [filter]
f=$1; flt=${2:-.}
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
python3 -c '
import sys, re
cur = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur = {"fn": t.split(":",1)[1].strip()}
    else:
        k, _, v = t.partition(":")
        if k.strip() in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs", "VGPRs Spill", "SGPRs Spill", "LDS Size [bytes/block]"):
            cur[k.strip().split(" ")[0]+("sp" if "Spill" in k else "")] = v.strip()
if cur: print(cur)
' | grep -E "$flt"
