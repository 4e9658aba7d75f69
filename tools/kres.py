"""Per-kernel register / scratch / LDS usage of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), demangled.
    python tools/kres.py 3dssd_amd/csrc/mlp_rowwave.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
extra = sys.argv[2:]
flags = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -fPIC".split()
if src.endswith("mlp_rowwave.hip"):
    flags += ["-mllvm", "-pragma-unroll-threshold=4000000"]
out = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: .*?:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except Exception:
            pass
        cur = {"name": re.sub(r"\(anonymous namespace\)::", "", name)}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    print("%-110s VGPR %-4s AGPR %-4s scratch %-6s occ %-3s LDS %s" % (r["name"][:110], r.get("VGPRs"), r.get("AGPRs"),
          r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
