"""VGPRs / scratch / occupancy per kernel from hipcc's -Rpass-analysis=kernel-resource-usage output (stdin or a file):
   cd oprl_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage -c X.hip -o /tmp/x.o 2> /tmp/res.txt
   python tools/kernel_regs.py /tmp/res.txt [name filter]"""
import re
import subprocess
import sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split()[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        pass
    if flt not in name:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    v, sc, oc = g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"V {v:>3} scr {sc:>4} occ {oc}  {name[:150]}")
