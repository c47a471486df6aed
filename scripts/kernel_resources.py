"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks (VGPRs, scratch, occupancy per kernel).
usage: python scripts/kernel_resources.py build.log [substring ...]   (kernels whose demangled name holds every substring; always lists spilling kernels)"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
want = sys.argv[2:]
blocks = re.split(r"Function Name: ", txt)
names = [b.split(" ")[0].strip() for b in blocks[1:]]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for b, d in zip(blocks[1:], dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    v, a, s, o, l = g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    d = d.replace("jw::", "").replace("(anonymous namespace)::", "")
    if s > 0 or (want and all(w in d for w in want)):
        print(f"V={v:4d} A={a:3d} scratch={s:5d} occ={o} lds={l:6d}  {d[:150]}")
