"""Per-kernel register / LDS / scratch / occupancy figures of the gfx950 code object, from the compiler's own
summary comments (hipcc -S).  Usage: python tools/kernel_resources.py [substring]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from numbotics_amd.csrc.build import FLAGS, hipcc
flags = [f for f in FLAGS if f not in ("-shared", "-fPIC")]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "nbk.s")
    subprocess.run([hipcc()] + flags + ["--cuda-device-only", "-S", os.path.join(ROOT, "numbotics_amd/csrc/nbk.hip"), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
pat = re.compile(r"^\s*\.amdhsa_kernel (\S+)$(.*?)^; Occupancy: (\d+)", re.M | re.S)
want = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"{'kernel':60s} vgpr agpr total scratch  lds  occ")
for m in pat.finditer(txt):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = name.split("(")[0]
    if want not in name:
        continue
    body = m.group(2)
    g = lambda k: int(re.search(rf"; {k}: (\d+)", body).group(1))
    print(f"{name[:60]:60s} {g('NumVgprs'):4d} {g('NumAgprs'):4d} {g('TotalNumVgprs'):5d} {g('ScratchSize'):7d} {g('LDSByteSize'):5d} {m.group(3):>4s}")
