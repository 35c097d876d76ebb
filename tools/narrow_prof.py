"""In-kernel phase timing of k_narrow (cycles per working wave, first chunk): NBK_ABLATE=128 python tools/narrow_prof.py"""
import os, sys, ctypes as C, numpy as np, torch
os.environ.setdefault("NBK_ABLATE", "128")
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from numbotics_amd.csrc import build as _b
from numbotics_amd import _lib as _l
_l.LIB_PATH = _b.build_ablate()          # the diagnostic build (-DNBK_ABLATE_BUILD): the product library has no ablation switches
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
from numbotics_amd import _lib
World()
arm, chain, obs = build_scene(sys.argv[1] if len(sys.argv) > 1 else 'c2')
sm, dev = arm._scene_device()
q = torch.from_numpy(sample_q(chain, 1_000_000, seed=1)).cuda()
lib = _lib.load()
out = (C.c_ulonglong * 16)()
dev.validity(q, 0.0, packed=True); torch.cuda.synchronize()
lib.nbk_debug_narrow_profile(out, 1)
for _ in range(1): dev.validity(q, 0.0, packed=True)
torch.cuda.synchronize()
lib.nbk_debug_narrow_profile(out, 0)
n = out[15]
names = ["count+item loads", "pair record + q row", "FK replay", "core construction", "pre-check + pool put + barrier", "GJK phase"]
tot = sum(out[i] for i in range(6))
print("working waves per launch:", n)
print("max wave lifetime %d ticks; first start -> last end of the launch %d ticks (the launch takes ~0.11 ms)" % (out[14], out[12] - out[13]))
for i, nm in enumerate(names):
    print("%-34s %9.0f cycles/wave  %5.1f %%" % (nm, out[i] / n, 100.0 * out[i] / tot))
print("total %.0f cycles/wave" % (tot / n))
