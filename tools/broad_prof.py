"""In-kernel phase profile of k_broad_f32: python tools/broad_prof.py <lib built with -DNBK_BF32_STAMP> [scene] [sharp]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from numbotics_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
World()
arm, chain, obs = build_scene(sys.argv[2] if len(sys.argv) > 2 else 'c2', bullet_margins=(len(sys.argv) < 4))
sm, dev = arm._scene_device()
qs = [torch.from_numpy(sample_q(chain, 1_000_000, seed=1 + i)).cuda() for i in range(5)]
lib = _lib.load()
out = (ctypes.c_ulonglong * 8)()
for i in range(5): dev.validity(qs[i], 0.0, packed=True)
torch.cuda.synchronize()
lib.nbk_debug_broad_profile(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(10): dev.validity(qs[i % 5], 0.0, packed=True)
e1.record(); torch.cuda.synchronize()
lib.nbk_debug_broad_profile(out, 0)
n = max(1, out[7])
names = ["prologue (q -> LDS)", "sweep", "world shapes", "robot-robot rows", "final flush + mask"]
tot = sum(out[i] for i in range(5))
print("step %.4f ms; waves %d; mean wave lifetime %.0f cycles" % (e0.elapsed_time(e1) / 10, n, tot / n))
for i, nm in enumerate(names):
    print("  %-22s %8.0f cycles per wave  %5.1f %%" % (nm, out[i] / n, 100.0 * out[i] / max(1, tot)))
