"""Prototype check: masks of a variant build against the oracle at positive thresholds, plus the undecided counter."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from numbotics_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
from oracle import cpu_oracle
World()
for scene in ('c2', 'c5m'):
    arm, chain, obs = build_scene(scene)
    sm, dev = arm._scene_device()
    lib = _lib.load()
    orc = cpu_oracle.Oracle(sm)
    out = (ctypes.c_ulonglong * 16)()
    B = 20000
    qn = sample_q(chain, B, seed=5)
    q = torch.from_numpy(qn).cuda()
    big = torch.from_numpy(sample_q(chain, 1_000_000, seed=6)).cuda()
    for thr in (1e-6, 1e-3, 0.02, 0.1):
        got = dev.validity(q, thr).cpu().numpy().astype(bool)
        ref = orc.validity(qn, thr, nthreads=8).astype(bool)
        lib.nbk_debug_narrow_profile(out, 1)
        dev.validity(big, thr, packed=True); torch.cuda.synchronize()
        lib.nbk_debug_narrow_profile(out, 0)
        print(scene, 'thr', thr, 'mismatch', None if ref is None else int((got != ref).sum()), 'undecided per 1e6 configs', out[11], flush=True)
