import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from numbotics_amd.csrc import build as _b
from numbotics_amd import _lib as _l
_l.LIB_PATH = _b.build_ablate()          # the diagnostic build (-DNBK_ABLATE_BUILD): the product library has no ablation switches
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
World()
arm, chain, obs = build_scene(sys.argv[1] if len(sys.argv)>1 else 'c2')
sm, dev = arm._scene_device()
q = torch.from_numpy(sample_q(chain, 1_000_000, seed=1)).cuda()
for _ in range(3): dev.validity(q, 0.0, packed=True)
torch.cuda.synchronize()
e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): w = dev.validity(q, 0.0, packed=True)
e1.record(); torch.cuda.synchronize()
print('NBK_ABLATE', os.environ.get('NBK_ABLATE'), 'ms', e0.elapsed_time(e1)/10)
