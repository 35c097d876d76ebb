import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
World()
arm, chain, obs = build_scene(sys.argv[1] if len(sys.argv)>1 else 'c2', bullet_margins=not (len(sys.argv) > 2 and sys.argv[2] == 'sharp'))   # [scene] [sharp]
sm, dev = arm._scene_device()
q = torch.from_numpy(sample_q(chain, 1_000_000, seed=1)).cuda()
for thr in (0.0, 1e-6, 0.01, -0.002):
    for _ in range(3): dev.validity(q, thr, packed=True)
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): w = dev.validity(q, thr, packed=True)
    e1.record(); torch.cuda.synchronize()
    print('thr %g: %.4f ms per 1e6' % (thr, e0.elapsed_time(e1)/10))
