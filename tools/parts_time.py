"""validity step time against the cap on narrowphase workgroups per sub-queue (debug option narrow_parts_max)"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from numbotics_amd import _lib
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
World()
arm, chain, obs = build_scene(sys.argv[1] if len(sys.argv) > 1 else 'c2')
sm, dev = arm._scene_device()
qs = [torch.from_numpy(sample_q(chain, 1_000_000, seed=1 + i)).cuda() for i in range(5)]
for pm in (32, 16, 12, 8, 6, 4):
    _lib.set_debug_option("narrow_parts_max", pm)
    for thr in (0.0, 1e-6):
        for i in range(5): dev.validity(qs[i], thr, packed=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40): w = dev.validity(qs[i % 5], thr, packed=True)
        e1.record(); torch.cuda.synchronize()
        print('parts_max %2d thr %g: %.4f ms per 1e6' % (pm, thr, e0.elapsed_time(e1) / 40))
