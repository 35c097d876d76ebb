"""Validity time against batch size (where the fused single-kernel path hands over to broadphase + narrowphase)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
World()
arm, chain, obs = build_scene('c2')
sm, dev = arm._scene_device()
qa = torch.from_numpy(sample_q(chain, 1 << 17, seed=1)).cuda()
for B in (64, 256, 1024, 2048, 4096, 8191, 8192, 16384, 65536, 131072):
    q = qa[:B]
    for _ in range(3): dev.validity(q, 0.0, packed=True)
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): dev.validity(q, 0.0, packed=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    print('B %6d  %.4f ms  %.3e configs/s  (min_B=%s)' % (B, ms, B/ms*1e3, os.environ.get('NBK_TWO_KERNEL_MIN_B','1')))
