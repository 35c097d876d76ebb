"""Validity rate for robots with more than 16 collision primitives (LDS float64 broadphase) vs fewer (register float32)."""
import os, sys, tempfile, numpy as np, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT','/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from numbotics_amd.physics import World, GraphChain
from numbotics_amd.physics.world import _reset_worlds
from numbotics_amd.robots import Arm
from random_scenes import random_urdf, random_obstacles
with tempfile.TemporaryDirectory() as d:
    for seed, n_links in ((120, 19), (121, 12), (122, 30), (124, 36), (123, 9)):
        _reset_worlds(); World()
        rng = np.random.default_rng(seed)
        chain = GraphChain.from_urdf(random_urdf(rng, n_links, os.path.join(d, 'f.urdf'), max_back=1 if n_links > 20 else 3))
        arm = Arm(chain)
        obs = random_obstacles(rng, 3)
        try:
            sm, dev = arm._scene_device()
        except Exception as e:
            print('links', n_links, 'not supported:', e); continue
        lim = np.asarray(chain.joint_limits, dtype=np.float64); lim = np.where(np.isfinite(lim), lim, np.sign(lim) * np.pi)
        q = torch.from_numpy(rng.uniform(lim[:, 0], lim[:, 1], (1_000_000, chain.dof))).cuda()
        for _ in range(2): m = dev.validity(q, 0.0)
        torch.cuda.synchronize()
        e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): m = dev.validity(q, 0.0)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/5
        print('links %2d dof %2d shapes %2d pairs %4d: %.3f ms per 1e6 (%.3e configs/s), colliding %.3f' % (n_links, chain.dof, sm.n_rshapes, sm.n_pairs, ms, 1e9/ms, m.float().mean().item()), flush=True)
