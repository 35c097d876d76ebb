import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene
World()
arm, chain, obs = build_scene('c3')
sm, dev = arm._scene_device()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = np.random.default_rng(3)
lim = chain.joint_limits
# end points U(limits); the goal is pulled towards the start so that |dq| <= pi (max_distance of _test_rrt.py:98)
s = rng.uniform(lim[:,0], lim[:,1], (E, 7)); g = rng.uniform(lim[:,0], lim[:,1], (E, 7))
d = np.linalg.norm(g - s, axis=1)
scale = np.minimum(1.0, rng.uniform(0.2, 1.0, E) * np.pi / d)
g = s + (g - s) * scale[:, None]
ts, tg = torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda()
for res in (0.01,):
    ok, end, ns = dev.edge_validity(ts, tg, res, np.pi)
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ok, end, ns = dev.edge_validity(ts, tg, res, np.pi)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/3
    tot = int(ns.sum().item())
    print('E', E, 'res', res, 'ms %.3f'%ms, 'edges/s %.3e'%(E/ms*1e3), 'samples(len T) %.3e'%tot, 'configs/s (len T) %.3e'%(tot/ms*1e3), 'valid frac %.3f'%ok.float().mean().item())
