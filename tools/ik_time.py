import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
World()
arm, chain, obs = build_scene('c2')
dev = arm._kin_device()
B = 200000
qt = sample_q(chain, B, seed=1, margin=0.2)
rng = np.random.default_rng(0)
for spread in (0.5, 3.0):
    q0 = torch.from_numpy(qt + rng.uniform(-spread, spread, qt.shape)).cuda()
    pose = arm.forward_kinematics(torch.from_numpy(qt).cuda(), 'tool_frame')
    ok, q, nrm, it = dev.ik(pose, q0, 'tool_frame'); torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ok, q, nrm, it = dev.ik(pose, q0, 'tool_frame')
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/3
    print('ik spread %.1f: %d problems %.3f ms -> %.3e problems/s, solved %.3f, mean steps %.1f, max %d -> %.3e LM steps/s' % (
        spread, B, ms, B/ms*1e3, ok.double().mean().item(), it.double().mean().item(), it.max().item(), it.sum().item()/ms*1e3))
