import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
World()
arm, chain, obs = build_scene('c1')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
q = torch.from_numpy(sample_q(chain, B, seed=1)).cuda()
for f in ('tool_frame',):
    for _ in range(3): arm.forward_kinematics(q, f)
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): T = arm.forward_kinematics(q, f)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    print(f, 'B', B, 'ms %.4f'%ms, 'GB/s %.0f'%(B*184/ms/1e6), 'poses/s %.3e'%(B/ms*1e3))
for _ in range(3): arm.jacobian(q, 'tool_frame')
torch.cuda.synchronize()
e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): J = arm.jacobian(q, 'tool_frame')
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/20
print('jacobian ms %.4f'%ms, 'GB/s %.0f'%(B*392/ms/1e6))
names = list(arm._kin.frames.keys())
for _ in range(3): arm.forward_kinematics_all(q)
torch.cuda.synchronize()
e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): TA, _n = arm.forward_kinematics_all(q)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/10
L = len(names)
print('all %d link poses ms %.4f'%(L, ms), 'GB/s %.0f'%(B*(56+128*L)/ms/1e6), 'poses/s %.3e'%(B*L/ms*1e3))
