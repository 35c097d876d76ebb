import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
from numbotics_amd.parallel import unpack_mask
from oracle.cpu_oracle import Oracle
World()
arm, chain, obs = build_scene('c2')
sm, dev = arm._scene_device()
orc = Oracle(sm)
B = 10_000_000
qh = sample_q(chain, B, seed=7)
q = torch.from_numpy(qh).cuda()
w = dev.validity(q, 0.0, packed=True); torch.cuda.synchronize()
e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): w = dev.validity(q, 0.0, packed=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/3
bits = unpack_mask(w.cpu().numpy(), B)
sl = np.arange(0, B, 97)
ok = np.array_equal(bits[sl], orc.validity(qh[sl], 0.0, nthreads=16))
print('B=1e7 ms %.3f -> %.3e configs/s, collision fraction %.4f, oracle slice (%d) bit-exact: %s, workspace MB %.0f' % (ms, B/ms*1e3, bits.mean(), sl.size, ok, dev.validity_workspace_bytes(B)/1e6))
