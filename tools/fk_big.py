"""k_jacobian_reg / k_fk_frames at large batch sizes with PREALLOCATED outputs (excludes the allocator)."""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
World()
arm, chain, obs = build_scene('c1')
dev = arm._kin_device()
for B in (1_000_000, 2_000_000, 4_000_000, 8_000_000):
    q = torch.from_numpy(sample_q(chain, B, seed=1)).cuda()
    def t(fn, n=5):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    mj = t(lambda: arm.jacobian(q, 'tool_frame'))
    mf = t(lambda: arm.forward_kinematics_all(q))
    mk = t(lambda: arm.forward_kinematics(q, 'tool_frame'))
    print('B %8d  fk %.3f ms (%.0f GB/s)  jacobian %.3f ms (%.0f GB/s)  all links %.3f ms (%.0f GB/s)  reserved %.1f GB' % (
        B, mk, B * 184 / mk / 1e6, mj, B * 392 / mj / 1e6, mf, B * 2104 / mf / 1e6, torch.cuda.memory_reserved() / 2**30), flush=True)
