"""Where the scalar in_collision(q) microseconds go: raw C call vs the Python layers above it."""
import os, sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
World()
arm, chain, obs = build_scene('c2')
sm, dev = arm._scene_device()
q = sample_q(chain, 2000, seed=1)
lib = dev._lib
out = C.c_int32(0)
def bench(name, fn, n=2000):
    for i in range(50): fn(i)
    t = time.perf_counter()
    for i in range(n): fn(i)
    print('%-44s %.1f us' % (name, (time.perf_counter() - t) / n * 1e6))
bench('C: nbk_validity_scalar_host', lambda i: lib.nbk_validity_scalar_host(dev._h, q[i].ctypes.data, 0.0, C.byref(out)))
bench('dev.validity_scalar(q)', lambda i: dev.validity_scalar(q[i], 0.0))
bench('arm.in_collision(q)', lambda i: arm.in_collision(q[i]))
qt = torch.from_numpy(q[:1]).cuda(); w = torch.zeros(1, dtype=torch.int64, device='cuda')
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def dev_call(i):
    lib.nbk_validity_batch(dev._h, qt.data_ptr(), 1, 0.0, w.data_ptr(), None, st); torch.cuda.synchronize()
bench('C: nbk_validity_batch(B=1) + synchronize', dev_call)
def launch_only(i):
    lib.nbk_validity_batch(dev._h, qt.data_ptr(), 1, 0.0, w.data_ptr(), None, st)
bench('C: nbk_validity_batch(B=1) enqueue only', launch_only); torch.cuda.synchronize()
