import os, sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
from numbotics_amd import _lib
World()
arm, chain, obs = build_scene('c2')
sm, dev = arm._scene_device()
B = 1_000_000
q = sample_q(chain, B, seed=1)
mask = np.empty((B,), dtype=np.uint8)
lib = _lib.load()
for _ in range(2):
    lib.nbk_validity_batch_host(dev._h, q.ctypes.data, B, 0.0, mask.ctypes.data)
t0 = time.perf_counter()
for _ in range(5): lib.nbk_validity_batch_host(dev._h, q.ctypes.data, B, 0.0, mask.ctypes.data)
dt = (time.perf_counter()-t0)/5
print('nbk_validity_batch_host (pageable numpy, hipMalloc+memcpy per call): %.3f ms -> %.3e configs/s'%(dt*1e3, B/dt))
# pinned + async via torch
qp = torch.from_numpy(q).pin_memory()
for _ in range(2): m = dev.validity(qp.cuda(non_blocking=True), 0.0).cpu()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): m = dev.validity(qp.cuda(non_blocking=True), 0.0, packed=True).cpu()
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
print('pinned H2D + kernels + packed mask D2H: %.3f ms -> %.3e configs/s'%(dt*1e3, B/dt))
