import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
World()
SCENE = sys.argv[1] if len(sys.argv) > 1 else 'c2'
arm, chain, obs = build_scene(SCENE)
sm, dev = arm._scene_device()
B = 200000 if SCENE == 'c2' else 50000
q = torch.from_numpy(sample_q(chain, B, seed=1)).cuda()
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
print('scene', SCENE, 'pairs', sm.n_pairs, 'hulls', sm.n_hulls)
Bv = 1000000
qv = torch.from_numpy(sample_q(chain, Bv, seed=1)).cuda()
for thr in (0.0, 1e-6):
    print('validity thr %g: ms %.3f for %d -> %.3e cfg/s' % (thr, (ms := t(lambda: dev.validity(qv, thr, packed=True), 10)), Bv, Bv / ms * 1e3))
print('closest ms %.3f for %d -> %.3e cfg/s'%((ms:=t(lambda: dev.closest(q))), B, B/ms*1e3))
print('pair_distances ms %.3f -> %.3e cfg/s'%((ms:=t(lambda: dev.pair_distances(q))), B/ms*1e3))
print('pair_distances+witness ms %.3f -> %.3e cfg/s'%((ms:=t(lambda: dev.pair_distances(q, witness=True))), B/ms*1e3))
print('proximity jacobian rows ms %.3f -> %.3e cfg/s'%((ms:=t(lambda: dev.proximity_jacobian(q))), B/ms*1e3))
q5 = q[:10071]
print('config 5 (10071 samples): proximity records + rows ms %.3f; mask ms %.3f'%(t(lambda: dev.proximity_jacobian(q5), 20), t(lambda: dev.validity(q5, 1e-6), 20)))
