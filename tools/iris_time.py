"""IRIS' bisection search (safe_sets.py:124-134) for M = 10 071 samples x 15 rounds: host loop vs device tensors vs one hipGraph.
    python tools/iris_time.py [scene]"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
from numbotics_amd.planning import counter_example_bisection
for scene in (sys.argv[1:] or ["c2", "c5m"]):
    from numbotics_amd.physics.world import _reset_worlds
    _reset_worlds(); World()
    arm, chain, obs = build_scene(scene)
    pts = sample_q(chain, 10071, seed=31)
    seed_q = np.zeros(chain.dof)
    pd = torch.from_numpy(pts).cuda()
    ref = counter_example_bisection(arm, seed_q, pts, 15, 1e-6)
    for name, fn in (("host loop (NumPy in/out, H2D + D2H per round)", lambda: counter_example_bisection(arm, seed_q, pts, 15, 1e-6)),
                     ("device tensors", lambda: counter_example_bisection(arm, seed_q, pd, 15, 1e-6)),
                     ("device tensors + hipGraph", lambda: counter_example_bisection(arm, seed_q, pd, 15, 1e-6, graph=True))):
        out = fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        same = np.array_equal(out.cpu().numpy() if torch.is_tensor(out) else out, ref)
        print(f"{scene}: {name:48s} median {np.median(ts) * 1e3:7.3f} ms  min {min(ts) * 1e3:7.3f} ms  bit-equal {same}")
