"""PRM roadmap construction: all candidate edges in one connect_batch call vs one connect call per edge."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
from numbotics_amd.planning.sampling_based import ConnectorParams, DiscreteConnector, EuclideanSpace, PlannerParams, PRM
World()
arm, chain, obs = build_scene('c3')
lim = np.asarray(chain.joint_limits, dtype=np.float64)
space = EuclideanSpace(lim[:, 0].copy(), lim[:, 1].copy())
conn = DiscreteConnector(ConnectorParams(resolution=0.01, max_distance=np.pi, arm=arm))
cand = sample_q(chain, 40000, seed=5)
free = cand[~np.asarray(arm.in_collision(cand))]
start, goal = free[0], free[1]
for n in (2000, 10000):
    params = PlannerParams(max_iters=n, k_nearest=50, goal_bias=0.0)
    samples = list(free[2:2 + n])
    prm = PRM(space, conn, params); prm.add_start(start); prm.add_goal(goal)
    prm.plan(samples); torch.cuda.synchronize()
    t0 = time.perf_counter(); V, nodes, ce, dist = prm.candidate_edges(samples); t1 = time.perf_counter()
    live = dist > np.finfo(np.float32).eps
    ok = conn.connect_batch(nodes[ce[live, 0]], nodes[ce[live, 1]], dist[live]); t2 = time.perf_counter()
    m = 300
    t3 = time.perf_counter()
    for a, b in ce[live][:m]: conn.connect(nodes[a], nodes[b], distance_func=space.distance)
    t4 = time.perf_counter()
    print('PRM %d samples, k=50: %d candidate edges; kNN (nbk_knn_prefix) + edge list %.3f s; connect_batch %.4f s (%.3e edges/s, PCIe incl.); '
          'scalar connect() %.3e edges/s -> %.1f s for the same roadmap; accepted %.3f' % (
          n, live.sum(), t1 - t0, t2 - t1, live.sum() / (t2 - t1), m / (t4 - t3), live.sum() * (t4 - t3) / m, ok.mean()))
