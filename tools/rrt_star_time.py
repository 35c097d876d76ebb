"""RRT* on the device: iterations per second with the neighbour connects of an iteration in one connect_batch call,
against one connect() call per neighbour (the reference's loop shape, each call already a whole edge per launch)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
from numbotics_amd.planning.sampling_based import ConnectorParams, DiscreteConnector, EuclideanSpace, PlannerParams, RRTStar
World()
arm, chain, obs = build_scene('c3')
lim = np.asarray(chain.joint_limits, dtype=np.float64)
space = EuclideanSpace(lim[:, 0].copy(), lim[:, 1].copy())
conn = DiscreteConnector(ConnectorParams(resolution=0.01, max_distance=1.0, arm=arm))


class PerEdge:
    """Same connector, connect_batch replaced by one scalar connect() per edge."""
    is_valid, steer = conn.is_valid, conn.steer

    def connect_batch(self, A, B, dist=None):
        return np.array([conn.connect(a, b, distance_func=space.distance) is not None for a, b in zip(A, B)])


cand = sample_q(chain, 20000, seed=5)
free = cand[~np.asarray(arm.in_collision(cand))]
start, goal = free[0], free[1]
for n, k in ((1000, 10), (3000, 50)):
    params = PlannerParams(max_iters=n, k_nearest=k, goal_bias=0.05, rewire_factor=5.0)
    rng = np.random.default_rng(3)
    samples = [goal.copy() if rng.random() < params.goal_bias else s for s in free[2:2 + n]]
    out = []
    for c in (conn, PerEdge()):
        t = RRTStar(space, c, params); t.add_start(start); t.add_goal(goal)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        t.plan(samples)
        out.append((time.perf_counter() - t0, t))
    (tb, a), (ts, b) = out
    assert a.parent == b.parent and a.cost == b.cost
    print('RRT* %d iterations, k=%d, resolution 0.01: %d vertices, %d connects, %d rewires; batched %.3f s (%.0f it/s), '
          'per-edge connect() %.3f s (%.0f it/s)' % (n, k, a.states.shape[0], a.n_candidate_edges, a.n_rewired, tb, n / tb, ts, n / ts))
