"""Latency of the scalar calls a planner makes one at a time (NumPy in, Python bool / arrays out)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
from numbotics_amd.planning.sampling_based import ConnectorParams, DiscreteConnector
World()
arm, chain, obs = build_scene('c2')
q = sample_q(chain, 2000, seed=1)
conn = DiscreteConnector(ConnectorParams(resolution=0.01, max_distance=np.pi, arm=arm))
def bench(name, fn, n=1000):
    for i in range(20): fn(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); print('%-38s %.1f us per call' % (name, (time.perf_counter() - t) / n * 1e6))
bench('arm.in_collision(q)', lambda i: arm.in_collision(q[i]))
bench('arm.forward_kinematics(q, frame)', lambda i: arm.forward_kinematics(q[i], 'tool_frame'))
bench('arm.jacobian(q, frame)', lambda i: arm.jacobian(q[i], 'tool_frame'))
bench('connector.connect(a, b)', lambda i: conn.connect(q[i], q[i + 1]))
bench('arm.closest_to(q)', lambda i: arm.closest_to(q[i]), 200)
