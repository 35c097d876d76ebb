"""Throughput of back-to-back validity steps issued on one stream vs round-robin on two / three streams (independent batches):
does the latency-bound narrowphase of one step overlap the issue-bound broadphase of the next?"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
World()
arm, chain, obs = build_scene(sys.argv[1] if len(sys.argv) > 1 else 'c2')
sm, dev = arm._scene_device()
qs = [torch.from_numpy(sample_q(chain, 1_000_000, seed=1 + i)).cuda() for i in range(6)]
for ns in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % ns]):
                dev.validity(qs[i % 6], 0.0, packed=True)
    run(2 * ns); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(60); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('%d stream(s): %.4f ms per step -> %.3e configs/s' % (ns, dt / 60 * 1e3, 60e6 / dt))
