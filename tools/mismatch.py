"""debug: where does the device mask differ from the oracle?  python tools/mismatch.py <scene> <thr> [sharp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
from oracle.cpu_oracle import Oracle
scene, thr = sys.argv[1], float(sys.argv[2])
World()
arm, chain, obs = build_scene(scene, bullet_margins=(len(sys.argv) < 4))
sm = arm.scene_model()
orc = Oracle(sm)
q = sample_q(chain, 20000, seed=2)
mask = arm.in_collision(q, thr)
ref = orc.validity(q, thr, nthreads=8)
bad = np.nonzero(mask != ref)[0]
print("mismatches", len(bad), "device-only hits", int((mask & ~ref).sum()), "missed hits", int((~mask & ref).sum()))
D = orc.pair_distances(q[bad])
for i, b in enumerate(bad[:12]):
    order = np.argsort(D[i])[:3]
    print(b, "dev", bool(mask[b]), "ref", bool(ref[b]), [(int(p), int(sm.pair_a[p]), int(sm.pair_b[p]), float(D[i][p])) for p in order])
