"""Randomised parity campaign (not part of the test suite): many random mechanisms / obstacle sets, large batches,
several thresholds, device masks (two-kernel path with the float32 broadphase) against the CPU oracle.
    python tools/fuzz_campaign.py [first_seed] [n_seeds] [configs_per_seed]
Every third seed builds its robot and obstacles WITH MESHES (random polytope files, scaled / offset / auto-centred / compound),
every second compiles the scene with bullet_margins=True (the default), the others with sharp shapes; NBK_FUZZ_ALL=1 adds every other entry point,
NBK_FUZZ_THRESHOLDS="0.05,0.2,-0.01,1e-3" replaces the four default thresholds."""
import os, sys, tempfile, numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from numbotics_amd.physics import World, GraphChain
from numbotics_amd.physics.world import _reset_worlds
from numbotics_amd.robots import Arm
from random_scenes import random_urdf, random_obstacles
from oracle.cpu_oracle import Oracle, build
build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 60000
THRESHOLDS = tuple(float(x) for x in os.environ.get("NBK_FUZZ_THRESHOLDS", "0,0.01,-0.002,1e-6").split(","))
bad = 0
total = 0
with tempfile.TemporaryDirectory() as d:
    for seed in range(first, first + n_seeds):
        _reset_worlds(); World()
        rng = np.random.default_rng(seed)
        n_links = int(rng.integers(3, 17))
        meshes = seed % 3 == 0
        chain = GraphChain.from_urdf(random_urdf(rng, n_links, os.path.join(d, "f.urdf"), meshes=meshes))
        if chain.dof == 0:
            continue
        arm = Arm(chain, bullet_margins=(seed % 2 == 0))
        obs = random_obstacles(rng, int(rng.integers(1, 9)), mesh_dir=d if meshes else None)
        sm = arm.scene_model()
        if sm.n_pairs == 0:
            continue
        orc = Oracle(sm)
        lim = np.asarray(chain.joint_limits, dtype=np.float64)
        lim = np.where(np.isfinite(lim), lim, np.sign(lim) * np.pi)
        q = rng.uniform(lim[:, 0], lim[:, 1], (B, chain.dof))
        line = f"seed {seed}: links {n_links} dof {chain.dof} shapes {sm.n_rshapes}+{sm.n_wshapes} pairs {sm.n_pairs} hulls {sm.n_hulls}" + (" bullet-margins" if seed % 2 == 0 else " sharp")
        for thr in THRESHOLDS:
            ref = orc.validity(q, thr, nthreads=16)
            got = np.asarray(arm.in_collision(q, thr))
            nbad = int((ref != got).sum())
            bad += nbad; total += B
            line += f" | thr {thr:g}: frac {ref.mean():.3f} mismatches {nbad}"
        if os.environ.get("NBK_FUZZ_ALL"):
            # every other entry point, bit for bit
            def same(a, b):
                a, b = np.asarray(a), np.asarray(b)
                return a.shape == b.shape and bool(((a.view(np.int64) == b.view(np.int64)) | (np.isnan(a) & np.isnan(b))).all()) if a.dtype == np.float64 else bool(np.array_equal(a, b))
            n = 1500
            try:
                dist_, w, rows = arm.proximity_jacobians(q[:n])
            except Exception as e:
                if 'UNSUPPORTED' not in str(e):
                    raise
                print(line + ' | per-pair distances unsupported for this robot (too many primitives for LDS)', flush=True)
                continue
            dr, wr, rr = orc.proximity_jacobian(q[:n])
            dmin, idx = arm.closest_distance(q[:n])
            dref, iref = orc.closest(q[:n])
            _, dev = arm._scene_device()
            ok, end, ns = dev.edge_validity(q[:400], q[400:800], 0.03, 1.5, mode="steer")
            okr, endr, nsr = orc.edge_validity(q[:400], q[400:800], 0.03, 1.5, mode="steer", nthreads=16)
            checks = {"dist": same(dist_, dr), "wit": same(w, wr), "rows": same(rows, rr), "closest": same(dmin, dref) and same(idx, iref),
                      "edges": same(ok, okr) and same(ns, nsr) and same(end, endr)}
            frame = list(arm._kin.frames)[-1]
            orc_k = Oracle(arm._kin)
            checks["fk"] = same(arm.forward_kinematics(q[:n], frame), orc_k.fk(q[:n], frame))
            if len(arm._kin.frames[frame].path):
                checks["jac"] = same(arm.jacobian(q[:n], frame), orc_k.jacobian(q[:n], frame))
                pose = orc_k.fk(q[:n], frame)
                q0 = q[:n] + rng.uniform(-0.3, 0.3, (n, chain.dof))
                r1 = arm._kin_device().ik(pose, q0, frame)
                r2 = orc_k.ik(pose, q0, frame)
                checks["ik"] = same(r1[0], r2[0]) and same(r1[1], r2[1]) and same(r1[3], r2[3])
            nb = sum(0 if v else 1 for v in checks.values())
            bad += nb
            line += " | " + " ".join(f"{k}:{'ok' if v else 'MISMATCH'}" for k, v in checks.items())
        print(line, flush=True)
print(f"TOTAL configurations checked {total}, mismatches {bad}")
sys.exit(1 if bad else 0)
