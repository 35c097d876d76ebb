"""Re-run one seed of tools/fuzz_campaign.py and dissect its mismatching configurations:
    python tools/fuzz_repro.py <seed> [configs]"""
import os, sys, tempfile, numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from numbotics_amd import _lib
from numbotics_amd.physics import World, GraphChain
from numbotics_amd.physics.world import _reset_worlds
from numbotics_amd.robots import Arm
from random_scenes import random_urdf, random_obstacles
from oracle.cpu_oracle import Oracle, build
build()
seed = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
np.set_printoptions(precision=17, linewidth=200)
with tempfile.TemporaryDirectory() as d:
    _reset_worlds(); World()
    rng = np.random.default_rng(seed)
    n_links = int(rng.integers(3, 17))
    meshes = seed % 3 == 0
    chain = GraphChain.from_urdf(random_urdf(rng, n_links, os.path.join(d, "f.urdf"), meshes=meshes))
    arm = Arm(chain, bullet_margins=(seed % 2 == 0))
    obs = random_obstacles(rng, int(rng.integers(1, 9)), mesh_dir=d if meshes else None)
    sm = arm.scene_model()
    orc = Oracle(sm)
    lim = np.asarray(chain.joint_limits, dtype=np.float64)
    lim = np.where(np.isfinite(lim), lim, np.sign(lim) * np.pi)
    q = rng.uniform(lim[:, 0], lim[:, 1], (B, chain.dof))
    _, dev = arm._scene_device()
    for thr in (0.0, 0.01, -0.002, 1e-6):
        ref = orc.validity(q, thr, nthreads=16)
        got = np.asarray(arm.in_collision(q, thr))
        idx = np.nonzero(ref != got)[0]
        print('thr', thr, 'mismatching rows', idx.tolist(), flush=True)
        for i in idx:
            qi = q[i:i + 1]
            print('  row', i, 'oracle', bool(ref[i]), 'device batch', bool(got[i]))
            for name, opts in (('fused', {'two_kernel_min_b': 10**9}), ('two-kernel', {}), ('two-kernel f64 broadphase', {'f64_broad': 1}),
                               ('two-kernel no_reg_broad', {'no_reg_broad': 1})):
                try:
                    import contextlib
                    with contextlib.ExitStack() as st:
                        for k, v in opts.items():
                            st.enter_context(_lib.debug_option(k, v))
                        print('    %-32s %s' % (name, bool(np.asarray(arm.in_collision(np.repeat(qi, 3, 0), thr))[0])))
                except Exception as e:
                    print('    %-32s failed: %s' % (name, e))
            r = orc.pair_distances(qi)
            dd = np.asarray(r[0] if isinstance(r, tuple) else r)[0]
            dg = np.asarray(arm.pair_distances(qi))[0]
            order = np.argsort(dd)[:4]
            for p in order:
                a, b = int(sm.pair_a[p]), int(sm.pair_b[p])
                print('    pair', int(p), 'shapes', a, b, 'oracle distance %.17g' % dd[p], '' if dg is None else 'device %.17g' % dg[p])
            print('    q =', repr(qi[0].tolist()))
