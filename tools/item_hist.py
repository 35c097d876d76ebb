import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from numbotics_amd.physics import World
from numbotics_amd.scenes import build_scene, sample_q
World()
for scene in ('c2','c3'):
    from numbotics_amd.physics.world import _reset_worlds
    _reset_worlds(); World()
    arm, chain, obs = build_scene(scene)
    sm, dev = arm._scene_device()
    B = 1_000_000
    q = torch.from_numpy(sample_q(chain, B, seed=1)).cuda()
    need = dev.validity_workspace_bytes(B)
    ws = torch.zeros((need,), dtype=torch.uint8, device='cuda')
    dev.validity(q, 0.0, packed=True, workspace=ws); torch.cuda.synchronize()
    W = sm.n_wshapes
    header = (256*128 + 4*(512 + 3*W*16 + 16) + 255) & ~255
    w64 = ws.view(torch.int64).cpu().numpy()
    counts = w64[:256*16:16]
    nblk = (B + 63)//64; cap = ((nblk + 255)//256)*64*sm.n_pairs
    items = w64[header//8:]
    allp = []
    for s in range(256):
        it = items[s*cap: s*cap + int(counts[s])]
        allp.append(it & 0xFFFFF)
    p = np.concatenate(allp)
    h = np.bincount(p, minlength=sm.n_pairs)
    order = np.argsort(-h)
    print(scene, 'items', p.size, 'per config', p.size/B, 'top shares', np.round(h[order[:8]]/p.size, 3), 'pairs with >0.5%:', int((h/p.size > 0.005).sum()), 'of', sm.n_pairs)
