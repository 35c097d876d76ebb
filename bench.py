#!/usr/bin/env python3
"""bench.py -- collision-checked configurations per second (BASELINE.json metric).

Workload (BASELINE.json configs[1]): Kinova-like 7-DoF arm, Arm.in_collision over a 1e6-q batch per GPU,
self pairs (default rule minus the removals of _test_rrt.py:38-61) + one Cube(half_extent=0.4) at
[1.0, 0.0, 0.2], threshold 0, in the DEFAULT shape mode: Bullet's own collision margins on every box / cylinder / hull
(Arm(chain) = bullet_margins=True -- what pybullet.createCollisionShape builds for the reference, numbotics/utils/shape.py:60-109);
`modes` carries the same step on the sharp analytic shapes (bullet_margins=False) beside it.  One "step" = one pass of the hot path over one batch: the validity kernels
writing the packed bit mask, and for N > 1 the RCCL all-gather of every rank's mask words.  The timed loop rotates
over 5 distinct q batches (5 x 56 MB = 280 MB > the 256 MiB Infinity Cache), so no step finds its input in cache.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B | --global-batch G]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` is whole-job configurations / wall time of the K timed steps (inputs resident in
HBM).  Beside it, all timed with HIP events on the launch stream, median and min over >= 10 repetitions:
  roofline              the validity step against HBM as the contract asks (56 B of q + 1 bit per configuration); the
                        step is arithmetic / latency-bound by construction; `valu_issue` prices it against vector
                        issue per kernel dtype (float32 wave64 = 2 cycles on a SIMD-32, float64 = 4)
  modes                 the headline step per shape mode (bullet margins = default = headline; sharp): which narrowphase build
                        ran, per-step kernel time median / min / max over >= 20 rotating batches (the tail shows), configs/s
  end_to_end            the same step from HOST memory: H2D of q (pinned and pageable) + kernels + D2H of the mask words
  edge_roofline         BASELINE config 3: E = 1e5 DiscreteConnector edges, resolution 0.01, 8-cube ring
  records               BASELINE config 5: M = 10 071 samples, every pair's distance / contact points / normal / (P,7) row,
                        on the primitive scene and on the mesh scene (compound-mesh collision shapes)
  config1_plumbing      config 1: tool_frame FK of 1024 q, CPU oracle rate beside one device call
  config4_shard         one GPU's share of config 4's 1e7 batch (1.25e6 q)
  config4_strong        N > 1 only, no flag needed: BASELINE config 4 itself -- 1e7 q split over the ranks, gathered mask checked
                        against the oracle on rank 0, ms per step with a serial and with an overlapped all-gather
  two_streams           the headline steps issued alternately on two HIP streams (independent batches): throughput with the
                        narrowphase of one step under the broadphase of the next
  fk_roofline, fk_all_links_roofline, jacobian_roofline   the HBM-bound kernels of the path
  cpu_baseline          the CPU oracle (a port of the same algorithm, oracle/) on this box's host cores, bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
N_ROTATE = 5                   # distinct q batches in the timed loop


def stats_ms(ms):
    ms = sorted(float(x) for x in ms)
    return {"median_ms": ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2]),
            "min_ms": ms[0], "mean_ms": float(np.mean(ms)), "reps": len(ms)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1_000_000, help="configurations per GPU per step (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0, help="total configurations per step, split over the ranks (strong scaling; "
                    "BASELINE config 4 is --global-batch 10000000)")
    ap.add_argument("--scene", default="c2")
    ap.add_argument("--sharp", action="store_true", help="headline on the sharp analytic shapes (bullet_margins=False) instead of the "
                    "default Bullet-margin shapes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline step (profiling runs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the control flow)")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0")
    ap.add_argument("--sync-gather", action="store_true", help="N > 1: wait for every step's mask all-gather before the next step's kernels "
                    "(default: the gather of step k overlaps the kernels of step k+1; everything completes inside the timed region)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the device path has no CPU fallback)"
    torch.cuda.set_device(0 if args.all_ranks_on_device0 else local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    from numbotics_amd.physics import World
    from numbotics_amd.physics.world import _reset_worlds
    from numbotics_amd.scenes import build_scene, sample_q
    from numbotics_amd.parallel import shard_bounds, shard_words, allgather_mask_words
    from numbotics_amd.csrc.build import source_digest

    World()
    arm, chain, obstacles = build_scene(args.scene, bullet_margins=not args.sharp)
    sm, dev = arm._scene_device()
    mode_name = "sharp (bullet_margins=False)" if args.sharp else "bullet margins (default: Arm(chain), bullet_margins=True)"
    strong = args.global_batch > 0
    total = args.global_batch if strong else args.batch * world
    lo, hi = shard_bounds(total, world, rank)
    B = hi - lo
    # every rank keeps only its shard (seeded per shard and per rotation slot)
    q_host = sample_q(chain, B, seed=1 + rank)
    qs = [torch.from_numpy(q_host).cuda()] + [torch.from_numpy(sample_q(chain, B, seed=1 + rank + 1000 * i)).cuda() for i in range(1, N_ROTATE)]
    q = qs[0]
    n_words = (B + 63) // 64
    g_words = shard_words(total, world)          # every rank contributes this many words (a short last shard is padded with zero bits)
    gathered = torch.zeros((world * g_words,), dtype=torch.int64, device="cuda") if world > 1 else None
    padded = torch.zeros((g_words,), dtype=torch.int64, device="cuda") if world > 1 and g_words != n_words else None
    # N > 1: the all-gather of a step's packed mask (a latency-bound 125 KB-per-rank message) runs on RCCL's stream while the
    # next step's kernels run; two receive buffers, a buffer is waited for before it is reused and all of them before the
    # clock stops.  gloo (control-flow rehearsal only) and --sync-gather keep the serial form.
    overlap = world > 1 and args.backend == "nccl" and not args.sync_gather
    bufs = [gathered, torch.zeros_like(gathered)] if overlap else None
    pending = [None, None]
    counter = [0]
    state = {"overlap": overlap}

    def gather(words):
        if padded is not None:                   # unequal shards (strong scaling of a batch that does not divide evenly)
            padded[:n_words].copy_(words)
            words = padded
        if not state["overlap"]:
            allgather_mask_words(words, gathered)
            return
        i = counter[0] & 1
        counter[0] += 1
        if pending[i] is not None:
            pending[i].wait()
        try:
            pending[i] = dist.all_gather_into_tensor(bufs[i], words.contiguous(), async_op=True)
        except (TypeError, RuntimeError, NotImplementedError) as exc:      # an API-level refusal is the same on every rank
            if counter[0] != 1:
                raise
            state["overlap"] = False
            print(f"[bench] overlapped gather unavailable ({exc}); using the serial form", file=sys.stderr)
            allgather_mask_words(words, gathered)

    def drain():
        for i in (0, 1):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None

    for k in range(args.warmup):
        words = dev.validity(qs[k % N_ROTATE], 0.0, packed=True)
        if world > 1:
            gather(words)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # one HIP event per step boundary on the launch stream: step k runs between ev[k] and ev[k + 1]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(args.steps):
        words = dev.validity(qs[k % N_ROTATE], 0.0, packed=True)
        ev[k + 1].record()
        if world > 1:
            gather(words)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    step_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]
    kern = stats_ms(step_ms)
    kern_ms = kern["mean_ms"]
    last_q_host = q_host if (args.steps - 1) % N_ROTATE == 0 else sample_q(chain, B, seed=1 + rank + 1000 * ((args.steps - 1) % N_ROTATE))

    # ---- BASELINE config 4, whenever N > 1 (no flag needed): a 1e7-q batch strong-scaled over the ranks, packed masks all-gathered ---------
    config4 = None
    if world > 1 and not strong:
        G4 = 10_000_000
        lo4, hi4 = shard_bounds(G4, world, rank)
        B4 = hi4 - lo4
        w4 = shard_words(G4, world)
        q4 = torch.from_numpy(sample_q(chain, max(B4, 1), seed=40 + rank)).cuda()[:B4]
        gath4 = [torch.zeros((world * w4,), dtype=torch.int64, device="cuda") for _ in range(2)]
        pad4 = torch.zeros((w4,), dtype=torch.int64, device="cuda")

        def step4(i, overlapped):
            wd = dev.validity(q4, 0.0, packed=True) if B4 > 0 else pad4[:0]
            if wd.numel() != w4:
                pad4.zero_(); pad4[:wd.numel()].copy_(wd); wd = pad4
            if overlapped and args.backend == "nccl":
                # (a private copy of the 156 KB of words: the next step rewrites its source while this gather may still read it)
                return dist.all_gather_into_tensor(gath4[i & 1], wd.clone().contiguous(), async_op=True)
            allgather_mask_words(wd, gath4[i & 1])
            return None

        def run4(n, overlapped):
            pend = [None, None]
            torch.cuda.synchronize(); dist.barrier()
            t_ = time.perf_counter()
            for i in range(n):
                if pend[i & 1] is not None:
                    pend[i & 1].wait()
                pend[i & 1] = step4(i, overlapped)
            for p_ in pend:
                if p_ is not None:
                    p_.wait()
            torch.cuda.synchronize(); dist.barrier()
            tt = torch.tensor([time.perf_counter() - t_], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item()) / n * 1e3
        run4(2, False)
        ms_serial = run4(5, False)
        ms_overlap = run4(5, True) if args.backend == "nccl" else None
        run4(1, False)                                       # leaves the gathered mask of this batch in gath4[0]
        config4 = {"what": "BASELINE config 4: 1e7 q strong-scaled over the ranks (shard_bounds: 64-aligned contiguous shards), one "
                           "all_gather_into_tensor of the packed mask words per step", "global_batch": G4, "ranks": world,
                   "batch_rank0": B4, "message_bytes_per_rank": 8 * w4, "ms_per_step_serial_gather": ms_serial,
                   "ms_per_step_overlapped_gather": ms_overlap,
                   "configs_per_s": G4 / ((ms_overlap or ms_serial) * 1e-3), "backend": args.backend}
        if rank == 0:
            # the gathered mask against the oracle on a strided slice of the GLOBAL batch (every rank's shard is re-generated here)
            from oracle.cpu_oracle import Oracle as _O4
            gw = gath4[0].cpu().numpy().view(np.uint64)
            ok4, n4 = True, 0
            for r_ in range(world):
                lo_r, hi_r = shard_bounds(G4, world, r_)
                if hi_r <= lo_r:
                    continue
                qr = sample_q(chain, hi_r - lo_r, seed=40 + r_)
                sl4 = np.arange(0, hi_r - lo_r, 997)
                wr = gw[r_ * w4:(r_ + 1) * w4]
                bits_r = ((wr[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool).reshape(-1)
                ok4 = ok4 and bool(np.array_equal(bits_r[sl4], _O4(sm).validity(qr[sl4], 0.0, nthreads=8)))
                n4 += int(sl4.size)
            config4["gathered_mask_vs_oracle"] = ("bit-exact" if ok4 else "MISMATCH") + f" on {n4} configurations of all {world} shards"
        del q4, gath4

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    def timed(fn, reps=10, warm=2):
        """per-repetition HIP-event times (ms) on the launch stream"""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        es = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in es:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return stats_ms([a.elapsed_time(b) for a, b in es])

    value = total * args.steps / elapsed
    # ---- parity spot check against the oracle (the checker, not the thing measured) -------------------
    from oracle.cpu_oracle import Oracle
    orc = Oracle(sm)
    w = words.cpu().numpy().view(np.uint64)
    bits = ((w[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool).reshape(-1)[:B]
    sl = np.arange(0, B, max(1, B // 20000))
    parity_ok = bool(np.array_equal(bits[sl], orc.validity(last_q_host[sl], 0.0, nthreads=8)))
    coll_frac = float(bits.mean())

    # ---- roofline of the dominant kernels (the validity step) --------------------------------------------------
    alg_bytes = B * (8.0 * chain.dof) + n_words * 8.0
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic, fk_traffic, traffic_src, valu, traffic_note = None, None, None, None, None
    import glob
    digest = source_digest()
    side = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if side:
        # HBM bytes from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 correction) of this same
        # command at batch 1e6, scaled to this run's batch.  bench.py cannot host the profiler itself, so the counters come
        # from a committed profile -- but ONLY of the kernel sources this run was built from (csrc_sha); anything else is stale
        with open(side[-1]) as f:
            tj = json.load(f)
        if tj.get("csrc_sha") == digest:
            scale = B / float(tj.get("batch", B))
            traffic = tj["validity_step_hbm_bytes"] * scale
            fk_traffic = tj["kernels"].get("k_fk", {}).get("hbm_bytes_corrected")
            fk_traffic = None if fk_traffic is None else fk_traffic * scale
            traffic_src = os.path.relpath(side[-1], ROOT)
            valu = tj.get("sq")
        else:
            traffic_note = (f"{os.path.relpath(side[-1], ROOT)} was measured on kernel sources {tj.get('csrc_sha')}, this run is "
                            f"{digest}: PMC figures withheld")
    narrow_build = dev.narrow_build(0.0)
    roofline = {"bound": "hbm", "kernel": f"k_broad_f32 + {narrow_build} (one nbk_validity_batch call)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note,
                "kernel_ms": kern_ms, "kernel_ms_median": kern["median_ms"], "kernel_ms_min": kern["min_ms"], "reps": kern["reps"],
                "algorithmic_bytes": alg_bytes, "algorithmic_bytes_per_config": 8.0 * chain.dof + 0.125,
                "kernel_ms_max": max(step_ms),
                "note": "arithmetic / latency-bound by arithmetic intensity (SURVEY.md F8); see DESIGN.md",
                "csrc_sha": digest, "valu": valu}
    if valu:
        # vector-ALU issue, per kernel dtype (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32,
        # float64 at half rate = 4): 256 CUs x 4 SIMDs at 2.4 GHz.  Instruction counts per wave from the SQ pass of the profile the
        # traffic comes from (same kernel sources), time from this run.  `frac` = the SIMD-cycles the step's VALU stream needs
        # over the SIMD-cycles the step lasted -- an upper bound (integer / move instructions of the float64 kernels priced at 4).
        step_k = [k for k in ("k_broad_f32", "k_broad_reg", "k_broad", "k_narrow") if k in valu]
        sc = B / float(tj.get("batch", B))
        per = {}
        need_cycles = 0.0
        for k in step_k:
            cyc = 2.0 if k == "k_broad_f32" else 4.0
            n = valu[k]["valu_insts_per_wave"] * valu[k]["waves"] * sc
            per[k] = {"dtype": "f32" if cyc == 2.0 else "f64", "cycles_per_wave_instruction": cyc, "wave_instructions": n,
                      "peak_wave_instructions_per_s": 1024 * 2.4e9 / cyc, "simd_issue_ms": n * cyc / (1024 * 2.4e9) * 1e3,
                      "salu_insts_per_wave": valu[k].get("salu_insts_per_wave"),
                      "salu_issue_us_one_unit_per_cu": valu[k].get("salu_issue_us_one_unit_per_cu"),
                      "s_waitcnt_share_of_wave_lifetime": valu[k].get("wait_fraction_of_wave_lifetime")}
            need_cycles += n * cyc
        insts = sum(v["wave_instructions"] for v in per.values())
        roofline["valu_issue"] = {"bound": "valu", "achieved": need_cycles / (kern_ms * 1e-3), "peak": 1024 * 2.4e9,
                                  "unit": "SIMD issue cycles/s", "frac": need_cycles / (kern_ms * 1e-3) / (1024 * 2.4e9),
                                  "wave_instructions_per_step": insts, "wave_instructions_per_config": insts / B, "kernels": per}

    out = {
        "metric": "collision-checked configs/sec, Kinova 7-DoF + obstacles",
        "value": value, "unit": "configs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"Arm.in_collision, Kinova-like 7-DoF (build-authored kinova_cyl.urdf), scene {args.scene}: "
                               f"{sm.n_pairs} primitive pairs ({sm.n_rshapes} robot shapes, {sm.n_wshapes} obstacle), "
                               f"{B} q per GPU per step, threshold 0, shape mode: {mode_name}, narrowphase build {narrow_build}, packed bit mask; the timed loop rotates over {N_ROTATE} distinct "
                               f"q batches ({N_ROTATE * B * 8 * chain.dof / 2**20:.0f} MiB)"
                               + (", RCCL all-gather of mask words" if world > 1 else ""),
                   "batch_per_gpu": B, "global_batch": total, "pairs": sm.n_pairs, "parallelism": f"dp{world}",
                   "shape_mode": "sharp" if args.sharp else "bullet_margins", "narrowphase_build": narrow_build,
                   "mask_gather": ("none" if world == 1 else ("overlapped with the next step" if state["overlap"] else "serial")),
                   "arithmetic": "float32 broadphase culls / certifies with a slack that only lets it decide what float64 would decide the "
                                 "same way; every other verdict is computed in float64; the mask is bit-exact vs the CPU oracle on the sample"},
        "roofline": roofline,
        "collision_fraction": coll_frac, "parity_vs_oracle": "bit-exact" if parity_ok else "MISMATCH",
        "parity_sample": int(sl.size),
        "parity_note": "GPU vs this build's CPU oracle (oracle/); parity of collision values vs the reference's PyBullet is UNPINNED "
                       "(third-party, absent, no reference fixture) -- FK / Jacobian / edge sampling are pinned by reference-generated golden vectors",
    }

    # ---- the headline step per shape mode: >= 20 steps rotating over the q batches, one HIP event pair per step (the tail shows) ----
    def mode_object(dev_x, thr=0.0, reps=20):
        for k in range(3):
            dev_x.validity(qs[k % N_ROTATE], thr, packed=True)
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        evs[0].record()
        for k in range(reps):
            dev_x.validity(qs[k % N_ROTATE], thr, packed=True)
            evs[k + 1].record()
        torch.cuda.synchronize()
        ms = [evs[k].elapsed_time(evs[k + 1]) for k in range(reps)]
        st = stats_ms(ms)
        return {"narrowphase_build": dev_x.narrow_build(thr), "kernel_ms_median": st["median_ms"], "kernel_ms_min": st["min_ms"],
                "kernel_ms_max": max(ms), "kernel_ms_mean": st["mean_ms"], "reps": reps, "rotating_batches": N_ROTATE,
                "configs_per_s": B / (st["median_ms"] * 1e-3)}
    modes = {"what": "one nbk_validity_batch call per step at threshold 0, 1e6 q per step rotating over the headline's batches; "
                     "HIP events on the launch stream around every step",
             "headline_mode": "sharp" if args.sharp else "bullet_margins"}
    modes["sharp" if args.sharp else "bullet_margins"] = mode_object(dev)
    arm.bullet_margins = bool(args.sharp)                 # the other mode: recompiles the scene into a second descriptor
    sm_o, dev_o = arm._scene_device()
    other = mode_object(dev_o)
    wo = dev_o.validity(qs[0], 0.0, packed=True).cpu().numpy().view(np.uint64)
    bo = ((wo[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool).reshape(-1)[:B]
    other_ok = bool(np.array_equal(bo[sl], Oracle(sm_o).validity(q_host[sl], 0.0, nthreads=8)))
    other["parity_vs_oracle"] = ("bit-exact" if other_ok else "MISMATCH") + f" on {sl.size} configurations"
    other["collision_fraction"] = float(bo.mean())
    modes["bullet_margins" if args.sharp else "sharp"] = other
    modes["sharp" if args.sharp else "bullet_margins"]["collision_fraction"] = coll_frac
    out["modes"] = modes
    if config4 is not None:
        out["config4_strong"] = config4
        parity_ok = parity_ok and "MISMATCH" not in config4.get("gathered_mask_vs_oracle", "")
    parity_ok = parity_ok and other_ok
    arm.bullet_margins = not args.sharp
    del dev_o

    if not args.no_extras:
        # ---- end to end from host memory: H2D q + kernels + D2H mask words (PCIe inclusive; never `value`) ---------------
        pin_q = torch.from_numpy(q_host).pin_memory()
        pin_w = torch.empty((n_words,), dtype=torch.int64).pin_memory()
        dq = torch.empty_like(q)
        page_q = torch.from_numpy(q_host)

        def e2e(src):
            dq.copy_(src, non_blocking=True)
            wds = dev.validity(dq, 0.0, packed=True)
            pin_w.copy_(wds, non_blocking=True)
        e_pin = timed(lambda: e2e(pin_q), reps=10)
        e_page = timed(lambda: e2e(page_q), reps=10)
        # ... and with q drawn on the device (scenes.sample_q_device): what a sampling planner pays when q never crosses PCIe
        from numbotics_amd.scenes import sample_q_device

        def e2e_dev():
            sample_q_device(chain, B, seed=7, out=dq)
            wds = dev.validity(dq, 0.0, packed=True)
            pin_w.copy_(wds, non_blocking=True)
        e_dev = timed(e2e_dev, reps=10)
        out["end_to_end"] = {"what": "H2D of q + nbk_validity_batch + D2H of the packed mask, per step", "batch": B,
                             "pinned": dict(e_pin, configs_per_s=B / (e_pin["median_ms"] * 1e-3)),
                             "pageable": dict(e_page, configs_per_s=B / (e_page["median_ms"] * 1e-3)),
                             "sampled_on_device": dict(e_dev, configs_per_s=B / (e_dev["median_ms"] * 1e-3),
                                                       what="q ~ U(limits) drawn on the device (torch generator) + the step + D2H of the mask"),
                             "h2d_bytes": B * 8.0 * chain.dof, "d2h_bytes": n_words * 8.0}
        del pin_q, dq, page_q

        # ---- the same steps issued round-robin on two HIP streams (independent batches, as a planner has them): every stream has
        # its own scratch set inside the descriptor, so the latency-bound narrowphase of one step runs under the issue-bound
        # broadphase of the next.  Wall clock over 40 steps; NOT the headline `value`, which stays one stream, one step at a time.
        sts = [torch.cuda.Stream(), torch.cuda.Stream()]
        for st_ in sts:
            st_.wait_stream(torch.cuda.current_stream())

        def issue(n):
            for i in range(n):
                with torch.cuda.stream(sts[i & 1]):
                    dev.validity(qs[i % N_ROTATE], 0.0, packed=True)
        issue(4)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        issue(40)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        out["two_streams"] = {"what": "40 validity steps (same batches as the headline) issued alternately on two HIP streams; wall clock, "
                              "barrier-free region bracketed by synchronize", "ms_per_step": dt2 / 40 * 1e3, "configs_per_s": 40 * B / dt2,
                              "vs_one_stream": (40 * B / dt2) / (B / (kern["median_ms"] * 1e-3))}

        # ---- the HBM-bound kernels of the path, rotating inputs ---------------------------------------------------------
        it = [0]

        def nxt():
            it[0] += 1
            return qs[it[0] % N_ROTATE]
        fk = timed(lambda: arm.forward_kinematics(nxt(), "tool_frame"), reps=20)
        fk_bytes = B * (8.0 * chain.dof + 128.0)
        fk_gbs = fk_bytes / (fk["median_ms"] * 1e-3) / 1e9
        out["fk_roofline"] = dict(fk, bound="hbm", kernel="k_fk", achieved=fk_gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=fk_gbs / HBM_PEAK_GBS,
                                  traffic=fk_traffic, kernel_ms=fk["median_ms"], algorithmic_bytes=fk_bytes,
                                  poses_per_s=B / (fk["median_ms"] * 1e-3), algorithmic_bytes_per_config=8.0 * chain.dof + 128.0)
        _, link_names = arm.forward_kinematics_all(q[:64])
        L = len(link_names)
        fa = timed(lambda: arm.forward_kinematics_all(nxt()), reps=10)
        fa_bytes = B * (8.0 * chain.dof + 128.0 * L)
        fa_gbs = fa_bytes / (fa["median_ms"] * 1e-3) / 1e9
        out["fk_all_links_roofline"] = dict(fa, bound="hbm", kernel="k_fk_frames", achieved=fa_gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                                            frac=fa_gbs / HBM_PEAK_GBS, traffic=None, kernel_ms=fa["median_ms"], algorithmic_bytes=fa_bytes,
                                            link_frames=L, poses_per_s=B * L / (fa["median_ms"] * 1e-3),
                                            algorithmic_bytes_per_config=8.0 * chain.dof + 128.0 * L)
        jc = timed(lambda: arm.jacobian(nxt(), "tool_frame"), reps=10)
        jac_bytes = B * (8.0 * chain.dof + 48.0 * chain.dof)
        jac_gbs = jac_bytes / (jc["median_ms"] * 1e-3) / 1e9
        out["jacobian_roofline"] = dict(jc, bound="hbm", kernel="k_jacobian_reg", achieved=jac_gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                                        frac=jac_gbs / HBM_PEAK_GBS, traffic=None, kernel_ms=jc["median_ms"], algorithmic_bytes=jac_bytes,
                                        jacobians_per_s=B / (jc["median_ms"] * 1e-3), algorithmic_bytes_per_config=56.0 * chain.dof)

        # ---- config 1 (plumbing): tool_frame FK of 1024 random q -- the CPU oracle's rate beside one device call ----------------
        q1h = sample_q(chain, 1024, seed=11, margin=0.1)
        from oracle.cpu_oracle import Oracle as _Orc
        ok1 = _Orc(arm._kin)
        t1 = time.perf_counter()
        n1 = 0
        while time.perf_counter() - t1 < 0.2:
            T_cpu = ok1.fk(q1h, "tool_frame")
            n1 += 1
        cpu1 = 1024 * n1 / (time.perf_counter() - t1)
        q1d = torch.from_numpy(q1h).cuda()
        g1 = timed(lambda: arm.forward_kinematics(q1d, "tool_frame"), reps=20)
        T_gpu = arm.forward_kinematics(q1d, "tool_frame").cpu().numpy()
        out["config1_plumbing"] = {"what": "BASELINE config 1: Kinova tool_frame FK, 1024 random q (q ~ U(lower + 0.1, upper - 0.1), _test_arm.py:58)",
                                   "cpu_oracle_poses_per_s": cpu1, "cpu_kind": "port (oracle/nbk_oracle.c orc_fk, one thread)",
                                   "gpu_call_median_ms": g1["median_ms"], "gpu_poses_per_s": 1024 / (g1["median_ms"] * 1e-3),
                                   "gpu_equals_oracle_bitwise": bool(np.array_equal(T_gpu, T_cpu))}

        # ---- config 4's per-GPU shard: 1e7 / 8 = 1.25e6 q ----------------------------------------------------------------
        if not strong and B == 1_000_000:
            q4 = torch.from_numpy(sample_q(chain, 1_250_000, seed=4)).cuda()
            c4 = timed(lambda: dev.validity(q4, 0.0, packed=True), reps=10)
            out["config4_shard"] = dict(c4, what="one GPU's share of BASELINE config 4 (1e7 q over 8 GPUs): 1.25e6 q, packed mask; the full "
                                        "config is `--gpus 8 --global-batch 10000000`", batch=1_250_000,
                                        configs_per_s=1_250_000 / (c4["median_ms"] * 1e-3))
            del q4
        del qs[1:]

        # ---- config 5: per-sample proximity records on M = 10 071 samples ---------------------------------------------------
        M = 10071
        rec = {"what": "BASELINE config 5 / SURVEY.md 8d C5: per sample every allowed pair's signed distance, contact points, normal "
                       "(80 B) and (P, n_q) proximity-Jacobian row -- nbk_proximity_jacobian_batch -- and the 1e-6 tolerance mask", "samples": M}
        q5 = torch.from_numpy(sample_q(chain, M, seed=31)).cuda()
        r5 = timed(lambda: dev.proximity_jacobian(q5), reps=10)
        m5 = timed(lambda: dev.validity(q5, 1e-6, packed=True), reps=10)
        per_cfg = 8.0 * chain.dof + sm.n_pairs * (80.0 + 8.0 * chain.dof)
        rec["primitive_scene"] = dict(r5, scene=args.scene, pairs=sm.n_pairs, bytes_per_config=per_cfg,
                                      achieved_GBps=M * per_cfg / (r5["median_ms"] * 1e-3) / 1e9, samples_per_s=M / (r5["median_ms"] * 1e-3),
                                      mask_median_ms=m5["median_ms"], mask_min_ms=m5["min_ms"])
        # ---- config 3: E = 1e5 edges, resolution 0.01, 8-cube ring ------------------------------------------------------------
        _reset_worlds()
        World()
        arm3, chain3, obs3 = build_scene("c3")
        sm3, dev3 = arm3._scene_device()
        E = 100_000
        rng = np.random.default_rng(3)
        lim = chain3.joint_limits
        s_ = rng.uniform(lim[:, 0], lim[:, 1], (E, chain3.dof))
        g_ = rng.uniform(lim[:, 0], lim[:, 1], (E, chain3.dof))
        d_ = np.linalg.norm(g_ - s_, axis=1)
        g_ = s_ + (g_ - s_) * np.minimum(1.0, rng.uniform(0.2, 1.0, E) * np.pi / d_)[:, None]     # |dq| <= pi = max_distance (_test_rrt.py:98)
        ts, tg = torch.from_numpy(s_).cuda(), torch.from_numpy(g_).cuda()
        ok, _, ns = dev3.edge_validity(ts, tg, 0.01, np.pi)
        n_samples = int(ns.sum().item())
        ed = timed(lambda: dev3.edge_validity(ts, tg, 0.01, np.pi), reps=10, warm=1)
        sl_e = np.arange(0, E, 250)
        okr = Oracle(sm3).edge_validity(s_[sl_e], g_[sl_e], 0.01, np.pi, nthreads=8)[0]
        edge_parity = bool(np.array_equal(ok.cpu().numpy()[sl_e], okr))
        eb = E * (16.0 * chain3.dof + 1.0)
        out["edge_roofline"] = dict(ed, what="BASELINE config 3: DiscreteConnector.connect over E edges (end points U(limits), |dq| <= pi), resolution "
                                    "0.01, 8 Cube(0.25) on a ring (SURVEY.md 8d C3): nbk_edge_validity_batch, asynchronous", edges=E, pairs=sm3.n_pairs,
                                    checked_samples=n_samples, edges_per_s=E / (ed["median_ms"] * 1e-3),
                                    checked_configs_per_s=n_samples / (ed["median_ms"] * 1e-3), bound="hbm", algorithmic_bytes=eb,
                                    algorithmic_bytes_per_edge=16.0 * chain3.dof + 1.0, achieved=eb / (ed["median_ms"] * 1e-3) / 1e9,
                                    peak=HBM_PEAK_GBS, unit="GB/s", frac=eb / (ed["median_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    valid_fraction=float(ok.float().mean().item()),
                                    parity_vs_oracle=("bit-exact" if edge_parity else "MISMATCH") + f" on {sl_e.size} edges")
        del ts, tg, dev3, arm3
        # ---- config 5 on compound-mesh collision shapes -------------------------------------------------------------------------
        _reset_worlds()
        World()
        arm5, chain5, obs5 = build_scene("c5m")
        sm5, dev5 = arm5._scene_device()
        q5h = sample_q(chain5, M, seed=31)
        q5m = torch.from_numpy(q5h).cuda()
        d5, w5, j5 = dev5.proximity_jacobian(q5m)
        r5m = timed(lambda: dev5.proximity_jacobian(q5m), reps=10)
        m5m = timed(lambda: dev5.validity(q5m, 1e-6, packed=True), reps=10)
        orc5 = Oracle(sm5)
        dr, wr, jr = orc5.proximity_jacobian(q5h[:200])
        mesh_parity = bool(np.array_equal(d5[:200].cpu().numpy(), dr) and np.array_equal(w5[:200].cpu().numpy(), wr)
                           and np.array_equal(j5[:200].cpu().numpy(), jr))
        per_cfg5 = 8.0 * chain5.dof + sm5.n_pairs * (80.0 + 8.0 * chain5.dof)
        rec["mesh_scene"] = dict(r5m, scene="c5m: mesh arm (10 hulls, one two-object compound link) among a rock, a five-object table, an STL "
                                 "wedge and a cube", pairs=sm5.n_pairs, hulls=sm5.n_hulls, hull_vertices=int(len(sm5.hull_verts)),
                                 bytes_per_config=per_cfg5, achieved_GBps=M * per_cfg5 / (r5m["median_ms"] * 1e-3) / 1e9,
                                 samples_per_s=M / (r5m["median_ms"] * 1e-3), mask_median_ms=m5m["median_ms"], mask_min_ms=m5m["min_ms"],
                                 parity_vs_oracle=("bit-exact" if mesh_parity else "MISMATCH") + " on 200 samples (distances, witnesses, rows)")
        out["records"] = rec
        if not (edge_parity and mesh_parity):
            parity_ok = False

    # ---- CPU baseline: the oracle (port) on this box's host cores, bounded sample ---------------------
    if not args.no_cpu_baseline and world == 1:            # (rank 0 at N = 1 only, as the bench contract says)
        # the box's CPU share for one GPU is 16 cores (gpurun guidance); never more threads than allowed CPUs
        cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)))
        n_s = min(B, 1_000_000)
        t_start = time.perf_counter()
        reps_cpu = 0
        while True:
            orc.validity(q_host[:n_s], 0.0, nthreads=cores)
            reps_cpu += 1
            if time.perf_counter() - t_start > 1.5 or reps_cpu >= 64:
                break
        cpu_t = time.perf_counter() - t_start
        t1s = time.perf_counter()
        n_1 = min(n_s, 100_000)
        orc.validity(q_host[:n_1], 0.0, nthreads=1)
        one_thread = n_1 / (time.perf_counter() - t1s)
        out["cpu_baseline"] = {"value": n_s * reps_cpu / cpu_t, "unit": "configs/s", "cores": cores, "kind": "port",
                               "sample": f"{reps_cpu} x the first {n_s} configurations of rank 0's batch, oracle/nbk_oracle.c "
                                         f"orc_validity with {cores} pthreads (~{cpu_t * cores:.0f} core-seconds)",
                               "single_thread_value": one_thread}
    else:
        out["cpu_baseline"] = None

    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if not parity_ok:
        raise SystemExit("GPU results differ from the oracle")


if __name__ == "__main__":
    main()
