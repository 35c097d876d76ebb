#!/usr/bin/env python3
"""bench.py -- collision-checked configurations per second (BASELINE.json metric).

Workload (BASELINE.json configs[1]): Kinova-like 7-DoF arm, Arm.in_collision over a 1e6-q batch per GPU,
self pairs (default rule minus the removals of _test_rrt.py:38-61) + one Cube(half_extent=0.4) at
[1.0, 0.0, 0.2], threshold 0.  One "step" = one pass of the hot path over the batch: the validity kernel
writing the packed bit mask, and for N > 1 the RCCL all-gather of every rank's mask words.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` prices the validity kernel against HBM as the contract asks
(algorithmic bytes = 56 B of q + 1 bit of mask per configuration); the kernel is VALU-bound by
construction, so the float64 VALU fraction is reported next to it, and the HBM-bound
pose-writing FK kernel gets its own roofline object.  `cpu_baseline` is the CPU oracle (a port of the
same algorithm, oracle/) timed on this box's host cores on a bounded sample.
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
FP64_VALU_PEAK_TFLOPS = 78.6   # vector float64, FMA = 2 flop


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1_000_000, help="configurations per GPU per step")
    ap.add_argument("--scene", default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    from numbotics_amd.scenes import build_scene, sample_q
    from numbotics_amd.parallel import shard_bounds, allgather_mask_words

    World()
    arm, chain, obstacles = build_scene(args.scene)
    sm, dev = arm._scene_device()
    B = args.batch
    total = B * world
    lo, hi = shard_bounds(total, world, rank)
    # every rank draws the same global batch description; each keeps only its shard (seeded per shard)
    q_host = sample_q(chain, hi - lo, seed=1 + rank)
    q = torch.from_numpy(q_host).cuda()
    n_words = (hi - lo + 63) // 64
    gathered = torch.zeros((world * n_words,), dtype=torch.int64, device="cuda") if world > 1 else None
    # N > 1: the all-gather of a step's packed mask (a latency-bound 125 KB-per-rank message) runs on RCCL's stream while the
    # next step's kernels run; two receive buffers, a buffer is waited for before it is reused and all of them before the
    # clock stops.  gloo (control-flow rehearsal only) and --sync-gather keep the serial form.
    overlap = world > 1 and args.backend == "nccl" and not args.sync_gather
    bufs = [gathered, torch.zeros_like(gathered)] if overlap else None
    pending = [None, None]
    counter = [0]

    state = {"overlap": overlap}

    def gather(words):
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

    def step():
        words = dev.validity(q, 0.0, packed=True)
        if world > 1:
            gather(words)
        return words

    for _ in range(args.warmup):
        words = step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        words = dev.validity(q, 0.0, packed=True)
        ev[k][1].record()
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
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = total * args.steps / elapsed
    # ---- parity spot check against the oracle (the checker, not the thing measured) -------------------
    from oracle.cpu_oracle import Oracle
    orc = Oracle(sm)
    w = words.cpu().numpy().view(np.uint64)
    bits = ((w[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool).reshape(-1)[:hi - lo]
    sl = np.arange(0, hi - lo, max(1, (hi - lo) // 20000))
    parity_ok = bool(np.array_equal(bits[sl], orc.validity(q_host[sl], 0.0, nthreads=8)))
    coll_frac = float(bits.mean())

    # ---- roofline of the dominant kernel (validity) ---------------------------------------------------
    alg_bytes = (hi - lo) * (8.0 * chain.dof) + n_words * 8.0
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic, fk_traffic, traffic_src, valu = None, None, None, None
    import glob
    side = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if side:
        # HBM bytes from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 correction) of this same
        # command at batch 1e6; scaled to this run's batch.  bench.py cannot host the profiler itself.
        with open(side[-1]) as f:
            tj = json.load(f)
        scale = (hi - lo) / float(tj.get("batch", hi - lo))
        traffic = tj["validity_step_hbm_bytes"] * scale
        fk_traffic = tj["kernels"]["k_fk"]["hbm_bytes_corrected"] * scale
        traffic_src = os.path.relpath(side[-1], ROOT)
        valu = tj.get("sq")
    roofline = {"bound": "hbm", "kernel": "k_broad + k_narrow (one nbk_validity_batch call)" if (hi - lo) >= 8192 else "k_validity",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": kern_ms,
                "algorithmic_bytes": alg_bytes, "algorithmic_bytes_per_config": 8.0 * chain.dof + 0.125,
                "note": "VALU-bound by arithmetic intensity (SURVEY.md F8); see DESIGN.md",
                "valu": valu}
    if valu:
        # the roof that does bound the step: vector-ALU issue (256 CUs x 4 SIMDs, one wave64 instruction per 4 cycles at
        # 2.4 GHz).  Instruction counts per wave from the SQ pass of the profile the traffic comes from, time from this run.
        step_k = [k for k in ("k_broad_f32", "k_broad_reg", "k_broad", "k_narrow") if k in valu]
        insts = sum(valu[k]["valu_insts_per_wave"] * valu[k]["waves"] for k in step_k) * scale
        peak = 1024 * 2.4e9 / 4.0
        roofline["valu_issue"] = {"bound": "valu", "achieved": insts / (kern_ms * 1e-3), "peak": peak, "unit": "wave-instructions/s",
                                  "frac": insts / (kern_ms * 1e-3) / peak, "wave_instructions_per_step": insts,
                                  "wave_instructions_per_config": insts / (hi - lo), "kernels": step_k}

    # ---- the HBM-bound kernel of the path: pose-writing FK of one frame -------------------------------
    T = arm.forward_kinematics(q, "tool_frame")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        T = arm.forward_kinematics(q, "tool_frame")
    e1.record()
    torch.cuda.synchronize()
    fk_ms = e0.elapsed_time(e1) / reps
    fk_bytes = (hi - lo) * (8.0 * chain.dof + 128.0)
    fk_gbs = fk_bytes / (fk_ms * 1e-3) / 1e9
    fk_roofline = {"bound": "hbm", "kernel": "k_fk", "achieved": fk_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": fk_gbs / HBM_PEAK_GBS, "traffic": fk_traffic, "kernel_ms": fk_ms, "algorithmic_bytes": fk_bytes,
                   "poses_per_s": (hi - lo) / (fk_ms * 1e-3), "algorithmic_bytes_per_config": 8.0 * chain.dof + 128.0}

    # ---- every link pose from one sweep (56 + 128 L bytes per configuration) ----------------
    TA, link_names = arm.forward_kinematics_all(q)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        TA, _ = arm.forward_kinematics_all(q)
    e1.record()
    torch.cuda.synchronize()
    fa_ms = e0.elapsed_time(e1) / 10
    L = len(link_names)
    fa_bytes = (hi - lo) * (8.0 * chain.dof + 128.0 * L)
    fk_all_roofline = {"bound": "hbm", "kernel": "k_fk_frames", "achieved": fa_bytes / (fa_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": fa_bytes / (fa_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel_ms": fa_ms,
                       "algorithmic_bytes": fa_bytes, "link_frames": L, "poses_per_s": (hi - lo) * L / (fa_ms * 1e-3),
                       "algorithmic_bytes_per_config": 8.0 * chain.dof + 128.0 * L}
    del TA

    # ---- geometric Jacobian of the tool frame (56 in + 6 x dof x 8 out per configuration) --------
    J = arm.jacobian(q, "tool_frame")
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        J = arm.jacobian(q, "tool_frame")
    e1.record()
    torch.cuda.synchronize()
    jac_ms = e0.elapsed_time(e1) / 10
    jac_bytes = (hi - lo) * (8.0 * chain.dof + 48.0 * chain.dof)
    jacobian_roofline = {"bound": "hbm", "kernel": "k_jacobian_reg", "achieved": jac_bytes / (jac_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": jac_bytes / (jac_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel_ms": jac_ms,
                         "algorithmic_bytes": jac_bytes, "jacobians_per_s": (hi - lo) / (jac_ms * 1e-3),
                         "algorithmic_bytes_per_config": 56.0 * chain.dof}
    del J

    # ---- CPU baseline: the oracle (port) on this box's host cores, bounded sample ---------------------
    cpu = None
    if not args.no_cpu_baseline:
        # the box's CPU share for one GPU is 16 cores (gpurun guidance); never more threads than allowed CPUs
        cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)))
        n_s = min(hi - lo, 1_000_000)
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
        cpu = {"value": n_s * reps_cpu / cpu_t, "unit": "configs/s", "cores": cores, "kind": "port",
               "sample": f"{reps_cpu} x the first {n_s} configurations of rank 0's batch, oracle/nbk_oracle.c "
                         f"orc_validity with {cores} pthreads (~{cpu_t * cores:.0f} core-seconds)",
               "single_thread_value": one_thread}

    out = {
        "metric": "collision-checked configs/sec, Kinova 7-DoF + obstacles",
        "value": value, "unit": "configs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"Arm.in_collision, Kinova-like 7-DoF (build-authored kinova_cyl.urdf), scene {args.scene}: "
                               f"{sm.n_pairs} primitive pairs ({sm.n_rshapes} robot shapes, {sm.n_wshapes} obstacle), "
                               f"{B} q per GPU per step, threshold 0, packed bit mask"
                               + (", RCCL all-gather of mask words" if world > 1 else ""),
                   "batch_per_gpu": B, "global_batch": total, "pairs": sm.n_pairs, "parallelism": f"dp{world}",
                   "mask_gather": ("none" if world == 1 else ("overlapped with the next step" if state["overlap"] else "serial")),
                   "arithmetic": "every verdict is decided in float64 (bit-exact vs the CPU oracle); the broadphase culls in float32 with a slack that only lets it cull what float64 would"},
        "roofline": roofline, "fk_roofline": fk_roofline, "fk_all_links_roofline": fk_all_roofline, "jacobian_roofline": jacobian_roofline,
        "cpu_baseline": cpu,
        "collision_fraction": coll_frac, "parity_vs_oracle": "bit-exact" if parity_ok else "MISMATCH",
        "parity_sample": int(sl.size),
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if not parity_ok:
        raise SystemExit("GPU mask differs from the oracle")


if __name__ == "__main__":
    main()
