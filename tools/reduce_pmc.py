"""Reduce rocprofv3 PMC passes to the per-launch figures bench.py and DESIGN.md quote.

    python tools/reduce_pmc.py --tag r01_f --batch 1000000

reads  profiles/<tag>_pmc_fetch_size_counter_collection.csv   (rocprofv3 --kernel-trace --pmc FETCH_SIZE)
       profiles/<tag>_pmc_write_size_counter_collection.csv   (... --pmc WRITE_SIZE)
       profiles/<tag>_pmc_sq_counter_collection.csv           (... --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
                                                                     SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU)
       profiles/<tag>_pmc_clk_counter_collection.csv          (... --pmc GRBM_GUI_ACTIVE; optional)
writes profiles/<tag>_traffic.json

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are KiB per
dispatch; on gfx950 wide coalesced reads are reported at 1/2, so fetch bytes = 2 * FETCH_SIZE * 1024.  Every
figure is the mean over the dispatches of that kernel in the run (all at the same batch).
"""
import argparse
import csv
import json
import os
from collections import defaultdict

import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from numbotics_amd.csrc.build import source_digest      # the json records which kernel sources the counters belong to
N_SIMD = 256 * 4
CLOCK_HZ = 2.4e9
# vector-ALU issue cost of one wave64 instruction on a CDNA4 SIMD-32 with other waves interleaved (MI355X_MICROARCH.md: "A wave
# (64 lanes) ... issues each VALU instruction over 2 cycles"; constants table: v_fma_f32 wave64 2 cyc): float32 / integer 2 cycles,
# float64 4 (half rate).  Kernels are priced by the type their arithmetic is in: the float32 broadphase at 2, everything else
# (float64 sweeps, GJK, FK) at 4 -- their integer / move instructions are then over-priced, which makes the utilisation an UPPER bound.
VALU_CYCLES = {"k_broad_f32": 2}
VALU_CYCLES_DEFAULT = 4
N_CU = 256

KERNELS = {"k_broad_f32": "nbk::k_broad_f32", "k_broad_reg": "nbk::k_broad_reg", "k_broad": "nbk::k_broad(", "k_narrow": "nbk::k_narrow",
           "k_fk_frames": "nbk::k_fk_frames", "k_fk": "nbk::k_fk<", "k_validity": "nbk::k_validity", "k_jacobian": "nbk::k_jacobian"}


def short(name):
    for k, pat in KERNELS.items():
        if pat in name:
            return k
    return None


def read(path):
    """{kernel: {counter: [value per dispatch]}}, {kernel: [duration ns]}, {kernel: grid}"""
    vals = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    grid = {}
    if not os.path.exists(path):
        return None, None, None
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            grid[k] = int(r["Grid_Size"])
    return vals, {k: list(v.values()) for k, v in dur.items()}, grid


def mean(x):
    return sum(x) / len(x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--batch", type=int, default=1_000_000)
    ap.add_argument("--also", default=None, help="tag of a second set of passes (tools/fk_time.py under NBK_PMC_CMD) whose kernels are merged in")
    a = ap.parse_args()
    p = lambda s: os.path.join(ROOT, "profiles", f"{a.tag}_pmc_{s}_counter_collection.csv")
    fetch, _, _ = read(p("fetch_size"))
    write, _, _ = read(p("write_size"))
    sq, sq_dur, grid = read(p("sq"))
    clk, clk_dur, _ = read(p("clk"))
    if a.also:
        p2 = lambda s: os.path.join(ROOT, "profiles", f"{a.also}_pmc_{s}_counter_collection.csv")
        f2, _, _ = read(p2("fetch_size"))
        w2, _, _ = read(p2("write_size"))
        s2, s2_dur, g2 = read(p2("sq"))
        for k in (f2 or {}):
            if k not in fetch:
                fetch[k] = f2[k]
        for k in (w2 or {}):
            if k not in write:
                write[k] = w2[k]
        for k in (s2 or {}):
            if sq is not None and k not in sq:
                sq[k], sq_dur[k], grid[k] = s2[k], s2_dur[k], g2[k]
    out = {"command": "rocprofv3 --kernel-trace --pmc <one counter set per pass> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
           "units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch (mean over dispatches); gfx950 correction per "
                    "MI355X_MICROARCH.md: wide coalesced reads report 1/2 -> fetch_bytes = 2*FETCH_SIZE*1024",
           "kernels": {}, "batch": a.batch, "csrc_sha": source_digest()}
    for k in sorted(set(fetch or {}) & set(write or {})):
        fk, wk = mean(fetch[k]["FETCH_SIZE"]), mean(write[k]["WRITE_SIZE"])
        out["kernels"][k] = {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "fetch_bytes_corrected": 2 * fk * 1024,
                             "write_bytes": wk * 1024, "hbm_bytes_corrected": 2 * fk * 1024 + wk * 1024}
    step = [k for k in ("k_broad_f32", "k_broad_reg", "k_broad", "k_narrow", "k_validity") if k in out["kernels"]]
    out["validity_step_kernels"] = step
    out["validity_step_hbm_bytes"] = sum(out["kernels"][k]["hbm_bytes_corrected"] for k in step)
    if sq:
        out["sq"] = {}
        for k, c in sq.items():
            waves = (grid[k] + 63) // 64
            insts = mean(c["SQ_INSTS_VALU"])
            d = {"waves": waves, "valu_insts_per_wave": insts / waves,
                 "lane_utilisation": mean(c["SQ_THREAD_CYCLES_VALU"]) / (insts * 64),
                 "valu_fraction_of_wave_lifetime": insts / mean(c["SQ_WAVE_CYCLES"]),
                 "wait_fraction_of_wave_lifetime": mean(c["SQ_WAIT_INST_ANY"]) / mean(c["SQ_WAVE_CYCLES"]),
                 "salu_insts_per_wave": mean(c["SQ_INSTS_SALU"]) / waves,
                 "lds_insts_per_wave": mean(c["SQ_INSTS_LDS"]) / waves,
                 "kernel_us_under_pmc": mean(sq_dur[k]) / 1e3}
            # SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY count quad-cycles.  Direct estimates over the launch at the nominal
            # 2.4 GHz engine clock: VALU instructions x their issue cycles over the SIMD-cycles available.
            cyc = mean(sq_dur[k]) * 1e-9 * CLOCK_HZ
            vc = VALU_CYCLES.get(k, VALU_CYCLES_DEFAULT)
            d["valu_cycles_per_wave_instruction"] = vc
            d["simd_valu_issue_utilisation"] = insts * vc / (N_SIMD * cyc)
            # the scalar side: ONE scalar unit per CU serves all of its waves; at one scalar instruction per cycle the launch's
            # SALU stream alone takes this long (a floor for the kernel if nothing else overlapped)
            salu = mean(c["SQ_INSTS_SALU"])
            d["salu_issues_per_cu"] = salu / N_CU
            d["salu_issue_us_one_unit_per_cu"] = salu / N_CU / CLOCK_HZ * 1e6
            d["salu_share_of_kernel_time"] = d["salu_issue_us_one_unit_per_cu"] / d["kernel_us_under_pmc"]
            d["s_waitcnt_share_of_wave_lifetime"] = d["wait_fraction_of_wave_lifetime"]
            d["resident_waves_per_simd"] = mean(c["SQ_WAVE_CYCLES"]) * 4 / (N_SIMD * cyc)
            if clk and k in clk and "GRBM_GUI_ACTIVE" in clk[k]:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs and includes the dispatch overhead around short kernels:
                # an upper bound on the clock, recorded as a sanity check of the nominal figure
                d["clock_ghz_upper_bound"] = mean(clk[k]["GRBM_GUI_ACTIVE"]) / 8 / mean(clk_dur[k])
            out["sq"][k] = d
        out["sq_source"] = f"profiles/{a.tag}_pmc_sq_counter_collection.csv (+ _clk_ for GRBM_GUI_ACTIVE)"
    dst = os.path.join(ROOT, "profiles", f"{a.tag}_traffic.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
