#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs of scripts/profile_round.sh into the summaries kept under profiles/.
  python scripts/summarize_profiles.py gpurun_out/prof r02
HBM bytes (scripts/pmc_kernel.py): reads = TCC_EA0_RDREQ x 128 B minus the 32-B requests counted at their size (gfx950
issues 128-B fabric reads; FETCH_SIZE tallies them at 64 B, MI355X_MICROARCH.md), writes = 64-B requests x 64 + the
others x 32 (this is WRITE_SIZE)."""
import csv, glob, json, os, re, shutil, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_kernel import summarize

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def keep_stats(sub, name):
    for f in glob.glob(os.path.join(src, sub, "stats", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
        return True
    return False


def log_line(sub, key):
    for ln in reversed(open(os.path.join(src, sub, "stats.log")).read().splitlines()):
        if key in ln:
            return ln
    return ""


out = {}
# ---- flip: bench.py --steps 3 --warmup 1 (4 launches of the bench's default shots per step)
keep_stats("flip", "flip")
k = summarize(os.path.join(src, "flip"), "flip4_kernel")
launches = k["time"]["calls"]
bench = json.loads(log_line("flip", '"metric"'))
shots = int(bench["config"]["shots_per_gpu_per_step"])
flip = {
    "command": "scripts/profile_round.sh: rocprofv3 --kernel-trace --stats, then one --pmc pass per counter group, -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bp",
    "kernel": bench["roofline"]["kernel"], "launches": launches, "shots_per_launch": shots,
    "kernel_stats_avg_ms": k["time"]["avg_ms"], "bench_kernel_ms_hip_events": bench["roofline"]["kernel_ms"],
    "hbm_read_bytes_per_launch": k["hbm_read_bytes"] / launches, "hbm_write_bytes_per_launch": k["hbm_write_bytes"] / launches,
    "traffic_bytes_per_launch": k["hbm_traffic_bytes"] / launches, "traffic_bytes_per_shot": k["hbm_traffic_bytes"] / launches / shots,
    "algorithmic_bytes_per_shot": 3144,
    "counters_per_launch": {n: v / launches for n, v in k["counters"].items()},
}
json.dump(flip, open(os.path.join(dst, "%s_flip_pmc.json" % tag), "w"), indent=1)
# pipe utilisation of the same kernel (the sq / wait passes of the same command), as bench.py's flip_binder() reads it
c = flip["counters_per_launch"]
cycles_xcd = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
pipes = {
    "command": flip["command"], "kernel": flip["kernel"], "launches_averaged": launches, "shots_per_launch": shots,
    "cycles_per_xcd": cycles_xcd,
    "valu_instructions_per_shot": c.get("SQ_INSTS_VALU", 0) / shots, "salu_instructions_per_shot": c.get("SQ_INSTS_SALU", 0) / shots,
    "lds_instructions_per_shot": c.get("SQ_INSTS_LDS", 0) / shots,
    # one vector instruction counted as four cycles of its SIMD (SQ_ACTIVE_INST_VALU counts quad-cycles); scripts/ubench_ops.hip
    # measures 1.9 - 3.5 cycles per wave-instruction with four waves per SIMD, so this figure overstates how full the pipes are
    "valu_pipe_busy": c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / max(cycles_xcd * 1024, 1),
    "lds_array_busy": c.get("SQ_LDS_IDX_ACTIVE", 0) / max(cycles_xcd * 256, 1),
    "lds_bank_conflict_share": c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
    "lds_array_cycles_per_shot": c.get("SQ_LDS_IDX_ACTIVE", 0) / shots,
}
json.dump(pipes, open(os.path.join(dst, "%s_flip_pipes.json" % tag), "w"), indent=1)
out["flip_ms"] = flip["kernel_stats_avg_ms"]
out["flip_traffic_per_shot"] = flip["traffic_bytes_per_shot"]

# ---- BP workloads: scripts/bp_one.py / css_one.py print the iteration counter
def bp(sub, name, kernel, edges, label):
    keep_stats(sub, name)
    k = summarize(os.path.join(src, sub), kernel)
    ln = log_line(sub, "iters")
    # iters = what the counters report (the reference's count); ran = iterations actually run (the LDS form ends
    # periodic runs early): the bytes are priced on ran
    if name == "css":
        m = re.search(r"shots (\d+) ms ([\d.]+) .* iters/shot ([\d.]+) fail \d+ ran/shot ([\d.]+)", ln)
        shots, ms, iters, ran = int(m.group(1)), float(m.group(2)), float(m.group(3)) * int(m.group(1)), float(m.group(4)) * int(m.group(1))
    else:
        kv = dict(zip(ln.split()[2::2], ln.split()[3::2]))
        shots, ms, iters, ran = int(kv["shots"]), float(kv["kernel_ms"]), float(kv["iters"]), float(kv["ran"])
    alg = ran * 32 * edges
    d = {
        "command": "scripts/profile_round.sh -> scripts/pmc_run.sh " + label, "kernel": kernel, "shots": shots, "shot_iterations_reported": iters, "shot_iterations_run": ran,
        "call_ms_hip_events_under_kernel_trace": ms, "kernel_time": k["time"],
        "algorithmic_bytes": alg, "algorithmic_GBs_whole_call": alg / ms / 1e6,
        "algorithmic_GBs_in_kernel": alg / k["time"]["total_ms"] / 1e6,
        "hbm_read_bytes": k.get("hbm_read_bytes"), "hbm_write_bytes": k.get("hbm_write_bytes"), "traffic_bytes": k.get("hbm_traffic_bytes"),
        "traffic_over_algorithmic": k.get("hbm_traffic_bytes", 0) / alg,
        "valu_pipe_busy": k["counters"].get("SQ_ACTIVE_INST_VALU", 0) * 4 / (k["counters"].get("GRBM_GUI_ACTIVE", 1) / 8 * 1024),
        # shader clock while the kernel ran: cycles of the counter pass / duration of the trace pass (same command)
        "shader_clock_ghz": k["counters"].get("GRBM_GUI_ACTIVE", 0) / 8 / max(k["time"]["total_ms"] * 1e6, 1),
        "valu_instructions_per_s": k["counters"].get("SQ_INSTS_VALU", 0) / max(k["time"]["total_ms"] * 1e-3, 1e-9),
        "counters": k["counters"],
    }
    json.dump(d, open(os.path.join(dst, "%s_%s_pmc.json" % (tag, name)), "w"), indent=1)
    out[name] = {"GBs_call": d["algorithmic_GBs_whole_call"], "GBs_kernel": d["algorithmic_GBs_in_kernel"], "traffic_ratio": d["traffic_over_algorithmic"],
                 "valu_busy": d["valu_pipe_busy"]}


bp("bp", "bp", "bp_lds_kernel<6, 3, true, true>", 30000, "python3 scripts/bp_one.py c2 8388608 0.02")
bp("c5", "bp_c5", "bps_iterate", 3000000, "python3 scripts/bp_one.py c5 70656 0.02")
bp("css", "css", "bp_lds_kernel<7, -4,", 2100, "python3 scripts/css_one.py 8388608")
bp("c1", "bp_c1", "bp_lds_kernel<4, 3,", 3600, "python3 scripts/bp_one.py c1 4194304 0.02")
print(json.dumps(out, indent=1))
