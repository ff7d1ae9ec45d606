#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs of scripts/profile_round.sh into the summaries kept under profiles/.
  python scripts/summarize_profiles.py gpurun_out/prof r01
HBM bytes: reads = TCC_EA0_RDREQ x 128 B minus the 32-B requests counted at their size (gfx950 issues
128-B fabric reads; FETCH_SIZE tallies them at 64 B, MI355X_MICROARCH.md), writes = 64-B requests x 64 +
the others x 32 (this is WRITE_SIZE)."""
import csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def counters(name, kernel):
    out = {}
    for f in glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                d = out.setdefault(int(r["Dispatch_Id"]), {})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [out[k] for k in sorted(out)]


def stats(name, kernel, dest):
    for f in glob.glob(os.path.join(src, name, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.reader(open(f)))
        with open(os.path.join(dst, dest), "w") as o:
            w = csv.writer(o)
            for r in rows:
                w.writerow(r)
        for r in rows[1:]:
            if kernel in r[0]:
                return {"calls": int(r[1]), "avg_ns": float(r[3])}
    return None


def traffic(rd, wr):
    rb = [d["TCC_EA0_RDREQ_sum"] * 128 - d.get("TCC_EA0_RDREQ_32B_sum", 0) * 96 for d in rd]
    wb = [d["TCC_EA0_WRREQ_64B_sum"] * 64 + (d["TCC_EA0_WRREQ_sum"] - d["TCC_EA0_WRREQ_64B_sum"]) * 32 for d in wr]
    return rb, wb


# ---- flip (bench.py: 1 warm-up + 3 timed launches of 2^20 shots)
shots = 1 << 20
rb, wb = traffic(counters("flip_rd", "flip2_kernel"), counters("flip_wr", "flip2_kernel"))
ks = stats("flip_stats", "flip2_kernel", "%s_flip_kernel_stats.csv" % tag)
flip = {
    "command": "scripts/profile_round.sh: rocprofv3 --pmc <one counter group per pass> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bp",
    "kernel": "flip2_kernel<3,16,true,false,true>", "shots_per_launch": shots,
    "kernel_stats_avg_ms": ks and ks["avg_ns"] / 1e6,
    "hbm_read_bytes_per_launch": sum(rb) / len(rb), "hbm_write_bytes_per_launch": sum(wb) / len(wb),
    "traffic_bytes_per_launch": sum(rb) / len(rb) + sum(wb) / len(wb),
    "traffic_bytes_per_shot": (sum(rb) / len(rb) + sum(wb) / len(wb)) / shots,
    "algorithmic_bytes_per_shot": 3144,
    "sq": counters("flip_sq", "flip2_kernel")[-1:], "lds_l2": counters("flip_lds", "flip2_kernel")[-1:],
}
json.dump(flip, open(os.path.join(dst, "%s_flip_pmc.json" % tag), "w"), indent=1)

# ---- bp (scripts/bp_one.py c2 8388608 0.02: a warm-up launch of 8192 shots, then the measured one)
rb, wb = traffic(counters("bp_rd", "bp_kernel"), counters("bp_wr", "bp_kernel"))
ks = stats("bp_stats", "bp_kernel", "%s_bp_kernel_stats.csv" % tag)
log = open(os.path.join(src, "bp_stats.log")).read()
line = [l for l in log.splitlines() if l.startswith("cfg c2")][-1].split()
kv = dict(zip(line[2::2], line[3::2]))
iters, ms = int(kv["iters"]), float(kv["kernel_ms"])
alg = iters * 32 * 30000
bp = {
    "command": "scripts/profile_round.sh: rocprofv3 --pmc <one counter group per pass> -- python3 scripts/bp_one.py c2 8388608 0.02",
    "kernel": "bp_kernel<6,3>, 8 waves per 64-shot group, 512 groups", "shots": int(kv["shots"]), "shot_iterations": iters,
    "kernel_ms_hip_events_under_kernel_trace": ms,
    "algorithmic_bytes": alg, "algorithmic_GBs": alg / ms / 1e6,
    "hbm_read_bytes": rb[-1], "hbm_write_bytes": wb[-1], "traffic_bytes": rb[-1] + wb[-1],
    "traffic_over_algorithmic": (rb[-1] + wb[-1]) / alg, "traffic_GBs": (rb[-1] + wb[-1]) / ms / 1e6,
    "sq": counters("bp_sq", "bp_kernel")[-1:],
}
json.dump(bp, open(os.path.join(dst, "%s_bp_pmc.json" % tag), "w"), indent=1)
print(json.dumps({"flip_traffic_per_shot": flip["traffic_bytes_per_shot"], "flip_ms": flip["kernel_stats_avg_ms"],
                  "bp_traffic_over_alg": bp["traffic_over_algorithmic"], "bp_GBs": bp["algorithmic_GBs"]}))
