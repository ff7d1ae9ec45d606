"""Counters of one kernel from a scripts/pmc_run.sh directory, as JSON: python scripts/pmc_kernel.py DIR KERNEL_SUBSTRING
HBM bytes as MI355X_MICROARCH.md prescribes for gfx950: reads = TCC_EA0_RDREQ x 128 B minus the 32-B requests counted
at their size, writes = 64-B requests x 64 + the others x 32."""
import csv, glob, json, os, sys
from collections import defaultdict


def kernel_counters(d, pat):
    acc, n = defaultdict(float), set()
    for f in glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                n.add((os.path.dirname(f), r["Dispatch_Id"]))
    return dict(acc)


def kernel_time(d, pat):
    for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Name"]:
                return {"calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6, "avg_ms": float(r["AverageNs"]) / 1e6}
    return None


def summarize(d, pat):
    c = kernel_counters(d, pat)
    out = {"kernel": pat, "time": kernel_time(d, pat), "counters": c}
    if "TCC_EA0_RDREQ_sum" in c:
        out["hbm_read_bytes"] = c["TCC_EA0_RDREQ_sum"] * 128 - c.get("TCC_EA0_RDREQ_32B_sum", 0) * 96
    if "TCC_EA0_WRREQ_sum" in c:
        out["hbm_write_bytes"] = c["TCC_EA0_WRREQ_64B_sum"] * 64 + (c["TCC_EA0_WRREQ_sum"] - c["TCC_EA0_WRREQ_64B_sum"]) * 32
    if "hbm_read_bytes" in out and "hbm_write_bytes" in out:
        out["hbm_traffic_bytes"] = out["hbm_read_bytes"] + out["hbm_write_bytes"]
    return out


if __name__ == "__main__":
    print(json.dumps(summarize(sys.argv[1], sys.argv[2]), indent=1))
