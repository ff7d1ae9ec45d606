#!/usr/bin/env python3
"""BASELINE.json configs[3]: logical-error-rate curve p = 0.005 ... 0.05 on the (3,6)-regular n = 10,000 code,
shots sharded over the GPUs of one node, failure counters reduced with ONE collective for the whole sweep.

  python examples/p_sweep.py [--shots 1000000] [--decoder bp|flip]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/p_sweep.py

Each point is the loop of examples/css_decoding.rs:14-21 for a classical code: sample, syndrome,
decode, count failures -- fused on the device.  The noise of a shot depends on (seed, global shot index) only, so
the curve does not depend on the number of GPUs."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

POINTS = (0.005, 0.01, 0.015, 0.02, 0.025, 0.03, 0.035, 0.04, 0.045, 0.05)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shots", type=int, default=1000000, help="shots per point, whole job")
    ap.add_argument("--decoder", choices=("bp", "flip"), default="bp")
    ap.add_argument("--seed", type=int, default=2024)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    import ldpc_amd
    from ldpc_amd import BpDecoder, FlipDecoder, LinearCode
    from ldpc_amd.dist import all_reduce_counters, shard_range

    ldpc_amd.set_device(local_rank)
    code = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
    lo, hi = shard_range(args.shots, rank, world)
    local = np.zeros((len(POINTS), ldpc_amd.N_COUNTERS), dtype=np.uint64)
    for k, p in enumerate(POINTS):
        dec = BpDecoder(code, p, 100) if args.decoder == "bp" else FlipDecoder(code)
        local[k] = dec.mc_run(p, args.seed + k, lo, hi - lo)
    total = all_reduce_counters(local)          # [points x 8] u64, one all-reduce (RCCL over xGMI)
    if rank == 0:
        print("# decoder=%s shots/point=%d gpus=%d" % (args.decoder, args.shots, world))
        print("# p  failures  logical_error_rate  95%_half_width  mean_work_per_shot")
        for p, c in zip(POINTS, total):
            n, f = int(c[ldpc_amd.CNT_SHOTS]), int(c[ldpc_amd.CNT_FAILURES])
            rate = f / n
            print("%.3f %9d %.6e %.2e %.2f" % (p, f, rate, 1.96 * (max(rate * (1 - rate), 1.0 / n) / n) ** 0.5,
                                              int(c[ldpc_amd.CNT_WORK]) / n))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
