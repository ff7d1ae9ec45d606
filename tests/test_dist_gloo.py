"""N > 1 path on CPU: world_size-2 gloo processes shard a Monte-Carlo run by global
shot index with ldpc_amd.dist and all-reduce the counters.  The per-rank `run`
callback is the CPU oracle here (no GPU in this container); on the GPU box the
same ldpc_amd.dist code wraps FlipDecoder.mc_run (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ldpc_amd import LinearCode
    from ldpc_amd.dist import all_reduce_counters, run_sharded, shard_range
    from oracle import oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    code = LinearCode.random_regular_code(120, 90, 3, 4, seed=1)
    ptr, idx = code.csr()
    oc = orc.Code(120, ptr=ptr.astype(np.int64), idx=idx.astype(np.uint32))
    reduced = run_sharded(lambda first, n: oc.mc_flip(0.03, 11, first, n), total, first_shot=5)
    # a whole p-sweep reduced with ONE collective: [points x 8]
    lo, hi = shard_range(total, rank, world)
    sweep = np.stack([oc.mc_flip(p, 11, 5 + lo, hi - lo) for p in (0.01, 0.03)])
    sweep = all_reduce_counters(sweep)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.concatenate([reduced[None, :], sweep]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 1001), (3, 100)])
def test_sharded_counters_equal_single_process(tmp_path, world, total):
    from ldpc_amd import LinearCode
    from oracle import oracle as orc

    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    code = LinearCode.random_regular_code(120, 90, 3, 4, seed=1)
    ptr, idx = code.csr()
    oc = orc.Code(120, ptr=ptr.astype(np.int64), idx=idx.astype(np.uint32))
    want = oc.mc_flip(0.03, 11, 5, total)
    want01 = oc.mc_flip(0.01, 11, 5, total)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "r%d.npy" % r))
        assert (got[0] == want).all()
        assert (got[1] == want01).all() and (got[2] == want).all()


def test_shard_range_partitions():
    from ldpc_amd.dist import shard_range
    for total in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
