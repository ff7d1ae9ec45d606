"""bench.py's launch contract, on CPU: `python bench.py --gpus N` with no launcher above it starts N ranks itself
(torch.distributed.run, gloo here), rank 0 prints ONE JSON line with n_gpus == N and the counters of all ranks, and a
failing rank fails the run.  The GPU step is stubbed (LDPC_BENCH_STUB=1): this tests the plumbing, not the kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, extra_env):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=300)


def test_bench_gpus_2_launches_two_ranks():
    r = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--shots-per-step", "1000"],
               {"LDPC_BENCH_BACKEND": "gloo", "LDPC_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["counters"]["shots"] == 2 * 3 * 1000          # both ranks' steps, all-reduced
    assert line["value"] > 0 and line["unit"] == "shots/s" and line["scaling"] == "weak"


def test_bench_single_rank_stub_line():
    r = _bench(["--steps", "2", "--warmup", "0", "--shots-per-step", "10"], {"LDPC_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["counters"]["shots"] == 20


def test_bench_fails_when_a_rank_fails():
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"LDPC_BENCH_BACKEND": "no-such-backend", "LDPC_BENCH_STUB": "1"})
    assert r.returncode != 0
