#!/usr/bin/env python3
"""Headline benchmark: decoded shots/s of the Monte-Carlo flip-decoding path on the
(3,6)-regular n=10,000 code at BSC p=0.02 (BASELINE.json configs[1]).

A step = one batch of shots through the whole hot path on each GPU, in ONE fused
kernel: Philox noise sampling -> syndrome H*e -> serial flip decoding -> failure
test, with the sampled error word, its syndrome and the decoded word written to
HBM (2*ceil(n/64)*8 + ceil(m/64)*8 = 3144 bytes per shot: SURVEY.md 8(d) "flip,
batched I/O", the algorithmic bytes the roofline figure is priced on).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--shots-per-step B]

N > 1: one rank per GPU under torch.distributed.run -- started by the caller, or,
when RANK is not set, by this script itself before anything touches the GPU
(`python bench.py --gpus 8` works on its own); shots are sharded by global shot
index (weak scaling: B shots per GPU per step), no data-path collective; the u64
counters are all-reduced over RCCL after the timed region.  Rank 0 prints one
JSON line.

LDPC_BENCH_BACKEND=gloo rehearses the multi-rank path where there are fewer GPUs
than ranks; LDPC_BENCH_STUB=1 replaces the GPU step by a constant (CPU test of the
launch / reduce / print logic only, tests/test_bench_launch.py).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_BITS, N_CHECKS, DV, DC = 10000, 5000, 3, 6
P = 0.02
CODE_SEED, NOISE_SEED = 1, 2024
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def _timed_sample(run, threads, budget_s, cap):
    """run(n_shots, nthreads) -> counters, on a sample sized to about budget_s seconds of wall time (at least `threads`
    shots, at most `cap`).  Returns (shots, seconds, counters)."""
    run(threads, threads)                                      # warm the thread pool / page the tables in
    t0 = time.perf_counter()
    run(4 * threads, threads)
    per_shot = max(time.perf_counter() - t0, 1e-4) / (4 * threads)
    shots = int(max(threads, min(cap, budget_s / per_shot)))
    t0 = time.perf_counter()
    cnt = run(shots, threads)
    return shots, time.perf_counter() - t0, cnt


def _host_threads():
    from oracle import oracle as orc
    return max(1, min(16, orc.max_threads(), os.cpu_count() or 1))


def cpu_baseline(code, budget_s=9.0):
    """The oracle's restatement of FlipDecoder::decode as written in the reference
    (flip.rs:21-43: rescan from bit 0 and a full H*e_bit per flip), timed on this
    host on a bounded sample of the same workload: all host threads the box gives a GPU (16) and one thread
    (SURVEY.md 8(d))."""
    from oracle import oracle as orc
    ptr, idx = code.csr()
    oc = orc.Code(code.len(), ptr=ptr.astype(np.int64), idx=idx.astype(np.uint32))
    threads = _host_threads()
    shots, dt, cnt = _timed_sample(lambda n, t: oc.mc_flip(P, NOISE_SEED, 0, n, reference_style=True, nthreads=t), threads, budget_s, 200000)
    shots1, dt1, _ = _timed_sample(lambda n, t: oc.mc_flip(P, NOISE_SEED, 0, n, reference_style=True, nthreads=t), 1, 2.0, 20000)
    # the same decoder with incremental bookkeeping (not how the reference does it), for context
    t0 = time.perf_counter()
    fast_shots = 256 * threads
    oc.mc_flip(P, NOISE_SEED, 0, fast_shots, reference_style=False, nthreads=threads)
    dt_fast = time.perf_counter() - t0
    return {
        "value": shots / dt, "unit": "shots/s", "cores": threads, "kind": "port",
        "sample": "%d shots of the bench workload ((3,6) n=10000, p=0.02, noise_seed=%d), reference-style "
                  "restatement of flip.rs:21-43 in C, %d OpenMP threads over shots, %.1f s" % (shots, NOISE_SEED, threads, dt),
        "failures_in_sample": int(cnt[1]),
        "one_thread": {"value": shots1 / dt1, "unit": "shots/s", "cores": 1, "sample": "%d shots, %.1f s" % (shots1, dt1)},
        "incremental_variant_shots_per_s": fast_shots / dt_fast,
    }, cnt


def cpu_baseline_bp(code, p, what, budget_s=3.0):
    """The oracle's restatement of BpDecoder (belief_propagation.rs:23-30, 100-180: two f64 arrays rebuilt per shot,
    glibc tanh / ln_1p, the full stop test per iteration) on a bounded sample of the same Monte-Carlo workload."""
    from oracle import oracle as orc
    ptr, idx = code.csr()
    oc = orc.Code(code.len(), ptr=ptr.astype(np.int64), idx=idx.astype(np.uint32))
    threads = _host_threads()
    shots, dt, cnt = _timed_sample(lambda n, t: oc.mc_bp(p, 100, NOISE_SEED, 0, n, nthreads=t), threads, budget_s, 1 << 20)
    shots1, dt1, _ = _timed_sample(lambda n, t: oc.mc_bp(p, 100, NOISE_SEED, 0, n, nthreads=t), 1, 1.0, 1 << 16)
    return {"value": shots / dt, "unit": "shots/s", "cores": threads, "kind": "port",
            "sample": "%d shots of %s, restatement of belief_propagation.rs in C (f64, glibc), %d OpenMP threads over shots, %.1f s"
                      % (shots, what, threads, dt),
            "iterations_per_shot_in_sample": int(cnt[4]) / shots, "failures_in_sample": int(cnt[1]),
            "one_thread": {"value": shots1 / dt1, "unit": "shots/s", "cores": 1, "sample": "%d shots, %.1f s" % (shots1, dt1)}}


def cpu_baseline_css(css, p, budget_s=3.0):
    """examples/css_decoding.rs:14-21 as the oracle restates it (depolarizing sample, two BP decodes, has_stabilizer),
    on a bounded sample."""
    from oracle import oracle as orc
    mats = [css.x_stabs_binary(), css.z_stabs_binary(), css.x_logicals_binary(), css.z_logicals_binary()]
    o = [orc.Code(m.len(), rows=m.parity_check_matrix()) for m in mats]
    threads = _host_threads()
    run = lambda n, t: orc.mc_css_bp(o[0], o[1], o[2], o[3], p, 100, NOISE_SEED, 0, n, nthreads=t)
    shots, dt, cnt = _timed_sample(run, threads, budget_s, 1 << 20)
    shots1, dt1, _ = _timed_sample(run, 1, 1.0, 1 << 16)
    return {"value": shots / dt, "unit": "shots/s", "cores": threads, "kind": "port",
            "sample": "%d shots of configs[2] (css_decoding.rs loop body restated in C), %d OpenMP threads over shots, %.1f s" % (shots, threads, dt),
            "logical_failures_in_sample": int(cnt[1]),
            "one_thread": {"value": shots1 / dt1, "unit": "shots/s", "cores": 1, "sample": "%d shots, %.1f s" % (shots1, dt1)}}


FLIP_PMC = "profiles/r03_flip_pmc.json"
FLIP_PIPES = "profiles/r03_flip_pipes.json"
BP_PMC = "profiles/r03_bp_pmc.json"
C5_PMC = "profiles/r03_bp_c5_pmc.json"
CSS_PMC = "profiles/r03_css_pmc.json"
C1_PMC = "profiles/r03_bp_c1_pmc.json"
CLOCK_HZ = 2.4e9                                 # MI355X_MICROARCH.md: shader clock
N_SIMD, N_CU = 1024, 256
VALU_F64_PEAK = N_SIMD * CLOCK_HZ / 4 / 1e9      # G wave-instructions / s: one f64 wave-instruction per SIMD per 4 cycles
LDS_PEAK_GBS = N_CU * 256 * CLOCK_HZ / 1e9       # MI355X_MICROARCH.md LDS: 256 B / clk / CU (ds_read_b64 / b128; 128 for b32)


def _profile(path):
    try:
        with open(os.path.join(ROOT, path)) as f:
            return json.load(f)
    except Exception:
        return None


def replayed_ratio(path):
    """traffic / algorithmic bytes of a recorded rocprofv3 PMC run of the same workload (profiles/), or None."""
    d = _profile(path)
    try:
        return float(d["traffic_over_algorithmic"])
    except Exception:
        return None


def lds_object(path, per_launch=False):
    """LDS-array utilisation of the recorded PMC run of this workload (replayed): SQ_LDS_IDX_ACTIVE counts all LDS-array
    cycles, SQ_LDS_BANK_CONFLICT the extra cycles of conflicts (MI355X_MICROARCH.md, LDS); the array moves up to 256 B per
    cycle and CU, so `achieved` = busy fraction x that peak is an upper estimate of the bytes moved."""
    d = _profile(path)
    try:
        c = d["counters_per_launch"] if per_launch else d["counters"]
        cu_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * N_CU
        busy = c["SQ_LDS_IDX_ACTIVE"] / cu_cycles
        return {"array_busy_frac": busy, "bank_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0),
                "achieved": busy * LDS_PEAK_GBS, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                "source": "%s: SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 x %d CUs) (replayed, not a counter of this run)" % (path, N_CU)}
    except Exception:
        return None


def bp_roofline(alg_bytes, ms, pmc, kernel="", reference_bytes=None, shots=None):
    """alg_bytes: 32 * E per shot-iteration of message arithmetic ACTUALLY RUN (ldpc_last_bp_executed_iterations: the LDS
    form serves a shot's first iteration from a table and ends a run that is found to repeat with period two early;
    neither is priced).  reference_bytes: the same figure on the iterations the counters report, i.e. what
    belief_propagation.rs:23-30 runs for these shots.
    Forms that keep a shot's messages in LDS (bp_lds_kernel) are bound by f64 vector issue, not HBM: their object is
    priced in vector wave-instructions per second against one per SIMD per four cycles, with the instructions per shot
    replayed from the recorded PMC pass of the same workload; the HBM-equivalent rate of the 32 E bytes stays as a
    secondary field, and `traffic` is what HBM really sees."""
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    ratio = replayed_ratio(pmc)
    src = None if ratio is None else "%s: measured traffic / algorithmic ratio x this run's algorithmic bytes (replayed, not a counter of this run)" % pmc
    hbm = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    if reference_bytes is not None:
        hbm["frac_on_reference_iterations"] = reference_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    if kernel.startswith("bp_lds_kernel"):
        d = _profile(pmc)
        hybrid = "hybrid" in kernel
        out = {"bound": "valu_f64", "achieved": None, "peak": VALU_F64_PEAK, "unit": "G wave-instr/s", "frac": None,
               "traffic": None if ratio is None else ratio * alg_bytes, "traffic_source": src,
               "hbm_equivalent": hbm, "lds": lds_object(pmc),
               "note": ("hybrid LDS form: about two thirds of the 32*E bytes per shot-iteration of SURVEY.md 8(d) are LDS traffic and one "
                        "third L2 traffic (the workgroup's own piece of global memory); " if hybrid else
                        "LDS-resident form: the 32*E bytes per shot-iteration of SURVEY.md 8(d) are LDS traffic, HBM sees the graph and the "
                        "outputs; ") + "hbm_equivalent is that rate over the HBM peak on the iterations of message arithmetic that were run"}
        try:
            per_shot = d["counters"]["SQ_INSTS_VALU"] / d["shots"]
            if shots:
                rate = per_shot * shots / (ms * 1e-3) / 1e9
                out.update({"achieved": rate, "frac": rate / VALU_F64_PEAK, "valu_instructions_per_shot": per_shot,
                            "valu_source": "%s: SQ_INSTS_VALU / shots of the recorded pass x this run's shots/s (replayed)" % pmc})
        except Exception:
            pass
        return out
    out = {"bound": "hbm", "traffic": None if ratio is None else ratio * alg_bytes, "traffic_source": src}
    out.update(hbm)
    return out


def flip_binder():
    """What binds the headline kernel (for the LDS-resident flip kernel the HBM fraction is small by
    construction): the instruction stream of a pass, with the counters of the tracked pass replayed and their source named."""
    text = ("not HBM: per CU eight flip waves run the serial chains of 64 shots (a pass = ~95 instructions for 8 flips, one L2 "
            "row and three LDS round trips, ~1100 cycles) and eight service waves prepare / write out shots through the "
            "other LDS slots; both kinds of wave are saturated (7.97 of 8 flip groups live in every pass, service waves "
            "96 % busy: profiles/r03_flip_waves.txt), 16 waves and 80 slots are all a CU's workgroup and LDS allow, with "
            "the LDS array 65 % and the VALU pipes 81 % busy")
    d = _profile(FLIP_PIPES)
    try:
        return {"what": text, "valu_pipe_busy": d["valu_pipe_busy"], "valu_instructions_per_shot": d["valu_instructions_per_shot"],
                "source": FLIP_PIPES + " (replayed, not a counter of this run)"}
    except Exception:
        return {"what": text, "valu_pipe_busy": None, "source": None}


def recorded_traffic(shots_per_launch):
    """HBM-side bytes per launch, NOT measured in this run: the per-shot figure of the rocprofv3 PMC passes over this
    same command (fabric read + write requests of the flip kernel, kept in FLIP_PMC) times this run's shots per
    launch.  Returns (bytes, source label); (None, None) when no profile is on record."""
    prof = _profile(FLIP_PMC)
    try:
        return prof["traffic_bytes_per_shot"] * shots_per_launch, "%s x shots per launch (replayed, not a counter of this run)" % FLIP_PMC
    except Exception:
        return None, None


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run and relay rank 0's
    line.  Runs before this process touches torch.cuda / HIP (a process that holds the GPU must not exec).  The
    rendezvous port is torchrun's own choice (--standalone), on 127.0.0.1."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def bp_line(code, shots=1 << 23, no_cpu=False):
    """Secondary measurement: the f64 BP kernel on the same code and channel (Monte-Carlo mode,
    max 100 iterations).  Algorithmic bytes = 32*E per shot-iteration."""
    import ldpc_amd
    from ldpc_amd import BpDecoder
    dec = BpDecoder(code, P, 100)
    dec.mc_run(P, NOISE_SEED, 1 << 40, 8192)
    cnt = dec.mc_run(P, NOISE_SEED, 0, shots)
    ms, ran, kernel = ldpc_amd.last_kernel_ms(), ldpc_amd.last_bp_executed_iterations(), ldpc_amd.last_kernel_name()
    return {"workload": "same code and channel, BpDecoder f64, max 100 iterations, %d shots" % shots,
            "shots_per_s": shots / (ms * 1e-3), "kernel": kernel, "kernel_ms": ms,
            "iterations_per_shot": int(cnt[ldpc_amd.CNT_WORK]) / shots, "iterations_run_per_shot": ran / shots,
            "failures": int(cnt[ldpc_amd.CNT_FAILURES]),
            "roofline": bp_roofline(ran * 32 * code.num_edges(), ms, BP_PMC, kernel, int(cnt[ldpc_amd.CNT_WORK]) * 32 * code.num_edges(), shots),
            "cpu_baseline": None if no_cpu else cpu_baseline_bp(code, P, "the same workload ((3,6) n=10000, p=0.02, BP max 100 iterations)")}


def bp_c1_line(shots=1 << 22, no_cpu=False):
    """Secondary measurement: configs[0]'s code, (3,4)-regular n = 1200, BSC p = 0.02, BpDecoder f64 max 100 iterations:
    the LDS-resident BP form (a shot's 28.8 KB of messages in LDS, one workgroup per shot in flight)."""
    import ldpc_amd
    from ldpc_amd import BpDecoder, LinearCode
    code = LinearCode.random_regular_code(1200, 900, 3, 4, seed=CODE_SEED)
    dec = BpDecoder(code, P, 100)
    dec.mc_run(P, NOISE_SEED, 1 << 40, 8192)
    cnt = dec.mc_run(P, NOISE_SEED, 0, shots)
    ms, ran, kernel = ldpc_amd.last_kernel_ms(), ldpc_amd.last_bp_executed_iterations(), ldpc_amd.last_kernel_name()
    return {"workload": "configs[0] code: random (3,4)-regular LDPC n=1200 m=900, BSC p=0.02, BpDecoder f64, max 100 iterations, "
                        "%d shots" % shots,
            "shots_per_s": shots / (ms * 1e-3), "kernel": kernel, "kernel_ms": ms,
            "iterations_per_shot": int(cnt[ldpc_amd.CNT_WORK]) / shots, "iterations_run_per_shot": ran / shots,
            "failures": int(cnt[ldpc_amd.CNT_FAILURES]),
            "roofline": bp_roofline(ran * 32 * code.num_edges(), ms, C1_PMC, kernel,
                                    int(cnt[ldpc_amd.CNT_WORK]) * 32 * code.num_edges(), shots),
            "cpu_baseline": None if no_cpu else cpu_baseline_bp(code, P, "configs[0] ((3,4) n=1200, p=0.02, BP max 100 iterations)", budget_s=2.0)}


def css_line(shots=1 << 23, no_cpu=False):
    """Secondary measurement: configs[2], the CSS code of examples/css_decoding.rs (hypergraph product of a
    (3,4)-regular n = 20 code with itself: 625 qubits), depolarizing p = 0.01, CssDecoder<BpDecoder> max 100
    iterations, X and Z halves decoded independently.  Algorithmic bytes = 32 * 2100 per shot-iteration of a half."""
    import ldpc_amd
    from ldpc_amd import CssCode, CssDecoder, LinearCode
    base = LinearCode.random_regular_code(20, 15, 3, 4, seed=CODE_SEED)
    css = CssCode.hypergraph_product(base, base)
    edges = css.x_stabs_binary().num_edges()
    dec = CssDecoder(css, 0.01, 100)
    dec.mc_run(0.01, NOISE_SEED, 1 << 40, 4096)
    cnt = dec.mc_run(0.01, NOISE_SEED, 0, shots)
    ms, ran, kernel = ldpc_amd.last_kernel_ms(), ldpc_amd.last_bp_executed_iterations(), ldpc_amd.last_kernel_name()
    return {"workload": "configs[2]: CSS hypergraph-product code, 625 qubits (Hx, Hz 300 x 625, row weight 7), depolarizing "
                        "p=0.01, BP f64 max 100 iterations on each half, logical failure test, %d shots" % shots,
            "shots_per_s": shots / (ms * 1e-3), "kernel": kernel, "kernel_ms": ms,
            "iterations_per_shot_both_halves": int(cnt[ldpc_amd.CNT_WORK]) / shots,
            # a shot's first iteration comes from a table; runs that never converge (5 % of the shots) alternate between two
            # states, which is found out exactly and ends them early
            "iterations_run_per_shot_both_halves": ran / shots,
            "logical_failures": int(cnt[ldpc_amd.CNT_FAILURES]),
            "roofline": bp_roofline(ran * 32 * edges, ms, CSS_PMC, kernel, int(cnt[ldpc_amd.CNT_WORK]) * 32 * edges, shots),
            "cpu_baseline": None if no_cpu else cpu_baseline_css(css, 0.01)}


def bp_c5_line(shots=70656):
    """Secondary measurement: configs[4], (3,6)-regular n = 10^6 (3 GB of f64 messages per 64-shot group: the
    streamed BP form, one grid-wide kernel per phase).  Time = hipEvents around the whole host-driven loop."""
    import ldpc_amd
    from ldpc_amd import BpDecoder, LinearCode
    code = LinearCode.random_regular_code(1000000, 500000, 3, 6, seed=CODE_SEED)
    dec = BpDecoder(code, P, 100)
    dec.mc_run(P, NOISE_SEED, 1 << 40, 64)
    cnt = dec.mc_run(P, NOISE_SEED, 0, shots)
    ms = ldpc_amd.last_kernel_ms()
    assert ldpc_amd.last_bp_executed_iterations() == int(cnt[ldpc_amd.CNT_WORK])   # (the streamed form runs every iteration it reports)
    return {"workload": "configs[4]: random (3,6)-regular LDPC n=1000000 m=500000, BSC p=0.02, BpDecoder f64, max 100 "
                        "iterations, %d shots" % shots,
            "shots_per_s": shots / (ms * 1e-3), "kernels": "bps_iterate<6,3> + set-up / stop-test / harvest kernels",
            "loop_ms": ms, "iterations_per_shot": int(cnt[ldpc_amd.CNT_WORK]) / shots,
            "failures": int(cnt[ldpc_amd.CNT_FAILURES]),
            "roofline": bp_roofline(int(cnt[ldpc_amd.CNT_WORK]) * 32 * code.num_edges(), ms, C5_PMC)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shots-per-step", type=int, default=1 << 22, help="shots per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bp", action="store_true", help="skip the secondary BP measurement")
    ap.add_argument("--decoder", choices=("flip", "bp"), default="flip",
                    help="flip (default): the headline line.  bp: the same contract with the f64 BP decoder as the step "
                         "(Monte-Carlo mode, counters only), e.g. for its scaling over N GPUs")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        # no launcher above us: become one (nothing here has touched the GPU yet); a failing rank fails the run
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    # one rank per GPU; LDPC_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get("LDPC_BENCH_BACKEND", "nccl")
    stub = os.environ.get("LDPC_BENCH_STUB") == "1"   # CPU test of the launch / reduce / print logic: no GPU step
    local_dev = 0 if stub else local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(backend=backend)
    if not stub:
        torch.cuda.set_device(local_dev)

    import ldpc_amd
    from ldpc_amd import FlipDecoder, LinearCode
    from ldpc_amd.dist import all_reduce_counters

    B = args.shots_per_step
    wn, wm = ldpc_amd.nwords(N_BITS), ldpc_amd.nwords(N_CHECKS)
    bytes_per_shot = 2 * wn * 8 + wm * 8
    use_bp = args.decoder == "bp"
    code = dec = None
    if not stub:
        ldpc_amd.set_device(local_dev)
        code = LinearCode.random_regular_code(N_BITS, N_CHECKS, DV, DC, seed=CODE_SEED)
        dec = FlipDecoder(code)
        if not use_bp:
            d_err = torch.empty((B, wn), dtype=torch.int64, device="cuda")
            d_syn = torch.empty((B, wm), dtype=torch.int64, device="cuda")
            d_out = torch.empty((B, wn), dtype=torch.int64, device="cuda")
        else:
            from ldpc_amd import BpDecoder
            bp_dec = BpDecoder(code, P, 100)

    def step(s):
        first = (s * world + rank) * B        # global shot index: disjoint across ranks and steps
        if stub:
            c = np.zeros(ldpc_amd.N_COUNTERS, dtype=np.uint64)
            c[ldpc_amd.CNT_SHOTS] = B
            return c
        if use_bp:
            return bp_dec.mc_run(P, NOISE_SEED, first, B)
        return dec.mc_run(P, NOISE_SEED, first, B, d_err, d_syn, d_out)

    def barrier():
        if world > 1:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    for s in range(args.warmup):
        step(s)
    barrier()
    t0 = time.perf_counter()
    counters = np.zeros(ldpc_amd.N_COUNTERS, dtype=np.uint64)
    kernel_ms = []
    executed = 0            # BP step: iterations of message arithmetic actually run on this rank (rooflines are priced on them)
    for s in range(args.steps):
        counters += step(args.warmup + s)
        kernel_ms.append(1.0 if stub else ldpc_amd.last_kernel_ms())
        if use_bp and not stub:
            executed += ldpc_amd.last_bp_executed_iterations()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" and not stub else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    counters = all_reduce_counters(counters, device="cpu" if stub or backend != "nccl" else None)

    if rank == 0:
        traffic, traffic_source = recorded_traffic(B)
        total_shots = B * world * args.steps
        avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
        achieved = B * bytes_per_shot / avg_kernel_s / 1e9
        line = {
            "metric": "decoded shots/sec (whole node), (3,6)-reg n=10k @p=0.02; % HBM roofline",
            "value": total_shots / elapsed,
            "unit": "shots/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: random (3,6)-regular LDPC n=10000 m=5000 (code_seed=1), BSC p=0.02, "
                            "Philox noise_seed=2024, flip decoder, fused sample+syndrome+decode+failure test, "
                            "error/syndrome/decoded words recorded to HBM",
                "shots_per_gpu_per_step": B,
                "parallelism": "shots sharded over %d GPU(s) by global shot index; counters all-reduced (RCCL)" % world,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": "stub" if stub else ldpc_amd.last_kernel_name(), "kernel_ms": avg_kernel_s * 1e3,
                "algorithmic_bytes_per_shot": bytes_per_shot,
                "binder": flip_binder(),
                "lds": lds_object(FLIP_PMC, per_launch=True),
            },
            "backend": "none" if world == 1 else backend,
            "world_size_seen": dist.get_world_size() if world > 1 and dist.is_initialized() else 1,
            "counters": {k: int(v) for k, v in zip(ldpc_amd.COUNTER_NAMES, counters)},
        }
        if stub:
            line["data"] = "stub (no GPU step: launch / reduce / print logic only)"
            print(json.dumps(line), flush=True)
        else:
            finish_line(line, args, world, use_bp, code, counters, avg_kernel_s, executed)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def finish_line(line, args, world, use_bp, code, counters, avg_kernel_s, executed):
    """Rank 0: the N = 1 extras (cpu_baseline, secondary BP / C5 / CSS objects) or the BP step's own roofline, then print."""
    import ldpc_amd
    no_cpu = args.no_cpu_baseline
    if not no_cpu and world == 1 and not use_bp:   # reported at N = 1 only
        cb, _ = cpu_baseline(code)
        line["cpu_baseline"] = cb
    if not args.no_bp and world == 1 and not use_bp:
        line["bp"] = bp_line(code, no_cpu=no_cpu)
        line["bp_c5"] = bp_c5_line()
        line["css"] = css_line(no_cpu=no_cpu)
        line["bp_c1"] = bp_c1_line(no_cpu=no_cpu)
    if use_bp:
        # BP step: the kernel that ran names its own roofline; algorithmic bytes = 32 E per shot-iteration of message
        # arithmetic run on rank 0, per launch; the reference's count comes from the all-reduced counter
        E = code.num_edges()
        kernel = ldpc_amd.last_kernel_name()
        alg = executed / args.steps * 32.0 * E
        ref = float(counters[ldpc_amd.CNT_WORK]) / world / args.steps * 32.0 * E
        line["dtype"] = "f64"
        line["config"]["workload"] = ("configs[1] with the BP decoder: random (3,6)-regular LDPC n=10000 m=5000 (code_seed=1), "
                                      "BSC p=0.02, Philox noise_seed=2024, BpDecoder f64 max 100 iterations, fused "
                                      "sample+syndrome+decode+failure test, counters only")
        line["roofline"] = bp_roofline(alg, avg_kernel_s * 1e3, BP_PMC, kernel, ref, args.shots_per_step)
        line["roofline"].update({"kernel": kernel, "kernel_ms": avg_kernel_s * 1e3, "algorithmic_bytes_per_shot_iteration": 32 * E,
                                 "iterations_run_per_launch_rank0": executed / args.steps})
        if world == 1 and not no_cpu:
            line["cpu_baseline"] = cpu_baseline_bp(code, P, "the same workload ((3,6) n=10000, p=0.02, BP max 100 iterations)")
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
