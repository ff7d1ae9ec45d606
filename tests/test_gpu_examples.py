"""The example mains (examples/css_decoding.rs:7-23, examples/random_linear_code.rs:5-17), run as the programs they
are: the C++ builds against the C ABI and the Python versions, their printed results compared with the library's
own calls and with the oracle.  examples/p_sweep.py (BASELINE.json configs[3]) likewise."""
import os
import subprocess
import sys

import numpy as np
import pytest

from ldpc_amd import CNT_FAILURES, CssCode, CssDecoder, FlipDecoder, LinearCode
from oracle import oracle as orc
from tests.gpu_util import oracle_code

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=300):
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    assert r.returncode == 0, (cmd, r.stdout[-2000:], r.stderr[-2000:])
    return r.stdout


def test_css_decoding_example_prints_the_failure_count():
    shots = 2000
    base = LinearCode.random_regular_code(20, 15, 3, 4, seed=1)
    css = CssCode.hypergraph_product(base, base)
    want = int(CssDecoder(css, 0.01, 100).mc_run(0.01, 2024, 0, shots)[CNT_FAILURES])
    exe = os.path.join(ROOT, "examples", "css_decoding")
    assert os.path.exists(exe), "examples/css_decoding is not built (make -C examples; __graft_entry__.build() does it)"
    assert int(_run([exe, str(shots)]).strip()) == want
    assert int(_run([sys.executable, "examples/css_decoding.py", str(shots)]).strip().splitlines()[-1]) == want
    # and the oracle's loop over the same shots (css_decoding.rs:14-22); float path: a shot or two may differ
    mats = [css.x_stabs_binary(), css.z_stabs_binary(), css.x_logicals_binary(), css.z_logicals_binary()]
    o = [orc.Code(css.len(), rows=mt.parity_check_matrix()) for mt in mats]
    ref = orc.mc_css_bp(o[0], o[1], o[2], o[3], 0.01, 100, 2024, 0, shots, nthreads=16)
    assert abs(want - int(ref[CNT_FAILURES])) <= 2, (want, ref)


def test_random_linear_code_example_prints_the_sampled_error():
    exe = os.path.join(ROOT, "examples", "random_linear_code")
    assert os.path.exists(exe), "examples/random_linear_code is not built"
    want = orc.words_to_pos(orc.sample_bsc_words(2024, 0, 0.2, 4), 4)
    got = _run([exe]).strip()
    assert got == "[" + ", ".join(str(x) for x in want) + "]", (got, want)
    got_py = _run([sys.executable, "examples/random_linear_code.py"]).strip()
    assert [int(x) for x in got_py.strip("[]").replace(",", " ").split()] == want, (got_py, want)


def test_p_sweep_example_matches_the_oracle_curve():
    # configs[3], flip decoder (integer: exact): 10 points p = 0.005 ... 0.05 on the C2 code, one process
    shots = 20000
    out = _run([sys.executable, "examples/p_sweep.py", "--shots", str(shots), "--decoder", "flip"], timeout=600)
    rows = [ln.split() for ln in out.splitlines() if ln and not ln.startswith("#")]
    assert len(rows) == 10
    code = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
    oc = oracle_code(code)
    for k, r in enumerate(rows):
        p = float(r[0])
        assert abs(p - 0.005 * (k + 1)) < 1e-9
        ref = oc.mc_flip(p, 2024 + k, 0, shots, nthreads=16)
        assert int(r[1]) == int(ref[CNT_FAILURES]), (p, r, ref)
