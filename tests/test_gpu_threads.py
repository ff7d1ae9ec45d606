"""Handles from several host threads (threading): calls on one handle are serialized by its mutex, calls on
different handles run side by side; either way the counters of disjoint shot ranges add up to the single call's."""
import threading

import numpy as np
import pytest

from ldpc_amd import BpDecoder, CssCode, CssDecoder, FlipDecoder, LinearCode

pytestmark = pytest.mark.gpu


def _threads(fns):
    out, err = [None] * len(fns), []

    def run(i, fn):
        try:
            out[i] = fn()
        except Exception as e:      # noqa: BLE001 - reported below
            err.append(e)

    ts = [threading.Thread(target=run, args=(i, fn)) for i, fn in enumerate(fns)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not err, err
    return out


def test_one_handle_many_threads_and_many_handles():
    code = LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)
    p, shots, parts = 0.05, 24000, 6
    step = shots // parts
    for make in (lambda: BpDecoder(code, p, 30), lambda: FlipDecoder(code)):
        whole = make().mc_run(p, 11, 0, shots)
        shared = make()
        same = _threads([lambda k=k: shared.mc_run(p, 11, k * step, step) for k in range(parts)])
        assert (np.sum(same, axis=0) == whole).all()
        own = _threads([lambda k=k: make().mc_run(p, 11, k * step, step) for k in range(parts)])
        assert (np.sum(own, axis=0) == whole).all()


def test_css_decoder_from_threads():
    base = LinearCode.random_regular_code(20, 15, 3, 4, seed=1)
    css = CssCode.hypergraph_product(base, base)
    dec = CssDecoder(css, 0.03, 40)
    whole = dec.mc_run(0.03, 5, 0, 8000)
    parts = _threads([lambda k=k: dec.mc_run(0.03, 5, k * 2000, 2000) for k in range(4)])
    assert (np.sum(parts, axis=0) == whole).all()
    # single-shot calls from threads, against the same calls in turn
    sx = [[0, 5], [3], [], [1, 2, 7]]
    sz = [[1], [], [4, 9], [0]]
    turn = [dec.correction_for(a, b) for a, b in zip(sx, sz)]
    side = _threads([lambda a=a, b=b: dec.correction_for(a, b) for a, b in zip(sx, sz)])
    assert turn == side
