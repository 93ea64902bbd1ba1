"""GPU test (-m gpu) of ultr_metrics_static, the entry point that allocates nothing (include/ultr_hip.h): through the
C ABI, against ultr_metrics (caller-owned workspace) on the same inputs, bit for bit, on two calls in a row -- the
second with a smaller batch, so that what the first call left in the library's device array must not leak into it.
Collected last (test_zz_*): the round's end runs pytest -x."""
import ctypes

import numpy as np
import pytest
import torch

from ultra_amd import _lib

pytestmark = pytest.mark.gpu

TOPN = [1, 3, 5, 10]


def _inputs(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    scores = torch.randn((L, B), generator=g)
    scores[1] = scores[0]  # ties: the uniforms decide
    labels = torch.randint(0, 5, (L, B), generator=g).float()
    docids = torch.randint(0, 50, (L, B), generator=g, dtype=torch.int32)  # == 49 marks a pad row
    lw = torch.rand(B, generator=g)
    u = torch.rand((B, L), generator=g)
    return [t.cuda() for t in (scores, labels, docids, lw, u)]


def _call(static, B, L, seed, ws=None):
    lib = _lib.lib()
    scores, labels, docids, lw, u = _inputs(B, L, seed)
    topn = (ctypes.c_int * len(TOPN))(*TOPN)
    out = torch.full((3 * len(TOPN) * 2,), -7.0, device='cuda')
    head = (_lib.ptr(scores), _lib.ptr(labels), _lib.ptr(docids), 49, _lib.ptr(lw), _lib.ptr(u), B, L, topn, len(TOPN),
            7, 4.0, _lib.ptr(out))
    if static:
        rc = lib.ultr_metrics_static(*head, _lib.current_stream())
    else:
        rc = lib.ultr_metrics(*head, _lib.ptr(ws), _lib.current_stream())
    _lib.check(rc, 'ultr_metrics')
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize('L', [10, 100])
def test_static_workspace_metrics_equal_ultr_metrics_twice_in_a_row(L):
    lib = _lib.lib()
    for B, seed in ((777, 1), (40, 2)):  # the second call is the smaller one
        ws = torch.zeros(lib.ultr_metrics_workspace_floats(B, len(TOPN)), device='cuda')
        want = _call(False, B, L, seed, ws)
        got = _call(True, B, L, seed)
        assert np.isfinite(want).all() and (want != -7.0).all()
        assert np.array_equal(got, want), (B, got, want)


def test_static_workspace_refuses_what_does_not_fit():
    lib = _lib.lib()
    cap = lib.ultr_metrics_static_max_floats()
    B = 8
    while lib.ultr_metrics_workspace_floats(B, len(TOPN)) <= cap:
        B *= 2
    with pytest.raises(_lib.UltrError, match='static array holds'):
        _call(True, B, 4, 3)


def test_static_workspace_metrics_inside_a_captured_graph():
    """After the per-device set-up call the entry point can be captured into a hipGraph (it allocates nothing and looks
    nothing up); two replays on new inputs in the same buffers give what ultr_metrics gives."""
    lib = _lib.lib()
    B, L = 300, 10
    _lib.check(lib.ultr_metrics_static_prepare(), 'ultr_metrics_static_prepare')
    bufs = _inputs(B, L, 10)
    topn = (ctypes.c_int * len(TOPN))(*TOPN)
    out = torch.zeros(3 * len(TOPN) * 2, device='cuda')
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            _lib.check(lib.ultr_metrics_static(_lib.ptr(bufs[0]), _lib.ptr(bufs[1]), _lib.ptr(bufs[2]), 49, _lib.ptr(bufs[3]),
                                               _lib.ptr(bufs[4]), B, L, topn, len(TOPN), 7, 4.0, _lib.ptr(out),
                                               _lib.current_stream()), 'ultr_metrics_static (captured)')
    ws = torch.zeros(lib.ultr_metrics_workspace_floats(B, len(TOPN)), device='cuda')
    for seed in (11, 12):
        for dst, src in zip(bufs, _inputs(B, L, seed)):
            dst.copy_(src)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        got = out.cpu().numpy().copy()
        assert np.array_equal(got, _call(False, B, L, seed, ws)), seed
