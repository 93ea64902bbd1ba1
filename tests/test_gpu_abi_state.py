"""GPU tests (-m gpu) of the library's parked requests and their caller-owned contexts (include/ultr_hip.h,
Conventions; ultr_pending_*): models on one host thread, stepped alternately -- each with a prefetching feed, so loss,
metric AND next-batch requests are parked and carried -- must train exactly as each does alone, whole steps
alternating or the phases of their steps interleaved; and a request parked under one context is invisible under
another."""
import numpy as np
import pytest
import torch

from tests.util import SyntheticData
from tests.test_gpu_step import settings, make, CASCADE, PBM

pytestmark = pytest.mark.gpu

SPECS = [
    ('DLA', 'hidden_layer_sizes=[64, 32]', 24, 10, 32, PBM),
    ('RegressionEM', 'hidden_layer_sizes=[48]', 17, 6, 20, PBM),
    ('PairDebias', 'hidden_layer_sizes=[64, 32]', 24, 8, 24, CASCADE),
    ('DLA', 'hidden_layer_sizes=[32]', 12, 20, 16, PBM),     # lists longer than 16: the per-launch plan
]


def build(spec, prefetch):
    algo, mhp, F, L, B, click = spec
    data = SyntheticData(150, L, F, seed=21, pad_frac=0.2)
    exp = settings(algo, '', mhp, Lmax=L, cutoff=L,
                   feed_hparams='click_model_json=%s,seed=5,prefetch=%s' % (click, 'true' if prefetch else 'false'))
    model, feed = make(algo, data, exp, B)
    rng = np.random.RandomState(4)
    return {'m': model, 'feed': feed, 'data': data, 'rng': rng, 'B': B, 'L': L, 'losses': []}


def uniforms(w):
    w['m'].metric_uniforms = torch.tensor(w['rng'].uniform(size=(w['B'], w['L'])).astype(np.float32)).cuda()


def whole_step(w):
    uniforms(w)
    fd, _ = w['feed'].get_batch(w['data'], check_validation=True)
    loss, _, summ = w['m'].step(None, fd, False)
    w['losses'].append((float(loss), summ.as_dict()))


def final_state(w):
    torch.cuda.synchronize()
    st = {'params': w['m'].model.params.clone()}
    st.update({k: v.clone() for k, v in w['m']._extra_state().items()})
    return st


def assert_same(a, b, la, lb):
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for (x, sx), (y, sy) in zip(la, lb):
        assert x == y or (np.isnan(x) and np.isnan(y))
        for k in sx:
            assert sx[k] == sy[k] or (np.isnan(sx[k]) and np.isnan(sy[k])), k


def test_models_stepped_alternately_on_one_thread_train_as_each_does_alone():
    solo = []
    for spec in SPECS:
        w = build(spec, prefetch=True)
        for _ in range(4):
            whole_step(w)
        solo.append((final_state(w), w['losses']))
        del w
    worlds = [build(spec, prefetch=True) for spec in SPECS]
    for _ in range(4):
        for w in worlds:
            whole_step(w)
    for w, (st, losses) in zip(worlds, solo):
        assert_same(final_state(w), st, w['losses'], losses)


@pytest.mark.parametrize('prefetch', [False, True])
def test_phases_of_steps_interleave(prefetch):
    """get_batch of every model, then step_local of every model, the collectives in reverse order, then every update:
    each model's requests (with prefetch=True also the next batch's assembly, parked during get_batch and carried by
    the model's own backward pass two phases later) live in the model's own context."""
    solo = []
    for spec in SPECS:
        w = build(spec, prefetch=prefetch)
        for _ in range(3):
            whole_step(w)
        solo.append((final_state(w), w['losses']))
    worlds = [build(spec, prefetch=prefetch) for spec in SPECS]
    for _ in range(3):
        feeds = []
        for w in worlds:
            uniforms(w)
            feeds.append(w['feed'].get_batch(w['data'], check_validation=True)[0])
        for w, fd in zip(worlds, feeds):
            w['m'].step_local(fd)
        for w in reversed(worlds):
            w['m'].step_reduce()
        outs = [w['m'].step_apply() for w in worlds]
        for w, (loss, _, summ) in zip(worlds, outs):
            w['losses'].append((float(loss), summ.as_dict()))
    for w, (st, losses) in zip(worlds, solo):
        assert_same(final_state(w), st, w['losses'], losses)


def test_a_request_is_seen_only_under_its_own_context():
    from ultra_amd import _lib as L
    lib = L.lib()
    a, b = L.PendingContext(), L.PendingContext()
    B, Lr = 8, 5
    scores = torch.randn(Lr, B, device='cuda')
    labels = (torch.rand(Lr, B, device='cuda') < 0.3).float()
    out = torch.zeros(8, device='cuda')
    topn = (L.ctypes.c_int * 1)(3)
    ws = torch.zeros(int(lib.ultr_metrics_workspace_floats(B, 1)), device="cuda")
    was = lib.ultr_pending_bind(a.handle)
    try:
        L.check(lib.ultr_metrics_deferred(L.ptr(scores), L.ptr(labels), None, 0, None, None, B, Lr, topn, 1, 2, 1.0,
                                          L.ptr(out), L.ptr(ws), L.current_stream()), 'ultr_metrics_deferred')
        assert a.count() == 1 and b.count() == 0
        b.bind()
        assert lib.ultr_metrics_cancel() == 0          # nothing parked HERE
        L.check(lib.ultr_metrics_flush(L.current_stream()))
        torch.cuda.synchronize()
        assert float(out.abs().sum()) == 0.0           # and nothing ran
        a.bind()
        L.check(lib.ultr_metrics_flush(L.current_stream()))
        torch.cuda.synchronize()
        assert a.count() == 0 and float(out.abs().sum()) > 0.0
        h = lib.ultr_pending_create()
        lib.ultr_pending_bind(h)
        lib.ultr_pending_destroy(h)                    # the bound context goes away: the thread's default takes over
        assert lib.ultr_pending_bind(None) is None
    finally:
        lib.ultr_pending_bind(was)
