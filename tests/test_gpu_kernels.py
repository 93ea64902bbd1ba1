"""GPU parity tests (-m gpu): every HIP kernel, called through the C ABI, against the CPU oracle.

Bars (north_star): clicks, accepted lists, sorted permutations bit-exact under the same uniform draws;
losses / gradients / propensities within 1e-5 relative in fp32 (tensor-wise L2-relative error: the
oracle is float64, the kernels are exact fp32 -- fmaf-chain MFMA -- so the bound is fp32 rounding).
"""
import ctypes
import os
import random

import numpy as np
import pytest
import torch

from oracle import ultr_oracle as O
from tests.util import load_json, SyntheticData, relerr

pytestmark = pytest.mark.gpu

RTOL = 1e-5  # the fp32 parity bound of BASELINE.json's north_star


@pytest.fixture(scope='module')
def L():
    from ultra_amd import _lib
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return _lib


_KEEP = []  # device tensors handed to the C ABI as raw pointers must outlive the (asynchronous) launch


@pytest.fixture(autouse=True)
def _release_after_test():
    yield
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    del _KEEP[:]


def keep(t):
    _KEEP.append(t)
    return t


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return keep(t.cuda())


def stream(L):
    return L.current_stream()


# ------------------------------------------------------------------------------------------------
# click kernels
# ------------------------------------------------------------------------------------------------
def _device_click_model(desc, Ls):
    """(kind, exam tensor, n_exam, click_prob tensor) as the feeds upload them (click_models.device_tables)."""
    from ultra_amd.utils import click_models as cm
    if desc['model_name'] == 'oracle':
        return 2, dev([1.0], torch.float64), 1, dev([1.0], torch.float64)
    d = {k: desc[k] for k in ('model_name', 'eta', 'click_prob', 'exam_prob')}
    if desc.get('extra') is not None:
        d['extra'] = desc['extra']
    kind, exam, n_exam, cp = cm.loadModelFromJson(d).device_tables(Ls)
    return kind, dev(exam, torch.float64), n_exam, dev(cp, torch.float64)


def test_click_kernels_bit_exact_on_reference_goldens(L):
    """Every click model of the reference (PBM, cascade, user browsing, trust bias, click matrix) on the clicks the
    reference module itself returned for the same labels and uniform draws."""
    g = load_json('click_goldens.json')
    assert len(g['models']) == 8
    for model in g['models']:
        for case in model['cases']:
            n = len(case['labels'])
            kind, exam, n_exam, cp = _device_click_model(model, n)
            lab, u = dev([case['labels']], torch.float32), dev([case['uniforms']], torch.float64)
            out = torch.empty((1, n), dtype=torch.float32, device='cuda')
            L.check(L.lib().ultr_click_sample(kind, L.ptr(lab), 1, n, L.ptr(exam), n_exam, L.ptr(cp), cp.numel(), L.ptr(u),
                                              0, 0, L.ptr(out), stream(L)), 'click')
            assert out.cpu().numpy()[0].tolist() == case['clicks'], (model['name'], case['seed'])
            if kind <= 1:  # the two original entry points
                fn = L.lib().ultr_click_cascade if kind == 1 else L.lib().ultr_click_pbm
                out2 = torch.empty_like(out)
                L.check(fn(L.ptr(lab), 1, n, L.ptr(exam), n_exam, L.ptr(cp), cp.numel(), L.ptr(u), 0, 0, L.ptr(out2),
                           stream(L)), 'click')
                assert torch.equal(out, out2)


@pytest.mark.parametrize('model_file', ['pbm_0.1_1.0_4_1.0.json', 'cascade_0.1_1.0_4_1.0.json', 'ubm_0.1_1_4_1.0.json'])
def test_click_kernel_long_lists(L, model_file):
    """Lists longer than 16 take the thread-per-list kernel (shorter ones the lane-per-element one): supplied uniforms
    against the oracle, and the Philox stream of a long list against its restatement."""
    g = load_json(model_file)
    rng = np.random.RandomState(29)
    n, Ls = 1500, 24
    labels = rng.choice(6, size=(n, Ls)).astype(np.float32)
    u = rng.random_sample((n, Ls)) * 0.8
    kind, exam, n_exam, cp = _device_click_model(g, Ls)
    out = torch.empty((n, Ls), dtype=torch.float32, device='cuda')
    L.check(L.lib().ultr_click_sample(kind, L.ptr(dev(labels)), n, Ls, L.ptr(exam), n_exam, L.ptr(cp), cp.numel(),
                                      L.ptr(dev(u)), 0, 0, L.ptr(out), stream(L)), 'click')
    want = np.array([O.sample_clicks_one_list(g['model_name'], g['click_prob'], g['exam_prob'], list(labels[i]), u[i])
                     for i in range(n)], dtype=np.float32)
    assert np.array_equal(out.cpu().numpy(), want)
    L.check(L.lib().ultr_click_sample(kind, L.ptr(dev(labels)), n, Ls, L.ptr(exam), n_exam, L.ptr(cp), cp.numel(), None,
                                      0xFEED, 2, L.ptr(out), stream(L)), 'click')
    got = out.cpu().numpy()
    for i in (0, 7, n - 1):
        ui = O.philox_uniforms(0xFEED, 2, i, 0, Ls)
        assert got[i].tolist() == [float(c) for c in O.sample_clicks_one_list(
            g['model_name'], g['click_prob'], g['exam_prob'], list(labels[i]), ui)]


@pytest.mark.parametrize('model_file', ['ubm_0.1_1_4_1.0.json', 'trust_bias.json', 'click_matrix_tiangong.json'])
def test_click_kernel_bulk_other_models(L, model_file):
    g = load_json(model_file)
    rng = np.random.RandomState(13)
    n, Ls = 4000, 10 if 'matrix' in model_file else 13   # 13 > 10: ranks beyond the examination tables
    n_grades = len(g['click_prob'])
    labels = rng.choice(n_grades if 'matrix' in model_file else n_grades + 1, size=(n, Ls)).astype(np.float32)
    labels[::17, 0] = -1.0
    u = rng.random_sample((n, Ls)) * 0.6
    kind, exam, n_exam, cp = _device_click_model(g, Ls)
    out = torch.empty((n, Ls), dtype=torch.float32, device='cuda')
    L.check(L.lib().ultr_click_sample(kind, L.ptr(dev(labels)), n, Ls, L.ptr(exam), n_exam, L.ptr(cp), cp.numel(),
                                      L.ptr(dev(u)), 0, 0, L.ptr(out), stream(L)), 'click')
    want = np.array([O.sample_clicks_one_list(g['model_name'], g['click_prob'], g['exam_prob'], list(labels[i]), u[i],
                                              g.get('extra')) for i in range(n)], dtype=np.float32)
    assert np.array_equal(out.cpu().numpy(), want) and 0.02 < want.mean() < 0.7


@pytest.mark.parametrize('cascade', [False, True])
def test_click_kernels_bulk_and_philox(L, cascade):
    g = load_json('cascade_0.1_1.0_4_1.0.json' if cascade else 'pbm_0.1_1.0_4_1.0.json')
    rng = np.random.RandomState(11)
    n, Ls = 5000, 10
    labels = rng.choice(6, size=(n, Ls)).astype(np.float32) - 0.0
    labels[::17, 0] = -1.0
    u = rng.random_sample((n, Ls))
    # make some draws sit exactly ON the threshold: u < p must be false there
    for i in range(0, n, 50):
        u[i, 3] = g['exam_prob'][3] * g['click_prob'][min(int(max(labels[i, 3], 0)), 4)]
    exam, cp = dev(g['exam_prob'], torch.float64), dev(g['click_prob'], torch.float64)
    fn = L.lib().ultr_click_cascade if cascade else L.lib().ultr_click_pbm
    out = torch.empty((n, Ls), dtype=torch.float32, device='cuda')
    L.check(fn(L.ptr(dev(labels)), n, Ls, L.ptr(exam), 10, L.ptr(cp), 5, L.ptr(dev(u)), 0, 0, L.ptr(out), stream(L)), 'click')
    want = np.array([O.sample_clicks_one_list(g['model_name'], g['click_prob'], g['exam_prob'], list(labels[i]), u[i])
                     for i in range(n)], dtype=np.float32)
    assert np.array_equal(out.cpu().numpy(), want)
    # Philox mode: the device stream equals the oracle's restatement of it, draw for draw
    seed, step = 0x1234567890ABCDEF, 7
    L.check(fn(L.ptr(dev(labels)), n, Ls, L.ptr(exam), 10, L.ptr(cp), 5, None, seed, step, L.ptr(out), stream(L)), 'click')
    got = out.cpu().numpy()
    for i in list(range(0, 40)) + [n - 1]:
        ui = O.philox_uniforms(seed, step, i, 0, Ls)
        assert got[i].tolist() == [float(c) for c in O.sample_clicks_one_list(
            g['model_name'], g['click_prob'], g['exam_prob'], list(labels[i]), ui)]
    rate = got.mean()
    assert 0.02 < rate < 0.6
    # empty input is a no-op
    assert fn(L.ptr(dev(labels)), 0, Ls, L.ptr(exam), 10, L.ptr(cp), 5, None, 0, 0, L.ptr(out), stream(L)) == 0


@pytest.mark.parametrize('model_file,Ls', [('pbm_0.1_1.0_4_1.0.json', 10), ('cascade_0.1_1.0_4_1.0.json', 10),
                                           ('ubm_0.1_1_4_1.0.json', 13), ('pbm_0.1_1.0_4_1.0.json', 16),
                                           ('cascade_0.1_1.0_4_1.0.json', 7), ('pbm_0.1_1.0_4_1.0.json', 1)])
def test_click_kernel_many_lists(L, model_file, Ls):
    """From 32768 lists on (Philox draws, L <= 16) a lane owns a whole list: the same clicks, bit for bit, as the
    lane-per-element kernel gives for a prefix of the lists, and as the oracle's restatement of the stream for the last
    ones (a ragged last tile: n is neither a multiple of 256 nor n L of 4)."""
    g = load_json(model_file)
    rng = np.random.RandomState(Ls)
    n, n_small = 40001 + (Ls & 1), 20000
    labels = rng.choice(6, size=(n, Ls)).astype(np.float32)
    labels[::17, 0] = -1.0
    kind, exam, n_exam, cp = _device_click_model(g, Ls)
    lab = dev(labels)
    out = torch.full((n, Ls), -1.0, dtype=torch.float32, device='cuda')
    small = torch.empty((n_small, Ls), dtype=torch.float32, device='cuda')
    seed, step = 0xABCDEF0123456789, 11
    L.check(L.lib().ultr_click_sample(kind, L.ptr(lab), n, Ls, L.ptr(exam), n_exam, L.ptr(cp), cp.numel(), None, seed, step,
                                      L.ptr(out), stream(L)), 'click')
    L.check(L.lib().ultr_click_sample(kind, L.ptr(lab), n_small, Ls, L.ptr(exam), n_exam, L.ptr(cp), cp.numel(), None, seed,
                                      step, L.ptr(small), stream(L)), 'click')
    got = out.cpu().numpy()
    assert np.array_equal(got[:n_small], small.cpu().numpy())
    assert set(np.unique(got).tolist()) <= {0.0, 1.0} and 0.02 < got.mean() < 0.7
    for i in (n_small, n - 300, n - 2, n - 1):
        ui = O.philox_uniforms(seed, step, i, 0, Ls)
        assert got[i].tolist() == [float(c) for c in O.sample_clicks_one_list(
            g['model_name'], g['click_prob'], g['exam_prob'], list(labels[i]), ui)]


# ------------------------------------------------------------------------------------------------
# batch assembly
# ------------------------------------------------------------------------------------------------
def _assemble(L, data, B, Ls, model, mode, seed=0, step=0, slot_base=0, u=None, slot_start=0, check=1,
              step_ctr=None, out=None, max_attempts=100):
    il, lab = dev(data.initial_list, torch.int32), dev(data.labels, torch.float32)
    kind, exam, n_exam, cp = _device_click_model(model, Ls)
    if out is None:
        out = (torch.full((Ls, B), -7, dtype=torch.int32, device='cuda'),
               torch.full((Ls, B), -7.0, dtype=torch.float32, device='cuda'),
               torch.full((B,), -7, dtype=torch.int32, device='cuda'),
               torch.zeros(L.ULTR_FEED_INFO_INTS, dtype=torch.int32, device='cuda'))
    ud = None if u is None else dev(u, torch.float64)
    n_cand = 0 if u is None else u.shape[0]
    n_rows = data.features.shape[0]
    L.check(L.lib().ultr_batch_assemble(
        L.ptr(il), L.ptr(lab), il.shape[0], il.shape[1], Ls, B, kind, L.ptr(exam), n_exam, L.ptr(cp), cp.numel(),
        mode, seed, step, L.ptr(step_ctr), slot_base, L.ptr(ud), n_cand, slot_start, check, max_attempts, n_rows,
        L.ptr(out[0]), L.ptr(out[1]), L.ptr(out[2]), L.ptr(out[3]), stream(L)), 'assemble')
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize('model_file', ['pbm_0.1_1.0_4_1.0.json', 'cascade_0.1_1.0_4_1.0.json', 'ubm_0.1_1_4_1.0.json',
                                        'trust_bias.json', 'click_matrix_tiangong.json'])
def test_batch_assemble_sequential_bit_exact(L, model_file):
    model = load_json(model_file)
    data = SyntheticData(300, 12, 8, seed=21, pad_frac=0.3)
    if 'matrix' in model_file:
        data.labels = np.minimum(data.labels, len(model['click_prob']) - 1)  # grades beyond the matrix raise in the reference
    Ls, B = 10, 256
    rng = np.random.RandomState(5)
    n_cand = 3000  # > 1024: several rounds of the single-workgroup loop
    u = rng.random_sample((n_cand, 1 + Ls))
    docids, clicks, idx, info = _assemble(L, data, B, Ls, model, 1, u=u)
    want_idx, want_clicks, want_n = O.get_batch_sequential(
        data.initial_list, data.labels, Ls, B, model['model_name'], model['click_prob'], model['exam_prob'],
        u.reshape(-1), True, model.get('extra'))
    assert info.cpu().tolist()[:2] == [want_n, B]
    assert idx.cpu().tolist() == want_idx
    assert np.array_equal(clicks.cpu().numpy().T, np.array(want_clicks, dtype=np.float32))
    il = data.initial_list[want_idx][:, :Ls]
    assert np.array_equal(docids.cpu().numpy().T, np.where(il < 0, data.features.shape[0], il))
    # shortfall: too few candidates -> partial fill, then a second call continues from slot_start
    out = _assemble(L, data, B, Ls, model, 1, u=u[:200])
    consumed, filled = out[3].cpu().tolist()[:2]
    assert consumed == 200 and 0 < filled < B
    out = _assemble(L, data, B, Ls, model, 1, u=u[200:], slot_start=filled, out=out)
    assert out[3].cpu().tolist()[:2] == [want_n - 200, B] and out[2].cpu().tolist() == want_idx


@pytest.mark.parametrize('model_file,Ls', [('ubm_0.1_1_4_1.0.json', 10), ('ubm_0.1_1_4_1.0.json', 12),
                                           ('trust_bias.json', 10), ('ubm_0.1_1_4_1.0.json', 20)])
def test_batch_assemble_philox_other_models(L, model_file, Ls):
    """The wave-per-slot assembly with the models whose click probability is not exam[rank] * click_prob[grade]: the
    user browsing model resolves its ranks in order inside the wave (every attempt of a round keeps its own last click)."""
    model = load_json(model_file)
    data = SyntheticData(400, 24, 8, seed=23, pad_frac=0.2)
    B, seed, step = 300, 0xABCDEF12345, 5
    docids, clicks, idx, info = _assemble(L, data, B, Ls, model, 0, seed=seed, step=step)
    w_idx, w_clicks, w_att = O.get_batch_slots(data.initial_list, data.labels, Ls, 48, model['model_name'],
                                               model['click_prob'], model['exam_prob'], seed, step, extra=model.get('extra'))
    assert idx.cpu().tolist()[:48] == w_idx
    assert np.array_equal(clicks.cpu().numpy().T[:48], np.array(w_clicks, dtype=np.float32))
    assert (clicks.sum(dim=0) > 0).all() and max(w_att) > 1


def test_batch_assemble_philox_slots_bit_exact_and_dp_invariant(L):
    model = load_json('pbm_0.1_1.0_4_1.0.json')
    data = SyntheticData(500, 10, 8, seed=22, pad_frac=0.2)
    Ls, B, seed, step = 10, 512, 0xC0FFEE1234, 3
    docids, clicks, idx, info = _assemble(L, data, B, Ls, model, 0, seed=seed, step=step)
    w_idx, w_clicks, w_att = O.get_batch_slots(data.initial_list, data.labels, Ls, 64, model['model_name'],
                                               model['click_prob'], model['exam_prob'], seed, step)
    assert idx.cpu().tolist()[:64] == w_idx
    assert np.array_equal(clicks.cpu().numpy().T[:64], np.array(w_clicks, dtype=np.float32))
    assert (clicks.sum(dim=0) > 0).all()
    assert sum(info.cpu().tolist()[64:64 + 32 * 16:32]) == sum(O.get_batch_slots(
        data.initial_list, data.labels, Ls, B, model['model_name'], model['click_prob'], model['exam_prob'], seed, step)[2])  # attempts
    # two "ranks" with B/2 lists each reproduce the one-big-batch run bit for bit
    a = _assemble(L, data, B // 2, Ls, model, 0, seed=seed, step=step, slot_base=0)
    b = _assemble(L, data, B // 2, Ls, model, 0, seed=seed, step=step, slot_base=B // 2)
    assert torch.equal(torch.cat([a[2], b[2]]), idx) and torch.equal(torch.cat([a[1], b[1]], dim=1), clicks)
    # device-resident step counter: added to `step`, bumped once per call
    ctr = torch.tensor([step], dtype=torch.int32, device='cuda')
    c = _assemble(L, data, B, Ls, model, 0, seed=seed, step=0, step_ctr=ctr)
    assert torch.equal(c[2], idx) and ctr.item() == step + 1
    # oracle mode feeds the labels themselves, lists without a relevant doc in the top L are rejected
    o = _assemble(L, data, 128, Ls, {'model_name': 'oracle', 'exam_prob': [1.0], 'click_prob': [1.0]}, 0, seed=1)
    got_idx = o[2].cpu().numpy()
    lab = np.where(data.initial_list[got_idx][:, :Ls] < 0, 0, data.labels[got_idx][:, :Ls])
    assert np.array_equal(o[1].cpu().numpy().T, lab.astype(np.float32)) and (lab.sum(1) > 0).all()
    # error behaviour: lists shorter than rank_list_size (ValueError in the reference feed)
    il = dev(data.initial_list, torch.int32)
    rc = L.lib().ultr_batch_assemble(L.ptr(il), L.ptr(il), 500, 10, 11, 4, 0, L.ptr(il), 1, L.ptr(il), 1, 0, 0, 0, None,
                                     0, None, 0, 0, 1, 100, 1, L.ptr(il), L.ptr(il), L.ptr(il), L.ptr(il), None)
    assert rc == -1 and b'no less than the required list size' in L.lib().ultr_last_error()


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
def _scores_clicks(rng, B, Ls):
    scores = rng.normal(scale=1.5, size=(B, Ls))
    clicks = (rng.uniform(size=(B, Ls)) < 0.25).astype(np.float64)
    clicks[clicks.sum(1) == 0, rng.randint(0, Ls)] = 1.0
    return scores, clicks


def _loss_ws(L, B, Ls):
    return keep(torch.zeros(L.lib().ultr_loss_workspace_floats(B, Ls), dtype=torch.float32, device='cuda'))


# L > 16: list per thread, capped grid; B >= 32768 with L <= 16 and the default softmax pair: one lane per list
@pytest.mark.parametrize('B,Ls', [(256, 10), (1000, 9), (3, 1), (5000, 16), (40000, 10), (300, 24), (70000, 20),
                                  (33001, 7), (140000, 16), (32768, 1)])
@pytest.mark.parametrize('loss_func,prob,maxw', [(0, 0, -1.0), (1, 0, -1.0), (0, 1, 3.0), (0, 0, 3.0)])
def test_dla_loss_kernel(L, B, Ls, loss_func, prob, maxw):
    rng = np.random.RandomState(B + Ls)
    scores, clicks = _scores_clicks(rng, B, Ls)
    W, b = rng.uniform(-0.7, 0.7, size=(Ls, 1)), rng.uniform(-0.5, 0.5, size=(1,))
    ref = O.dla_losses(scores, clicks, W, b, 'sigmoid_loss' if loss_func else 'softmax_loss',
                       'sigmoid' if prob else 'softmax', maxw)
    s, c = dev(scores.T, torch.float32), dev(clicks.T, torch.float32)
    pp = dev(np.concatenate([W[:, 0], b]), torch.float32)
    d = torch.empty_like(s)
    pg = torch.empty(Ls + 1, dtype=torch.float32, device='cuda')
    lw = torch.empty(B, dtype=torch.float32, device='cuda')
    stats = torch.zeros(4 + 3 * Ls, dtype=torch.float32, device='cuda')
    ws = _loss_ws(L, B, Ls)
    for _ in range(2):  # twice: the arrival counter must be left clean
        L.check(L.lib().ultr_loss_dla_fwdbwd(L.ptr(s), L.ptr(c), B, Ls, L.ptr(pp), loss_func, prob, maxw, L.ptr(d),
                                             L.ptr(pg), L.ptr(lw), L.ptr(stats), L.ptr(ws), stream(L)), 'dla')
    st = stats.cpu().numpy().astype(np.float64)
    assert abs(st[0] / st[1] - ref['rank_loss']) <= RTOL * abs(ref['rank_loss']) + 1e-12   # L = 1: the loss is 0
    assert abs(st[2] / st[3] - ref['exam_loss']) <= RTOL * abs(ref['exam_loss']) + 1e-12
    assert relerr(d.cpu().numpy().T / st[1], ref['d_output']) < RTOL or Ls == 1
    pgn = pg.cpu().numpy().astype(np.float64) / st[3]
    assert relerr(pgn[:Ls], ref['d_prop_W'][:, 0]) < 5 * RTOL or Ls == 1   # differences of O(B) sums: a few ulps more
    assert abs(pgn[Ls] - ref['d_prop_b'][0]) <= 5 * RTOL * np.abs(ref['d_prop_W']).sum()
    assert relerr(st[4:4 + Ls], ref['propensity_weights'][0]) < RTOL
    assert relerr(st[4 + Ls:4 + 2 * Ls] / B, ref['relevance_weights'].mean(0)) < RTOL
    want_lw = (ref['propensity_weights'] * np.clip(clicks, 0, 1)).mean(1)
    assert relerr(lw.cpu().numpy(), want_lw) < RTOL


# L <= 16: lane per (list, rank); beyond: thread per list; from 32768 lists of at most 16 on (softmax loss): lane per list
@pytest.mark.parametrize('B,Ls,loss_func', [(256, 10, 0), (777, 7, 0), (256, 10, 1), (20000, 10, 0), (130, 16, 1),
                                            (128, 24, 0), (96, 24, 1), (40000, 10, 0), (33001, 16, 0), (32768, 3, 0),
                                            (40000, 10, 1)])
def test_ipw_loss_kernel(L, B, Ls, loss_func):
    rng = np.random.RandomState(3)
    scores, clicks = _scores_clicks(rng, B, Ls)
    ipw = load_json('randomized_pbm_0.1_1.0_4_1.0.json')['IPW_list']
    pw_ref = np.array([O.propensity_for_one_list(ipw, list(c)) for c in clicks])
    s, c = dev(scores.T, torch.float32), dev(clicks.T, torch.float32)
    pw = torch.empty_like(s)
    L.check(L.lib().ultr_ipw_weights(L.ptr(c), B, Ls, L.ptr(dev(ipw[:Ls], torch.float32)), Ls, L.ptr(pw), stream(L)), 'ipww')
    assert np.array_equal(pw.cpu().numpy().T, pw_ref.astype(np.float32))
    if loss_func == 0:
        loss, d_ref, tot = O.softmax_loss(scores, clicks, pw_ref)
    else:
        loss, d_ref = O.sigmoid_loss_on_list(scores, clicks, pw_ref)
        tot = B
        d_ref = d_ref * B
    d = torch.empty_like(s)
    stats = torch.zeros(2 + Ls, dtype=torch.float32, device='cuda')
    lw = torch.empty(B, dtype=torch.float32, device='cuda')
    L.check(L.lib().ultr_loss_ipw_fwdbwd(L.ptr(s), L.ptr(c), L.ptr(pw), B, Ls, loss_func, L.ptr(d), L.ptr(stats),
                                         L.ptr(lw), L.ptr(_loss_ws(L, B, Ls)), stream(L)), 'ipw')
    st = stats.cpu().numpy().astype(np.float64)
    assert abs(st[0] / st[1] - loss) < RTOL * abs(loss)
    if loss_func == 0:
        assert relerr(d.cpu().numpy().T / st[1], d_ref) < RTOL
    else:
        assert relerr(d.cpu().numpy().T, d_ref) < RTOL and st[1] == B
    assert np.allclose(st[2:], pw_ref.max(0), rtol=1e-6)
    assert relerr(lw.cpu().numpy(), (pw_ref * np.clip(clicks, 0, 1)).mean(1)) < RTOL


@pytest.mark.parametrize('B,Ls', [(256, 10), (1500, 6), (20000, 10), (100, 16), (128, 24)])  # as for IPW
def test_rem_loss_kernel(L, B, Ls):
    rng = np.random.RandomState(9)
    scores, clicks = _scores_clicks(rng, B, Ls)
    p = rng.uniform(0.2, 0.95, size=(1, Ls))
    bias = 0.3
    u = rng.uniform(size=(B, Ls)).astype(np.float32)
    loss, d_ref, new_p, y_ref = O.rem_step(scores.astype(np.float32).astype(np.float64) + np.float32(bias), clicks, p,
                                           0.05, u.astype(np.float64))
    s, c = dev(scores.T, torch.float32), dev(clicks.T, torch.float32)
    prop = dev(p[0], torch.float32)
    d, y = torch.empty_like(s), torch.empty_like(s)
    lw = torch.empty(B, dtype=torch.float32, device='cuda')
    stats = torch.zeros(3 + 3 * Ls, dtype=torch.float32, device='cuda')
    db = torch.zeros(1, dtype=torch.float32, device='cuda')
    L.check(L.lib().ultr_loss_rem_fwdbwd(L.ptr(s), L.ptr(c), B, Ls, L.ptr(prop), L.ptr(dev([bias], torch.float32)),
                                         L.ptr(dev(u.T)), 0, 0, None, 0, L.ptr(d), L.ptr(y), L.ptr(lw), L.ptr(stats),
                                         L.ptr(db), L.ptr(_loss_ws(L, B, Ls)), stream(L)), 'rem')
    st = stats.cpu().numpy().astype(np.float64)
    yk = y.cpu().numpy().T
    # a Bernoulli label can only differ where p_r1 - u is within fp32 rounding of 0
    assert (yk != y_ref).mean() < 1e-3
    same = (yk == y_ref).all(axis=1)
    assert relerr((d.cpu().numpy().T / B)[same], d_ref[same]) < RTOL
    assert st[1] == B and abs(db.item() - d.sum().item()) < 1e-3 * B
    old = torch.empty(Ls, dtype=torch.float32, device='cuda')
    L.check(L.lib().ultr_rem_update_propensity(L.ptr(prop), Ls, L.ptr(stats), 0.05, L.ptr(old), stream(L)), 'remup')
    assert relerr(prop.cpu().numpy(), new_p[0]) < RTOL and relerr(old.cpu().numpy(), p[0]) < 1e-7
    assert relerr(st[2 + Ls:2 + 2 * Ls], yk.sum(0)) < 1e-6 and relerr(st[2 + 2 * Ls:2 + 3 * Ls], clicks.sum(0)) < 1e-6
    assert relerr(lw.cpu().numpy(), (np.clip(clicks, 0, 1) / p).mean(1)) < RTOL
    # Philox-drawn labels: right marginal frequency, different across steps
    L.check(L.lib().ultr_loss_rem_fwdbwd(L.ptr(s), L.ptr(c), B, Ls, L.ptr(dev(p[0], torch.float32)), None, None, 5, 1,
                                         None, 0, L.ptr(d), L.ptr(y), None, L.ptr(stats), None,
                                         L.ptr(_loss_ws(L, B, Ls)), stream(L)), 'rem')
    y1 = y.clone()
    L.check(L.lib().ultr_loss_rem_fwdbwd(L.ptr(s), L.ptr(c), B, Ls, L.ptr(dev(p[0], torch.float32)), None, None, 5, 2,
                                         None, 0, L.ptr(d), L.ptr(y), None, L.ptr(stats), None,
                                         L.ptr(_loss_ws(L, B, Ls)), stream(L)), 'rem')
    assert not torch.equal(y1, y) and ((y1 == 0) | (y1 == 1)).all() and (y1 >= c).all()


@pytest.mark.parametrize('B,Ls', [(40000, 10), (33001, 16), (32768, 3)])
def test_rem_loss_kernel_many_lists(L, B, Ls):
    """From 32768 lists on (Philox draws, L <= 16) a lane owns a whole list: labels and gradients bit-identical to the
    16-lane kernel on a prefix of the lists (same draws, same element arithmetic), batch statistics against float64 sums
    of the kernel's own elementwise outputs."""
    rng = np.random.RandomState(B % 1000)
    scores, clicks = _scores_clicks(rng, B, Ls)
    p = rng.uniform(0.2, 0.95, size=Ls)
    bias = 0.25
    s, c = dev(scores.T, torch.float32), dev(clicks.T, torch.float32)
    prop, bd = dev(p, torch.float32), dev([bias], torch.float32)

    def run(s_, c_, n):
        d, y = torch.empty_like(s_), torch.empty_like(s_)
        lw = torch.empty(n, dtype=torch.float32, device='cuda')
        stats = torch.zeros(3 + 3 * Ls, dtype=torch.float32, device='cuda')
        db = torch.zeros(1, dtype=torch.float32, device='cuda')
        ws = _loss_ws(L, n, Ls)
        for _ in range(2):  # twice: the arrival counter must be left clean
            L.check(L.lib().ultr_loss_rem_fwdbwd(L.ptr(s_), L.ptr(c_), n, Ls, L.ptr(prop), L.ptr(bd), None, 0x5EED, 9, None, 0,
                                                 L.ptr(d), L.ptr(y), L.ptr(lw), L.ptr(stats), L.ptr(db), L.ptr(ws), stream(L)),
                    'rem')
        return d, y, lw, stats.cpu().numpy().astype(np.float64), db.item()
    d, y, lw, st, db = run(s, c, B)
    n_small = 20000
    d2, y2, lw2, _, _ = run(s[:, :n_small].contiguous(), c[:, :n_small].contiguous(), n_small)
    assert torch.equal(d[:, :n_small], d2) and torch.equal(y[:, :n_small], y2)
    assert relerr(lw[:n_small].cpu().numpy(), lw2.cpu().numpy()) < 1e-6
    yk, x = y.cpu().numpy().T.astype(np.float64), scores.astype(np.float32).astype(np.float64) + np.float32(bias)
    g = 1.0 / (1.0 + np.exp(-x))
    ck = clicks.astype(np.float64)
    assert st[1] == B
    assert abs(st[0] - (np.maximum(x, 0) - x * yk + np.log1p(np.exp(-np.abs(x)))).sum()) < RTOL * st[0]
    assert relerr(st[2:2 + Ls], (ck + (1 - ck) * p * (1 - g) / (1 - p * g)).sum(0)) < RTOL
    assert np.array_equal(st[2 + Ls:2 + 2 * Ls], yk.sum(0)) and np.array_equal(st[2 + 2 * Ls:2 + 3 * Ls], ck.sum(0))
    assert abs(st[2 + 3 * Ls] - d.double().sum().item()) < 1e-3 * B and abs(db - st[2 + 3 * Ls]) < 1e-3
    assert relerr(lw.cpu().numpy(), (np.clip(ck, 0, 1) / p).mean(1)) < RTOL
    assert ((yk == 0) | (yk == 1)).all() and (yk >= ck).all()


# short-list kernel: L <= 16; from 32768 lists of at most 12 on: one lane per list (pd_loss_lists_body)
@pytest.mark.parametrize('B,Ls', [(256, 10), (900, 5), (64, 32), (20000, 10), (333, 16), (40000, 10), (33001, 12), (32768, 2),
                                  (40000, 13)])
def test_pd_loss_kernel(L, B, Ls):
    rng = np.random.RandomState(13)
    scores, clicks = _scores_clicks(rng, B, Ls)
    tp, tm = rng.uniform(0.5, 1.5, size=(1, Ls)), rng.uniform(0.5, 1.5, size=(1, Ls))
    loss, d_ref, ntp, ntm, pair = O.pd_step(scores, clicks, tp, tm, 0.05, 1)
    s, c = dev(scores.T, torch.float32), dev(clicks.T, torch.float32)
    tpd, tmd = dev(tp[0], torch.float32), dev(tm[0], torch.float32)
    d = torch.empty_like(s)
    stats = torch.zeros(Ls * Ls + 1, dtype=torch.float32, device='cuda')
    L.check(L.lib().ultr_loss_pd_fwdbwd(L.ptr(s), L.ptr(c), B, Ls, L.ptr(tpd), L.ptr(tmd), L.ptr(d), L.ptr(stats),
                                        L.ptr(_loss_ws(L, B, Ls)), stream(L)), 'pd')
    st = stats.cpu().numpy().astype(np.float64)
    assert relerr(st[:Ls * Ls].reshape(Ls, Ls), pair) < RTOL
    assert abs(st[Ls * Ls] - loss[0, 0]) < RTOL * abs(loss[0, 0])
    assert relerr(d.cpu().numpy().T, d_ref) < RTOL
    old = torch.empty(2 * Ls, dtype=torch.float32, device='cuda')
    L.check(L.lib().ultr_pd_update_ratios(L.ptr(tpd), L.ptr(tmd), Ls, L.ptr(stats), 0.05, 1.0, L.ptr(old), stream(L)), 'pdup')
    assert relerr(tpd.cpu().numpy(), ntp[0]) < RTOL and relerr(tmd.cpu().numpy(), ntm[0]) < RTOL
    assert abs(stats[Ls * Ls].item() - loss[0, 0]) < RTOL * abs(loss[0, 0])


def test_pd_no_pair_at_position0_gives_nan_like_reference(L):
    # q17: ratios divide by position-0 sums; no clicked-over-unclicked pair at rank 0 -> 0/0
    B, Ls = 8, 4
    scores = np.zeros((B, Ls))
    clicks = np.zeros((B, Ls))
    clicks[:, 2] = 1.0
    tpd, tmd = torch.ones(Ls, device='cuda'), torch.ones(Ls, device='cuda')
    stats = torch.zeros(Ls * Ls + 1, device='cuda')
    d = torch.empty((Ls, B), device='cuda')
    L.check(L.lib().ultr_loss_pd_fwdbwd(L.ptr(dev(scores.T, torch.float32)), L.ptr(dev(clicks.T, torch.float32)), B, Ls,
                                        L.ptr(tpd), L.ptr(tmd), L.ptr(d), L.ptr(stats), L.ptr(_loss_ws(L, B, Ls)),
                                        stream(L)), 'pd')
    L.check(L.lib().ultr_pd_update_ratios(L.ptr(tpd), L.ptr(tmd), Ls, L.ptr(stats), 0.05, 1.0, None, stream(L)), 'pdup')
    with np.errstate(all='ignore'):
        _, _, ntp, ntm, _ = O.pd_step(scores, clicks, np.ones((1, Ls)), np.ones((1, Ls)), 0.05, 1)
    assert np.array_equal(np.isnan(tpd.cpu().numpy()), np.isnan(ntp[0]))
    assert np.array_equal(np.isinf(tpd.cpu().numpy()), np.isinf(ntp[0]))


# ------------------------------------------------------------------------------------------------
# metrics
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B,Lx', [(256, 10), (37, 9), (20, 1), (50, 33), (9, 120), (5, 700)])
def test_sort_permutation_bit_exact_with_ties(L, B, Lx):
    from ultra_amd.utils.metrics import sort_by_scores_indices
    rng = np.random.RandomState(Lx)
    scores = np.round(rng.normal(size=(B, Lx)), 1).astype(np.float32)  # many exact ties
    scores[0, :] = 0.0
    u = rng.uniform(size=(B, Lx)).astype(np.float32)
    u[1, :] = 0.5                                                     # ties in the shuffle too -> index order
    for topn in (None, 3):
        got = sort_by_scores_indices(dev(scores), dev(u), topn).cpu().numpy()
        assert np.array_equal(got, O.sort_permutation(scores, u, topn))
    got = sort_by_scores_indices(dev(scores), None, None).cpu().numpy()
    assert np.array_equal(got, O.sort_permutation(scores, None, None))


@pytest.mark.parametrize('B,Lx', [(256, 10), (100, 9), (33, 40), (6, 300),
                                  (40000, 10), (33001, 7), (32768, 16)])  # from 32768 lists on: one lane per list
def test_metrics_kernel(L, B, Lx):
    from ultra_amd.utils import metrics as M
    rng = np.random.RandomState(B)
    labels = rng.choice(3, size=(B, Lx), p=[0.7, 0.2, 0.1]).astype(np.float64)
    labels[3 % B] = 0.0                                   # a list without any relevant doc
    scores = np.round(rng.normal(size=(B, Lx)), 2)
    n_rows = 1000
    docids = rng.randint(0, n_rows, size=(B, Lx))
    docids[:, Lx - 1] = np.where(rng.uniform(size=B) < 0.5, n_rows, docids[:, Lx - 1])  # some pads
    labels[docids == n_rows] = 0.0
    u = rng.uniform(size=(B, Lx)).astype(np.float32)
    w = rng.uniform(0.0, 2.0, size=(B, 1))
    w[5 % B] = 0.0
    topn = [1, 3, 5, 10]
    plan = M.MetricPlan(['mrr', 'ndcg', 'err'], topn)
    out = torch.zeros(plan.n_out, dtype=torch.float32, device='cuda')
    ws = M.metrics_workspace(B, plan, 'cuda')
    s32 = scores.astype(np.float32)
    for _ in range(2):
        M.batch_metrics(plan, dev(s32.T), dev(labels.T, torch.float32), dev(docids.T, torch.int32), n_rows,
                        dev(w[:, 0], torch.float32), out, ws, dev(u), max_label=2.0)
    got = out.cpu().numpy().astype(np.float64)
    masked = O.remove_padding(s32.astype(np.float64), docids, n_rows)
    for i, k in enumerate(topn):
        for weighted, ww in ((False, None), (True, w)):
            assert abs(got[plan.slot('ndcg', i, weighted)] - O.ndcg(labels, masked, ww, k, u, u)) < 2e-6
            assert abs(got[plan.slot('err', i, weighted)] - O.err(labels, masked, 2.0, ww, k, u)) < 2e-6
            assert abs(got[plan.slot('mrr', i, weighted)] - O.mrr(labels, masked, ww, u)) < 2e-6
    # invalid (negative) labels are reset and pushed to the end (metrics.py:241-250)
    labels2 = labels.copy()
    labels2[:, 0] = -1.0
    plan2 = M.MetricPlan(['ndcg'], [5])
    out2 = torch.zeros(plan2.n_out, dtype=torch.float32, device='cuda')
    M.batch_metrics(plan2, dev(s32.T), dev(labels2.T, torch.float32), None, 0, None, out2, keep(M.metrics_workspace(B, plan2, 'cuda')),
                    dev(u), max_label=2.0)
    assert abs(out2[0].item() - O.ndcg(labels2, s32.astype(np.float64), None, 5, u, u)) < 2e-6
    # no tie-break uniforms at all: equal scores keep their index order
    plan3 = M.MetricPlan(['mrr', 'ndcg'], [3, 10])
    out3 = torch.zeros(plan3.n_out, dtype=torch.float32, device='cuda')
    M.batch_metrics(plan3, dev(s32.T), dev(labels.T, torch.float32), None, 0, None, out3, keep(M.metrics_workspace(B, plan3, 'cuda')),
                    None, max_label=2.0)
    got3 = out3.cpu().numpy().astype(np.float64)
    assert abs(got3[plan3.slot('ndcg', 1, False)] - O.ndcg(labels, s32.astype(np.float64), None, 10)) < 2e-6
    assert abs(got3[plan3.slot('ndcg', 0, False)] - O.ndcg(labels, s32.astype(np.float64), None, 3)) < 2e-6
    assert abs(got3[plan3.slot('mrr', 0, False)] - O.mrr(labels, s32.astype(np.float64), None)) < 2e-6
    # the factory API of the reference (metrics.py:64)
    fn = M.make_ranking_metric_fn('ndcg', 3)
    v = fn(dev(labels, torch.float32), dev(s32), None)
    assert abs(v.item() - O.ndcg(labels, s32.astype(np.float64), None, 3)) < 2e-6


# ------------------------------------------------------------------------------------------------
# optimiser
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n,strategy,clip,l2', [(236608, 0, 5.0, 0.0), (236611, 0, 0.5, 1e-3), (11, 0, 5.0, 0.0),
                                                (5000, 1, 0.0, 0.0), (1 << 21, 0, 1.0, 0.0)])
def test_clip_adagrad(L, n, strategy, clip, l2):
    rng = np.random.RandomState(n % 1000)
    theta, acc = rng.normal(size=n), rng.uniform(0.1, 1.0, size=n)
    g = rng.normal(scale=0.01, size=n)
    denom = 7.5
    th, ac, gr = dev(theta, torch.float32), dev(acc, torch.float32), dev(g, torch.float32)
    dn = dev([denom], torch.float32)
    nrm = torch.zeros(1, device='cuda')
    ws = torch.zeros(2048, device='cuda')
    L.check(L.lib().ultr_clip_adagrad(L.ptr(th), L.ptr(ac) if strategy == 0 else None, L.ptr(gr), n, L.ptr(dn), 2.0,
                                      l2, clip, 0.05, strategy, L.ptr(nrm), L.ptr(ws), stream(L)), 'opt')
    t32, a32, g32 = theta.astype(np.float32).astype(np.float64), acc.astype(np.float32).astype(np.float64), \
        g.astype(np.float32).astype(np.float64)
    gg = g32 * 2.0 / denom + l2 * t32
    want_norm = np.sqrt((gg ** 2).sum())
    if clip > 0:
        gg = gg * clip / max(want_norm, clip)
    if strategy == 0:
        want_t, want_a = O.adagrad_update(t32, a32, gg, 0.05)
        assert relerr(ac.cpu().numpy(), want_a) < 1e-6
    else:
        want_t = O.sgd_update(t32, gg, 0.05)
    assert abs(nrm.item() - want_norm) < RTOL * want_norm
    assert relerr(th.cpu().numpy(), want_t) < 1e-6


@pytest.mark.parametrize('kind,Ls,n', [('rem', 10, 50000), ('rem', 10, 1000), ('rem', 64, 50000), ('pd', 10, 50000),
                                       ('pd', 16, 9000), ('pd', 10, 1000), ('pd', 24, 50000), ('rem', 10, 0)])
def test_em_update_rides_with_the_optimiser(L, kind, Ls, n):
    """The parked EM-state update (ultr_*_deferred) against the immediate entry point: carried by the update launch of
    the next ultr_clip_adagrad3 (n > 4096), launched first for a small group, at once for long lists, and by
    ultr_em_update_flush when nothing carries it (n == 0 here).  Bit-exact: both run the same arithmetic."""
    rng = np.random.RandomState(Ls + n % 97)
    if kind == 'rem':
        state = [rng.uniform(0.2, 0.95, size=Ls)]
        stats = np.concatenate([[0.7, 256.0], rng.uniform(10, 200, size=3 * Ls), [0.0]])
        n_old = Ls
    else:
        state = [rng.uniform(0.5, 1.5, size=Ls), rng.uniform(0.5, 1.5, size=Ls)]
        stats = np.concatenate([rng.uniform(0.1, 5.0, size=Ls * Ls), [0.0]])
        n_old = 2 * Ls
    nn = max(n, 8)
    theta, acc, g = rng.normal(size=nn), rng.uniform(0.1, 1.0, size=nn), rng.normal(scale=0.01, size=nn)
    dn = dev([7.5], torch.float32)

    def run(deferred):
        st = [dev(x, torch.float32) for x in state]
        sd = dev(stats, torch.float32)
        row = torch.zeros(8 + n_old, device='cuda')  # [norm | .. | old state (8..)]
        dst = torch.full((8 + n_old,), -5.0, device='cuda')
        th, ac, gr = dev(theta, torch.float32), dev(acc, torch.float32), dev(g, torch.float32)
        ws = torch.zeros(2048, device='cuda')
        sfx = '_deferred' if deferred else ''
        if kind == 'rem':
            L.check(getattr(L.lib(), 'ultr_rem_update_propensity' + sfx)(
                L.ptr(st[0]), Ls, L.ptr(sd), 0.05, L.ptr(row[8:]), stream(L)), 'remup')
        else:
            L.check(getattr(L.lib(), 'ultr_pd_update_ratios' + sfx)(
                L.ptr(st[0]), L.ptr(st[1]), Ls, L.ptr(sd), 0.05, 1.0, L.ptr(row[8:]), stream(L)), 'pdup')
        if n > 4096:
            L.check(L.lib().ultr_clip_adagrad3(
                L.ptr(th), L.ptr(ac), L.ptr(gr), n, L.ptr(dn), 2.0, 0.0, 1.5, 0.05, 0, L.ptr(row[0:1]), L.ptr(ws),
                None, None, None, 0, None, 1.0, 0.0, 0.0, 0.0, None, None, 0, 0,
                L.ptr(row), L.ptr(dst), row.numel(), stream(L)), 'opt3')
        elif n > 0:
            L.check(L.lib().ultr_clip_adagrad(L.ptr(th), L.ptr(ac), L.ptr(gr), n, L.ptr(dn), 2.0, 0.0, 1.5, 0.05, 0,
                                              L.ptr(row[0:1]), L.ptr(ws), stream(L)), 'opt')
        else:
            L.check(L.lib().ultr_em_update_flush(stream(L)), 'flush')
        torch.cuda.synchronize()
        return [x.cpu().numpy() for x in st] + [sd.cpu().numpy(), row.cpu().numpy(), dst.cpu().numpy(), th.cpu().numpy()]
    want, got = run(False), run(True)
    for w, g_ in zip(want, got):
        assert np.array_equal(w, g_, equal_nan=True)
    if n > 4096:  # the published row holds the pre-update state and the norm
        assert np.array_equal(got[-2][8:], np.concatenate(state).astype(np.float32)) and got[-2][0] == got[-3][0] > 0


@pytest.mark.parametrize('with_small,with_sumsq,with_publish', [(True, True, True), (False, True, False),
                                                                (True, False, True), (False, False, True)])
def test_clip_adagrad3_fused_tail(L, with_small, with_sumsq, with_publish):
    """ultr_clip_adagrad3 against two plain ultr_clip_adagrad calls: norm partials in place of the norm pass, the small
    group riding along, the results row published by the update launch (here into device memory)."""
    rng = np.random.RandomState(5)
    n, n_cov, ns = 50000 + 64, 50000, 11
    theta, acc, g = rng.normal(size=n), rng.uniform(0.1, 1.0, size=n), rng.normal(scale=0.01, size=n)
    ths, acs, gs = rng.normal(size=ns), rng.uniform(0.1, 1.0, size=ns), rng.normal(size=ns)
    row = torch.arange(32, dtype=torch.float32, device='cuda')  # results row: [.. | norm (20) | small norm (21) | ..]
    dn, dns = dev([7.5], torch.float32), dev([3.0], torch.float32)

    def fresh():
        return [dev(x, torch.float32) for x in (theta, acc, g, ths, acs, gs)]
    # reference: the plain entry point, one call per group
    th0, ac0, g0, ths0, acs0, gs0 = fresh()
    n0, n0s = torch.zeros(1, device='cuda'), torch.zeros(1, device='cuda')
    ws = torch.zeros(2048, device='cuda')
    L.check(L.lib().ultr_clip_adagrad(L.ptr(th0), L.ptr(ac0), L.ptr(g0), n, L.ptr(dn), 2.0, 0.0, 1.5, 0.05, 0,
                                      L.ptr(n0), L.ptr(ws), stream(L)), 'opt')
    L.check(L.lib().ultr_clip_adagrad(L.ptr(ths0), L.ptr(acs0), L.ptr(gs0), ns, L.ptr(dns), 1.0, 0.0, 0.7, 0.01, 0,
                                      L.ptr(n0s), L.ptr(ws), stream(L)), 'opt')
    th1, ac1, g1, ths1, acs1, gs1 = fresh()
    parts = g1[:n_cov].double().view(50, -1).pow(2).sum(1).float().contiguous()  # 50 partial sums of squares
    dst = torch.full((32,), -5.0, device='cuda')
    small = [L.ptr(ths1), L.ptr(acs1), L.ptr(gs1), ns, L.ptr(dns), 1.0, 0.0, 0.7, 0.01, L.ptr(row[21:22])] if with_small \
        else [None, None, None, 0, None, 1.0, 0.0, 0.0, 0.0, None]
    L.check(L.lib().ultr_clip_adagrad3(
        L.ptr(th1), L.ptr(ac1), L.ptr(g1), n, L.ptr(dn), 2.0, 0.0, 1.5, 0.05, 0, L.ptr(row[20:21]), L.ptr(ws), *small,
        L.ptr(parts) if with_sumsq else None, 50 if with_sumsq else 0, n_cov,
        L.ptr(row) if with_publish else None, L.ptr(dst) if with_publish else None, 32 if with_publish else 0,
        stream(L)), 'opt3')
    torch.cuda.synchronize()
    assert relerr(th1.cpu().numpy(), th0.cpu().numpy()) < 1e-6 and relerr(ac1.cpu().numpy(), ac0.cpu().numpy()) < 1e-6
    assert abs(row[20].item() - n0.item()) < RTOL * n0.item()
    if with_small:
        assert relerr(ths1.cpu().numpy(), ths0.cpu().numpy()) < 1e-6
        assert abs(row[21].item() - n0s.item()) < RTOL * n0s.item()
    if with_publish:
        want = row.clone()
        assert torch.equal(dst, want), (dst, want)  # the fresh norms included
    else:
        assert (dst == -5.0).all()


# ------------------------------------------------------------------------------------------------
# DNN forward / backward
# ------------------------------------------------------------------------------------------------
def _dnn_case(L, F, hidden, act, norm, B, Ls, seed, pad_frac=0.2):
    from ultra_amd.ranking_model import DNN
    from ultra_amd.ranking_model.DNN import feature_row_stats
    rng = np.random.RandomState(seed)
    data = SyntheticData(max(B // 2, 4), Ls, F, seed=seed, pad_frac=pad_frac, sparse=0.5)
    n_rows = data.features.shape[0]
    lists = rng.randint(0, data.initial_list.shape[0], size=B)
    docids = np.where(data.initial_list[lists] < 0, n_rows, data.initial_list[lists]).astype(np.int32)  # [B, Ls]
    hs = 'hidden_layer_sizes=[%s],activation_func=%s,norm=%s' % (', '.join(map(str, hidden)), act, norm) if hidden \
        else 'activation_func=%s,norm=%s' % (act, norm)
    model = DNN(hs)
    if not hidden:
        model.hparams.hidden_layer_sizes = []
    model.create_variables(F, 'cuda', seed=seed)
    ps = O.init_dnn_params(rng, F, hidden, dtype=np.float32)
    for p in ps:
        p['gamma'] = rng.uniform(0.5, 1.5, size=p['gamma'].shape).astype(np.float32)
        p['beta'] = rng.uniform(-0.3, 0.3, size=p['beta'].shape).astype(np.float32)
    model.load_numpy_params(ps)
    ldx = (F + 3) // 4 * 4
    table = torch.zeros((n_rows, ldx), dtype=torch.float32, device='cuda')
    table[:, :F] = dev(data.features)
    stats = feature_row_stats(table, F) if norm == 'layer' else None
    d_lb = dev(docids.T)
    return model, ps, data, table, stats, n_rows, docids, d_lb, rng


DNN_CASES = [
    (136, [512, 256, 128], 'elu', 'layer', 256, 10),   # BASELINE config 2 (MSLR-10K shape)
    (136, [512, 256, 128], 'elu', 'layer', 100, 9),    # toy-like: L = 9, ragged M
    (700, [512, 256, 128], 'elu', 'layer', 64, 10),    # Yahoo shape
    (220, [512, 256, 128], 'elu', 'layer', 64, 10),    # Istella-S shape
    (46, [64, 32], 'relu', 'none', 50, 7),             # no normalisation
    (10, [100, 30], 'tanh', 'layer', 33, 5),           # widths that are not multiples of 32
    (21, [50], 'selu', 'layer', 40, 3),                # F not a multiple of 4
    (30, [20, 20, 20, 20], 'sigmoid', 'layer', 17, 4), # deeper, tiny
    (136, [], 'elu', 'layer', 40, 10),                 # Linear-like: no hidden layer
    (64, [1024], 'elu', 'layer', 40, 3),               # maximum width
]


@pytest.mark.parametrize('F,hidden,act,norm,B,Ls', DNN_CASES)
def test_dnn_forward_backward(L, F, hidden, act, norm, B, Ls):
    model, ps, data, table, stats, n_rows, docids, d_lb, rng = _dnn_case(L, F, hidden, act, norm, B, Ls, seed=F + B)
    scores = model.forward_lists(table, stats, n_rows, d_lb)
    ps64 = [{k: v.astype(np.float64) for k, v in p.items()} for p in ps]
    x = O.gather_rows(data.features.astype(np.float64), docids.T.reshape(-1))
    ref, cache = O.dnn_forward(ps64, x, act, norm)
    got = scores.cpu().numpy().reshape(-1)
    assert relerr(got, ref[:, 0]) < RTOL
    # pad rows (doc id == n_rows) score like the zero feature row
    d_out = rng.normal(size=(Ls * B)).astype(np.float32)
    grads = torch.full((model.n_total,), 7.0, dtype=torch.float32, device='cuda')
    model.backward(dev(d_out).view(Ls, B), grads)
    g_ref = O.dnn_backward(ps64, cache, d_out.astype(np.float64)[:, None], act, norm)
    g_got = model.unflatten(grads)
    total = np.sqrt(sum((g_ref[j][k] ** 2).sum() for j in range(len(ps)) for k in g_ref[j]))
    for j in range(len(ps)):
        for k in ('W', 'b') + (('gamma', 'beta') if norm == 'layer' else ()):
            err = np.linalg.norm(g_got[j][k].reshape(g_ref[j][k].shape) - g_ref[j][k])
            assert err < RTOL * max(np.linalg.norm(g_ref[j][k]), 1e-3 * total), (j, k, err)
    # padding of the flat buffer stays exactly zero, the spare tail is untouched
    flat = grads.cpu().numpy()
    used = np.zeros(model.n_total, bool)
    for ent in model.slices:
        o, ln = ent['W']
        w = np.zeros((ent['K'], ent['ldw']), bool)
        w[:, :ent['N']] = True
        used[o:o + ln] = w.reshape(-1)
        for nm in ('b', 'gamma', 'beta'):
            o, ln = ent[nm]
            used[o:o + ln] = True
    assert (flat[:model.desc.n_params][~used[:model.desc.n_params]] == 0).all()
    assert (flat[model.desc.n_params:] == 7.0).all()
    # the same pass with norm partials (ultr_dnn_backward_sumsq): identical gradient, partials add up to its square norm
    grads2 = torch.full((model.n_total,), 7.0, dtype=torch.float32, device='cuda')
    sumsq = torch.full((model.sumsq_count(),), -1.0, dtype=torch.float32, device='cuda')
    model.backward(dev(d_out).view(Ls, B), grads2, sumsq)
    assert torch.equal(grads2, grads)
    want_sq = float((flat[:model.desc.n_params].astype(np.float64) ** 2).sum())
    assert abs(float(sumsq.double().sum()) - want_sq) <= 1e-5 * max(want_sq, 1e-30)


@pytest.mark.parametrize('F,hidden,B,Ls,loss_func,prob,maxw', [
    (136, [512, 256, 128], 256, 10, 0, 0, -1.0),   # BASELINE config 2
    (136, [512, 256, 128], 100, 9, 1, 0, -1.0),    # ragged M, sigmoid loss
    (46, [64, 32], 50, 16, 0, 1, 3.0),             # longest short list, clipped weights
    (136, [], 40, 10, 0, 0, -1.0),                 # no hidden layer
    (64, [300], 4100, 3, 0, 0, -1.0),              # more lists than one grid of loss workgroups covers at once
])
def test_dla_loss_rides_with_backward(L, F, hidden, B, Ls, loss_func, prob, maxw):
    """ultr_loss_dla_deferred: the loss carried by the backward pass's first launch (which then evaluates the d_scores of
    its rows itself) against a loss launch of its own followed by the same backward pass -- same statistics, d_scores,
    tower gradient, list weights and parameter gradient, bit for bit; and the fall-backs (flush, another d_scores buffer)."""
    model, ps, data, table, stats, n_rows, docids, d_lb, rng = _dnn_case(L, F, hidden, 'elu', 'layer', B, Ls, seed=F + B)
    scores = model.forward_lists(table, stats, n_rows, d_lb)
    clicks = dev(_scores_clicks(rng, B, Ls)[1].T, torch.float32)
    pp = dev(np.concatenate([rng.uniform(-0.7, 0.7, size=Ls), rng.uniform(-0.5, 0.5, size=1)]), torch.float32)
    lib = L.lib()

    def run(mode):
        d = torch.full((Ls, B), 3.0, dtype=torch.float32, device='cuda')
        pg = torch.zeros(Ls + 1, dtype=torch.float32, device='cuda')
        lw = torch.zeros(B, dtype=torch.float32, device='cuda')
        st = torch.zeros(4 + 3 * Ls, dtype=torch.float32, device='cuda')
        ws = _loss_ws(L, B, Ls)
        grads = torch.full((model.n_total,), 7.0, dtype=torch.float32, device='cuda')
        fn = lib.ultr_loss_dla_fwdbwd if mode == 'plain' else lib.ultr_loss_dla_deferred
        for _ in range(2):  # twice: arrival counters must be left clean
            L.check(fn(L.ptr(scores), L.ptr(clicks), B, Ls, L.ptr(pp), loss_func, prob, maxw, L.ptr(d), L.ptr(pg),
                       L.ptr(lw), L.ptr(st), L.ptr(ws), stream(L)), 'dla')
            if mode == 'flush':
                L.check(lib.ultr_loss_flush(stream(L)), 'flush')
            if mode == 'other_buffer':  # the pass reads another d_scores buffer: the parked loss runs first, on its own
                model.backward(torch.zeros_like(d), torch.empty_like(grads))
            model.backward(d, grads)
        L.check(lib.ultr_loss_flush(stream(L)), 'flush')  # nothing left parked
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in (d, pg, lw, st, grads)]
    want = run('plain')
    assert np.abs(want[0] - 3.0).min() > 0 and np.abs(want[4][:model.desc.n_params]).max() > 0
    for mode in ('ride', 'flush', 'other_buffer'):
        got = run(mode)
        for a, b, name in zip(got, want, ('d_scores', 'prop_grads', 'list_weights', 'stats', 'grads')):
            assert np.array_equal(a, b), (mode, name, np.abs(a - b).max())


@pytest.mark.parametrize('kind', ['ipw', 'ipw_sigmoid', 'ipw_table', 'rem_uniforms', 'rem_philox', 'pd'])
@pytest.mark.parametrize('F,hidden,B,Ls', [(136, [512, 256, 128], 256, 10), (46, [64, 32], 77, 16), (136, [], 40, 10)])
def test_ipw_and_rem_losses_ride_with_backward(L, kind, F, hidden, B, Ls):
    """ultr_loss_ipw_deferred / ultr_loss_rem_deferred against a loss launch of its own followed by the same backward
    pass: statistics, d_scores, list weights, (REM) the bias gradient and ranker labels, and the parameter gradient,
    bit for bit."""
    model, ps, data, table, stats, n_rows, docids, d_lb, rng = _dnn_case(L, F, hidden, 'elu', 'layer', B, Ls, seed=F + B)
    scores = model.forward_lists(table, stats, n_rows, d_lb)
    clicks = dev(_scores_clicks(rng, B, Ls)[1].T, torch.float32)
    pw = dev(rng.uniform(0.5, 3.0, size=(Ls, B)) * (clicks.cpu().numpy() > 0), torch.float32)
    prop = dev(rng.uniform(0.2, 0.95, size=Ls), torch.float32)
    bias = dev([0.3], torch.float32)
    u = dev(rng.uniform(size=(Ls, B)), torch.float32)
    tplus = dev(rng.uniform(0.8, 2.0, size=Ls), torch.float32)
    lib = L.lib()

    def run(ride):
        d = torch.full((Ls, B), 3.0, dtype=torch.float32, device='cuda')
        lw = torch.zeros(B, dtype=torch.float32, device='cuda')
        st = torch.zeros(8 + 4 * Ls + Ls * Ls, dtype=torch.float32, device='cuda')
        ws = _loss_ws(L, B, Ls)
        grads = torch.full((model.n_total,), 7.0, dtype=torch.float32, device='cuda')
        ylab = torch.zeros((Ls, B), dtype=torch.float32, device='cuda')
        dbias = torch.zeros(1, dtype=torch.float32, device='cuda')
        rng2 = np.random.RandomState(99)
        for _ in range(2):
            if kind == 'ipw_table':  # weights from the estimator's table, formed inside the carrying launch
                tab = dev(rng2.uniform(1.0, 9.0, size=max(Ls - 3, 1)), torch.float32)   # shorter than the list: last entry repeats
                if ride:
                    L.check(lib.ultr_loss_ipw_table_deferred(L.ptr(scores), L.ptr(clicks), L.ptr(tab), tab.numel(), L.ptr(ylab),
                                                             B, Ls, 0, L.ptr(d), L.ptr(st), L.ptr(lw), L.ptr(ws), stream(L)), 'ipw')
                else:
                    L.check(lib.ultr_ipw_weights(L.ptr(clicks), B, Ls, L.ptr(tab), tab.numel(), L.ptr(ylab), stream(L)), 'w')
                    L.check(lib.ultr_loss_ipw_fwdbwd(L.ptr(scores), L.ptr(clicks), L.ptr(ylab), B, Ls, 0, L.ptr(d), L.ptr(st),
                                                     L.ptr(lw), L.ptr(ws), stream(L)), 'ipw')
            elif kind.startswith('ipw'):
                fn = lib.ultr_loss_ipw_deferred if ride else lib.ultr_loss_ipw_fwdbwd
                L.check(fn(L.ptr(scores), L.ptr(clicks), L.ptr(pw), B, Ls, 1 if kind == 'ipw_sigmoid' else 0, L.ptr(d),
                           L.ptr(st), L.ptr(lw), L.ptr(ws), stream(L)), 'ipw')
            elif kind == 'pd':
                fn = lib.ultr_loss_pd_deferred if ride else lib.ultr_loss_pd_fwdbwd
                L.check(fn(L.ptr(scores), L.ptr(clicks), B, Ls, L.ptr(tplus), L.ptr(prop), L.ptr(d), L.ptr(st), L.ptr(ws),
                           stream(L)), 'pd')
            else:
                fn = lib.ultr_loss_rem_deferred if ride else lib.ultr_loss_rem_fwdbwd
                L.check(fn(L.ptr(scores), L.ptr(clicks), B, Ls, L.ptr(prop), L.ptr(bias),
                           L.ptr(u) if kind == 'rem_uniforms' else None, 0x5EED, 9, None, 1000, L.ptr(d), L.ptr(ylab),
                           L.ptr(lw), L.ptr(st), L.ptr(dbias), L.ptr(ws), stream(L)), 'rem')
            model.backward(d, grads)
        L.check(lib.ultr_loss_flush(stream(L)), 'flush')
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in (d, lw, st, grads, ylab, dbias)]
    want, got = run(False), run(True)
    assert np.abs(want[0] - 3.0).min() > 0 and np.abs(want[3][:model.desc.n_params]).max() > 0
    for a, b, name in zip(got, want, ('d_scores', 'list_weights', 'stats', 'grads', 'ranker_labels', 'd_bias')):
        assert np.array_equal(a, b), (kind, name, np.abs(a - b).max())


def test_dnn_build_reference_api(L):
    from ultra_amd.ranking_model import DNN
    rng = np.random.RandomState(0)
    B, Ls, F = 32, 4, 24
    model = DNN('hidden_layer_sizes=[32, 16]')
    xs = [rng.uniform(-1, 1, size=(B, F)).astype(np.float32) for _ in range(Ls)]
    outs = model.build([dev(x) for x in xs], is_training=True)
    assert len(outs) == Ls and tuple(outs[0].shape) == (B, 1)
    ps = [{k: v.astype(np.float64) for k, v in p.items()} for p in model.unflatten(model.params)]
    ref, _ = O.dnn_forward(ps, np.concatenate(xs, 0).astype(np.float64))
    assert relerr(torch.cat(outs, 0).cpu().numpy(), ref) < RTOL
    assert 'ranking_model/dnn_W_0:0' in model.model_parameters
    # biases are Glorot-uniform, not zero (tf.get_variable default initializer)
    b0 = model.model_parameters['ranking_model/dnn_b_0:0'].cpu().numpy()[:32]
    assert np.abs(b0).max() > 0 and np.abs(b0).max() <= np.sqrt(6.0 / 64) + 1e-6
    assert DNN('norm=batch').create_variables(8, 'cuda') is not None  # batch normalisation: tests/test_gpu_batchnorm.py


@pytest.mark.parametrize('env', [{'ULTR_FWD_CHAIN': '0'}, {'ULTR_BWD_CHAIN': '1'},
                                 {'ULTR_METRICS_RIDE': '0'}, {'ULTR_FUSED_TAIL': '0'}, {'ULTR_LOSS_RIDE': '0'}])
def test_alternate_kernel_paths_in_a_fresh_process(env):
    """The library picks its launch plan from the shapes; the plans that the default shapes no longer reach (per-layer
    forward kernels, the backward chain, a metric launch of its own, norm pass and result copy as launches of their own) are switched
    on through their environment variables -- read once per process, hence the subprocess -- and must pass the same
    DNN forward/backward and training-step parity tests."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-p', 'no:cacheprovider',
                        'tests/test_gpu_kernels.py', 'tests/test_gpu_step.py', '-k',
                        'dnn_forward_backward or dla_steps_match or naive_steps_match'],
                       cwd=root, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------
# online simulation: re-rank by score + click model
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('model_file,B,Lmax,Ls', [('pbm_0.1_1.0_4_1.0.json', 200, 12, 10), ('cascade_0.1_1.0_4_1.0.json', 64, 9, 9),
                                                  ('ubm_0.1_1_4_1.0.json', 100, 30, 10), ('ubm_0.1_1_4_1.0.json', 70, 14, 10), ('trust_bias.json', 50, 300, 10),
                                                  ('trust_bias.json', 33, 16, 16),
                                                  ('oracle', 40, 12, 5)])
def test_online_rerank_click_kernel(L, model_file, B, Lmax, Ls):
    """ultr_online_rerank_click against the reference's per-list loop (deterministic_online_simulation_feed.py:129-187):
    list length from the last real document, stable descending sort with exact score ties, permuted doc ids, clicks on
    the shown ranks from supplied uniforms, zeros beyond, padding untouched -- bit for bit."""
    model = {'model_name': 'oracle', 'click_prob': [1.0], 'exam_prob': [1.0]} if model_file == 'oracle' else load_json(model_file)
    rng = np.random.RandomState(B + Lmax)
    n_rows = 5000
    lens = rng.randint(0, Lmax + 1, size=B)
    lens[:3] = [Lmax, 0, 1]
    docids = np.full((B, Lmax), n_rows, dtype=np.int32)
    labels = np.zeros((B, Lmax), dtype=np.float32)
    for i in range(B):
        docids[i, :lens[i]] = rng.choice(n_rows, size=lens[i], replace=False)
        labels[i, :lens[i]] = rng.choice(5, size=lens[i])
    scores = np.round(rng.normal(size=(B, Lmax)), 1).astype(np.float32)   # one decimal: plenty of exact ties
    u = rng.random_sample((B, Ls)) * 0.7
    kind, exam, n_exam, cp = _device_click_model(model, Ls)
    out_d = torch.full((Lmax, B), -5, dtype=torch.int32, device='cuda')
    out_c = torch.full((Lmax, B), -5.0, dtype=torch.float32, device='cuda')
    out_p = torch.full((Lmax, B), -5, dtype=torch.int32, device='cuda')
    L.check(L.lib().ultr_online_rerank_click(
        L.ptr(dev(docids.T)), L.ptr(dev(labels.T)), L.ptr(dev(scores.T)), B, Lmax, Ls, n_rows, kind, L.ptr(exam), n_exam,
        L.ptr(cp), cp.numel(), L.ptr(dev(u)), 0, 0, None, 0, 0, 100, L.ptr(out_d), L.ptr(out_c), L.ptr(out_p), stream(L)), 'online')
    want_d, want_c, want_p = O.online_rerank_click(
        docids.tolist(), labels.tolist(), scores.tolist(), n_rows, Ls, model['model_name'], model['click_prob'],
        model['exam_prob'], u, model.get('extra'), oracle_mode=model_file == 'oracle')
    assert np.array_equal(out_p.cpu().numpy().T, np.array(want_p))
    assert np.array_equal(out_d.cpu().numpy().T, np.array(want_d))
    assert np.array_equal(out_c.cpu().numpy().T, np.array(want_c, dtype=np.float32))
    if model_file == 'oracle':
        return
    # Philox rounds with check_validation: every list that shows a document ends up with a click, in place
    d_io, c_io = dev(docids.T).clone(), dev(labels.T).clone()
    L.check(L.lib().ultr_online_rerank_click(
        L.ptr(d_io), L.ptr(c_io), L.ptr(dev(scores.T)), B, Lmax, Ls, n_rows, kind, L.ptr(exam), n_exam, L.ptr(cp),
        cp.numel(), None, 99, 4, None, 0, 1, 100, L.ptr(d_io), L.ptr(c_io), None, stream(L)), 'online')
    assert torch.equal(d_io, out_d)
    got = c_io.cpu().numpy().T
    shown_relevant = np.array([(np.array(want_d[i][:Ls]) < n_rows).any() for i in range(B)])
    assert (got[shown_relevant].sum(1) >= 1).all() and (got[~shown_relevant] == 0).all()
    assert (got[:, Ls:] == 0).all() and set(np.unique(got)) <= {0.0, 1.0}
    if model['model_name'] == 'cascade_model':
        assert (got.sum(1) <= 1).all()


def test_online_sample_click_kernel_plackett_luce(L):
    """ultr_online_sample_click: with supplied noise uniforms the ranking equals the oracle's sort of Gumbel-perturbed
    scores; with the Philox stream the first rank follows the Plackett-Luce law p ~ exp(tau * score)
    (stochastic_online_simulation_feed.py:140-158) and two steps give different rankings."""
    model = load_json('pbm_0.1_1.0_4_1.0.json')
    rng = np.random.RandomState(3)
    B, Lmax, Ls, n_rows, tau = 300, 12, 10, 1000, 2.0
    lens = rng.randint(1, Lmax + 1, size=B)
    docids = np.full((B, Lmax), n_rows, dtype=np.int32)
    labels = np.zeros((B, Lmax), dtype=np.float32)
    for i in range(B):
        docids[i, :lens[i]] = rng.choice(n_rows, size=lens[i], replace=False)
        labels[i, :lens[i]] = rng.choice(5, size=lens[i])
    scores = rng.normal(size=(B, Lmax)).astype(np.float32)
    u, ur = rng.random_sample((B, Ls)), rng.random_sample((B, Lmax))
    kind, exam, n_exam, cp = _device_click_model(model, Ls)
    out = [torch.empty((Lmax, B), dtype=t, device='cuda') for t in (torch.int32, torch.float32, torch.int32)]
    common = lambda uu: (L.ptr(dev(docids.T)), L.ptr(dev(labels.T)), L.ptr(dev(scores.T)), B, Lmax, Ls, n_rows, kind,  # noqa: E731
                         L.ptr(exam), n_exam, L.ptr(cp), cp.numel(), uu)
    L.check(L.lib().ultr_online_sample_click(tau, L.ptr(dev(ur)), *common(L.ptr(dev(u))), 0, 0, None, 0, 0, 100,
                                             L.ptr(out[0]), L.ptr(out[1]), L.ptr(out[2]), stream(L)), 'online')
    want_d, want_c, want_p = O.online_rerank_click(
        docids.tolist(), labels.tolist(), scores.tolist(), n_rows, Ls, model['model_name'], model['click_prob'],
        model['exam_prob'], u, tau=tau, rank_uniforms=ur)
    assert np.array_equal(out[2].cpu().numpy().T, np.array(want_p))
    assert np.array_equal(out[0].cpu().numpy().T, np.array(want_d))
    assert np.array_equal(out[1].cpu().numpy().T, np.array(want_c, dtype=np.float32))
    # Philox noise: one list replicated 20000 times, the document ranked first follows softmax(tau * score)
    n, s4 = 20000, np.array([0.3, -0.2, 0.9, 0.0], dtype=np.float32)
    d4 = np.tile(np.array([[5, 6, 7, 8, n_rows, n_rows]], dtype=np.int32), (n, 1))
    sc = np.tile(np.concatenate([s4, [9.0, 9.0]]).astype(np.float32), (n, 1))    # pads score high and must stay put
    o2 = [torch.empty((6, n), dtype=t, device='cuda') for t in (torch.int32, torch.float32, torch.int32)]
    firsts = []
    for step in (1, 2):
        L.check(L.lib().ultr_online_sample_click(
            tau, None, L.ptr(dev(d4.T)), L.ptr(dev(np.zeros((6, n), np.float32))), L.ptr(dev(sc.T)), n, 6, 4, n_rows, kind,
            L.ptr(exam), n_exam, L.ptr(cp), cp.numel(), None, 77, step, None, 0, 0, 100, L.ptr(o2[0]), L.ptr(o2[1]),
            L.ptr(o2[2]), stream(L)), 'online')
        perm = o2[2].cpu().numpy().T
        assert (np.sort(perm[:, :4], axis=1) == np.arange(4)).all() and (perm[:, 4:] == [4, 5]).all()
        firsts.append(perm[:, 0].copy())
    p = np.exp(tau * s4.astype(np.float64))
    p /= p.sum()
    freq = np.bincount(firsts[0], minlength=4) / n
    assert np.abs(freq - p).max() < 4 * np.sqrt(0.25 / n), (freq, p)
    assert (firsts[0] != firsts[1]).mean() > 0.3


@pytest.mark.parametrize('model_file,stochastic', [('pbm_0.1_1.0_4_1.0.json', False), ('ubm_0.1_1_4_1.0.json', False),
                                                   ('cascade_0.1_1.0_4_1.0.json', True)])
def test_online_kernels_short_and_long_list_layouts_agree(L, model_file, stochastic):
    """Lists of at most 16 candidates take the 16-lanes-per-list kernel, longer ones a workgroup per list.  The same
    lists with one more (padding) column go through the other kernel and must come out identical: same ranking, same
    Philox draws for the Plackett-Luce noise and for every re-sampling round of the clicks."""
    model = load_json(model_file)
    rng = np.random.RandomState(17)
    B, Lmax, Ls, n_rows = 333, 16, 10, 4000
    lens = rng.randint(0, Lmax + 1, size=B)
    docids = np.full((B, Lmax + 1), n_rows, dtype=np.int32)
    labels = np.zeros((B, Lmax + 1), dtype=np.float32)
    for i in range(B):
        docids[i, :lens[i]] = rng.choice(n_rows, size=lens[i], replace=False)
        labels[i, :lens[i]] = rng.choice(5, size=lens[i])
    scores = np.round(rng.normal(size=(B, Lmax + 1)), 1).astype(np.float32)
    kind, exam, n_exam, cp = _device_click_model(model, Ls)
    outs = []
    for width in (Lmax, Lmax + 1):
        o = [torch.full((width, B), -5, dtype=t, device='cuda') for t in (torch.int32, torch.float32, torch.int32)]
        args = (L.ptr(dev(docids[:, :width].T)), L.ptr(dev(labels[:, :width].T)), L.ptr(dev(scores[:, :width].T)), B, width, Ls,
                n_rows, kind, L.ptr(exam), n_exam, L.ptr(cp), cp.numel(), None, 4242, 3, None, 1000, 1, 100, L.ptr(o[0]),
                L.ptr(o[1]), L.ptr(o[2]), stream(L))
        if stochastic:
            L.check(L.lib().ultr_online_sample_click(1.5, None, *args), 'online')
        else:
            L.check(L.lib().ultr_online_rerank_click(*args), 'online')
        outs.append([t.cpu().numpy()[:Lmax] for t in o])
    for a, b, name in zip(outs[0], outs[1], ('docids', 'clicks', 'order')):
        assert np.array_equal(a, b), name
    shown = np.minimum(lens, Ls) > 0
    assert (outs[0][1][:Ls].sum(0)[shown] >= 1).all()


@pytest.mark.parametrize('model_file,stochastic,Lmax', [('pbm_0.1_1.0_4_1.0.json', False, 10), ('ubm_0.1_1_4_1.0.json', False, 16),
                                                        ('cascade_0.1_1.0_4_1.0.json', True, 10),
                                                        ('cascade_0.1_1.0_4_1.0.json', False, 13), ('oracle', False, 10)])
def test_online_kernel_many_lists_layout_agrees(L, model_file, stochastic, Lmax):
    """From 131072 lists of at most 16 candidates on a lane owns a whole list.  The same lists padded to 17 columns go
    through the workgroup-per-list kernel and must come out identical: ranking, Plackett-Luce noise, every re-sampling
    round of the clicks."""
    model = {'model_name': 'oracle'} if model_file == 'oracle' else load_json(model_file)
    rng = np.random.RandomState(Lmax)
    B, wide, Ls, n_rows = 140001, 17, 10, 4000
    lens = rng.randint(0, Lmax + 1, size=B)
    docids = np.full((B, wide), n_rows, dtype=np.int32)
    labels = np.zeros((B, wide), dtype=np.float32)
    live = np.arange(wide)[None, :] < lens[:, None]
    docids[live] = rng.randint(0, n_rows, size=int(live.sum()))
    labels[live] = rng.choice(5, size=int(live.sum()))
    docids[::7, 1] = n_rows                       # holes inside a list: the length is 1 + the last real position
    scores = np.round(rng.normal(size=(B, wide)), 1).astype(np.float32)
    kind, exam, n_exam, cp = _device_click_model(model, Ls)
    outs = []
    for width in (Lmax, wide):
        o = [torch.full((width, B), -5, dtype=t, device='cuda') for t in (torch.int32, torch.float32, torch.int32)]
        args = (L.ptr(dev(docids[:, :width].T)), L.ptr(dev(labels[:, :width].T)), L.ptr(dev(scores[:, :width].T)), B, width,
                min(Ls, Lmax), n_rows, kind, L.ptr(exam), n_exam, L.ptr(cp), cp.numel(), None, 777, 5, None, 2000, 1, 100,
                L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), stream(L))
        if stochastic:
            L.check(L.lib().ultr_online_sample_click(1.5, None, *args), 'online')
        else:
            L.check(L.lib().ultr_online_rerank_click(*args), 'online')
        outs.append([t.cpu().numpy()[:Lmax] for t in o])
    for a, b, name in zip(outs[0], outs[1], ('docids', 'clicks', 'order')):
        assert np.array_equal(a, b), name
    assert (outs[0][0] != -5).all() and (outs[0][1] != -5).all()


@pytest.mark.parametrize('B,Lx', [(200, 10), (40000, 10), (25, 40)])  # 16 lanes per list, one lane per list, LDS bitonic sort
def test_metrics_kernel_draws_its_own_tie_break_uniforms(L, B, Lx):
    """ultr_metrics_ties: ties between equal scores are broken by Philox uniforms drawn inside the kernel (the reference
    draws fresh tf.random_uniform values per call, metric_utils.py:76-81); the oracle regenerates the same uniforms, and the
    device counter makes the second call draw new ones."""
    from ultra_amd.utils import metrics as M
    rng = np.random.RandomState(B + Lx)
    labels = rng.choice(3, size=(B, Lx), p=[0.6, 0.25, 0.15]).astype(np.float64)
    scores = np.round(rng.normal(size=(B, Lx)), 0).astype(np.float32)         # integers: ties everywhere
    topn = [1, 3, 5, 10]
    plan = M.MetricPlan(['mrr', 'ndcg', 'err'], topn)
    out = torch.zeros(plan.n_out, dtype=torch.float32, device='cuda')
    ws = M.metrics_workspace(B, plan, 'cuda')
    ctr = torch.zeros(1, dtype=torch.int32, device='cuda')
    seed, slot_base = 0x1234567887654321, 1000
    s_dev, l_dev = dev(scores.T), dev(labels.T, torch.float32)
    seen = []
    for call in range(2):
        M.batch_metrics(plan, s_dev, l_dev, None, 0, None, out, ws, None, max_label=2.0, ties=(seed, 7, ctr, slot_base))
        got = out.cpu().numpy().astype(np.float64)
        assert int(ctr.item()) == call + 1
        u = O.tie_uniforms(seed, 7 + call, slot_base, B, Lx)
        for i, k in enumerate(topn):
            assert abs(got[plan.slot('ndcg', i, False)] - O.ndcg(labels, scores.astype(np.float64), None, k, u, u)) < 2e-6
            assert abs(got[plan.slot('err', i, False)] - O.err(labels, scores.astype(np.float64), 2.0, None, k, u)) < 2e-6
            assert abs(got[plan.slot('mrr', i, False)] - O.mrr(labels, scores.astype(np.float64), None, u)) < 2e-6
        seen.append(got.copy())
    assert not np.array_equal(seen[0], seen[1])                               # fresh draws on the second call
