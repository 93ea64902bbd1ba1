#!/usr/bin/env python3
"""Achieved HBM GB/s of the bandwidth-bound kernels of the path (click, loss, metric, optimiser), at sizes where
they are no longer launch-bound (north_star: "achieved HBM GB/s against peak for the loss, click and metric
kernels"; SURVEY.md 8d: K1, K4-K12 are HBM-bound and at B <= 4096 launch-latency-bound, so their fraction is
reported from a large-batch run).  Algorithmic bytes per unit are SURVEY 8d's:

  click        8 L B per candidate list (read labels fp32, write clicks)
  DLA loss     12 L B per list (read scores and clicks, write d_scores); IPW loss 16 L B (the weights as well)
  metrics      8 L B per list (read scores and labels) + 4 B per metric
  online feed  20 Lmax B per list (doc ids, labels, scores in; doc ids, clicks out)
  optimiser    20 B per parameter (read g, acc, theta; write acc, theta) + 8 B for the norm pass

    python tools/bench_hbm_kernels.py > profiles/r01_hbm_kernels.txt      (on the GPU box)
"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ultra_amd import _lib  # noqa: E402
from ultra_amd.utils import metrics as M  # noqa: E402

PEAK_GBS = 8000.0  # MI355X HBM3E (MI355X_MICROARCH.md)


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def report(name, units, unit_name, nbytes, sec):
    print('%-22s %10d %-10s %8.1f MB  %9.2f us  %8.1f GB/s  %5.1f %% of HBM peak' % (
        name, units, unit_name, nbytes / 1e6, sec * 1e6, nbytes / sec / 1e9, 100 * nbytes / sec / 1e9 / PEAK_GBS))


def main():
    L = _lib
    lib = L.lib()
    st = L.current_stream()
    dev = 'cuda'
    Ls = 10
    g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'pbm_0.1_1.0_4_1.0.json')))
    exam = torch.tensor(g['exam_prob'], dtype=torch.float64, device=dev)
    cp = torch.tensor(g['click_prob'], dtype=torch.float64, device=dev)
    print('# kernel                   units               bytes       time       achieved   fraction  (event-timed, back to back)')
    for logB in (12, 16, 18, 20):
        B = 1 << logB
        labels = torch.randint(0, 5, (B, Ls), device=dev).float()
        clicks = torch.empty_like(labels)
        report('click_pbm (Philox)', B, 'lists', 8 * Ls * B, timed(lambda: L.check(lib.ultr_click_pbm(
            L.ptr(labels), B, Ls, L.ptr(exam), exam.numel(), L.ptr(cp), cp.numel(), None, 2024, 1, L.ptr(clicks), st), 'click')))
        scores = torch.randn((Ls, B), device=dev)
        c_lb = (torch.rand((Ls, B), device=dev) < 0.2).float()
        pp = torch.zeros(Ls + 1, device=dev)
        d = torch.empty_like(scores)
        pg = torch.empty(Ls + 1, device=dev)
        lw = torch.empty(B, device=dev)
        stats = torch.zeros(4 + 3 * Ls, device=dev)
        ws = torch.zeros(lib.ultr_loss_workspace_floats(B, Ls), device=dev)
        report('dla_loss fwd+bwd', B, 'lists', 12 * Ls * B, timed(lambda: L.check(lib.ultr_loss_dla_fwdbwd(
            L.ptr(scores), L.ptr(c_lb), B, Ls, L.ptr(pp), 0, 0, -1.0, L.ptr(d), L.ptr(pg), L.ptr(lw), L.ptr(stats), L.ptr(ws),
            st), 'dla')))
        pw = torch.rand((Ls, B), device=dev) + 0.5
        istats = torch.zeros(2 + Ls, device=dev)
        report('ipw_loss fwd+bwd', B, 'lists', 16 * Ls * B, timed(lambda: L.check(lib.ultr_loss_ipw_fwdbwd(
            L.ptr(scores), L.ptr(c_lb), L.ptr(pw), B, Ls, 0, L.ptr(d), L.ptr(istats), L.ptr(lw), L.ptr(ws), st), 'ipw')))
        prop = torch.full((Ls,), 0.6, device=dev)
        rstats = torch.zeros(3 + 3 * Ls, device=dev)
        report('rem_loss fwd+bwd', B, 'lists', 12 * Ls * B, timed(lambda: L.check(lib.ultr_loss_rem_fwdbwd(
            L.ptr(scores), L.ptr(c_lb), B, Ls, L.ptr(prop), None, None, 2024, 1, None, 0, L.ptr(d), None, L.ptr(lw),
            L.ptr(rstats), None, L.ptr(ws), st), 'rem')))
        tpd, tmd = torch.rand(Ls, device=dev) + 0.5, torch.rand(Ls, device=dev) + 0.5
        pstats = torch.zeros(Ls * Ls + 1, device=dev)
        report('pd_loss fwd+bwd', B, 'lists', 12 * Ls * B, timed(lambda: L.check(lib.ultr_loss_pd_fwdbwd(
            L.ptr(scores), L.ptr(c_lb), B, Ls, L.ptr(tpd), L.ptr(tmd), L.ptr(d), L.ptr(pstats), L.ptr(ws), st), 'pd')))
        plan = M.MetricPlan(['mrr', 'ndcg'], [3, 5, 10])
        out = torch.zeros(plan.n_out, device=dev)
        mws = M.metrics_workspace(B, plan, dev)
        lab_lb = torch.randint(0, 5, (Ls, B), device=dev).float()
        docids = torch.zeros((Ls, B), dtype=torch.int32, device=dev)
        report('metrics mrr+ndcg@3,5,10', B, 'lists', (8 + 4) * Ls * B + 4 * plan.n_out, timed(
            lambda: M.batch_metrics(plan, scores, lab_lb, docids, 10, lw, out, mws, None, max_label=4.0)))
        # online simulation: re-rank by score + click model, one workgroup per list (20 Lmax bytes per list: doc ids,
        # labels and scores in, doc ids and clicks out)
        dl = torch.randint(0, 1000, (Ls, B), device=dev, dtype=torch.int32)
        od, oc = torch.empty_like(dl), torch.empty_like(lab_lb)
        report('online rerank + click', B, 'lists', 20 * Ls * B, timed(lambda: L.check(lib.ultr_online_rerank_click(
            L.ptr(dl), L.ptr(lab_lb), L.ptr(scores), B, Ls, Ls, 1000, 0, L.ptr(exam), exam.numel(), L.ptr(cp), cp.numel(), None,
            2024, 1, None, 0, 1, 100, L.ptr(od), L.ptr(oc), None, st), 'online')))
    for n in (236608, 1 << 22, 1 << 26):
        th, ac, gr = torch.randn(n, device=dev), torch.full((n,), 0.1, device=dev), torch.randn(n, device=dev) * 1e-3
        dn = torch.ones(1, device=dev)
        nrm = torch.zeros(1, device=dev)
        ows = torch.zeros(2048, device=dev)
        report('clip + adagrad', n, 'params', 28 * n, timed(lambda: L.check(lib.ultr_clip_adagrad(
            L.ptr(th), L.ptr(ac), L.ptr(gr), n, L.ptr(dn), 1.0, 0.0, 5.0, 0.05, 0, L.ptr(nrm), L.ptr(ows), st), 'opt')))


if __name__ == '__main__':
    main()
