#!/usr/bin/env python3
"""Headline benchmark: train query-lists/sec of the ULTRA training step (DNN + DLA, MSLR-WEB10K shape).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input resident in HBM:
ClickSimulationFeed.get_batch (device click simulation + batch assembly) -> DLA.step (DNN forward, DLA
loss fwd+bwd, DNN backward, [all-reduce], clip + Adagrad on both parameter groups, in-step MRR/NDCG@3,5,10
plain + weighted).  Rank 0 prints ONE JSON line (contract in the task statement) including

  roofline     -- the dominant kernel's algorithmic FLOP/s against the dense fp32-MFMA peak, its duration
                  measured with HIP events on the launch stream (ultr_profile_*), over K more steps of the
                  same workload right after the timed region (event pairs would perturb the timed region);
  cpu_baseline -- the op-for-op torch-CPU restatement of the reference graph + the reference's interpreted
                  feed loop (oracle/, kind "port"), on a bounded sample, rank 0 at N=1 only.
"""
import argparse
import contextlib
import ctypes
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.util import GOLDEN, SyntheticData  # noqa: E402

CONFIGS = {
    # name: (features, list size, queries, hidden, algorithm, click model json)
    'mslr10k': dict(F=136, L=10, Q=6000, algo='DLA', click='pbm_0.1_1.0_4_1.0.json'),
    'yahoo': dict(F=700, L=10, Q=20000, algo='RegressionEM', click='pbm_0.1_1.0_4_1.0.json'),
    'istella': dict(F=220, L=10, Q=20000, algo='PairDebias', click='cascade_0.1_1.0_4_1.0.json'),
    'mslr30k': dict(F=136, L=10, Q=19000, algo='DLA', click='pbm_0.1_1.0_4_1.0.json'),
    # BASELINE config 0: example/toy offline_exp_pipeline -- the reference's own CPU-runnable case, on its toy libsvm
    # data (27 queries, lists of up to 9; tests/golden/toy_libsvm is a copy of example/toy/data)
    'toy': dict(F=136, L=9, Q=27, algo='IPWrank', click='pbm_0.1_1.0_4_1.0.json', data_dir='toy_libsvm'),
}
HIDDEN = [512, 256, 128]
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (16 x the fp32 matrix rate)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def train_flops_per_list(F, L):
    """L * (6 P_W - 2 F H0): fwd + wgrad for every layer, dgrad for all but layer 0."""
    pw = F * 512 + 512 * 256 + 256 * 128 + 128
    return L * (6 * pw - 2 * F * 512)


def kernel_flops_per_launch(name, M):
    """GEMM launches are reported as '<kernel>/K<k>N<n>' (csrc/dnn.hip): 2 M k n flops each; the forward chain
    (all hidden layers in one launch) as '<kernel>/P<sum of k n>': 2 M P."""
    m = re.search(r'/K(\d+)N(\d+)$', name)
    if m:
        return 2.0 * M * int(m.group(1)) * int(m.group(2))
    m = re.search(r'/P(\d+)(?:M(\d+))?$', name)   # 'M<rows>': a launch that covers part of the batch
    if m:
        return 2.0 * (int(m.group(2)) if m.group(2) else M) * int(m.group(1))
    return None


def traffic_table(args):
    """Per-kernel HBM-side bytes per launch from the newest committed PMC passes of THIS workload
    (tools/collect_profiles.sh, tools/make_traffic.py): profiles/rNN_traffic.json is the default command's,
    rNN_b4096_traffic.json the mslr30k / 4096-list one, rNN_bf16_traffic.json the default shape in bf16."""
    pdir = os.path.join(ROOT, 'profiles')
    if not os.path.isdir(pdir):
        return None
    if args.config == 'mslr10k' and args.batch == 256:
        infix = '_bf16' if args.dtype == 'bf16' else ''
    elif args.config == 'mslr30k' and args.batch == 4096 and args.dtype != 'bf16':
        infix = '_b4096'
    else:
        return None
    import re
    files = sorted(f for f in os.listdir(pdir) if re.fullmatch(r'r\d+%s_traffic\.json' % infix, f))
    if not files:
        return None
    return json.load(open(os.path.join(pdir, files[-1])))['kernels']


def measured_traffic(table, kernel):
    if not table:
        return None
    for name, ent in table.items():  # template instances are listed as 'name<args>'
        if name == kernel or name.startswith(kernel + '<'):
            return ent['hbm_bytes']
    return None


def step_traffic(table, kernels, nprof):
    """HBM-side bytes of one step: every profiled launch of the step priced at its kernel's measured bytes per launch
    (None if a kernel of the step is missing from the table)."""
    if not table:
        return None
    total = 0.0
    for name, (cnt, _) in kernels.items():
        t = measured_traffic(table, name.split('/')[0])
        if t is None:
            return None
        total += t * cnt / float(nprof)
    return int(total)


def build_world(cfg, args, device):
    from ultra_amd.utils import find_class, metrics
    metrics.RankingMetricKey.MAX_LABEL = 4.0
    if cfg.get('data_dir'):
        from ultra_amd.utils import read_data
        data = read_data(os.path.join(GOLDEN, cfg['data_dir']) + '/', 'train')
        data.pad(data.rank_list_size)
        metrics.RankingMetricKey.MAX_LABEL = 4.0
        # the arrays the CPU restatement indexes
        il = np.full((len(data.initial_list), cfg['L']), -1, dtype=np.int64)
        lab = np.zeros((len(data.initial_list), cfg['L']), dtype=np.float32)
        for i, (ids, labs) in enumerate(zip(data.initial_list, data.labels)):
            il[i, :len(ids)] = ids
            lab[i, :min(len(labs), cfg['L'])] = labs[:cfg['L']]
        data.cpu_view = (np.asarray(data.features, dtype=np.float32), il, lab)
    else:
        data = SyntheticData(cfg['Q'], cfg['L'], cfg['F'], seed=1234, pad_frac=0.1)
    exp = {'ranking_model': 'ultra_amd.ranking_model.DNN',
           'ranking_model_hparams': 'hidden_layer_sizes=[512, 256, 128],compute_dtype=%s' % (
               'bf16' if args.dtype == 'bf16' else 'f32'),
           'learning_algorithm': 'ultra_amd.learning_algorithm.' + cfg['algo'],
           'learning_algorithm_hparams': args.algo_hparams,
           'metrics': ['mrr', 'ndcg'], 'metrics_topn': [3, 5, 10],
           'max_candidate_num': cfg['L'], 'selection_bias_cutoff': cfg['L'], 'seed': 1,
           'metric_shuffle_ties': bool(args.shuffle_ties)}
    if cfg['algo'] == 'IPWrank':
        exp['learning_algorithm_hparams'] = 'propensity_estimator_json=%s' % os.path.join(
            GOLDEN, 'randomized_pbm_0.1_1.0_4_1.0.json')
    model = find_class(exp['learning_algorithm'])(data, exp, False)
    feed = find_class('ultra_amd.input_layer.ClickSimulationFeed')(
        model, args.batch, 'click_model_json=%s,seed=2024,prefetch=%s' % (
            os.path.join(GOLDEN, cfg['click']), 'true' if args.prefetch else 'false'), None)
    return data, exp, model, feed


def cpu_baseline(cfg, args, data, model):
    """The reference's path restated on the CPU: interpreted feed loop (click_simulation_feed.py:107-162) +
    the TF graph op for op in torch fp32 (oracle/torch_graph.py).  Bounded sample."""
    import random
    from oracle import torch_graph as TG
    from oracle import ultr_oracle as O
    # Threads: what this process may actually use -- OMP_NUM_THREADS when the environment sets it (the GPU box exports
    # its CPU share that way: os.cpu_count() there is the whole machine's), else the affinity mask.  Nothing is capped
    # silently: the source of the number, os.cpu_count() and the CPU model go into the output.
    host_cpus = os.cpu_count() or 1
    if os.environ.get('OMP_NUM_THREADS', '').isdigit():
        cores, cores_from = int(os.environ['OMP_NUM_THREADS']), 'OMP_NUM_THREADS'
    else:
        cores, cores_from = len(os.sched_getaffinity(0)), 'sched_getaffinity'
    cores = max(1, min(cores, host_cpus))
    torch.set_num_threads(cores)
    cpu_model = 'unknown'
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                cpu_model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    log('  cpu_baseline: %s, os.cpu_count()=%d, torch threads=%d (from %s)' % (cpu_model, host_cpus, cores, cores_from))
    g = json.load(open(os.path.join(GOLDEN, cfg['click'])))
    kind = {'DLA': 'dla', 'IPWrank': 'ipw', 'RegressionEM': 'rem', 'PairDebias': 'pd'}[cfg['algo']]
    ps = model.model.unflatten(model.model.params)
    kw = {}
    if kind == 'dla':
        pp = model.prop_params.cpu().numpy()
        kw = dict(prop_W=pp[:cfg['L'], None], prop_b=pp[cfg['L']:])
    ref = TG.TorchReferenceStep(kind, ps, cfg['L'], {}, dtype=torch.float32, max_label=4.0, **kw)
    feats_np, il_np, lab_np = getattr(data, 'cpu_view', (data.features, data.initial_list, data.labels))
    features = [list(map(float, row)) for row in feats_np]  # list of lists, like Raw_data.features
    il, labels = np.asarray(il_np).tolist(), np.asarray(lab_np).tolist()
    L, B = cfg['L'], args.batch
    ipw = json.load(open(os.path.join(GOLDEN, 'randomized_pbm_0.1_1.0_4_1.0.json')))['IPW_list']
    random.seed(1)

    def one_step():
        docid_inputs, letor, clicks = [], [], []
        while len(docid_inputs) < B:  # click_simulation_feed.py:131-137
            i = int(random.random() * len(il))
            label_list = [0 if il[i][x] < 0 else labels[i][x] for x in range(L)]
            c = O.sample_clicks_one_list(g['model_name'], g['click_prob'], g['exam_prob'], label_list,
                                         [random.random() for _ in range(L)])
            if sum(c) == 0:
                continue
            base = len(letor)
            for x in range(L):
                if il[i][x] >= 0:
                    letor.append(features[il[i][x]])
            docid_inputs.append([-1 if il[i][x] < 0 else base + x for x in range(L)])
            clicks.append(c)
        n = len(letor)
        table = torch.from_numpy(np.array(letor).astype(np.float32))  # float64 ndarray, cast on feed (q1)
        d = [torch.tensor([n if docid_inputs[b][l] < 0 else docid_inputs[b][l] for b in range(B)]) for l in range(L)]
        c = [torch.tensor([float(clicks[b][l]) for b in range(B)]) for l in range(L)]
        pw = None
        if kind == 'ipw':
            pw = [torch.tensor([O.propensity_for_one_list(ipw, clicks[b])[l] for b in range(B)]) for l in range(L)]
        t_feed = time.time()
        ref.step(table, d, c, pw_lists=pw)
        return t_feed
    one_step()  # warm-up
    t0, n, feed_s = time.time(), 0, 0.0
    while n < 3 or time.time() - t0 < args.cpu_seconds:
        t_in = time.time()
        feed_s += one_step() - t_in
        n += 1
    dt = time.time() - t0
    return {'value': round(n * B / dt, 1), 'unit': 'query-lists/s', 'cores': cores, 'kind': 'port',
            'cores_from': cores_from, 'host_cpus': host_cpus, 'cpu_model': cpu_model,
            'feed_ms_per_step': round(1e3 * feed_s / n, 3), 'graph_ms_per_step': round(1e3 * (dt - feed_s) / n, 3),
            'sample': '%d steps of batch %d (%.1f s): python feed loop (single-threaded, %.0f%% of the time) + torch-CPU '
                      'fp32 graph of the reference on %d threads' % (n, B, dt, 100 * feed_s / dt, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default='mslr10k', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=256, help='query lists per step per GPU')
    ap.add_argument('--graph', type=int, default=-1,
                    help='0: eager launches; 1: the whole step as one hipGraph (single GPU only); 2: two hipGraphs '
                         '(before / after the collective) with the RCCL all-reduce issued between them.  Default: 1 on '
                         'one GPU, 2 on several')
    ap.add_argument('--graph_steps', type=int, default=0,
                    help='--graph 1 only: consecutive steps captured into one hipGraph (0 = four, whatever --steps is; '
                         'steps beyond a multiple are launched eagerly)')
    ap.add_argument('--force_collective', action='store_true',
                    help='single-GPU rehearsal of the multi-GPU launch path: a 1-rank RCCL group, all-reduce issued anyway')
    ap.add_argument('--algo_hparams', default='')
    ap.add_argument('--override', default='', help='diagnostics only: e.g. algo=PairDebias,Q=20000 on top of --config')
    ap.add_argument('--prefetch', type=int, default=1,
                    help='feed hparam prefetch: batch t+1 is assembled inside the last launch of step t (every step still '
                         'assembles exactly one batch)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help='f32: the parity path (headline).  bf16: the opt-in fast path -- forward and dgrad GEMMs of the '
                         'hidden layers on bf16 MFMA with fp32 accumulation, the rest fp32 (bound: DESIGN.md section 3); '
                         'its roofline is priced against the bf16 matrix peak')
    ap.add_argument('--shuffle_ties', type=int, default=1,
                    help='1 (the reference default): the in-step metrics break score ties with fresh uniforms; 0: index order')
    ap.add_argument('--clock_warm_ms', type=float, default=250.0,
                    help='untimed steps of the chosen launch path for this long right before the timed region (after the '
                         '--warmup steps): the GPU raises its engine clock over ~100 ms of sustained load, and a '
                         'short timed region straight after start-up would measure that ramp (0: none)')
    ap.add_argument('--cpu_seconds', type=float, default=12.0)
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_profile', action='store_true')
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner when its first
    # communicator comes up), so file descriptor 1 is pointed at stderr for the run and the line goes to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        torch.distributed.init_process_group('nccl', device_id=device)
    elif args.force_collective:
        torch.distributed.init_process_group('nccl', init_method='tcp://127.0.0.1:29577', rank=0, world_size=1,
                                             device_id=device)
    cfg = dict(CONFIGS[args.config])
    for kv in filter(None, args.override.split(',')):  # diagnostics: swap one field of the named config
        k, v = kv.split('=')
        cfg[k] = type(cfg[k])(v)
    with contextlib.redirect_stdout(sys.stderr):  # the plugin classes print progress lines like the reference does
        data, exp, model, feed = build_world(cfg, args, device)
    dp = world > 1 or args.force_collective
    if args.force_collective:
        model.force_dp_plan = True  # the launch plan of several ranks (norm inside the update launch, reduced metric rows)
    use_graph = args.graph if args.graph >= 0 else 1
    # Steps per captured graph: FOUR whatever --steps is (so that a 20-step and a 200-step run are the same experiment);
    # steps beyond a multiple of four are launched eagerly.  --graph_steps overrides.
    if args.graph_steps <= 0:
        args.graph_steps = 4

    def collective():
        if args.force_collective and world == 1:
            torch.distributed.all_reduce(model._train_bufs['reduce'])
        else:
            model.step_reduce()

    def step():
        input_feed, _ = feed.get_batch(data, check_validation=True)
        if args.force_collective and world == 1:
            model.step_local(input_feed)
            collective()
            return model.step_apply()
        return model.step(None, input_feed, False)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()

    for _ in range(max(args.warmup, 3)):
        last = step()
    barrier()

    # Launch paths: eager launches (the host enqueues ahead of the GPU), or hipGraph replay.  Which one is faster
    # depends on the host's enqueue cost against the step's GPU time, so with --graph -1 both are rehearsed for a
    # few steps and the timed region uses the faster one; "launch" in the output names it.
    paths = {'eager': step}
    last_of = {}
    steps_per_call = {}
    if use_graph:
        try:
            model.enable_graph_replay(True)
            step()
            torch.cuda.synchronize()
            captured = False
            if use_graph == 1:
                # --graph_steps consecutive steps per graph: kernels inside a graph run back to back, while two replays
                # are a few microseconds apart.  Every captured step publishes into its own pinned ring slot, so
                # each step's loss and summary stay readable after a replay.  With several ranks the RCCL all-reduce
                # is captured INSIDE the graph (torch's process group supports capture); if RCCL refuses, the split
                # plan below takes over.
                try:
                    if args.graph_steps > 1:
                        model.enable_graph_replay(False)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        for _ in range(args.graph_steps):
                            last_of['hipGraph replay'] = step()
                    name = 'hipGraph replay' + (' (RCCL all-reduce captured)' if dp else '')
                    last_of[name] = last_of['hipGraph replay']
                    steps_per_call[name] = args.graph_steps
                    paths[name] = graph.replay
                    captured = True
                except Exception as e:  # noqa: BLE001
                    if not dp:
                        raise
                    log('capturing the collective failed (%s: %s); two graphs around an eager all-reduce' % (type(e).__name__, e))
                    torch.cuda.synchronize()
                    model.enable_graph_replay(True)
                    step()
                    torch.cuda.synchronize()
            if not captured:
                # feed + forward + loss + metrics + backward | all-reduce (eager, RCCL's own stream) | update + publish
                graph, graph_apply = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    input_feed, _ = feed.get_batch(data, check_validation=True)
                    model.step_local(input_feed)
                with torch.cuda.graph(graph_apply):
                    last_of['two hipGraphs around one RCCL all-reduce'] = model.step_apply()

                def run_split():
                    graph.replay()
                    collective()
                    graph_apply.replay()
                paths['two hipGraphs around one RCCL all-reduce'] = run_split
            list(paths.values())[-1]()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001 - report and time the eager launch path instead
            log('hipGraph capture failed (%s: %s); timing eager launches' % (type(e).__name__, e))
            paths = {'eager': step}
            model.enable_graph_replay(False)
            torch.cuda.synchronize()
    if args.graph >= 0 and len(paths) > 1:
        del paths['eager']                       # an explicit --graph 1/2 is honoured
    launch = list(paths)[0]
    if len(paths) > 1:
        trial = {}
        for name, fn in paths.items():
            barrier()
            t0 = time.perf_counter()
            for _ in range(max(1, 32 // steps_per_call.get(name, 1))):
                fn()
            torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) * 30.0 / (max(1, 32 // steps_per_call.get(name, 1)) * steps_per_call.get(name, 1))],
                             dtype=torch.float64, device=device)
            if world > 1:
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)  # every rank picks the same path
            trial[name] = float(t.item())
            log('  launch path %-45s %8.1f us/step over 30 steps' % (name, trial[name] / 30 * 1e6))
        launch = min(trial, key=trial.get)
    run = paths[launch]

    spc = steps_per_call.get(launch, 1)
    rest = args.steps % spc                                # steps beyond a multiple of the graph's: eager launches
    # Engine clock: the power management raises it over ~100 ms of sustained load (measured: 20 timed steps take 84.3 us
    # each after 5 warm-up steps, 80.9 after 1000), so the chosen path runs untimed for --clock_warm_ms first.  The
    # number of calls is the same on every rank (they may contain a collective).
    clock_warm_steps = 0
    if args.clock_warm_ms > 0:
        barrier()
        t0 = time.perf_counter()
        for _ in range(4):
            run()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / 4.0], dtype=torch.float64, device=device)
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        n_calls = int(min(50000, max(1, args.clock_warm_ms * 1e-3 / max(float(t.item()), 1e-6))))
        for _ in range(n_calls):
            run()
        clock_warm_steps = (4 + n_calls) * spc
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps // spc):                     # exactly args.steps steps
        r = run()
    for _ in range(rest):
        r_rest = step()
    barrier()
    dt = time.perf_counter() - t0
    last = r if launch == 'eager' else last_of[launch]
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    loss_value = float(last[0])
    lists = args.steps * args.batch * world
    value = lists / dt
    # several ranks: every replica must hold the same parameters after the timed steps (one bucket applied to the same
    # variables on every rank) -- a captured collective that did not do its job would show here
    replicas_equal = None
    if world > 1:
        chk = torch.stack([model.model.params.double().sum(), model.model.params.double().abs().sum()])
        got = [torch.zeros_like(chk) for _ in range(world)]
        torch.distributed.all_gather(got, chk)
        replicas_equal = all(bool(torch.equal(g, got[0])) for g in got)

    # ---- per-kernel durations with HIP events on the launch stream (same workload, eager launches)
    roofline, kernels = None, {}
    if not args.no_profile:
        from ultra_amd import _lib
        lib = _lib.lib()
        model.enable_graph_replay(False)
        lib.ultr_profile_enable(1)
        nprof = min(args.steps, 50)
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 16)
        lib.ultr_profile_report(buf, len(buf))
        lib.ultr_profile_enable(0)
        for line in buf.value.decode().strip().split('\n'):
            name, cnt, ms = line.rsplit(None, 2)
            kernels[name] = (int(cnt), float(ms))
        M = args.batch * cfg['L']
        best = None
        tot_ms = sum(ms for _, ms in kernels.values())
        for name, (cnt, ms) in sorted(kernels.items(), key=lambda kv: -kv[1][1]):
            fl = kernel_flops_per_launch(name, M)
            log('  %-52s %5d launches  %8.3f us/launch  %8.3f us/step  %5.1f%%%s' % (
                name, cnt, 1e3 * ms / cnt, 1e3 * ms / nprof, 100 * ms / tot_ms,
                '' if fl is None else '  %.1f TFLOP/s' % (fl / (ms / cnt * 1e-3) / 1e12)))
            if fl is not None and best is None:
                best = (name, fl, ms / cnt)
        log('  sum of kernel time per step: %.1f us' % (1e3 * tot_ms / nprof))
        if best:
            name, fl, ms = best
            achieved = fl / (ms * 1e-3) / 1e12
            kname = name.split('/')[0]
            peak = PEAK_BF16_MFMA_TFLOPS if kname.endswith('_bf16') else PEAK_FP32_MFMA_TFLOPS
            roofline = {'kernel': kname, 'gemm': name.split('/')[1] + (' (sum of K N over the hidden layers)' if '/P' in name else ''), 'bound': 'mfma', 'achieved': round(achieved, 3),
                        'peak': peak, 'unit': 'TFLOP/s',
                        'frac': round(achieved / peak, 4), 'traffic': measured_traffic(traffic_table(args), kname),
                        'launch_us': round(1e3 * ms, 3), 'flops_per_launch': fl}
    step_gflop = train_flops_per_list(cfg['F'], cfg['L']) * args.batch / 1e9
    if roofline:
        # the whole step against the same peak: training flops of one step / measured step time / peak
        # (always against the fp32 matrix peak: the weight gradients -- a third of the flops -- run fp32 MFMA in both modes)
        roofline['step_frac'] = round(step_gflop / (1e3 * dt / args.steps) / PEAK_FP32_MFMA_TFLOPS, 4)  # GFLOP / ms = TFLOP/s
        roofline['launches_per_step'] = round(sum(c for c, _ in kernels.values()) / float(nprof), 2)
        roofline['step_traffic_bytes'] = step_traffic(traffic_table(args), kernels, nprof)

    out = {
        'metric': 'train query-lists/sec (DNN+DLA, MSLR-10K shape)' if args.config == 'mslr10k'
        else 'train query-lists/sec (%s)' % args.config,
        'value': round(value, 1), 'unit': 'query-lists/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 5), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic' if not cfg.get('data_dir') else 'toy libsvm data of the reference (example/toy/data)',
        'config': {'workload': '%s shape: %d features, list %d, batch %d lists/GPU, DNN[512,256,128] + %s, %s clicks, '
                               '%d %s queries resident in HBM' % (
                                   args.config, cfg['F'], cfg['L'], args.batch, cfg['algo'],
                                   'PBM' if 'pbm' in cfg['click'] else 'cascade', cfg['Q'], 'toy-data' if cfg.get('data_dir') else 'synthetic'),
                   'global_batch': args.batch * world, 'list_size': cfg['L'], 'parallelism': 'dp%d' % world,
                   'launch': launch, 'steps_per_graph': spc, 'clock_warm_steps': clock_warm_steps, 'feed_prefetch': bool(args.prefetch),
                   'replicas_equal': replicas_equal,
                   'train_gflop_per_step_per_gpu': round(train_flops_per_list(cfg['F'], cfg['L']) * args.batch / 1e9, 3)},
        'final_loss': loss_value,
    }
    if roofline:
        out['roofline'] = roofline
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(cfg, args, data, model)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
