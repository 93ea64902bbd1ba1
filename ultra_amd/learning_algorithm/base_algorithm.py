"""Plugin base class for the unbiased learning-to-rank algorithms, MI355X edition.

Reference: ultra/learning_algorithm/base_algorithm.py -- abstract API :27-62,
remove_padding_for_metric_eval :64-87, ranking_model/get_ranking_scores :89-134, losses :136-251.
The constructor / ``step(session, input_feed, forward_only)`` contract is unchanged; ``session``
is opaque (pass None).  Where the reference builds a TF graph, this class owns device buffers and
launches the HIP kernels of libultr_hip.so on torch's current stream:

    feed tensors -> ultr_dnn_forward -> ultr_loss_*_fwdbwd -> ultr_dnn_backward
                 -> [one RCCL all-reduce of the flat bucket] -> ultr_clip_adagrad -> ultr_metrics

The ``bucket`` is one flat fp32 buffer [ranker grads | additive statistics | local values]; the
first two parts are what a data-parallel run all-reduces (SUM) once per step, so normalisers, EM
statistics and gradients are combined before any division -- N ranks x B lists equal one rank x
N*B lists.
"""
from abc import ABC, abstractmethod

import ctypes

import numpy as np
import os
import torch

from .. import _lib, parallel
from ..utils import data_utils, find_class, hparams as hp, metrics as dev_metrics
from ..ranking_model.DNN import feature_row_stats


class Placeholder(object):
    """Stands in for a tf.placeholder: the feeds only use ``.name`` as the feed-dict key."""

    def __init__(self, name):
        self.name = name + ':0'


class HostVariable(object):
    """Stands in for a non-trainable tf.Variable read with ``.eval()`` (main.py:143,149)."""

    def __init__(self, value):
        self.value = value

    def eval(self, session=None):
        return self.value


class Saver(object):
    """Checkpointing in place of tf.train.Saver (dla.py:222, main.py:68-74,181-194): torch.save of the
    algorithm's flat buffers.  The TF checkpoint format is not reproduced."""

    def __init__(self, algo):
        self.algo = algo

    def save(self, session, path, global_step=None):
        step = global_step.eval() if hasattr(global_step, 'eval') else global_step
        full = '%s-%s' % (path, step) if step is not None else path
        torch.save(self.algo.state_dict(), full + '.pt')
        return full

    def restore(self, session, path):
        self.algo.load_state_dict(torch.load(path if path.endswith('.pt') else path + '.pt', weights_only=True))


class DeviceScalar(object):
    """A step result that stays on the GPU until somebody looks at it (float(), arithmetic, format)."""

    def __init__(self, fetch):
        self._fetch, self._v = fetch, None

    def __float__(self):
        if self._v is None:
            self._v = float(self._fetch())
            self._fetch = None
        return self._v

    item = __float__

    def __add__(self, o): return float(self) + o
    __radd__ = __add__
    def __sub__(self, o): return float(self) - o
    def __rsub__(self, o): return o - float(self)
    def __mul__(self, o): return float(self) * o
    __rmul__ = __mul__
    def __truediv__(self, o): return float(self) / o
    def __rtruediv__(self, o): return o / float(self)
    def __eq__(self, o): return float(self) == o
    def __lt__(self, o): return float(self) < o
    def __gt__(self, o): return float(self) > o
    def __format__(self, spec): return format(float(self), spec)
    def __repr__(self): return repr(float(self))


class DeviceBatch(object):
    """The feed of one step, resident on the device."""
    __slots__ = ('table', 'table_stats', 'n_rows', 'docids', 'labels', 'B', 'L')

    def __init__(self, table, table_stats, n_rows, docids, labels):
        self.table, self.table_stats, self.n_rows = table, table_stats, n_rows
        self.docids, self.labels = docids, labels
        self.L, self.B = docids.shape


DEVICE_BATCH_KEY = 'ultra_amd:device_batch'


class BaseAlgorithm(ABC):
    PADDING_SCORE = -100000

    @abstractmethod
    def __init__(self, data_set, exp_settings, forward_only=False):
        pass

    @abstractmethod
    def step(self, session, input_feed, forward_only):
        """-> (loss, None, summary) when training, (None, scores [B, max_candidate_num], summary) otherwise."""
        pass

    # ---- shared construction ---------------------------------------------------------------
    def _setup(self, data_set, exp_settings, forward_only, hparams):
        print(exp_settings['learning_algorithm_hparams'])
        hparams.parse(exp_settings['learning_algorithm_hparams'])
        self.hparams = hparams
        self.exp_settings = exp_settings
        self.forward_only = forward_only
        self.max_candidate_num = exp_settings['max_candidate_num']
        self.feature_size = data_set.feature_size
        self.rank_list_size = None if forward_only else exp_settings['selection_bias_cutoff']
        if not torch.cuda.is_available():
            raise _lib.UltrError('ultra_amd needs an MI355X (HIP) device: there is no CPU path')
        _lib.lib()
        self.pending_ctx = _lib.PendingContext()       # this model's parked requests (include/ultr_hip.h, Conventions)
        self.device = torch.device('cuda', torch.cuda.current_device())
        # feed-dict keys (dla.py:85-94)
        self.is_training = Placeholder('is_train')
        self.letor_features = Placeholder('letor_features')
        self.docid_inputs = [Placeholder('docid_input%d' % i) for i in range(self.max_candidate_num)]
        self.labels = [Placeholder('label%d' % i) for i in range(self.max_candidate_num)]
        self.learning_rate = HostVariable(float(hparams.learning_rate))
        self.global_step = HostVariable(0)
        # the ranking model, by dotted name (base_algorithm.py:125-128)
        name = exp_settings['ranking_model']
        if name.startswith('ultra.'):  # the reference's own settings files select the drop-in class
            name = 'ultra_amd.' + name[len('ultra.'):]
        self.model = find_class(name)(exp_settings['ranking_model_hparams'])
        self.model.create_variables(self.feature_size, self.device, seed=exp_settings.get('seed'))
        self.n_params = int(self.model.desc.n_params)   # what the DNN kernels see
        self.n_total = int(self.model.n_total)          # + the spare tail for extra trainables
        self.n_stats = self._n_stats()
        self.eval_plan = dev_metrics.MetricPlan(exp_settings['metrics'], exp_settings['metrics_topn'])
        self.train_plan = self.eval_plan
        # Ties between equal scores are broken by fresh uniforms on every metric evaluation, as in the reference
        # (metric_utils.py:49-86: shuffle_ties=True is the default of every metric); the metric kernels draw them
        # themselves (Philox keyed by the experiment seed, a device counter they advance).  'metric_shuffle_ties': false
        # in the settings switches to index order.
        self.shuffle_ties = bool(exp_settings.get('metric_shuffle_ties', True))
        self._tie_seed = (int(exp_settings.get('seed') or 0) * 0x9E3779B97F4A7C15 + 0x7F4A7C15) % (1 << 64) or 1
        self.world = torch.distributed.get_world_size() if torch.distributed.is_available() and \
            torch.distributed.is_initialized() else 1
        self.rank = torch.distributed.get_rank() if self.world > 1 else 0
        self._eval_bufs = {}
        self._train_bufs = None
        self._tie_ctr = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.saver = Saver(self)
        if not forward_only:
            if hparams.grad_strategy == 'sgd':
                self.strategy = _lib.OPT_SGD
                self.acc = None
            else:
                self.strategy = _lib.OPT_ADAGRAD
                self.acc = torch.full((self.n_total,), 0.1, dtype=torch.float32, device=self.device)
            self.opt_ws = torch.zeros(2048, dtype=torch.float32, device=self.device)

    # ---- feed handling ------------------------------------------------------------------------
    def _to_device_batch(self, input_feed, list_size):
        """Accept the device feed of ultra_amd.input_layer as is, or a reference-style feed of numpy
        arrays (click_simulation_feed.py:158-162: float64 features, float32-typed doc ids, q1)."""
        batch = input_feed.get(DEVICE_BATCH_KEY)
        if batch is not None:
            if batch.L == list_size:
                return batch
            return DeviceBatch(batch.table, batch.table_stats, batch.n_rows, batch.docids[:list_size],
                               batch.labels[:list_size])
        feats = input_feed[self.letor_features.name]
        if not torch.is_tensor(feats):
            feats = torch.as_tensor(np.asarray(feats, dtype=np.float32).reshape(-1, self.feature_size))
        n_rows = feats.shape[0]
        ldx = (self.feature_size + 3) // 4 * 4
        table = torch.zeros((max(n_rows, 1), ldx), dtype=torch.float32, device=self.device)
        if n_rows:
            table[:n_rows, :self.feature_size] = feats.to(self.device, torch.float32)
        table = table[:n_rows] if n_rows else table[:0]
        stats = feature_row_stats(table, self.feature_size) if self.model.desc.norm == _lib.NORM_LAYER else None

        def stack(placeholders, dtype):
            cols = [torch.as_tensor(np.asarray(input_feed[p.name])) if not torch.is_tensor(input_feed[p.name])
                    else input_feed[p.name] for p in placeholders[:list_size]]
            return torch.stack([c.to(self.device) for c in cols], dim=0).to(dtype).contiguous()
        return DeviceBatch(table, stats, n_rows, stack(self.docid_inputs, torch.int32),
                           stack(self.labels, torch.float32))

    # ---- reference helper API ------------------------------------------------------------------
    def ranking_model(self, list_size, batch, out=None, is_training=True):
        """Scores of the first ``list_size`` positions, position-major [list_size, B]
        (base_algorithm.py:89-134: zero pad row, per-position gather, model.build, concat; is_training is the feed's
        is_train placeholder, dla.py:103-117 -- it only matters under norm='batch')."""
        return self.model.forward_lists(batch.table, batch.table_stats, batch.n_rows, batch.docids[:list_size], out,
                                        is_training=is_training)

    # ---- buffers ---------------------------------------------------------------------------------
    N_LOCAL = 192                               # non-additive values (norms, pre-update EM state)

    def _n_stats(self):
        """Floats of additive statistics the loss kernel produces (multiple of 4)."""
        return 4 + 4 * _lib.ULTR_MAX_LIST + 64

    def _train_buffers(self, B):
        tb = self._train_bufs
        if tb is not None and tb['B'] == B:
            return tb
        L, dev = self.rank_list_size, self.device
        n_met = self.train_plan.n_out
        n0, n1 = self.n_total, self.n_total + self.n_stats
        n_bucket = n1 + n_met + self.N_LOCAL
        tb = {'B': B}
        # Data parallel: the in-step metrics' partial rows (the metric workspace) sit directly in FRONT of the bucket, so
        # that the step's one all-reduce carries them as well: column sums are additive, and the update launch adds the
        # all-reduced rows exactly as it adds one rank's rows -- no launch of its own before the collective.
        n_ws = int(_lib.lib().ultr_metrics_workspace_floats(B, len(self.train_plan.topn)))
        tb['met_rows_reduced'] = self._dp_plan() and n_ws <= 16384
        pre = (n_ws + 63) // 64 * 64 if tb['met_rows_reduced'] else 0
        full = torch.zeros(pre + n_bucket, dtype=torch.float32, device=dev)
        tb['bucket'] = full[pre:]
        tb['reduce'] = full[:pre + n1 + n_met]       # what the collective sums: [metric rows | grads | stats | metrics]
        tb['grads'] = tb['bucket'][:n0]
        tb['stats'] = tb['bucket'][n0:n1]
        tb['metrics'] = tb['bucket'][n1:n1 + n_met]
        tb['local'] = tb['bucket'][n1 + n_met:]
        tb['n_reduce'] = n1 + n_met
        tb['scores'] = torch.empty((L, B), dtype=torch.float32, device=dev)
        tb['d_scores'] = torch.empty((L, B), dtype=torch.float32, device=dev)
        tb['list_weights'] = torch.empty(B, dtype=torch.float32, device=dev)
        tb['loss_ws'] = torch.zeros(_lib.lib().ultr_loss_workspace_floats(B, L), dtype=torch.float32, device=dev)
        tb['met_ws'] = full[:n_ws] if tb['met_rows_reduced'] else dev_metrics.metrics_workspace(B, self.train_plan, dev)
        tb['sumsq'] = torch.zeros(self.model.sumsq_count(), dtype=torch.float32, device=dev)
        tb['host_dev'] = None  # device-side addresses of the pinned result slots, looked up on first use
        n_host = self.n_stats + n_met + self.N_LOCAL
        tb['host'] = torch.zeros((64, n_host), dtype=torch.float32).pin_memory()
        tb['events'] = [None] * 64
        tb['gen'] = [0] * 64     # how often each slot has been handed out: a late reader notices a recycled slot
        tb['slot'] = 0
        tb['fixed_slot'] = getattr(self, '_fixed_slot_default', False)
        self._train_bufs = tb
        return tb

    # opt-in: a parallel hipGraph branch costs more than the 19 us metric kernel wins back at batch 256
    side_metrics = os.environ.get('ULTR_SIDE_METRICS', '0') == '1'

    def _side_stream(self):
        s = getattr(self, '_side', None)
        if s is None:
            s = self._side = torch.cuda.Stream(device=self.device)
        return s

    # True: take the data-parallel launch plan (gradient norm inside the update launch, metric rows inside the reduced
    # range) even with one rank -- bench.py's one-GPU rehearsal of the multi-GPU launch path sets it.
    force_dp_plan = False

    def _dp_plan(self):
        return self.world > 1 or self.force_dp_plan

    def _all_reduce(self, tb):
        parallel.all_reduce_bucket(tb['reduce'], tb['reduce'].numel(), self.world)

    # ULTR_METRICS_RIDE=0: the metric kernel gets a launch of its own instead of riding with the backward pass
    ride_metrics = os.environ.get('ULTR_METRICS_RIDE', '1') != '0'

    def _train_metrics(self, tb, batch, scores, weighted, deferred=False):
        u = getattr(self, 'metric_uniforms', None)  # [B, L] tie-break uniforms supplied by the caller (tests)
        tb['metric_u'] = u  # kept alive until the backward pass has carried a deferred request
        ties = (self._tie_seed, 0, self._tie_ctr, self.rank * batch.B) if (u is None and self.shuffle_ties) else None
        dev_metrics.batch_metrics(self.train_plan, scores, batch.labels, batch.docids, batch.n_rows,
                                  tb['list_weights'] if weighted else None, tb['metrics'], tb['met_ws'], u,
                                  deferred=deferred, ties=ties)

    # ULTR_LOSS_RIDE=0: the loss kernel gets a launch of its own instead of riding with the backward pass's first one
    ride_loss = os.environ.get('ULTR_LOSS_RIDE', '1') != '0'

    def _loss_rides(self):
        """The loss may be parked for the backward pass's first launch when nothing needs its outputs before that: the
        in-step metrics, which read the list weights, must ride with a later launch of the same pass."""
        return self.ride_loss and self.ride_metrics and not self.side_metrics

    # ULTR_FUSED_TAIL=0: norm pass and result copy as launches of their own
    fused_tail = os.environ.get('ULTR_FUSED_TAIL', '1') != '0'

    def _next_slot(self, tb):
        return 0 if tb['fixed_slot'] else tb['slot']

    def _arm_tail(self, tb):
        """Options for ultr_clip_adagrad3: norm partials left by the backward pass, and the results row written
        straight into the pinned host slot by the update launch."""
        if not self.fused_tail:
            return None
        tail = {'sumsq': tb['sumsq'] if tb.get('sumsq_valid') else None, 'n_covered': self.n_params,
                'grid_norm': self._dp_plan(),  # the gradient is final only after the collective: norm inside the update launch
                'fused': self.model.fused_update_target() if tb.get('fused') else None}
        if tb['host_dev'] is None:
            base = _lib.host_device_ptr(tb['host'])
            tb['host_dev'] = base.value if base is not None else 0
        if tb['host_dev']:
            row = tb['host'].shape[1]
            tail['pub'] = (tb['bucket'][self.n_total:], tb['host_dev'] + 4 * row * self._next_slot(tb), row)
        return tail

    def _publish(self, tb, copied=False):
        """Async copy of [stats | metrics | local] to a pinned ring slot; returns fetch() -> numpy row."""
        slot = self._next_slot(tb)
        tb['slot'] = (tb['slot'] + 1) % 64
        host = tb['host'][slot]
        if not copied:
            host.copy_(tb['bucket'][self.n_total:], non_blocking=True)
        capturing = torch.cuda.is_current_stream_capturing()
        ev = None
        if not capturing:
            ev = torch.cuda.Event()
            ev.record()
        tb['gen'][slot] += 1
        stamp = tb['gen'][slot]
        ring = not tb['fixed_slot']

        def fetch():
            # Results are read lazily from a ring of 64 pinned rows: reading one more than 64 steps after its step
            # would silently return a later step's row.  (Under hipGraph replay every step lands in ONE slot by
            # design -- replays are invisible to the host -- and a result means "the latest replay".)
            if ring and tb['gen'][slot] != stamp:
                raise RuntimeError('this step result was first read more than 64 steps after its step: its pinned '
                                   'slot has been reused (read losses / summaries within 64 steps)')
            if ev is not None:
                ev.synchronize()
            else:
                torch.cuda.current_stream().synchronize()
            return host.numpy().copy()
        fetch.stamp = stamp
        return fetch

    # ---- training step skeleton -------------------------------------------------------------------
    WEIGHTED_METRICS = True

    def _train_step(self, input_feed):
        self.step_local(input_feed)
        self.step_reduce()
        return self.step_apply()

    # The three phases of a training step, callable one by one so that a data-parallel driver can replay the two
    # compute phases as hipGraphs and issue the collective between them (bench.py does; RCCL stays outside capture).
    def sync_replicas(self, src=0):
        """Data parallel: every replica takes rank ``src``'s variables -- ranker parameters, optimiser accumulators,
        the algorithm's own state (propensity tower, EM state) and the step count.  The step itself needs no broadcast
        (every rank applies the same all-reduced bucket to the same variables), but that argument starts from equal
        variables: equal seeds give them, a per-rank seed or a checkpoint restored on one rank only does not.  Runs
        once before the first training step and after load_state_dict; no-op without a process group."""
        self._replicas_synced = True
        if self.world <= 1 or not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return
        tensors = [self.model.params]
        if getattr(self, 'acc', None) is not None:
            tensors.append(self.acc)
        tensors += [v for _, v in sorted(self._replica_state().items())]
        for t in tensors:
            torch.distributed.broadcast(t, src=src)
        gs = torch.tensor([self.global_step.value], dtype=torch.int64, device=self.device)
        torch.distributed.broadcast(gs, src=src)
        self.global_step.value = int(gs.item())
        self.model._wt_state = None  # the fused step's transposed weight copies follow the new parameters

    def _replica_state(self):
        """Device tensors beside params / acc that must be equal on every replica (default: the checkpointed extras)."""
        return self._extra_state()

    def step_local(self, input_feed):
        """Everything before the collective: scores, loss + d_scores, in-step metrics, flat un-normalised gradient."""
        self.pending_ctx.bind()
        if self.world > 1 and not getattr(self, '_replicas_synced', False):
            self.sync_replicas()
        L = self.rank_list_size
        batch = self._to_device_batch(input_feed, L)
        tb = self._train_buffers(batch.B)
        tb['fused'] = self._fused_ok(batch)
        if tb['fused']:
            self._step_local_fused(tb, batch)
            self._pending = (tb, batch)
            return
        scores = self.ranking_model(L, batch, tb['scores'])
        try:
            self._loss(tb, batch, scores)                      # d_scores + additive statistics (may be parked)
            self._after_scores(tb, batch, scores)
            self._backward_and_metrics(tb, batch, scores)
        finally:
            # a parked loss that no backward pass carried (an error in between) is launched while its buffers are alive
            _lib.lib().ultr_loss_flush(_lib.current_stream())
        self._pending = (tb, batch)

    def _after_scores(self, tb, batch, scores):
        """Hook: rank-local summaries an algorithm forms from the scores with plain tensor ops (once they exist)."""

    # ULTR_FUSED_STEP=0: forward, loss and backward pass as separate calls even for short lists
    fused_step = os.environ.get('ULTR_FUSED_STEP', '1') != '0'

    # The 16-row fused step's workgroups own one MFMA tile of whole lists and each streams every weight: it wins while
    # the batch fits the chip a few times over (measured against the per-launch plan: 90 against 131 us at 2560 rows,
    # 279 against 297 at 10240, 414 against 405 at 15360, 1094 against 808 at 40960); beyond that the per-layer kernels,
    # which share weight tiles between workgroups, are the better plan.  ULTR_FUSED_MAX_ROWS overrides the threshold.
    # Shapes the 32-row fused step takes (the default scorer in fp32: DNN.fused_rows) stay fused at every size: 713 us
    # against 802 us for the per-launch plan at 40960 rows, 224 against 302 us for the 16-row step at 10240.
    fused_max_rows = int(os.environ.get('ULTR_FUSED_MAX_ROWS', '12288'))

    def _fused_ok(self, batch):
        """Short lists run forward + loss gradient + backward as ONE call (ultr_dnn_train_fused)."""
        # (the bf16 step streams half the bytes and spends next to nothing on MFMA: it stays ahead of the fp32 per-launch
        # plan at every size -- 613 against 810 us at 40960 rows -- so only the fp32 step hands over)
        bf16 = self.model.desc.compute_dtype == _lib.DTYPE_BF16
        if not (self.fused_step and self._loss_rides() and self.model.fused_supported(batch.B, batch.L)):
            return False
        return bf16 or batch.B * batch.L <= self.fused_max_rows or self.model.fused_rows(batch.B, batch.L) == 32

    def _step_local_fused(self, tb, batch):
        scores = tb['scores']
        lib = _lib.lib()
        try:
            self._loss(tb, batch, scores)                      # parked: the fused call differentiates it
            self._train_metrics(tb, batch, scores, self.WEIGHTED_METRICS, deferred=True)
            tb['sumsq_valid'] = self.fused_tail and not self._dp_plan()
            self.model.train_fused(batch.table, batch.table_stats, batch.n_rows, batch.docids[:self.rank_list_size],
                                   scores, tb['grads'], tb['sumsq'] if tb['sumsq_valid'] else None)
            _lib.check(lib.ultr_metrics_flush(_lib.current_stream()), 'ultr_metrics_flush')  # no-op if carried
            if self._dp_plan() and not tb['met_rows_reduced']:
                # the fused step leaves the metrics' column sums to the update launch; when the partial rows are too many
                # to travel with the collective, the sums are formed now and travel instead
                _lib.check(lib.ultr_metrics_tail_flush(_lib.current_stream()), 'ultr_metrics_tail_flush')
            self._after_scores(tb, batch, scores)
        except Exception:
            lib.ultr_metrics_cancel()                          # the scores these requests describe never came to be
            lib.ultr_loss_cancel()
            raise

    def _backward_and_metrics(self, tb, batch, scores):
        # The in-step metrics need only the scores and the list weights: they run on a side stream (a parallel
        # branch of the hipGraph when the step is captured) next to the backward pass, and are joined before
        # the all-reduce, which carries them together with the gradient and the loss statistics.
        if self.side_metrics:
            main = torch.cuda.current_stream()
            side = self._side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._train_metrics(tb, batch, scores, self.WEIGHTED_METRICS)
            tb['sumsq_valid'] = False
            self.model.backward(tb['d_scores'], tb['grads'])   # flat un-normalised gradient
            main.wait_stream(side)
        else:
            # the request is parked and carried by the backward pass inside one of its own launches
            self._train_metrics(tb, batch, scores, self.WEIGHTED_METRICS, deferred=self.ride_metrics)
            # single GPU: the gradient is final after the backward pass, so its norm partials come out of it
            tb['sumsq_valid'] = self.fused_tail and not self._dp_plan()
            self.model.backward(tb['d_scores'], tb['grads'], tb['sumsq'] if tb['sumsq_valid'] else None)
            if self.ride_metrics:
                _lib.check(_lib.lib().ultr_metrics_flush(_lib.current_stream()), 'ultr_metrics_flush')  # no-op if carried

    def step_reduce(self):
        """DP: the one collective of the step (sum of [grads | statistics | metrics] over ranks)."""
        self._all_reduce(self._pending[0])

    def step_apply(self):
        """EM state, clip + Adagrad, publish; -> (loss, None, summary)."""
        tb, batch = self._pending
        self.pending_ctx.bind()
        self._tail = self._arm_tail(tb)                        # consumed by the (last) clip + optimiser call of _update
        self._update(tb, batch)                                # EM state, clip + Adagrad
        _lib.check(_lib.lib().ultr_metrics_tail_flush(_lib.current_stream()), 'ultr_metrics_tail_flush')  # no-op if carried
        fetch = self._publish(tb, copied=self._tail is not None and self._tail.get('published', False))
        self._tail = None
        self.global_step.value += 1
        n_stats, n_met, world = self.n_stats, self.train_plan.n_out, float(self.world)
        tags = self.train_plan.tags(self.WEIGHTED_METRICS)
        cache = {}

        def row():
            if 'r' not in cache:
                r = fetch()
                cache['r'] = (r[:n_stats], r[n_stats:n_stats + n_met] / world, r[n_stats + n_met:])
            return cache['r']

        def summary():
            stats, met, local = row()
            out = self._summary(stats, local, batch.B * self.world)
            for tag, slot in tags:
                out[tag] = met[slot]
            return out
        loss = DeviceScalar(lambda: self._loss_value(*row()[0:3:2]))
        return loss, None, data_utils.Summary(fetch=summary)

    # ---- evaluation step (shared by every algorithm; dla.py:103-115,329-341) -------------------------
    def _eval_step(self, input_feed):
        Lx = self.max_candidate_num
        self.pending_ctx.bind()
        batch = self._to_device_batch(input_feed, Lx)
        B = batch.B
        eb = self._eval_bufs.get(B)
        if eb is None:
            eb = {'scores': torch.empty((Lx, B), dtype=torch.float32, device=self.device),
                  'out': torch.zeros(self.eval_plan.n_out, dtype=torch.float32, device=self.device),
                  'ws': dev_metrics.metrics_workspace(B, self.eval_plan, self.device)}
            self._eval_bufs[B] = eb
        scores = self.ranking_model(Lx, batch, eb['scores'], is_training=False)
        ties = (self._tie_seed, 0, self._tie_ctr, 0) if self.shuffle_ties else None
        dev_metrics.batch_metrics(self.eval_plan, scores, batch.labels, batch.docids, batch.n_rows, None, eb['out'],
                                  eb['ws'], getattr(self, 'metric_uniforms_eval', None), ties=ties)
        out = eb['out'].clone()
        tags = self.eval_plan.tags(False)

        def fetch():
            vals = out.cpu().numpy()
            return {tag: vals[slot] for tag, slot in tags}
        return None, scores.t().contiguous().cpu().numpy(), data_utils.Summary(fetch=fetch)

    # ---- optimiser helper --------------------------------------------------------------------------
    def _params_written_outside_fused(self, theta):
        """An update call that does NOT refresh the fused step's streamed weight copies has written ``theta``: if that
        is the scorer's parameter buffer the copies are stale from now on.  (The C calls write in place without
        touching torch's version counter, so ``DNN.train_fused`` cannot see the write by itself.)"""
        if theta is not None and theta.data_ptr() == self.model.params.data_ptr():
            self.model._wt_state = None

    def _clip_adagrad(self, theta, acc, grads, n, denom, scale_const, l2, clip_norm, lr, out_norm):
        tail = getattr(self, '_tail', None)
        if tail is not None and n > 4096:
            return self._clip_adagrad3((theta, acc, grads, n, denom, scale_const, l2, clip_norm, lr, out_norm), None, tail)
        self._params_written_outside_fused(theta)
        _lib.check(_lib.lib().ultr_clip_adagrad(
            _lib.ptr(theta), _lib.ptr(acc), _lib.ptr(grads), int(n), _lib.ptr(denom), float(scale_const), float(l2),
            float(clip_norm), float(lr), self.strategy, _lib.ptr(out_norm), _lib.ptr(self.opt_ws),
            _lib.current_stream()), 'ultr_clip_adagrad')

    def _clip_adagrad2(self, main, small):
        """Two independent clip + optimiser groups in the launches of one (ultr_clip_adagrad2); each argument is the
        tuple _clip_adagrad takes.  Falls back to two calls when the main group is too small to host the rider."""
        if main[3] <= 4096:
            self._clip_adagrad(*small)
            self._clip_adagrad(*main)
            return
        tail = getattr(self, '_tail', None)
        if tail is not None:
            return self._clip_adagrad3(main, small, tail)
        (th, ac, gr, n, den, sc, l2, clip, lr, out), (ths, acs, grs, ns, dens, scs, l2s, clips, lrs, outs) = main, small
        self._params_written_outside_fused(th)
        self._params_written_outside_fused(ths)
        _lib.check(_lib.lib().ultr_clip_adagrad2(
            _lib.ptr(th), _lib.ptr(ac), _lib.ptr(gr), int(n), _lib.ptr(den), float(sc), float(l2), float(clip), float(lr),
            self.strategy, _lib.ptr(out), _lib.ptr(self.opt_ws),
            _lib.ptr(ths), _lib.ptr(acs), _lib.ptr(grs), int(ns), _lib.ptr(dens), float(scs), float(l2s), float(clips),
            float(lrs), _lib.ptr(outs), _lib.current_stream()), 'ultr_clip_adagrad2')

    def _clip_adagrad3(self, main, small, tail):
        """The fused tail of the step (ultr_clip_adagrad3): optional small group, norm partials from the backward
        pass instead of a norm launch, results row written to the pinned host slot by the update launch."""
        th, ac, gr, n, den, sc, l2, clip, lr, out = main
        ths, acs, grs, ns, dens, scs, l2s, clips, lrs, outs = small if small is not None else (None,) * 4 + (None, 1.0, 0.0, 0.0, 0.0, None)
        sumsq = tail.get('sumsq')
        if sumsq is not None and (gr.data_ptr() != self._train_bufs['grads'].data_ptr() or n < tail['n_covered']):
            sumsq = None  # the partials describe the step's flat gradient buffer only
        pub = tail.get('pub')
        fused = tail.get('fused')
        if fused is not None and th.data_ptr() == self.model.params.data_ptr():
            B, L, ws = fused  # the scorer's update also refreshes the fused step's transposed weight copies
            _lib.check(_lib.lib().ultr_clip_adagrad_fused(
                self.model.desc, B, L, _lib.ptr(ws),
                _lib.ptr(th), _lib.ptr(ac), _lib.ptr(gr), int(n), _lib.ptr(den), float(sc), float(l2), float(clip), float(lr),
                self.strategy, _lib.ptr(out), _lib.ptr(self.opt_ws),
                _lib.ptr(ths), _lib.ptr(acs), _lib.ptr(grs), int(ns or 0), _lib.ptr(dens), float(scs), float(l2s), float(clips),
                float(lrs), _lib.ptr(outs),
                _lib.ptr(sumsq), int(sumsq.numel()) if sumsq is not None else (-1 if tail.get('grid_norm') else 0), int(tail['n_covered']),
                _lib.ptr(pub[0]) if pub else None, ctypes.c_void_p(pub[1]) if pub else None, int(pub[2]) if pub else 0,
                _lib.current_stream()), 'ultr_clip_adagrad_fused')
            self.model.mark_wt_fresh(ws)
            if pub:
                tail['published'] = True
            return
        self._params_written_outside_fused(th)
        self._params_written_outside_fused(ths)
        _lib.check(_lib.lib().ultr_clip_adagrad3(
            _lib.ptr(th), _lib.ptr(ac), _lib.ptr(gr), int(n), _lib.ptr(den), float(sc), float(l2), float(clip), float(lr),
            self.strategy, _lib.ptr(out), _lib.ptr(self.opt_ws),
            _lib.ptr(ths), _lib.ptr(acs), _lib.ptr(grs), int(ns or 0), _lib.ptr(dens), float(scs), float(l2s), float(clips),
            float(lrs), _lib.ptr(outs),
            _lib.ptr(sumsq), int(sumsq.numel()) if sumsq is not None else (-1 if tail.get('grid_norm') else 0), int(tail['n_covered']),
            _lib.ptr(pub[0]) if pub else None, ctypes.c_void_p(pub[1]) if pub else None, int(pub[2]) if pub else 0,
            _lib.current_stream()), 'ultr_clip_adagrad3')
        if pub:
            tail['published'] = True

    # ---- checkpoint state ---------------------------------------------------------------------------
    def state_dict(self):
        sd = {'params': self.model.params.clone(), 'global_step': torch.tensor(self.global_step.value)}
        if getattr(self, 'acc', None) is not None:
            sd['acc'] = self.acc.clone()
        for k, v in self._extra_state().items():
            sd[k] = v.clone()
        return sd

    def load_state_dict(self, sd):
        self.model.params.copy_(sd['params'])
        self.global_step.value = int(sd['global_step'])
        if 'acc' in sd and getattr(self, 'acc', None) is not None:
            self.acc.copy_(sd['acc'])
        for k, v in self._extra_state().items():
            if k in sd:
                v.copy_(sd[k])
        self._replicas_synced = False  # a restore on some ranks only must not leave the replicas apart

    def _extra_state(self):
        return {}

    def enable_graph_replay(self, flag=True):
        """Under hipGraph capture the result row must land in ONE pinned slot (replays reuse addresses)."""
        if self._train_bufs is not None:
            self._train_bufs['fixed_slot'] = flag
        self._fixed_slot_default = flag
