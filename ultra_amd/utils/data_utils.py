"""Data set container and summary helpers for the plugin API.

Reference: ultra/utils/data_utils.py -- ``Raw_data`` (:25-346), ``merge_TFSummary`` (:349-369),
``read_data`` (:378-380).  Parsing is one-off host work (not on the hot path); what matters for
the device path is the container's contract: ``features`` (row-indexable, zero pad row appended
by ``pad``), ``initial_list`` (doc row ids, -1 = pad), ``labels``, ``feature_size``,
``rank_list_size``.  TensorFlow summaries are replaced by the duck-typed ``Summary`` below
(``.value[i].tag`` / ``.simple_value``), which is all the reference's callers touch
(main.py:169-188, data_utils.py:359-364).
"""
import json
import os

from . import metrics


class SummaryValue(object):
    __slots__ = ('tag', 'simple_value')

    def __init__(self, tag, simple_value):
        self.tag = tag
        self.simple_value = simple_value

    def __repr__(self):
        return '%s:%g' % (self.tag, self.simple_value)


def sanitize_tag(tag):
    """TF1 summary names: characters outside [A-Za-z0-9_.\\-/] become '_' (spaces included)."""
    return ''.join(c if (c.isalnum() or c in '_.-/') else '_' for c in tag)


class Summary(object):
    """Stand-in for tf.Summary: an ordered list of (tag, simple_value).

    Built lazily from device results: ``fetch`` is a zero-argument callable returning
    {tag: float}; it runs (and synchronises with the GPU) the first time ``.value`` is read,
    so a training loop that never looks at its summaries never waits for them.
    """

    def __init__(self, value=None, fetch=None):
        self._value = list(value) if value is not None else None
        self._fetch = fetch

    @property
    def value(self):
        if self._value is None:
            self._value = [SummaryValue(sanitize_tag(k), float(v)) for k, v in self._fetch().items()]
            self._fetch = None
        return self._value

    def as_dict(self):
        return {v.tag: v.simple_value for v in self.value}


def merge_TFSummary(summary_list, weights):
    """Weighted mean of each tag over the summaries (data_utils.py:349-369)."""
    total, wsum, order = {}, {}, []
    for summary, w in zip(summary_list, weights):
        for e in summary.value:
            if e.tag not in total:
                total[e.tag], wsum[e.tag] = 0.0, 0.0
                order.append(e.tag)
            total[e.tag] += e.simple_value * w
            wsum[e.tag] += w
    return Summary(value=[SummaryValue(t, total[t] / max(0.0000001, wsum[t])) for t in order])


class Raw_data(object):
    """One split of an LTR data set (ULTRA format or libsvm format)."""

    def __init__(self, data_path=None, file_prefix=None, rank_cut=None):
        self.data_path, self.file_prefix = data_path, file_prefix
        self.feature_size = -1
        self.rank_list_size = -1
        self.removed_feature_ids = []
        self.features, self.dids = [], []
        self.initial_list, self.qids, self.labels = [], [], []
        self.initial_scores, self.initial_list_lengths = [], []
        if data_path is None:
            return
        stem = os.path.join(data_path + file_prefix, file_prefix)
        if os.path.isfile(stem + '.feature'):
            self._load_ultra(data_path, stem, rank_cut)
        elif os.path.isfile(stem + '.txt'):
            self._load_libsvm(data_path, stem, file_prefix, rank_cut)
        assert self.feature_size > 0, 'No valid feature has been found.'
        assert len(self.qids) > 0, 'No valid query has been found.'
        assert len(self.dids) > 0, 'No valid doc has been found.'

    # -- settings.json: feature_size, removed_feature_ids, max_label (:68-93)
    def _load_settings(self, data_path):
        with open(data_path + 'settings.json') as fin:
            settings = json.load(fin)
        self.feature_size = settings['feature_size']
        removed = sorted(settings.get('removed_feature_ids', []), reverse=True)
        self.removed_feature_ids = [r for r in removed if r <= self.feature_size]
        metrics.RankingMetricKey.MAX_LABEL = settings['max_label']  # side effect the ERR metric relies on (:92)

    def _feature_row(self, tokens):
        row = [0.0] * self.feature_size
        for tok in tokens:
            idx, val = tok.split(':')
            idx = int(idx) - 1
            if idx < self.feature_size:
                row[idx] = float(val)  # idx == -1 wraps to the last column, as in the reference (:134-136)
        for rf in self.removed_feature_ids:
            del row[rf - 1]
        return row

    def _load_ultra(self, data_path, stem, rank_cut):
        """<prefix>.feature / .init_list / .labels (:95-185)."""
        self._load_settings(data_path)
        with open(stem + '.feature') as fin:
            for line in fin:
                arr = line.strip().split(' ')
                self.dids.append(arr[0])
                self.features.append(self._feature_row(arr[1:]))
        self.feature_size -= len(self.removed_feature_ids)
        with open(stem + '.init_list') as fin:
            for line in fin:
                arr = line.strip().split(' ')
                self.qids.append(arr[0])
                ids = [int(x) for x in arr[1:]]
                self.initial_list.append(ids[:rank_cut] if rank_cut else ids)
                self.rank_list_size = max(self.rank_list_size, len(self.initial_list[-1]))
        with open(stem + '.labels') as fin:
            for line in fin:
                self.labels.append([float(x) for x in line.strip().split(' ')[1:][:self.rank_list_size]])
        self._finish()

    def _load_libsvm(self, data_path, stem, file_prefix, rank_cut):
        """<prefix>.txt, lines ``label qid:<q> idx:val ...``; lists follow file order (:187-265)."""
        self._load_settings(data_path)
        index_of = {}
        with open(stem + '.txt') as fin:
            for line_num, line in enumerate(fin):
                arr = line.strip().split(' ')
                qid = arr[1].split(':')[1]
                if qid not in index_of:
                    index_of[qid] = len(index_of)
                    self.qids.append(qid)
                    self.initial_list.append([])
                    self.labels.append([])
                q = index_of[qid]
                if rank_cut and len(self.initial_list[q]) >= rank_cut:
                    continue
                # doc ids are LINE numbers while features only grow for kept lines (:235-247);
                # the two agree as long as no line was skipped by rank_cut, as in the reference
                self.initial_list[q].append(line_num)
                self.labels[q].append(int(arr[0]))
                self.dids.append('%s_%s_%d' % (file_prefix, qid, line_num))
                self.features.append(self._feature_row(arr[2:]))
        self.feature_size -= len(self.removed_feature_ids)
        self._finish()

    def _finish(self):
        self.initial_list_lengths = [len(x) for x in self.initial_list]
        self.rank_list_size = max([self.rank_list_size] + self.initial_list_lengths)
        self.remove_invalid_data()

    def remove_invalid_data(self):
        """Drop queries with < 2 docs or no relevant doc; feature rows stay (ids remain valid) (:267-322)."""
        keep = [i for i in range(len(self.qids))
                if len(self.initial_list[i]) >= 2 and sum(self.labels[i]) > 0]
        self.qids = [self.qids[i] for i in keep]
        self.initial_list = [self.initial_list[i] for i in keep]
        self.labels = [self.labels[i] for i in keep]
        if self.initial_scores:
            self.initial_scores = [self.initial_scores[i] for i in keep]
        self.initial_list_lengths = [len(x) for x in self.initial_list]
        self.rank_list_size = max([self.rank_list_size] + self.initial_list_lengths)

    def pad(self, rank_list_size, pad_tails=True):
        """Append the zero feature row and pad every list with -1 to ``rank_list_size`` (:324-346)."""
        self.rank_list_size = rank_list_size
        self.features.append([0.0] * self.feature_size)
        for i, lst in enumerate(self.initial_list):
            short = rank_list_size - len(lst)
            if short > 0:
                self.initial_list[i] = lst + [-1] * short if pad_tails else [-1] * short + lst


def read_data(data_path, file_prefix, rank_cut=None):
    return Raw_data(data_path, file_prefix, rank_cut)


# ---- ranked-list output (host side; reference: ultra/utils/data_utils.py:383-501) -------------------------------
def _complete_order(order, n):
    """De-duplicate `order` (first occurrence wins) and append the positions it never mentions, ascending."""
    seen, out = set(), []
    for idx in list(order) + list(range(n)):
        if idx not in seen:
            seen.add(idx)
            out.append(idx)
    return out


def generate_ranklist(data, rerank_lists):
    """qid -> document ids in the order given by positions into the initial list; pad entries dropped (:383-425)."""
    if len(rerank_lists) != len(data.initial_list):
        raise ValueError('The number of queries in rerank ranklists number must be equal to the initial list, %d != %d.'
                         % (len(rerank_lists), len(data.initial_list)))
    out = {}
    for qid, initial, rerank in zip(data.qids, data.initial_list, rerank_lists):
        if len(rerank) != len(initial):
            raise ValueError('The number of docs in each rerank ranklists must be equal to the initial list, %d != %d.'
                             % (len(rerank), len(initial)))
        out[qid] = [data.dids[initial[idx]] for idx in _complete_order(rerank, len(rerank)) if initial[idx] >= 0]
    return out


def generate_ranklist_by_scores(data, rerank_scores):
    """qid -> [(document id, score)] by descending score, ties in initial order; pad entries dropped (:428-477)."""
    if len(rerank_scores) != len(data.initial_list):
        raise ValueError('Rerank ranklists number must be equal to the initial list, %d != %d.'
                         % (len(rerank_scores), len(data.initial_list)))
    out = {}
    for qid, initial, scores in zip(data.qids, data.initial_list, rerank_scores):
        if len(scores) != len(initial):
            raise ValueError('Rerank ranklists length must be equal to the gold list, %d != %d.'
                             % (len(scores), len(initial)))
        order = sorted(range(len(scores)), key=lambda k: scores[k], reverse=True)  # stable: ties keep initial order
        out[qid] = [(data.dids[initial[idx]], scores[idx]) for idx in order if initial[idx] >= 0]
    return out


def output_ranklist(data, rerank_scores, output_path, file_name='test'):
    """TREC run file `<output_path><file_name>.ranklist`: "qid Q0 did rank score Model" (:480-501)."""
    ranked = generate_ranklist_by_scores(data, rerank_scores)
    with open(output_path + file_name + '.ranklist', 'w') as fout:
        for qid in data.qids:
            for rank, (did, score) in enumerate(ranked[qid]):
                fout.write('%s Q0 %s %d %s Model\n' % (qid, did, rank + 1, str(score)))
