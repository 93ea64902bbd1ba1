"""Click-model descriptions for the device feed (reference: ultra/utils/click_models.py).

These classes hold the model TABLES (exam_prob, click_prob, eta) and their JSON form; the
sampling itself runs in the HIP kernels of csrc/feed.hip.  ``sampleClicksForOneList`` is kept
for callers outside the training step (it consumes ``random.random()`` exactly like the
reference, click_models.py:86-116 / :144-155 / :204-218 / :258-266 / :291-298) and is never used by
ultra_amd's feeds.  ``device_tables(list_size)`` is what the feeds upload: the ULTR_CLICK_* kind and the
float64 tables in the layout ``ultr_click_sample`` documents.
"""
import json
import random

PBM_ORIGINAL_EXAM_PROB = [0.68, 0.61, 0.48, 0.34, 0.28, 0.20, 0.11, 0.10, 0.08, 0.06]  # :82-83


class ClickModel(object):
    model_name = 'click_model'
    device_kind = None  # ULTR_CLICK_* of the HIP kernel, None = not on the device path

    def __init__(self, neg_click_prob=0.0, pos_click_prob=1.0, relevance_grading_num=1, eta=1.0):
        self.exam_prob = None
        self.setExamProb(eta)
        self.setClickProb(neg_click_prob, pos_click_prob, relevance_grading_num)

    def setClickProb(self, neg_click_prob, pos_click_prob, relevance_grading_num):
        """ERR-style graded click probabilities (:51-57)."""
        step = (pos_click_prob - neg_click_prob) / (pow(2, relevance_grading_num) - 1)
        self.click_prob = [neg_click_prob - step + pow(2, g) * step for g in range(relevance_grading_num + 1)]

    def setExamProb(self, eta):
        self.eta = eta

    def getModelJson(self):
        return {'model_name': self.model_name, 'eta': self.eta, 'click_prob': self.click_prob,
                'exam_prob': self.exam_prob}

    def getExamProb(self, rank):
        return self.exam_prob[min(rank, len(self.exam_prob) - 1)]

    def _click_p(self, label):
        grade = int(label) if label > 0 else 0
        return self.click_prob[min(grade, len(self.click_prob) - 1)]

    def sampleClick(self, rank, relevance_label):
        exam_p, click_p = self.getExamProb(rank), self._click_p(relevance_label)
        return (1 if random.random() < exam_p * click_p else 0), exam_p, click_p

    def estimatePropensityWeightsForOneList(self, click_list, use_non_clicked_data=False):
        return [1.0 / self.getExamProb(r) * self.getExamProb(0) if (use_non_clicked_data or c > 0) else 0.0
                for r, c in enumerate(click_list)]

    def device_tables(self, list_size):
        """(ULTR_CLICK_* kind, exam table flattened, rows of that table, click_prob) for the HIP kernels."""
        return self.device_kind, [float(x) for x in self.exam_prob], len(self.exam_prob), \
            [float(x) for x in self.click_prob]


class PositionBiasedModel(ClickModel):
    model_name = 'position_biased_model'
    device_kind = 0

    def setExamProb(self, eta):
        """exam = built-in table ** eta (:80-84) -- also what a dynamic-eta bump rebuilds (q5)."""
        self.eta = eta
        self.original_exam_prob = list(PBM_ORIGINAL_EXAM_PROB)
        self.exam_prob = [pow(x, eta) for x in self.original_exam_prob]

    def sampleClicksForOneList(self, label_list):
        triples = [self.sampleClick(r, lab) for r, lab in enumerate(label_list)]
        return [t[0] for t in triples], [t[1] for t in triples], [t[2] for t in triples]


class CascadeModel(ClickModel):
    model_name = 'cascade_model'
    device_kind = 1

    def setExamProb(self, eta):
        self.eta = eta
        self.exam_prob = [1.0] * 10  # :200-202

    def sampleClicksForOneList(self, label_list):
        clicks, exams, click_ps = [], [], []
        seen = False
        for r, lab in enumerate(label_list):
            c, e, p = self.sampleClick(r, lab)  # a uniform is drawn at every rank (:208)
            clicks.append(0.0 if seen else c)
            exams.append(0.0 if seen else e)
            click_ps.append(p)
            seen = seen or c > 0
        return clicks, exams, click_ps


UBM_ORIGINAL_EXAM_TABLE = [  # :127-138: row = rank, column = distance to the last click - 1
    [1.0],
    [0.98, 1.0],
    [1.0, 0.62, 0.95],
    [1.0, 0.77, 0.42, 0.82],
    [1.0, 0.92, 0.55, 0.31, 0.69],
    [1.0, 0.96, 0.63, 0.4, 0.22, 0.54],
    [1.0, 0.99, 0.73, 0.46, 0.29, 0.17, 0.47],
    [1.0, 1.0, 0.89, 0.52, 0.35, 0.24, 0.14, 0.43],
    [1.0, 1.0, 0.95, 0.68, 0.4, 0.29, 0.19, 0.12, 0.41],
    [1.0, 1.0, 1.0, 0.96, 0.52, 0.36, 0.27, 0.18, 0.12, 0.43]]


class UserBrowsingModel(ClickModel):
    """Examination depends on the rank and on the distance to the last click above it (:119-191)."""
    model_name = 'user_browsing_model'
    device_kind = 4

    def setExamProb(self, eta):
        self.eta = eta
        self.original_rd_exam_table = [list(row) for row in UBM_ORIGINAL_EXAM_TABLE]
        self.exam_prob = [[pow(x, eta) for x in row] for row in self.original_rd_exam_table]

    def getExamProb(self, rank, last_click_rank):
        distance = rank - last_click_rank
        if rank < len(self.exam_prob):
            return self.exam_prob[rank][distance - 1]
        if distance > rank:
            return self.exam_prob[-1][-1]
        return self.exam_prob[-1][distance - 1 if distance < len(self.exam_prob[-1]) - 1 else -2]

    def sampleClick(self, rank, last_click_rank, relevance_label):
        exam_p, click_p = self.getExamProb(rank, last_click_rank), self._click_p(relevance_label)
        return (1 if random.random() < exam_p * click_p else 0), exam_p, click_p

    def sampleClicksForOneList(self, label_list):
        clicks, exams, click_ps = [], [], []
        last = -1
        for r, lab in enumerate(label_list):
            c, e, p = self.sampleClick(r, last, lab)
            if c > 0:
                last = r
            clicks.append(c)
            exams.append(e)
            click_ps.append(p)
        return clicks, exams, click_ps

    def estimatePropensityWeightsForOneList(self, click_list, use_non_clicked_data=False):
        weights, last = [], -1
        for r, c in enumerate(click_list):
            weights.append(1.0 / self.getExamProb(r, last) if (use_non_clicked_data or c > 0) else 0.0)
            if c > 0:
                last = r
        return weights

    def device_tables(self, list_size):
        n = len(self.exam_prob)
        if n < 2 or any(len(row) != r + 1 for r, row in enumerate(self.exam_prob)):
            raise ValueError('user_browsing_model: exam_prob must be triangular (row r holds r + 1 distances, at least '
                             'two rows) for the device feed')
        flat = []
        for row in self.exam_prob:
            flat.extend([float(x) for x in row] + [0.0] * (n - len(row)))
        return self.device_kind, flat, n, [float(x) for x in self.click_prob]


class ClickMatrixModel(ClickModel):
    """click_prob[grade][rank] is the click probability itself (:244-270)."""
    model_name = 'click_matrix_model'
    device_kind = 3

    def setExamProb(self, eta):
        self.eta = eta
        self.exam_prob = []

    def setClickProb(self, neg_click_prob, pos_click_prob, relevance_grading_num):
        ClickModel.setClickProb(self, neg_click_prob, pos_click_prob, relevance_grading_num)

    def sampleClicksForOneList(self, label_list):
        clicks = []
        for r, lab in enumerate(label_list):
            grade = int(lab) if lab > 0 else 0
            clicks.append(1 if random.random() < self.click_prob[grade][r] else 0)
        return clicks, [], []

    def estimatePropensityWeightsForOneList(self, click_list, use_non_clicked_data=False):
        raise RuntimeError('click_matrix_model has no propensity estimate (the reference raises too)')

    def device_tables(self, list_size):
        rows = self.click_prob
        if not rows or not isinstance(rows[0], (list, tuple)):
            raise ValueError('click_matrix_model: click_prob must be a [grade][rank] matrix')
        width = min(len(row) for row in rows)
        if width < list_size:  # the reference indexes click_prob[grade][rank] and fails with IndexError
            raise IndexError('click_matrix_model: click_prob has %d ranks, the list size is %d' % (width, list_size))
        # grades beyond the matrix raise IndexError in the reference; the device clamps to the last row instead
        flat = [float(x) for row in rows for x in row[:width]]
        return self.device_kind, flat, width, [1.0] * len(rows)


class TrustBiasedModel(ClickModel):
    """PBM examination, click probability mixed with a rank-dependent trust term (Vardasbi et al.; :273-322)."""
    model_name = 'trust_biased_model'
    device_kind = 3

    def setExamProb(self, eta):
        self.eta = eta
        self.original_exam_prob = list(PBM_ORIGINAL_EXAM_PROB)
        self.exam_prob = [pow(x, eta) for x in self.original_exam_prob]

    def getClickProb(self, rank, label):
        primitive = self.click_prob[min(label, len(self.click_prob) - 1)]
        ep_plus = 1 - (rank + 2) / 100
        ep_minus = self.extra['epsilon_1_neg'] / (rank + 1)
        return primitive * ep_plus + (1 - primitive) * ep_minus

    def sampleClick(self, rank, relevance_label):
        grade = int(relevance_label) if relevance_label > 0 else 0
        exam_p, click_p = self.getExamProb(rank), self.getClickProb(rank, grade)
        return (1 if random.random() < exam_p * click_p else 0), exam_p, click_p

    def sampleClicksForOneList(self, label_list, feature_list=None):
        triples = [self.sampleClick(r, lab) for r, lab in enumerate(label_list)]
        return [t[0] for t in triples], [t[1] for t in triples], [t[2] for t in triples]

    def estimatePropensityWeightsForOneList(self, click_list, use_non_clicked_data=False):
        raise RuntimeError('trust_biased_model has no propensity estimate (the reference raises too)')

    def device_tables(self, list_size):
        if not hasattr(self, 'extra'):  # the reference fails with AttributeError at the first sampleClick
            raise AttributeError("trust_biased_model needs extra['epsilon_1_neg'] (set by loadModelFromJson)")
        width = max(list_size, len(self.exam_prob))
        flat = [self.getExamProb(r) * self.getClickProb(r, g) for g in range(len(self.click_prob)) for r in range(width)]
        return self.device_kind, flat, width, [1.0] * len(self.click_prob)


_BY_NAME = {'position_biased_model': PositionBiasedModel, 'cascade_model': CascadeModel,
            'user_browsing_model': UserBrowsingModel, 'click_matrix_model': ClickMatrixModel,
            'trust_biased_model': TrustBiasedModel}


def loadModelFromJson(model_desc):
    """Model from its JSON description (:7-22); eta / click_prob / exam_prob are taken verbatim."""
    model = _BY_NAME.get(model_desc['model_name'], PositionBiasedModel)()  # unknown names fall back to PBM (:8)
    model.eta = model_desc['eta']
    model.click_prob = model_desc['click_prob']
    model.exam_prob = model_desc['exam_prob']
    if 'extra' in model_desc:
        model.extra = model_desc['extra']
    return model


def loadModelFromFile(path):
    with open(path) as fin:
        return loadModelFromJson(json.load(fin))
