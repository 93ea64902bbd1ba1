"""Click-model descriptions for the device feed (reference: ultra/utils/click_models.py).

These classes hold the model TABLES (exam_prob, click_prob, eta) and their JSON form; the
sampling itself runs in the HIP kernels of csrc/feed.hip.  ``sampleClicksForOneList`` is kept
for callers outside the training step (it consumes ``random.random()`` exactly like the
reference, click_models.py:86-116 / :204-218) and is never used by ultra_amd's feeds.
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


_BY_NAME = {'position_biased_model': PositionBiasedModel, 'cascade_model': CascadeModel}


def loadModelFromJson(model_desc):
    """Model from its JSON description (:7-22); eta / click_prob / exam_prob are taken verbatim."""
    name = model_desc['model_name']
    if name not in _BY_NAME:
        raise NotImplementedError('click model %r is outside the MI355X hot path (PBM and cascade only)' % name)
    model = _BY_NAME[name]()
    model.eta = model_desc['eta']
    model.click_prob = model_desc['click_prob']
    model.exam_prob = model_desc['exam_prob']
    if 'extra' in model_desc:
        model.extra = model_desc['extra']
    return model


def loadModelFromFile(path):
    with open(path) as fin:
        return loadModelFromJson(json.load(fin))
