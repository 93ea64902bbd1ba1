"""Inverse-propensity weight tables (reference: ultra/utils/propensity_estimator.py:10-93,149-167).

Only what IPWrank's training step needs: load ``IPW_list`` from JSON and expose it for the
device lookup kernel (ultr_ipw_weights).  The 10^7-session randomised estimation loop
(:95-132) is an offline tool and out of scope.
"""
import json

from . import click_models as CM


class BasicPropensityEstimator(object):
    def __init__(self, file_name=None):
        self.IPW_list = None
        self.click_model = None
        if file_name:
            self.loadEstimatorFromFile(file_name)

    def loadEstimatorFromFile(self, file_name):
        with open(file_name) as fin:
            data = json.load(fin)
        self.IPW_list = data['IPW_list']
        if 'click_model' in data:
            try:
                self.click_model = CM.loadModelFromJson(data['click_model'])
            except NotImplementedError:
                self.click_model = None

    def ipw_table(self, length):
        """Weights for ranks 0..length-1: IPW_list[min(r, len-1)] (:37-40)."""
        return [self.IPW_list[min(r, len(self.IPW_list) - 1)] for r in range(length)]

    def getPropensityForOneList(self, click_list, use_non_clicked_data=False):
        table = self.ipw_table(len(click_list))
        return [table[r] if (use_non_clicked_data or click_list[r] > 0) else 0.0 for r in range(len(click_list))]

    def outputEstimatorToFile(self, file_name):
        with open(file_name, 'w') as fout:
            fout.write(json.dumps({'IPW_list': self.IPW_list}, indent=4, sort_keys=True))


class RandomizedPropensityEstimator(BasicPropensityEstimator):
    pass


class OraclePropensityEstimator(BasicPropensityEstimator):
    """Weights straight from the click model's examination table (:149-167)."""

    def __init__(self, click_model):
        self.click_model = click_model
        self.IPW_list = None

    def ipw_table(self, length):
        return self.click_model.estimatePropensityWeightsForOneList([1] * length)

    def getPropensityForOneList(self, click_list, use_non_clicked_data=False):
        return self.click_model.estimatePropensityWeightsForOneList(click_list, use_non_clicked_data)
