"""ultra_amd.utils -- same surface as the reference's ultra.utils for the hot path."""
from . import hparams
from . import click_models
from . import metrics
from . import propensity_estimator
from . import data_utils
from .sys_tools import find_class, create_object
from .data_utils import (Raw_data, read_data, merge_TFSummary, Summary, SummaryValue, generate_ranklist,
                         generate_ranklist_by_scores, output_ranklist)
from .metrics import make_ranking_metric_fn, RankingMetricKey
