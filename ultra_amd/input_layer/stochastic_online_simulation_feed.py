"""Online simulation feed with a SAMPLED ranking: Plackett-Luce over the current model's scores, then clicks.

Reference: ultra/input_layer/stochastic_online_simulation_feed.py -- hparams :47-61 (``tau`` scales the scores),
simulate_clicks_online :108-200 (plackett_luce_sampling :140-158), get_batch :202-276, get_next_batch :278-343.
Everything but the way a list is re-ranked is the deterministic feed's; see that module.  The ranking is drawn on
the device (``ultr_online_sample_click``: a sort of Gumbel-perturbed scores, the same distribution as the
reference's ``np.random.choice`` without replacement); numpy's own random stream is not reproduced.
"""
from ..utils import hparams as hp
from .deterministic_online_simulation_feed import DeterministicOnlineSimulationFeed


class StochasticOnlineSimulationFeed(DeterministicOnlineSimulationFeed):
    BANNER = 'Create online stochastic simluation feed'
    STOCHASTIC = True

    def _default_hparams(self):
        return hp.HParams(
            click_model_json='./example/ClickModel/pbm_0.1_1.0_4_1.0.json',
            tau=1.0,                         # scalar for the probability distribution
            oracle_mode=False,
            dynamic_bias_eta_change=0.0,
            dynamic_bias_step_interval=1000,
            device_rng=True,                 # (new) False: click draws from Python's random stream (the ranking stays
                                             # a device draw)
            seed=0)                          # (new) Philox key of the device stream
