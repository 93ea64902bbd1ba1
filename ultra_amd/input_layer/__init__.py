from .base_input_feed import BaseInputFeed
from .click_simulation_feed import ClickSimulationFeed
from .direct_label_feed import DirectLabelFeed
