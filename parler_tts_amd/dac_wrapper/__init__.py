from .configuration_dac import DACConfig
from .modeling_dac import DACModel
