"""MI355X-native batched RMSA / DeepRMSA / QoT-aware step() path.  Import as ``optical_rl_gym_amd``."""
from ._exports import *  # noqa: F401,F403
from ._exports import __all__  # noqa: F401
