"""Importable name for the package whose sources live in ``optical-rl-gym-qot-aware_amd/``
(the hyphenated directory name cannot be written in an ``import`` statement)."""
import os as _os

__path__[:] = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                             "optical-rl-gym-qot-aware_amd")]

from ._exports import *  # noqa: E402,F401,F403
from ._exports import __all__  # noqa: E402,F401
