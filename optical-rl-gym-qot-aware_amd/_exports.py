from . import _lib
from ._lib import OrlgError
from .batched import DEFAULT_BIT_RATES, BatchedDeepRMSAEnv, BatchedRMSAEnv
from .topology import FrozenTopology, Modulation, Path, Service, TopologyView, selection_tables

__all__ = ["FrozenTopology", "Modulation", "Path", "Service", "TopologyView", "selection_tables",
           "BatchedRMSAEnv", "BatchedDeepRMSAEnv", "DEFAULT_BIT_RATES", "OrlgError", "_lib"]
