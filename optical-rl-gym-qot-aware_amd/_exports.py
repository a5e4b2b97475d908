from . import _lib, envs
from ._lib import OrlgError
from .batched import DEFAULT_BIT_RATES, BatchedDeepRMSAEnv, BatchedRMSAEnv
from .envs import (DeepRMSAEnv, PathOnlyFirstFitAction, RMSAEnv, SimpleMatrixObservation, deeprmsa_shortest_available_path_first_fit,
                   deeprmsa_shortest_path_first_fit, evaluate_heuristic, least_loaded_path_first_fit,
                   random_policy, shortest_available_path_first_fit, shortest_path_first_fit)
from .monitor import evaluate_heuristic_batched, evaluate_phy_heuristic_batched, write_monitor_csv
from .osnr import gn_gate_parameters, gn_osnr, modulation_level_from_gsnr
from .phy import BatchedPhyRMSAEnv
from .phy_env import (PhyRMSAEnv, phy_aware_bmfa_rmsa, phy_aware_bmfa_rss_rmsa, phy_aware_bmff_rmsa, phy_aware_faff_rmsa,
                      phy_aware_faff_rss_rmsa, phy_aware_sapbm_rmsa,
                      sapff_rmsa, use_existing_channels)
from .registry import ENV_IDS, env_class, make, register_with_gym
from .topology import FrozenTopology, Modulation, Path, Service, TopologyView, selection_tables

__all__ = ["ENV_IDS", "env_class", "make", "register_with_gym", "FrozenTopology", "Modulation", "Path", "Service", "TopologyView", "selection_tables",
           "BatchedRMSAEnv", "BatchedDeepRMSAEnv", "BatchedPhyRMSAEnv", "PhyRMSAEnv", "phy_aware_bmfa_rmsa", "phy_aware_bmfa_rss_rmsa", "phy_aware_bmff_rmsa", "phy_aware_sapbm_rmsa", "phy_aware_faff_rmsa", "phy_aware_faff_rss_rmsa", "sapff_rmsa", "use_existing_channels", "gn_osnr", "gn_gate_parameters", "evaluate_heuristic_batched", "evaluate_phy_heuristic_batched", "write_monitor_csv", "modulation_level_from_gsnr", "DEFAULT_BIT_RATES", "OrlgError", "_lib", "envs",
           "RMSAEnv", "DeepRMSAEnv", "SimpleMatrixObservation", "PathOnlyFirstFitAction", "shortest_path_first_fit", "shortest_available_path_first_fit",
           "least_loaded_path_first_fit", "deeprmsa_shortest_path_first_fit",
           "deeprmsa_shortest_available_path_first_fit", "random_policy", "evaluate_heuristic"]
