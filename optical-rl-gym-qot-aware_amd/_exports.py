from .topology import FrozenTopology, Modulation, Path, Service, TopologyView, selection_tables

__all__ = ["FrozenTopology", "Modulation", "Path", "Service", "TopologyView", "selection_tables"]
