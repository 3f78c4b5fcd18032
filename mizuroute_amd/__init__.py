"""mizuroute_amd -- MI355X-native river-routing hot path (drop-in for mizuRoute's main_route).

The compute path is the HIP library built from mizuroute_amd/csrc (C-ABI: include/mzr.h).
There is no CPU fallback: creating a RoutingDomain without the library or without a GPU raises.
"""
from .api import (RoutingDomain, MzrError, lib_path, load_library, build_library,
                  SUM, IRF, KWT, KW, MC, DW)
from .synthetic import RiverNetwork, make_network, make_remap, make_runoff, make_source_runoff, make_star_network

__all__ = ["RoutingDomain", "MzrError", "lib_path", "load_library", "build_library",
           "RiverNetwork", "make_network", "make_remap", "make_runoff", "make_source_runoff", "make_star_network", "SUM", "IRF", "KWT", "KW", "MC", "DW"]
