"""MI355X-native per-scan particle-filter update of KumarRobotics/top_down_renderer.

Host-side mirrors of the reference's hot-path classes over hand-written HIP kernels (libtdr_hip.so, include/tdr.h).
Importing the package does not need a GPU; constructing any of the classes does (there is no CPU fallback).
"""
from .active_localizer import ActiveLocalizer  # noqa: F401
from .particle_filter import FilterParams, ParticleFilter  # noqa: F401
from .scan_renderer import ScanRenderer, ScanRendererPolar  # noqa: F401
from .synth import STATE_DTYPE  # noqa: F401
from .top_down_map import Params, TopDownMap, TopDownMapPolar  # noqa: F401
from .top_down_render_core import CoreConfig, PoseEst, TopDownRenderCore  # noqa: F401

__all__ = ["ActiveLocalizer", "FilterParams", "ParticleFilter", "ScanRenderer", "ScanRendererPolar", "Params", "TopDownMap",
           "TopDownMapPolar", "STATE_DTYPE", "CoreConfig", "PoseEst", "TopDownRenderCore"]
