"""TopDownRenderCore — the ROS-free part of the reference's orchestrator `TopDownRender`
(include/top_down_render/top_down_render.h:36-108, src/top_down_render.cpp) over the Python classes of this package: what
the node does with the hot-path classes once a cloud and a motion prior have arrived.  The C++ form of the same class is
include/top_down_render/top_down_render_core.h; both follow

    initialize      :81, 115-117   samplePtsPolar(100 x 25) / ParticleFilter / ScanRendererPolar
    takeStep        :505-560       render at current_range_scale_, updateFilter, publishPoseEst
    updateFilter    :413-425       propagate(trans, yaw), update(top_down, top_down_geo, res)
    publishPoseEst  :331-365       range-scale stepping (+0.05 / -0.02 inside [range_scale_min_, range_scale_max_]), the
                                   freezeScale trigger (cov(3,3) < 0.003 scale), the convergence gate

The node changes `res` on EVERY step, so a filter is scored with a different range scale scan after scan
(tests/test_takestep_loop.py).  The scan stays on the device between the renderer and the filter."""
from dataclasses import dataclass, field

import numpy as np

from .particle_filter import ParticleFilter
from .scan_renderer import ScanRendererPolar


@dataclass
class CoreConfig:                       # the node's parameters that reach the step (src/top_down_render.cpp:45-53)
    particle_count: int = 20000         # :53
    range_scale_min: float = 0.5        # :45
    range_scale_max: float = 4.0        # :46
    target_uncertainty_m: float = 2.5   # top_down_render.h:80
    theta_bins: int = 100               # hard-coded 100 x 25 in the node (:115, 530, 534); parameters here
    range_bins: int = 25
    seed: int = 0                       # ParticleFilter's seed


@dataclass
class PoseEst:                          # what publishPoseEst computed this step
    cov: np.ndarray = field(default_factory=lambda: np.zeros((4, 4), np.float32))   # computeMeanCov (:333)
    ml_state: np.ndarray = None         # meanLikelihood (:354); None when the filter holds no particle (:347-350)
    scale: float = -1.0                 # filter_->scale() at :335
    range_scale: float = 0.0            # current_range_scale_ AFTER the step's adjustment: the next scan's res
    froze_scale: bool = False           # freezeScale() was called in this step (:356-359)
    converged: bool = False             # is_converged_ (:362-364; sticky)


class TopDownRenderCore:
    def __init__(self, cfg=None, kernels=None):
        self.cfg = cfg or CoreConfig()
        self.k = kernels
        self.current_range_scale_ = np.float32(self.cfg.range_scale_max)   # :47
        self.last_res_ = np.float32(0)
        self.is_converged_ = False                                         # top_down_render.h:83
        self.map_ = self.filter_ = self.renderer_ = None

    def initialize(self, map, filter_params, flatten_lut, **filter_kw):
        """:81, 115-117 with a TopDownMapPolar the host built."""
        c = self.cfg
        self.map_ = map
        self.ang_res = np.float32(2 * np.pi / c.theta_bins)
        map.samplePtsPolar((c.theta_bins, c.range_bins), self.ang_res)                                     # :115
        self.filter_ = ParticleFilter(c.particle_count, map, filter_params, seed=c.seed, kernels=self.k, **filter_kw)   # :116
        self.renderer_ = ScanRendererPolar(flatten_lut, kernels=self.k)                                    # :117
        self.renderer_.set_output_shape(map.numClasses(), c.theta_bins, c.range_bins)

    def takeStep(self, cloud, trans, yaw):
        """takeStep (:505-560).  trans / yaw: the motion prior's delta projected to the plane like updateFilter does
        (:418-420, projectPrior).  Returns the step's PoseEst, or None when the step was skipped (no map yet, :508-511)."""
        if self.map_ is None or not self.map_.haveMap():
            return None
        self.last_res_ = self.current_range_scale_
        self.renderer_.renderSemanticTopDown(cloud, float(self.current_range_scale_), self.ang_res)         # :539
        self.updateFilter(self.renderer_.last_scan(), None, float(self.current_range_scale_), trans, yaw)   # :559
        return self.publishPoseEst()                                                                        # :560

    def updateFilter(self, top_down, top_down_geo, res, trans, yaw):
        self.filter_.propagate(trans, yaw)                                                                  # :423
        self.filter_.update(top_down, top_down_geo, res)                                                    # :425

    def publishPoseEst(self):
        """publishPoseEst (:331-365) without the publishing."""
        f, c = self.filter_, self.cfg
        e = PoseEst()
        e.cov = np.asarray(f.computeMeanCov(), np.float32)                                                  # :333
        scale = np.float32(f.scale())                                                                       # :335
        scale_2 = np.float32(scale * scale)
        e.scale = float(scale)
        spread = np.float32(max(e.cov[0, 0], e.cov[1, 1]) / scale_2)
        # (the node compares against std::pow(float, int), a double, and steps its float member by double constants)
        if float(spread) > float(np.float32(c.target_uncertainty_m)) ** 2 and self.current_range_scale_ < np.float32(c.range_scale_max):
            self.current_range_scale_ = np.float32(np.float64(self.current_range_scale_) + 0.05)             # :341 widen
        elif self.current_range_scale_ > np.float32(c.range_scale_min):
            self.current_range_scale_ = np.float32(np.float64(self.current_range_scale_) - 0.02)             # :344 shrink
        e.range_scale = float(self.current_range_scale_)
        e.converged = self.is_converged_
        if f.numParticles() < 1:                                                                            # :347-350
            return e
        e.ml_state = np.asarray(f.meanLikelihood(), np.float32)                                             # :354
        if float(e.cov[3, 3]) < 0.003 * float(e.ml_state[3]) and not f.isScaleFrozen():                      # :356
            f.freezeScale()                                                                                 # :359
            e.froze_scale = True
        if (np.float32(e.cov[0, 0] / scale_2) < 40 and np.float32(e.cov[1, 1] / scale_2) < 40 and e.cov[2, 2] < 0.5
                and f.scale() > 0):                                                                         # :363
            self.is_converged_ = True
        e.converged = self.is_converged_
        return e

    @staticmethod
    def projectPrior(R, t):
        """The plane projection of a 3-D motion prior (updateFilter, :418-420): R 3 x 3, t the translation."""
        R = np.asarray(R, np.float32).reshape(3, 3)
        return (float(t[0]), float(t[1])), float(np.arctan2(R[1, 0], R[0, 0]))

    def currentRangeScale(self):
        return float(self.current_range_scale_)

    def lastRes(self):
        return float(self.last_res_)

    def isConverged(self):
        return self.is_converged_
