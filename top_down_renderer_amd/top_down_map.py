"""TopDownMap / TopDownMapPolar — host-side mirrors of the reference classes (include/top_down_render/top_down_map.h:52-102,
top_down_map_polar.h:6-22) holding the map on the device for the HIP kernels.

Map content is taken in the form the reference's load-time code produces (src/top_down_map.cpp:289-326): per-class
truncated distance maps + unknown mask.  Loading SVG/PNG maps and the on-disk caches is load-time work outside the
per-scan path.
"""
from dataclasses import dataclass, field

import numpy as np
import torch


@dataclass
class Params:
    """TopDownMap::Params (top_down_map.h:54-62), minus the load-time colour LUT."""
    map_path: str = ""
    flatten_lut: list = field(default_factory=list)
    num_classes: int = 0
    exclusive_classes: list = field(default_factory=list)
    resolution: float = 1.0
    out_of_bounds_const: float = 5.0   # read but unused by the reference: every OOB write is a literal 0


class TopDownMap:
    def __init__(self, params, class_maps=None, class_mask=None, kernels=None):
        """class_maps: (ncls, H, W) float32 indexed [cls, row(y), col(x)]; class_mask: (H, W) uint8, 1 = unknown."""
        if kernels is None:
            from .kernels import HipKernels
            kernels = HipKernels()   # raises without the built extension / a GPU: no CPU fallback
        self.k = kernels
        self.params_ = params
        self.map_center_ = (0, 0)
        self.have_map_ = False
        self.dev = None
        self.geo_constant_one_ = False   # True after updateMap(label image): the reference leaves geo_maps_ at 1 there
        if class_maps is not None:
            self._load(class_maps, class_mask)

    def _load(self, class_maps, class_mask):
        class_maps = np.ascontiguousarray(class_maps, np.float32)
        class_mask = np.ascontiguousarray(class_mask, np.uint8)
        ncls, H, W = class_maps.shape
        if self.params_.num_classes and self.params_.num_classes != ncls:
            raise ValueError("num_classes does not match the class maps")
        self.params_.num_classes = ncls
        self.rows, self.cols = H, W
        # host copy in the reference's column-major layout (class_maps_): needed by getClassesAtPoint / particle init
        self.maps_cm_host = np.ascontiguousarray(np.transpose(class_maps, (0, 2, 1)))
        self.dev = self.k.make_map(class_maps, class_mask, self.params_.resolution)
        self.have_map_ = True

    # top_down_map.cpp:146-157
    def updateMap(self, class_maps, class_mask=None, map_center=(0, 0)):
        """updateMap(label_img, map_center) like the reference (a cv::Mat of class ids), or
        updateMap(class_maps, class_mask, map_center) with distance maps computed elsewhere."""
        if class_mask is None or np.ndim(class_maps) == 2:
            if class_mask is not None and np.ndim(class_mask) == 1:
                map_center = class_mask
            return self.loadCompressedRasterMap(class_maps, map_center)
        self.map_center_ = (int(map_center[0]), int(map_center[1]))
        old = self.dev
        self.geo_constant_one_ = False
        self._load(class_maps, class_mask)
        if old is not None and getattr(old, "nb", 0):
            self.k.set_polar_table(self.dev, old.nb, old.nr, old.ang_res)

    # top_down_map.cpp:116-144 + 289-326, and updateMap :146-157 for a class-index image (cv::Mat CV_8UC1)
    def loadCompressedRasterMap(self, label_img, map_center=(0, 0)):
        """label_img: (H, W) uint8 image of raw class ids, row 0 = top like a cv::Mat; params_.flatten_lut maps raw id
        -> flattened class.  Class binary maps, the exact Euclidean distance transform, truncation at 50 and the
        unknown mask are all computed on the GPU."""
        p = self.params_
        if not p.num_classes or not len(p.flatten_lut):
            raise ValueError("Params.num_classes and Params.flatten_lut are needed to ingest a label image")
        old = self.dev
        self.geo_constant_one_ = True            # loadCompressedRasterMap :126-133, never recomputed by updateMap
        self.dev = self.k.make_map_from_labels(label_img, p.flatten_lut, p.num_classes, p.resolution)
        self.rows, self.cols = self.dev.rows, self.dev.cols
        maps_cm, _ = self.k.unpack_map(self.dev)
        self.maps_cm_host = maps_cm            # column-major host copy (class_maps_), for getClassesAtPoint / init
        self.map_center_ = (int(map_center[0]), int(map_center[1]))
        # `if (!class_maps_[1].isZero(0)) have_map_ = true` (:150): a map without any road is not usable
        if p.num_classes > 1 and bool((self.maps_cm_host[1] != 0).any()):
            self.have_map_ = True
        if old is not None and getattr(old, "nb", 0):
            self.k.set_polar_table(self.dev, old.nb, old.nr, old.ang_res)

    # top_down_map.cpp:197-224 — the raster cache: a directory of class<i>.png (8-bit grey, 0 inside the class, flipped)
    def saveRasterizedMaps(self, path):
        import os
        os.makedirs(path, exist_ok=True)
        maps_cm, mask_cm = self.k.unpack_map(self.dev)
        for c in range(self.dev.ncls):
            inside = (maps_cm[c].T == 0) & (mask_cm.T == 0)                  # [row][col], row 0 = bottom
            self.k.png_write_gray8(os.path.join(path, f"class{c}.png"), np.where(inside, 0, 255).astype(np.uint8)[::-1])

    def loadRasterizedMaps(self, map_path, map_center=(0, 0)):
        """The constructor's path for a raster-cache directory (:42-58): rasters -> distance maps on the GPU."""
        import os
        p = self.params_
        planes = np.stack([self.k.png_read_gray8(os.path.join(map_path, f"class{c}.png")) for c in range(p.num_classes)])
        old = self.dev
        self.geo_constant_one_ = False
        self.dev = self.k.make_map_from_rasters(planes, p.resolution)
        self.rows, self.cols = self.dev.rows, self.dev.cols
        maps_cm, _ = self.k.unpack_map(self.dev)
        self.maps_cm_host = maps_cm
        self.map_center_ = (int(map_center[0]), int(map_center[1]))
        self.have_map_ = True                                                # :63
        if old is not None and getattr(old, "nb", 0):
            self.k.set_polar_table(self.dev, old.nb, old.nr, old.ang_res)

    # top_down_map.cpp:159-175
    def getClassesAtPoint(self, center):
        cx, cy = center
        res = np.float32(self.params_.resolution)
        if not (isinstance(cx, (int, np.integer)) and isinstance(cy, (int, np.integer))):
            # Vector2f overload (:172-175) converts to an index and then calls the Vector2i overload
            cx, cy = int(np.float32(cx) / res), int(np.float32(cy) / res)
        c0, c1 = int(np.float32(cx) / res), int(np.float32(cy) / res)   # :160
        out = []
        for cls in range(self.params_.num_classes):
            if 0 <= c0 < self.cols and 0 <= c1 < self.rows and self.maps_cm_host[cls, c0, c1] < 1:
                out.append(cls)
        return out

    # top_down_map.cpp:429-459
    def getLocalMap(self, center, rot, res, shape):
        """Cartesian window of one pose: (dists [ncls] list of (rows, cols) arrays, mask (rows, cols) uint8, 1 = unknown
        / outside).  `shape` = (rows, cols) stands for the sizes the reference reads off the caller's arrays."""
        rows, cols = int(shape[0]), int(shape[1])
        d, k = self.k.local_map(self.dev, False, float(center[0]), float(center[1]), float(rot), float(res), rows, cols)
        d = d.cpu().numpy().reshape(self.dev.ncls, cols, rows)
        return [d[c].T.copy() for c in range(self.dev.ncls)], k.cpu().numpy().reshape(cols, rows).T.copy()

    # top_down_map.cpp:461-481
    def getLocalGeoMap(self, center, rot, res, shape):
        """The Cartesian window gathered from the two geometric layers geo_maps_: list of two (rows, cols) arrays."""
        rows, cols = int(shape[0]), int(shape[1])
        g = self.geo_dev()
        d, _ = self.k.local_map(g, False, float(center[0]), float(center[1]), float(rot), float(res), rows, cols)
        d = d.cpu().numpy().reshape(2, cols, rows)
        return [d[c].T.copy() for c in range(2)]

    def geo_dev(self):
        """geo_maps_ on the device (built on first use): derived from the class maps like the static-map constructor does
        (src/top_down_map.cpp:48-58), or the constant 1 the updateMap path leaves (:126-133)."""
        return self.dev.geo_map(self.k, self.geo_constant_one_)

    # --- window (image) shape of the scan the filter scores against -------------------------------------------------
    polar = False

    def setWindow(self, rows, cols):
        """Cartesian window shape (rows = y, cols = x) for the Cartesian score (BASELINE config 4)."""
        self.win_rows, self.win_cols = int(rows), int(cols)

    def window_shape(self):
        return (self.win_rows, self.win_cols)

    def scan_handle(self, scan):
        """Accepts what ParticleFilter::update is handed in the reference (a list of per-class column-major images),
        an (ncls, rows*cols) array / device tensor, or an already packed device scan; returns the packed device scan
        the scoring kernel reads."""
        ncls = self.numClasses()
        nb, nr = self.window_shape()
        if isinstance(scan, tuple) and scan[0] == "pk":
            return scan[1]
        if isinstance(scan, (list, tuple)):
            if len(scan) < ncls:
                raise ValueError("fewer scan images than map classes")
            scan = np.stack([np.asarray(s, np.float32).reshape(nb, nr, order="A").ravel(order="F") for s in scan[:ncls]])
        if isinstance(scan, np.ndarray):
            if scan.shape != (ncls, nb * nr):
                raise ValueError(f"scan shape {scan.shape} != {(ncls, nb * nr)}")
            scan = self.k.to_device(np.ascontiguousarray(scan, np.float32))
        if isinstance(scan, torch.Tensor):
            if tuple(scan.shape) != (ncls, nb * nr):
                raise ValueError(f"scan shape {tuple(scan.shape)} != {(ncls, nb * nr)}")
            return self.k.pack_scan(scan.contiguous(), ncls, nb, nr)
        raise TypeError("unsupported scan type")

    def numClasses(self):
        return self.params_.num_classes

    def size(self):
        return (self.cols, self.rows)

    def mapCenter(self):
        return self.map_center_

    def resolution(self):
        return self.params_.resolution

    def haveMap(self):
        return self.have_map_


class TopDownMapPolar(TopDownMap):
    def __init__(self, params, class_maps=None, class_mask=None, kernels=None):
        super().__init__(params, class_maps, class_mask, kernels)
        if self.have_map_:
            self.samplePtsPolar((100, 50), np.float32(2 * np.pi / 100))   # top_down_map_polar.cpp:3-5

    def samplePtsPolar(self, shape, ang_res):
        """top_down_map_polar.cpp:7-19; shape = (theta bins, range bins)."""
        self.k.set_polar_table(self.dev, int(shape[0]), int(shape[1]), float(ang_res))

    # top_down_map_polar.cpp:21-53 (and the 3-argument overload with scale = 1, :78-81)
    def getLocalMap(self, center, scale_or_res, res=None):   # noqa: D401  (getLocalMap(center, scale, res) / (center, res))
        scale, res = (1.0, scale_or_res) if res is None else (scale_or_res, res)
        d, k = self.k.local_map(self.dev, True, float(center[0]), float(center[1]), float(scale), float(res))
        nb, nr = self.nb, self.nr
        d = d.cpu().numpy().reshape(self.dev.ncls, nr, nb)
        return [d[c].T.copy() for c in range(self.dev.ncls)], k.cpu().numpy().reshape(nr, nb).T.copy()

    # top_down_map_polar.cpp:55-76 (and the 3-argument overload, :83-86)
    def getLocalGeoMap(self, center, scale_or_res, res=None):   # noqa: D401
        scale, res = (1.0, scale_or_res) if res is None else (scale_or_res, res)
        d, _ = self.k.local_map(self.geo_dev(), True, float(center[0]), float(center[1]), float(scale), float(res))
        d = d.cpu().numpy().reshape(2, self.nr, self.nb)
        return [d[c].T.copy() for c in range(2)]

    polar = True

    @property
    def nb(self):
        return self.dev.nb

    @property
    def nr(self):
        return self.dev.nr

    def window_shape(self):
        return (self.nb, self.nr)
