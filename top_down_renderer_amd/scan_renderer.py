"""ScanRenderer / ScanRendererPolar — host-side mirrors of the reference classes (include/top_down_render/scan_renderer.h:14-23,
scan_renderer_polar.h:15-22) on top of the LDS-tile raster kernel."""
import numpy as np
import torch


class ScanRenderer:
    def __init__(self, flatten_lut, kernels=None):
        if kernels is None:
            from .kernels import HipKernels
            kernels = HipKernels()
        self.k = kernels
        lut = np.asarray(flatten_lut, np.int32).ravel()
        if lut.size != 256:
            raise ValueError("flatten_lut must have 256 entries (src/top_down_render.cpp:57)")
        self.flatten_lut_ = self.k.to_device(lut)
        self._last = None

    def _points(self, cloud):
        """cloud: (n,4) packed x,y,z,class or (n,8) pcl::PointXYZI-strided float32, host array or device tensor."""
        if isinstance(cloud, np.ndarray):
            cloud = self.k.to_device(np.ascontiguousarray(cloud, np.float32))
        if cloud.dim() != 2 or cloud.shape[1] not in (4, 8):
            raise ValueError("cloud must be (n,4) xyzi or (n,8) pcl::PointXYZI")
        stride = cloud.shape[1]
        return cloud.contiguous(), cloud.shape[0], stride, (3 if stride == 4 else 4)

    @staticmethod
    def _fill(imgs, img_dev, rows, cols):
        if imgs is None:
            return
        host = img_dev.cpu().numpy()
        for c, im in enumerate(imgs):   # caller-owned outputs, written in place like the reference
            if c >= host.shape[0]:
                break
            np.copyto(im, host[c].reshape(rows, cols, order="F"))

    def renderSemanticTopDown(self, cloud, res, imgs=None):
        """src/scan_renderer.cpp:55-78.  imgs: list of (rows, cols) arrays filled in place (may be None: the render
        then stays on the device, see last_scan())."""
        if imgs is not None and len(imgs) < 1:
            return   # :57
        pts, n, stride, ioff = self._points(cloud)
        ncls, (rows, cols) = self._shape(imgs)
        img, pk = self.k.raster_cart(pts, n, stride, ioff, float(res), self.flatten_lut_, ncls, rows, cols)
        self._last = (img, pk)
        self._fill(imgs, img, rows, cols)

    def _organised(self, cloud, width, height):
        pts, n, stride, _ = self._points(cloud)
        if width is None:
            width, height = n, 1          # an unorganised cloud: pcl sets width = size, height = 1
        if width * height != n:
            raise ValueError("cloud has %d points, width x height = %d x %d" % (n, width, height))
        return pts, stride, int(width), int(height)

    def renderGeometricTopDown(self, cloud, res, imgs, width=None, height=None):
        """src/scan_renderer.cpp:7-53: ground (imgs[0]) / obstacle (imgs[1]) counts; every column of the organised
        cloud (cloud->at(idx, idy) = element idy*width + idx) is one scan line.  Returns the (2, rows*cols) device images."""
        if len(imgs) < 2:
            return None   # :8
        pts, stride, width, height = self._organised(cloud, width, height)
        rows, cols = imgs[0].shape
        img = self.k.raster_geo(pts, stride, width, height, float(res), 0.0, rows, cols, polar=False)
        self._fill(imgs[:2], img, rows, cols)
        for im in imgs[2:]:
            im[...] = 0          # `imgs[i].setZero()` for every image (:12-14)
        return img

    def _shape(self, imgs):
        if imgs is None:
            return self.default_shape
        return len(imgs), imgs[0].shape

    def set_output_shape(self, ncls, rows, cols):
        """Shape used when renderSemanticTopDown is called without host images."""
        self.default_shape = (int(ncls), (int(rows), int(cols)))

    def last_scan(self):
        """The last render as a packed device scan, accepted by ParticleFilter.update without a host round trip."""
        return ("pk", self._last[1])

    def last_images(self):
        return self._last[0]


class ScanRendererPolar(ScanRenderer):
    def renderSemanticTopDown(self, cloud, res, ang_res, imgs=None):
        """src/scan_renderer_polar.cpp:83-109."""
        if imgs is not None and len(imgs) < 1:
            return   # :85
        pts, n, stride, ioff = self._points(cloud)
        ncls, (nb, nr) = self._shape(imgs)
        img, pk = self.k.raster_polar(pts, n, stride, ioff, float(res), float(ang_res), self.flatten_lut_, ncls, nb, nr)
        self._last = (img, pk)
        self._fill(imgs, img, nb, nr)

    def renderGeometricTopDown(self, cloud, res, ang_res, imgs, width=None, height=None):
        """src/scan_renderer_polar.cpp:6-81: per theta bin the returns sorted by range descending and walked (equal
        ranges in input order).  imgs[0] ground, imgs[1] obstacles, (theta bins x range bins)."""
        if len(imgs) < 2:
            return None   # :8
        pts, stride, width, height = self._organised(cloud, width, height)
        nb, nr = imgs[0].shape
        img = self.k.raster_geo(pts, stride, width, height, float(res), float(ang_res), nb, nr, polar=True)
        self._fill(imgs[:2], img, nb, nr)
        for im in imgs[2:]:
            im[...] = 0
        return img
