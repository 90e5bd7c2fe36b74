"""Seeded synthetic maps, scans and particle sets of the shapes BASELINE.json names.

NumPy/SciPy only; used by bench.py and the tests.  Nothing here is on the timed path.

Map content follows what the reference's load-time code produces (src/top_down_map.cpp:116-144, 289-326):
per-class `min(50, resolution * L2 distance to the nearest cell of that class)`, 0 where no class is present,
and a u8 mask that is 1 on such unknown cells.  Class index 1 is "road" (src/state_particle.cpp:29).
"""
from dataclasses import dataclass

import numpy as np

STATE_DTYPE = np.dtype(
    [("init_x_px", "<f4"), ("init_y_px", "<f4"), ("dx_m", "<f4"), ("dy_m", "<f4"), ("theta", "<f4"),
     ("scale", "<f4"), ("have_init", "u1"), ("pad", "u1", (3,))]
)


@dataclass
class Config:
    name: str
    n_pts: int
    ncls: int
    nb: int          # image rows: theta bins (polar) / y rows (Cartesian)
    nr: int          # image cols: range bins (polar) / x cols (Cartesian)
    map_size: int
    n_particles: int
    polar: bool = True
    have_init: bool = True
    seed: int = 1234
    res: float = 1.0           # metres per range bin
    map_resolution: float = 1.0

    @property
    def ang_res(self):
        return np.float32(2 * np.pi / self.nb)


CONFIGS = {
    # BASELINE.json configs[0..4]
    "c1": Config("c1", 10_000, 3, 128, 128, 1000, 1_000, seed=1234),
    "c2": Config("c2", 100_000, 6, 256, 256, 4000, 100_000, seed=1235),
    "c3": Config("c3", 100_000, 6, 256, 256, 4000, 1_000_000, seed=1236),
    "c4": Config("c4", 100_000, 6, 512, 512, 8000, 200_000, polar=False, seed=1237),
    "c5": Config("c5", 100_000, 6, 256, 256, 4000, 2_000_000, have_init=False, seed=1238),
    # the reference node's own defaults: 20 000 particles (src/top_down_render.cpp:53), 100x25 polar image
    # (samplePtsPolar(Vector2i(100, 25), 2*pi/100), :115); latency-bound on a GPU
    "ref": Config("ref", 30_000, 6, 100, 25, 2000, 20_000, seed=1239, res=4.0),
    # micro shape for fixtures and pure-Python cross-checks
    "micro": Config("micro", 64, 3, 16, 8, 48, 32, seed=1230),
}


def make_label_image(size, ncls, rng):
    """(H, W) int8 label image, -1 = unlabelled (~10 %).  Class 1 = road grid; other classes = rectangles."""
    H = W = size
    lab = np.full((H, W), -1, np.int8)
    others = [c for c in range(ncls) if c != 1]
    # background: coarse random blocks of the non-road classes
    blk = max(8, size // 40)
    gh = (H + blk - 1) // blk
    coarse = rng.integers(0, len(others), (gh, gh))
    bg = np.asarray(others, np.int8)[coarse]
    lab[:, :] = np.kron(bg, np.ones((blk, blk), np.int8))[:H, :W]
    # a few hundred random axis-aligned rectangles
    nrect = max(20, (size * size) // 40000)
    for _ in range(nrect):
        h, w = rng.integers(max(4, size // 100), max(8, size // 12), 2)
        y, x = rng.integers(0, H - 1), rng.integers(0, W - 1)
        lab[y:y + h, x:x + w] = others[rng.integers(0, len(others))]
    # unlabelled holes, ~10 % of the area
    nhole = max(6, (size * size) // 100000)
    hole_side = int(np.sqrt(0.10 * size * size / nhole))
    for _ in range(nhole):
        y, x = rng.integers(0, H - 1), rng.integers(0, W - 1)
        lab[y:y + hole_side, x:x + hole_side] = -1
    # road grid: 8-12 px wide roads every ~150 px (scaled down for tiny maps)
    pitch = min(150, max(12, size // 4))
    for y in range(pitch // 2, H, pitch):
        wd = int(rng.integers(8, 13)) if size >= 200 else 3
        lab[y:y + wd, :] = 1
    for x in range(pitch // 2, W, pitch):
        wd = int(rng.integers(8, 13)) if size >= 200 else 3
        lab[:, x:x + wd] = 1
    return lab


def label_to_maps(lab, ncls, resolution=1.0):
    """Label image -> (class_maps (ncls,H,W) f32, class_mask (H,W) u8) like computeDists (top_down_map.cpp:289-326)."""
    from scipy.ndimage import distance_transform_edt

    H, W = lab.shape
    maps = np.empty((ncls, H, W), np.float32)
    unknown = lab < 0
    for c in range(ncls):
        binary = lab != c  # 0 inside the class, 1 elsewhere
        if binary.all():
            d = np.full((H, W), 50.0, np.float32)
        else:
            d = distance_transform_edt(binary).astype(np.float32)
        d = np.minimum(d * np.float32(resolution), np.float32(50))
        d[unknown] = 0
        maps[c] = d
    return maps, unknown.astype(np.uint8)


def make_map(cfg, rng=None):
    rng = np.random.default_rng(cfg.seed) if rng is None else rng
    lab = make_label_image(cfg.map_size, cfg.ncls, rng)
    maps, mask = label_to_maps(lab, cfg.ncls, cfg.map_resolution)
    return lab, maps, mask


def make_lut(ncls):
    """256-entry class-id -> flattened-class LUT (-1 = ignore), like flatten_lut_ (src/top_down_render.cpp:56-62).
    Raw ids 0..ncls-1 map to themselves, raw id ncls and everything above map to -1."""
    lut = -np.ones(256, np.int32)
    lut[:ncls] = np.arange(ncls)
    return lut


def pick_true_pose(lab, rng, margin):
    """A pose on a road cell, at least `margin` px from the border."""
    H, W = lab.shape
    ys, xs = np.nonzero(lab[margin:H - margin, margin:W - margin] == 1)
    k = rng.integers(0, len(ys))
    return float(xs[k] + margin) + 0.25, float(ys[k] + margin) + 0.25, float(rng.uniform(-np.pi, np.pi))


def make_scan(cfg, lab, pose, rng, scale=1.0):
    """(n,4) float32 points x,y,z,class-id as seen from `pose` = (cx_px, cy_px, theta)."""
    n = cfg.n_pts
    cx, cy, th = pose
    rmax = cfg.nr * cfg.res if cfg.polar else 0.5 * min(cfg.nb, cfg.nr) * cfg.res
    # 64 "rings": range clusters like the beams of a spinning LiDAR, half-normal envelope
    ring_r = np.abs(rng.normal(0, 0.35 * rmax, 64)).clip(0.5, rmax * 0.999)
    r = ring_r[rng.integers(0, 64, n)] + rng.normal(0, 0.02 * rmax, n)
    r = np.clip(np.abs(r), 0.5, rmax * 0.999)
    ang = rng.uniform(-np.pi, np.pi, n)
    x = (r * np.sin(ang)).astype(np.float32)
    y = (r * np.cos(ang)).astype(np.float32)
    z = rng.normal(0, 1, n).astype(np.float32)
    # label from the map around the true pose: scan angle a looks along map direction (a - theta)
    phi = ang - th
    row = np.rint(cy + np.cos(phi) * r * scale).astype(np.int64)
    col = np.rint(cx + np.sin(phi) * r * scale).astype(np.int64)
    H, W = lab.shape
    inb = (row >= 0) & (row < H) & (col >= 0) & (col < W)
    cls = np.full(n, cfg.ncls, np.int64)  # raw id `ncls` -> LUT -1
    cls[inb] = lab[row[inb], col[inb]]
    cls[cls < 0] = cfg.ncls
    drop = rng.random(n) < 0.05  # 5 % labels mapped to -1 by the LUT
    cls[drop] = cfg.ncls + rng.integers(0, 3, int(drop.sum()))
    origin = rng.random(n) < 0.01  # 1 % points exactly (0,0,z): exercises the skip
    x[origin] = 0
    y[origin] = 0
    return np.stack([x, y, z, cls.astype(np.float32)], axis=1).astype(np.float32)


def make_particles(cfg, lab, pose, rng, n=None, sigma_px=30.0, sigma_deg=10.0, uniform_frac=0.10, scale=1.0):
    """Structured array of reference `State`s: Gaussian about the true pose plus a uniform fraction."""
    n = cfg.n_particles if n is None else n
    H, W = lab.shape
    st = np.zeros(n, STATE_DTYPE)
    nu = int(round(uniform_frac * n))
    cx, cy, th = pose
    sig = sigma_px * min(1.0, cfg.map_size / 1000.0)
    st["init_x_px"] = rng.normal(cx, sig, n)
    st["init_y_px"] = rng.normal(cy, sig, n)
    st["theta"] = rng.normal(th, np.deg2rad(sigma_deg), n)
    uni = rng.permutation(n)[:nu]
    st["init_x_px"][uni] = rng.uniform(0, W, nu)
    st["init_y_px"][uni] = rng.uniform(0, H, nu)
    st["theta"][uni] = rng.uniform(-np.pi, np.pi, nu)
    st["scale"] = scale
    st["have_init"] = 1 if cfg.have_init else 0
    return st


def make_cluster_particles(cfg, lab, rng, n_clusters=8, per_cluster=None, sigma_px=40.0):
    """BASELINE config 5: `n_clusters` init clusters on road cells spread over the map, have_init = False."""
    per_cluster = cfg.n_particles // n_clusters if per_cluster is None else per_cluster
    H, W = lab.shape
    st = np.zeros(n_clusters * per_cluster, STATE_DTYPE)
    for g in range(n_clusters):
        cx, cy, _ = pick_true_pose(lab, rng, margin=min(H // 8, 300))
        sl = slice(g * per_cluster, (g + 1) * per_cluster)
        st["init_x_px"][sl] = rng.normal(cx, sigma_px, per_cluster)
        st["init_y_px"][sl] = rng.normal(cy, sigma_px, per_cluster)
    st["scale"] = 1.0
    st["have_init"] = 0
    return st


@dataclass
class Scene:
    cfg: Config
    lab: np.ndarray
    class_maps: np.ndarray   # (ncls, H, W) f32, [cls, row(y), col(x)]
    class_mask: np.ndarray   # (H, W) u8, 1 = unknown
    lut: np.ndarray          # (256,) i32
    pose: tuple
    pts: np.ndarray          # (n_pts, 4) f32
    states: np.ndarray       # STATE_DTYPE


def make_scene(name_or_cfg, n_particles=None, with_particles=True):
    cfg = CONFIGS[name_or_cfg] if isinstance(name_or_cfg, str) else name_or_cfg
    rng = np.random.default_rng(cfg.seed)
    lab, maps, mask = make_map(cfg, rng)
    rmax = cfg.nr * cfg.res if cfg.polar else 0.5 * max(cfg.nb, cfg.nr) * cfg.res
    margin = int(min(cfg.map_size // 4, rmax + 40))
    pose = pick_true_pose(lab, rng, margin)
    pts = make_scan(cfg, lab, pose, rng)
    if not with_particles:
        states = np.zeros(0, STATE_DTYPE)
    elif cfg.have_init:
        states = make_particles(cfg, lab, pose, rng, n=n_particles)
    else:
        n = cfg.n_particles if n_particles is None else n_particles
        states = make_cluster_particles(cfg, lab, rng, per_cluster=max(1, n // 8))
    return Scene(cfg, lab, maps, mask, make_lut(cfg.ncls), pose, pts, states)
