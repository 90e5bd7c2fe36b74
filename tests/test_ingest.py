"""Map ingest: the .eig cache format, and label image -> per-class truncated distance maps.
CPU: format + oracle restatement properties.  GPU: the device ingest (exact EDT) against the oracle, bit for bit."""
import os
import tempfile

import numpy as np
import pytest

from oracle import np_oracle as no
from top_down_renderer_amd import eig_io, synth


def _label_image(rng, h, w, ncls, raw_ids):
    lab = synth.make_label_image(max(h, w), ncls, rng)[:h, :w]
    img = np.where(lab >= 0, np.asarray(raw_ids)[np.maximum(lab, 0)], 200).astype(np.uint8)   # 200: unlabelled
    return img[::-1].copy()   # cv::Mat: row 0 = top, the map's row 0 = bottom


def test_eig_format_round_trip_and_layout():
    d = tempfile.mkdtemp()
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    eig_io.write_eig(os.path.join(d, "a.eig"), a)
    raw = open(os.path.join(d, "a.eig"), "rb").read()
    assert np.frombuffer(raw[:16], "<i8").tolist() == [3, 4]                      # Index rows, Index cols
    assert np.frombuffer(raw[16:], "<f4").tolist() == a.ravel(order="F").tolist()   # column-major scalars
    assert np.array_equal(eig_io.read_eig(os.path.join(d, "a.eig"), np.float32), a)
    assert np.array_equal(no.read_eig(os.path.join(d, "a.eig"), np.float32), a)     # the oracle reads the same bytes
    m = (np.arange(20).reshape(5, 4) % 2).astype(np.uint8)
    no.write_eig(os.path.join(d, "m.eig"), m)
    assert np.array_equal(eig_io.read_eig(os.path.join(d, "m.eig"), np.uint8), m)
    with pytest.raises(ValueError):
        eig_io.read_eig(os.path.join(d, "m.eig"), np.float32)                     # wrong scalar size is caught
    maps = np.random.default_rng(0).random((3, 6, 5)).astype(np.float32)
    eig_io.save_cached_maps(os.path.join(d, "cache"), "/some/map.png", maps, np.zeros((6, 5), np.uint8), 1.0)
    got, mask = eig_io.load_cached_maps(os.path.join(d, "cache"), 3)
    assert np.array_equal(got, maps) and mask.shape == (6, 5)
    assert open(os.path.join(d, "cache", "cached_data.txt")).read().split("\n")[:3] == ["/some/map.png", "3", "1"]


def test_oracle_ingest_matches_generator_convention():
    rng = np.random.default_rng(3)
    lab = synth.make_label_image(200, 4, rng)
    raw_ids = [7, 3, 11, 5]
    lut = -np.ones(16, np.int32)
    lut[raw_ids] = np.arange(4)
    img = np.where(lab >= 0, np.asarray(raw_ids)[np.maximum(lab, 0)], 15).astype(np.uint8)[::-1].copy()
    maps, mask = no.load_compressed_raster_map(img, lut, 4, 1.0)
    ref_maps, ref_mask = synth.label_to_maps(lab, 4, 1.0)
    assert np.array_equal(mask, ref_mask) and np.array_equal(maps, ref_maps)
    assert maps.max() <= 50 and np.all(maps[:, mask == 1] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,ncls,resol", [(300, 420, 4, 1.0), (97, 64, 6, 1.0), (120, 90, 3, 0.5), (200, 260, 2, 2.0)])
def test_gpu_ingest_matches_oracle_exactly(h, w, ncls, resol):
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd.kernels import HipKernels
    k = HipKernels()
    rng = np.random.default_rng(h + w)
    raw_ids = list(rng.permutation(40)[:ncls])
    lut = -np.ones(64, np.int32)
    lut[raw_ids] = np.arange(ncls)
    img = _label_image(rng, h, w, ncls, raw_ids)
    ref_maps, ref_mask = no.load_compressed_raster_map(img, lut, ncls, resol)
    m = pkg.TopDownMapPolar(pkg.Params(resolution=resol, num_classes=ncls, flatten_lut=list(lut)), kernels=k)
    assert not m.haveMap()
    m.updateMap(img, (5, 7))
    assert m.haveMap() and m.mapCenter() == (5, 7)
    assert m.size() == (ref_maps.shape[2], ref_maps.shape[1])
    maps_cm, mask_cm = k.unpack_map(m.dev)
    assert np.array_equal(mask_cm.T, ref_mask)
    assert np.array_equal(np.transpose(maps_cm, (0, 2, 1)), ref_maps)           # bit-exact distances
    # a class that is absent everywhere saturates at 50 on known cells
    img2 = np.full((40, 50), raw_ids[0], np.uint8)
    m.updateMap(img2, (0, 0))
    maps2, mask2 = k.unpack_map(m.dev)
    assert not mask2.any() and np.all(maps2[0] == 0) and np.all(maps2[1:] == 50)


@pytest.mark.gpu
def test_ingested_map_scores_like_an_uploaded_one(oracle):
    """updateMap(label image) on the GPU followed by a filter update == uploading the oracle's distance maps."""
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd.kernels import HipKernels
    k = HipKernels()
    sc = synth.make_scene("c1", n_particles=512)
    cfg = sc.cfg
    lut = synth.make_lut(cfg.ncls)
    img = np.where(sc.lab >= 0, sc.lab, 250).astype(np.uint8)[::-1].copy()
    m1 = pkg.TopDownMapPolar(pkg.Params(resolution=1.0, num_classes=cfg.ncls, flatten_lut=list(lut[:16])), kernels=k)
    m1.updateMap(img, (0, 0))
    m1.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    m2 = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), sc.class_maps, sc.class_mask, kernels=k)
    m2.samplePtsPolar((cfg.nb, cfg.nr), cfg.ang_res)
    scan = oracle.raster_polar(sc.pts, cfg.res, cfg.ang_res, sc.lut, cfg.ncls, cfg.nb, cfg.nr)
    out = []
    for m in (m1, m2):
        f = pkg.ParticleFilter(512, m, pkg.FilterParams(fixed_scale=1.0), kernels=k, init_particles=False)
        f.set_states(sc.states)
        f.update(scan, None, cfg.res)
        out.append(f.raw_weights())
    assert np.array_equal(out[0], out[1], equal_nan=True)
