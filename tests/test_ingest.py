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


# ---- the raster cache: class<i>.png (src/top_down_map.cpp:197-224) ---------------------------------------------------------
def _png_bytes(img, filters, colour_type=0, bit_depth=8, interlace=0, break_crc=False):
    """A PNG file made by hand (zlib + struct): one filter type per scanline, cycling through `filters`."""
    import struct
    import zlib
    h, w = img.shape
    raw = bytearray()
    prev = np.zeros(w, np.int64)
    for y in range(h):
        ft = filters[y % len(filters)]
        cur = img[y].astype(np.int64)
        left = np.concatenate([[0], cur[:-1]])
        upleft = np.concatenate([[0], prev[:-1]])
        if ft == 0:
            pred = np.zeros(w, np.int64)
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) // 2
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        raw.append(ft)
        raw += bytes(((cur - pred) % 256).astype(np.uint8))
        prev = cur

    def chunk(t, d):
        crc = zlib.crc32(t + d) ^ (1 if break_crc and t == b"IDAT" else 0)
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", crc)
    z = zlib.compress(bytes(raw), 9)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, colour_type, 0, 0, interlace)) +
            chunk(b"tEXt", b"Comment\x00by hand") + chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) +
            chunk(b"IEND", b""))


def _png_decode(data):
    """A PNG reader of the test's own (filter 0 only: what the library writes)."""
    import struct
    import zlib
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, w, h = 8, b"", 0, 0
    while at < len(data):
        n, t = struct.unpack(">I", data[at:at + 4])[0], data[at + 4:at + 8]
        d = data[at + 8:at + 8 + n]
        assert zlib.crc32(t + d) == struct.unpack(">I", data[at + 8 + n:at + 12 + n])[0]
        if t == b"IHDR":
            w, h, bd, ct, cm, fm, il = struct.unpack(">IIBBBBB", d)
            assert (bd, ct, cm, fm, il) == (8, 0, 0, 0, 0)
        elif t == b"IDAT":
            idat += d
        at += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w + 1)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].copy()


def test_png_codec_of_the_raster_cache(tmp_path):
    """8-bit greyscale PNG over zlib (csrc/tdr_png.cpp), no GPU: files written by hand with every filter type, split IDAT
    and an ancillary chunk are read exactly; what the library writes is a PNG an independent reader decodes; other kinds of
    PNG and damaged files are refused."""
    import ctypes as C
    from top_down_renderer_amd import _lib, build
    build.build()
    L = _lib.load()
    rng = np.random.default_rng(8)
    vp = C.c_void_p

    def read(path, cap=None):
        w, h = C.c_int(0), C.c_int(0)
        out = np.zeros(1 << 16 if cap is None else cap, np.uint8)
        rc = L.tdr_png_read_gray8_host(str(path).encode(), out.ctypes.data_as(vp), len(out), C.byref(w), C.byref(h))
        return rc, w.value, h.value, out
    for shape, filters in (((13, 29), [0, 1, 2, 3, 4]), ((1, 1), [4]), ((40, 7), [3, 3, 4]), ((5, 64), [2, 1])):
        img = rng.integers(0, 256, shape).astype(np.uint8)
        img[rng.random(shape) < 0.5] = 255
        p = tmp_path / "hand.png"
        p.write_bytes(_png_bytes(img, filters))
        rc, w, h, out = read(p)
        assert rc == 0 and (h, w) == shape and np.array_equal(out[: w * h].reshape(h, w), img)
        q = tmp_path / "lib.png"
        assert L.tdr_png_write_gray8_host(str(q).encode(), img.ctypes.data_as(vp), shape[1], shape[0]) == 0
        assert np.array_equal(_png_decode(q.read_bytes()), img)
        rc, w, h, out = read(q)
        assert rc == 0 and np.array_equal(out[: w * h].reshape(h, w), img)
    img = rng.integers(0, 256, (6, 6)).astype(np.uint8)
    bad = {"rgb": dict(colour_type=2), "16 bit": dict(bit_depth=16), "interlaced": dict(interlace=1), "crc": dict(break_crc=True)}
    for name, kw in bad.items():
        p = tmp_path / "bad.png"
        p.write_bytes(_png_bytes(img, [0], **kw))
        assert read(p)[0] == -1, name
        assert b"png" in L.tdr_last_error()
    p.write_bytes(_png_bytes(img, [0])[:-30])
    assert read(p)[0] == -1                                   # truncated
    p.write_bytes(_png_bytes(img, [1]))
    rc, w, h, _ = read(p, cap=10)
    assert rc == -1 and (w, h) == (6, 6)                      # buffer too small: the size is still reported
    assert read(tmp_path / "missing.png")[0] == -1
    # a header that announces 2^24 x 2^24 pixels over a few bytes of image data (valid CRCs): refused before anything of
    # that size is allocated — an error code, not std::bad_alloc through the C ABI
    import struct
    import zlib
    good = _png_bytes(img, [0])
    ihdr_at = good.index(b"IHDR")
    hdr = struct.pack(">II", 1 << 24, 1 << 24) + good[ihdr_at + 12: ihdr_at + 17]
    bomb = good[: ihdr_at + 4] + hdr + struct.pack(">I", zlib.crc32(b"IHDR" + hdr)) + good[ihdr_at + 21:]
    p.write_bytes(bomb)
    rc, w, h, _ = read(p)
    assert rc == -1 and b"announces" in L.tdr_last_error()


@pytest.mark.gpu
def test_raster_cache_round_trip_and_overlapping_classes(tmp_path):
    """saveRasterizedMaps / loadRasterizedMaps through the C ABI: a map ingested from a label image, saved as class<i>.png
    and loaded back is the same map (records array-equal); rasters whose classes OVERLAP (an SVG map's polygons may) give the
    distance maps and mask of computeDists — checked against scipy's exact Euclidean distance transform."""
    import ctypes as C
    from scipy.ndimage import distance_transform_edt
    import torch
    from top_down_renderer_amd import synth
    from top_down_renderer_amd._lib import check
    from top_down_renderer_amd.kernels import HipKernels
    k = HipKernels()
    L, vp = k.lib, C.c_void_p
    rng = np.random.default_rng(12)
    h, w, ncls = 150, 210, 4
    lab = synth.make_label_image(max(h, w), ncls, rng)[:h, :w]
    img = np.where(lab < 0, 255, lab).astype(np.uint8)[::-1].copy()       # label image, row 0 = top
    lut = np.full(256, -1, np.int32)
    lut[:ncls] = np.arange(ncls)

    def records(m):
        d = np.zeros((ncls, 30 * 40), np.float32)
        mk = np.zeros(30 * 40, np.uint8)
        out = []
        for cx, cy in ((60.0, 50.0), (5.0, 140.0), (200.0, 10.0)):
            check(L.tdr_map_local_map(m, 0, C.c_float(cx), C.c_float(cy), C.c_float(1.7), C.c_float(1.0), 30, 40,
                                      d.ctypes.data_as(vp), mk.ctypes.data_as(vp)))
            out.append((d.copy(), mk.copy()))
        return out
    m = vp()
    check(L.tdr_map_create(C.byref(m)))
    check(L.tdr_map_set_labels(m, img.ctypes.data_as(vp), h, w, lut.ctypes.data_as(vp), 256, ncls, C.c_float(1.0), 0, 0))
    d = str(tmp_path / "site_raster_cache").encode()
    check(L.tdr_map_save_rasters(m, d))
    m2 = vp()
    check(L.tdr_map_create(C.byref(m2)))
    check(L.tdr_map_load_rasters(m2, d, ncls, C.c_float(1.0), 0, 0))
    for (a, am), (b, bm) in zip(records(m), records(m2)):
        assert np.array_equal(a, b) and np.array_equal(am, bm)
    # overlapping classes, grey values off 0 / 255, a hole no class covers
    planes = np.full((3, 90, 120), 255, np.uint8)                 # as stored: row 0 = top
    planes[0, 10:50, 10:70] = 0
    planes[1, 30:80, 40:100] = 0                                  # overlaps class 0
    planes[2, 60:85, 5:30] = 100                                  # <= 127: inside
    planes[2, 0:5, :] = 200                                       # > 127: outside, yet not 255: the cell counts as known
    dp = k.to_device(planes.reshape(-1)).view(torch.uint8) if False else torch.from_numpy(planes.reshape(-1)).to(k.device)
    rows, cols = 90, 120
    rec = k.zeros((int(L.tdr_map_rec_floats_total(3, rows, cols)),))
    ws = torch.zeros(int(L.tdr_map_ingest_workspace_bytes(3, rows, cols)), dtype=torch.uint8, device=k.device)
    check(L.tdr_k_map_from_rasters(vp(dp.data_ptr()), 3, rows, cols, C.c_float(0.5), vp(rec.data_ptr()), vp(ws.data_ptr()), None))
    k.synchronize()
    rf = int(L.tdr_rec_floats(3))
    got = rec.cpu().numpy().reshape(rows + 2, cols + 2, rf)[1:-1, 1:-1]
    flipped = planes[:, ::-1, :]                                  # map row 0 = bottom image row (:217)
    unknown = (flipped == 255).all(axis=0)
    for c in range(3):
        inside = flipped[c] <= 127
        dist = distance_transform_edt(~inside).astype(np.float32) * np.float32(0.5)
        want = np.where(unknown, 0.0, np.minimum(dist, 50.0)).astype(np.float32)
        assert np.array_equal(got[:, :, c], want), c
    assert np.array_equal(got[:, :, rf - 1] == 0, unknown)
    L.tdr_map_destroy(m)
    L.tdr_map_destroy(m2)
    # the Python classes: the same round trip through TopDownMap.saveRasterizedMaps / loadRasterizedMaps
    import top_down_renderer_amd as pkg
    pm = pkg.TopDownMap(pkg.Params(resolution=1.0, num_classes=ncls, flatten_lut=list(range(ncls))), kernels=k)
    pm.loadCompressedRasterMap(img)
    pd = str(tmp_path / "py_raster_cache")
    pm.saveRasterizedMaps(pd)
    pm2 = pkg.TopDownMap(pkg.Params(resolution=1.0, num_classes=ncls), kernels=k)
    pm2.loadRasterizedMaps(pd)
    assert torch.equal(pm.dev.rec, pm2.dev.rec)
