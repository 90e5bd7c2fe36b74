"""Hand-derived micro vectors for the rows of SURVEY.md §8 that carry the result: A5 (window gather), A9 (cost of one
rotation / weight), A12 (weight statistics) and A14 (systematic resample).

Every expected value below was worked ON PAPER from the text of the reference — the arithmetic is written out in the
comments with the reference lines it follows — NOT produced by oracle/oracle.cpp, oracle/np_oracle.py or the HIP
library.  It is a third derivation: the C oracle, its NumPy twin and the HIP path are each checked against it.  The
sizes are what a person can do by hand: a 4 x 2 polar image (4 directions, 2 rings), a 5 x 5 map with 2 classes,
5 particles.

The only thing taken as given is the sample table of A4 (TopDownMapPolar::samplePtsPolar, src/top_down_map_polar.cpp:7-19)
for shape (4, 2), ang_res = pi/2, resolution 1:  k = i + 4 j,  theta_i = (i - 1.5) pi/2 = -135, -45, 45, 135 degrees,
r_j = j,  row 0 = cos(theta_i) r_j,  row 1 = sin(theta_i) r_j,  i.e. for j = 1 (with h = 0.70710678):
    i = 0: (-h, -h)    i = 1: (+h, -h)    i = 2: (+h, +h)    i = 3: (-h, +h)         and (0, 0) for j = 0.
"""
import numpy as np
import pytest

H = 0.70710678
F32 = np.float32
NB, NR, NCLS = 4, 2, 2
ANG_RES = np.pi / 2
RES = 2.0                       # metres per bin: the ring-1 samples sit 2 h = 1.41421356 cells from the centre

# ---- the map (5 x 5, indexed [class][row][col]) ---------------------------------------------------------------------------
# class 0: M0(r, c) = 10 r + c + 1;  class 1: M1(r, c) = 50 - (10 r + c);  cells (3, 1) and (2, 3) are unknown (mask 1) and,
# like every unknown cell computeDists leaves behind (src/top_down_map.cpp:289-326), hold distance 0 in every class.
MAPS = np.zeros((NCLS, 5, 5), F32)
for _r in range(5):
    for _c in range(5):
        MAPS[0, _r, _c] = 10 * _r + _c + 1
        MAPS[1, _r, _c] = 50 - (10 * _r + _c)
MASK = np.zeros((5, 5), np.uint8)
for _r, _c in ((3, 1), (2, 3)):
    MASK[_r, _c] = 1
    MAPS[:, _r, _c] = 0

# ---- A5: TopDownMapPolar::getLocalMap (src/top_down_map_polar.cpp:21-53) -------------------------------------------------
# pts = tab * scale * res (:28) = +-1.41421356 on ring 1, 0 on ring 0; row 0 += center[1] / resolution (:29),
# row 1 += center[0] / resolution (:30); pts.round() (:31, half away from zero); in bounds (0 <= r < 5, 0 <= c < 5) ->
# class_maps[cls](r, c) else 0 (:33-42); mask = class_mask(r, c) in bounds else 1 (:44-52).  scale = 1, resolution = 1.
#
# P1, centre (cx, cy) = (2, 2): ring 0 -> (2, 2) four times.  Ring 1:
#   i = 0: (2 - 1.4142, 2 - 1.4142) = (0.5858, 0.5858) -> (1, 1)      i = 1: (3.4142, 0.5858) -> (3, 1)  [unknown cell]
#   i = 2: (3.4142, 3.4142) -> (3, 3)                                i = 3: (0.5858, 3.4142) -> (1, 3)
#   class 0: M0(2,2) = 23 x4, then M0(1,1) = 12, (3,1) -> 0, M0(3,3) = 34, M0(1,3) = 14
#   class 1: M1(2,2) = 28 x4, then M1(1,1) = 39, (3,1) -> 0, M1(3,3) = 17, M1(1,3) = 37
# P2, centre (4, 0.25): ring 0 -> (0.25, 4) -> (0, 4): M0 = 5, M1 = 46.  Ring 1:
#   i = 0: (0.25 - 1.4142, 4 - 1.4142) = (-1.1642, 2.5858) -> (-1, 3) out of bounds
#   i = 1: (1.6642, 2.5858) -> (2, 3)  [unknown cell: distance 0, mask 1]
#   i = 2: (1.6642, 5.4142) -> (2, 5) out (5 columns)       i = 3: (-1.1642, 5.4142) -> (-1, 5) out
# P3, centre (-3, -3): every sample has a negative row and column: all zero, mask all 1.
CENTRES = {"P1": (2.0, 2.0), "P2": (4.0, 0.25), "P3": (-3.0, -3.0)}
WINDOWS = {                                  # k = i + 4 j
    "P1": (np.asarray([[23, 23, 23, 23, 12, 0, 34, 14], [28, 28, 28, 28, 39, 0, 17, 37]], F32),
           np.asarray([0, 0, 0, 0, 0, 1, 0, 0], np.uint8)),
    "P2": (np.asarray([[5, 5, 5, 5, 0, 0, 0, 0], [46, 46, 46, 46, 0, 0, 0, 0]], F32),
           np.asarray([0, 0, 0, 0, 1, 1, 1, 1], np.uint8)),
    "P3": (np.zeros((2, 8), F32), np.ones(8, np.uint8)),
}

# ---- A9: StateParticle::getCostForRot + the weight (src/state_particle.cpp:112-155, :212) --------------------------------
# Scan images S_c(a, j) (4 directions x 2 rings), k = a + 4 j:
#   S0 = [[1, 0], [0, 2], [0, 0], [3, 0]]        S1 = [[0, 1], [0, 0], [2, 0], [0, 1]]
SCAN = np.zeros((NCLS, NB * NR), F32)
for (_c, _a, _j), _v in {(0, 0, 0): 1, (0, 1, 1): 2, (0, 3, 0): 3, (1, 0, 1): 1, (1, 2, 0): 2, (1, 3, 1): 1}.items():
    SCAN[_c, _a + NB * _j] = _v
CLASS_WEIGHTS = [1.0, 0.5]
REG = 0.15
# rot_shift = int(round(rot * 4 / 2 / pi)) brought into [0, 4) (:123-128); scan rows [0, s) meet window rows [4 - s, 4)
# and scan rows [s, 4) window rows [0, 4 - s) (:136-142): scan row a meets window row (a - s) mod 4.
# cost += dot_c * 0.01 * w_c, normalization += sum(S_c * maskf), maskf = 1 - mask (:199); return cost / normalization;
# if sum(maskf) / 8 < 0.5 -> NaN (:117-120); weight = 1 / (cost + regularization) (:212).
#
# P1: W0(i, .) = [23, 12], [23, 0], [23, 34], [23, 14];  W1(i, .) = [28, 39], [28, 0], [28, 17], [28, 37];
#     maskf(i, .) = [1, 1], [1, 0], [1, 1], [1, 1];  sum(maskf) / 8 = 7 / 8.
#  rot = pi/2: s = round(1.0) = 1, a -> i: 0 -> 3, 1 -> 0, 2 -> 1, 3 -> 2
#     class 0: S0(0,0) W0(3,0) + S0(1,1) W0(0,1) + S0(3,0) W0(2,0) = 1*23 + 2*12 + 3*23 = 116
#     class 1: S1(0,1) W1(3,1) + S1(2,0) W1(1,0) + S1(3,1) W1(2,1) = 37 + 2*28 + 17 = 110
#     cost = 116 * 0.01 * 1 + 110 * 0.01 * 0.5 = 1.71;  norm = (1 + 2 + 3) + (1 + 2 + 1) = 10
#     cost / norm = 0.171;  weight = 1 / 0.321 = 3.1152648
#  rot = 0: s = 0, a -> a
#     class 0: 1*W0(0,0) + 2*W0(1,1) + 3*W0(3,0) = 23 + 0 + 69 = 92;   class 1: W1(0,1) + 2 W1(2,0) + W1(3,1) = 39 + 56 + 37 = 132
#     cost = 0.92 + 0.66 = 1.58;  norm = (1 + 2*0 + 3) + (1 + 2 + 1) = 8;  0.1975;  weight = 1 / 0.3475 = 2.8776978
#  rot = -pi/2: s = round(-1.0) = -1 -> 3, a -> (a + 1) mod 4
#     class 0: 1*W0(1,0) + 2*W0(2,1) + 3*W0(0,0) = 23 + 68 + 69 = 160;  class 1: W1(1,1) + 2 W1(3,0) + W1(0,1) = 0 + 56 + 39 = 95
#     cost = 1.60 + 0.475 = 2.075;  norm = (1 + 2 + 3) + (0 + 2 + 1) = 9;  0.2305556;  weight = 1 / 0.3805556 = 2.6277372
# P2: maskf = 1 on ring 0, 0 on ring 1: sum / 8 = 0.5 exactly, NOT < 0.5: scored.  rot = pi/2 (s = 1):
#     class 0: 1*5 + 2*0 + 3*5 = 20;  class 1: 0 + 2*46 + 0 = 92;  cost = 0.20 + 0.46 = 0.66
#     norm = (1 + 0 + 3) + (0 + 2 + 0) = 6;  0.11;  weight = 1 / 0.26 = 3.8461538
# P3: sum(maskf) = 0 -> NaN cost -> NaN weight;  with force_on_map (:163-168) its centre is off the map: weight 0.
COSTS = [("P1", np.pi / 2, 0.171, 1 / 0.321), ("P1", 0.0, 0.1975, 1 / 0.3475), ("P1", -np.pi / 2, 2.075 / 9, 1 / (2.075 / 9 + 0.15)),
         ("P2", np.pi / 2, 0.11, 1 / 0.26), ("P3", 0.0, np.nan, np.nan)]

# ---- A12: weight statistics (src/particle_filter.cpp:107-147) ------------------------------------------------------------
# raw = [0.5, NaN, 0.25, 0.125, 0.125]: sum over the non-NaN = 1.0, num_valid = 4, mean = 0.25 (:108-117);
# below the mean: 0.125, 0.125 -> (0.125)^2 * 2 / 2 -> bottom_stddev = 0.125 (:118-126); NaN -> mean - stddev = 0.125
# (:131-133); weights [0.5, 0.125, 0.25, 0.125, 0.125] / 1.125 = [4, 1, 2, 1, 1] / 9 (:135).
# last_dist = [0.2, 0.1, 0, 1, 0.05]: d = min(5 last_dist, 1) = [1, 0.5, 0, 1, 0.25]; w = d w + (1 - d) / 5 (:138-141):
#   4/9 = 40/90;  1/18 + 1/10 = 14/90;  0 + 1/5 = 18/90;  1/9 = 10/90;  1/36 + 3/20 = 16/90;  sum = 98/90
# renormalised (:142): [40, 14, 18, 10, 16] / 98;  argmax = particle 0 (:145-147).
STATS = [
    (np.asarray([0.5, np.nan, 0.25, 0.125, 0.125], F32), np.asarray([0.2, 0.1, 0.0, 1.0, 0.05], F32),
     np.asarray([40, 14, 18, 10, 16], np.float64) / 98, 0),
    # sum == 0 -> every weight 1 (:129-130) -> 1/3 each; d = 1 leaves them; the first maximum wins
    (np.asarray([0.0, np.nan, 0.0], F32), np.ones(3, F32), np.full(3, 1 / 3), 0),
    # nothing below the mean (num_under < 1) -> every weight 1, the NaN included
    (np.asarray([0.25, 0.25, np.nan], F32), np.ones(3, F32), np.full(3, 1 / 3), 0),
]

# ---- A14: systematic resample (src/particle_filter.cpp:171-185) ----------------------------------------------------------
# sample_i = (i + shift) / N'; j = first index whose running sum of weights EXCEEDS the sample, or the last index.
# w = [40, 14, 18, 10, 16] / 98: running sums 0.40816, 0.55102, 0.73469, 0.83673, 1.0
#   N' = 5, shift 0.5: samples 0.1 0.3 0.5 0.7 0.9            -> 0 0 1 2 4
#   N' = 3, shift 0.9: samples 0.3 0.63333 0.96667            -> 0 2 4
#   N' = 8, shift 0:   samples 0 .125 .25 .375 .5 .625 .75 .875 -> 0 0 0 0 1 2 3 4
# w = [0.25, 0.25] (not normalised), N' = 2, shift 0.5: samples 0.25, 0.75: 0.25 > 0.25 is false -> j = 1 (0.5 > 0.25);
#   0.75 is never exceeded -> the last index, 1.
W5 = (np.asarray([40, 14, 18, 10, 16], np.float64) / 98).astype(F32)
RESAMPLE = [(W5, 5, 0.5, [0, 0, 1, 2, 4]), (W5, 3, 0.9, [0, 2, 4]), (W5, 8, 0.0, [0, 0, 0, 0, 1, 2, 3, 4]),
            (np.asarray([0.25, 0.25], F32), 2, 0.5, [1, 1])]


def _states(names, rots):
    from top_down_renderer_amd import STATE_DTYPE
    st = np.zeros(len(names), STATE_DTYPE)
    for q, (name, rot) in enumerate(zip(names, rots)):
        st["init_x_px"][q], st["init_y_px"][q] = CENTRES[name]
        st["theta"][q] = rot
    st["scale"] = 1.0
    st["have_init"] = 1
    return st


def _close(got, want, rtol=2e-6):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(want)), (got, want)
    ok = ~np.isnan(want)
    assert np.allclose(got[ok], want[ok], rtol=rtol, atol=0), (got, want)


def test_table_is_the_one_the_derivation_assumes(oracle):
    from oracle import np_oracle
    want = np.zeros((2, 8))
    want[:, 4:] = [[-H, H, H, -H], [-H, -H, H, H]]
    tab_c = np.asarray(oracle.polar_table(NB, NR, ANG_RES, 1.0)).reshape(8, 2).T          # the C oracle keeps [k][2]
    tab_np = np.asarray(np_oracle.polar_table(NB, NR, ANG_RES, 1.0)).reshape(2, 8)
    for tab in (tab_c, tab_np):
        assert np.allclose(tab, want, rtol=0, atol=1e-6)


def test_hand_vectors_against_the_c_oracle(oracle):
    om = oracle.OracleMap(MAPS, MASK, 1.0)
    tab = oracle.polar_table(NB, NR, ANG_RES, 1.0)
    for name, (cx, cy) in CENTRES.items():                                         # A5
        d, m = oracle.local_map_polar(om, tab, cx, cy, 1.0, RES)
        assert np.array_equal(d, WINDOWS[name][0]) and np.array_equal(m, WINDOWS[name][1]), name
    for name, rot, cost, _ in COSTS:                                               # A9, cost of one rotation
        win, mask = WINDOWS[name]
        got = oracle.cost_for_rot(SCAN, win, (1 - mask).astype(F32), NB, NR, np.asarray(CLASS_WEIGHTS, F32), rot)
        _close([got], [cost])
    st = _states([c[0] for c in COSTS], [c[1] for c in COSTS])                      # A8 + A9, the weight
    fp = oracle.make_params(NCLS, regularization=REG, class_weights=CLASS_WEIGHTS)
    _close(oracle.compute_weights(om, tab, NB, NR, SCAN, RES, fp, st.copy()), [c[3] for c in COSTS])
    fp = oracle.make_params(NCLS, regularization=REG, class_weights=CLASS_WEIGHTS, force_on_map=True)
    _close(oracle.compute_weights(om, tab, NB, NR, SCAN, RES, fp, st.copy()), [c[3] for c in COSTS[:4]] + [0.0])
    for raw, ld, want, best in STATS:                                              # A12
        w, arg, _ = oracle.update_weights(raw, ld)
        _close(w, want)
        assert arg == best
    for w, n_new, shift, want in RESAMPLE:                                         # A14: the literal O(N N') loop and the prefix form
        assert list(oracle.resample_literal(w, n_new, shift)) == want
        assert list(oracle.resample_prefix(w, n_new, shift)) == want


def test_hand_vectors_against_the_numpy_twin():
    from oracle import np_oracle as npo
    tab = npo.polar_table(NB, NR, ANG_RES, 1.0)
    for name, (cx, cy) in CENTRES.items():
        d, m = npo.local_map_polar(MAPS, MASK, 1.0, tab, cx, cy, 1.0, RES)
        assert np.array_equal(d, WINDOWS[name][0]) and np.array_equal(m, WINDOWS[name][1]), name
    for name, rot, cost, _ in COSTS:
        win, mask = WINDOWS[name]
        _close([npo.cost_for_rot(SCAN, win, (1 - mask).astype(F32), NB, NR, CLASS_WEIGHTS, rot)], [cost])
    for raw, ld, want, best in STATS:
        w, arg = npo.update_weights(raw, ld)
        _close(w, want)
        assert arg == best
    for w, n_new, shift, want in RESAMPLE:
        assert list(npo.resample(w, n_new, shift)) == want


@pytest.mark.gpu
def test_hand_vectors_against_the_hip_path():
    import torch
    import top_down_renderer_amd as pkg
    from top_down_renderer_amd.kernels import HipKernels
    k = HipKernels()
    m = pkg.TopDownMapPolar(pkg.Params(resolution=1.0), MAPS, MASK, kernels=k)
    m.samplePtsPolar((NB, NR), ANG_RES)
    for name, centre in CENTRES.items():                                           # A5
        d, mk = m.getLocalMap(centre, 1.0, RES)
        assert np.array_equal(np.stack([x.T.ravel() for x in d]), WINDOWS[name][0]), name
        assert np.array_equal(mk.T.ravel(), WINDOWS[name][1]), name
    st = _states([c[0] for c in COSTS], [c[1] for c in COSTS])                      # A8 + A9 through the scoring kernels
    scan_imgs = [SCAN[c].reshape(NR, NB).T.copy() for c in range(NCLS)]            # [direction][ring] images
    for force, want in ((False, [c[3] for c in COSTS]), (True, [c[3] for c in COSTS[:4]] + [0.0])):
        for compact in (1, 0):
            before = k.lib.tdr_config_compact(-1)
            try:
                k.lib.tdr_config_compact(compact)
                f = pkg.ParticleFilter(len(st), m, pkg.FilterParams(fixed_scale=1.0, regularization=REG,
                                                                    class_weights=CLASS_WEIGHTS, force_on_map=force),
                                       kernels=k, init_particles=False)
                f.set_states(st)
                f.update(scan_imgs, None, RES)
                _close(f.raw_weights(), want)
            finally:
                k.lib.tdr_config_compact(before)
    for raw, ld, want, best in STATS:                                              # A12
        n = len(raw)
        w, info = k.zeros((n,)), k.zeros((65536,))
        k.update_weights(k.to_device(raw), k.to_device(ld), n, w, info)
        _close(w.cpu().numpy(), want)
        assert int(info[:1].cpu().view(torch.int32).item()) == best
    for w, n_new, shift, want in RESAMPLE:                                         # A14
        n = len(w)
        runmax, idx = k.zeros((n,)), k.zeros((n_new,), torch.int32)
        k.prefix(k.to_device(w), n, runmax)
        k.resample(runmax, n, n_new, shift, 0, n_new, idx)
        assert list(idx.cpu().numpy()) == want
