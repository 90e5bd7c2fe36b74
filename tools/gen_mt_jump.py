#!/usr/bin/env python3
"""Writes top_down_renderer_amd/csrc/tdr_mt_jump.h: jump-ahead polynomials of std::mt19937 for a fixed stride, so that the
device can fill DISJOINT stretches of the generator's stream in parallel (csrc/tdr_rng.hip: mt_jump_kernel, mt_fill_kernel).

    python3 tools/gen_mt_jump.py        (deterministic; the header is committed; tests/test_rng.py checks it)

The reference draws all of a step's noise serially from ONE std::mt19937 (src/particle_filter.cpp:86-92,
src/state_particle.cpp:64-73): 8.4 words per particle, i.e. 13 500 state blocks of 624 words for a million particles — 4.7 ms
on the one wave that can run the recurrence.  The recurrence is linear over GF(2): with A the one-word step of the
19937-bit state and phi its characteristic polynomial, the state J words ahead is g(A) s with g = t^J mod phi (Haramoto,
Matsumoto, Nishimura, Panneton, L'Ecuyer 2008).  For J = STRIDE * 624 * 2^m (m = 0 .. LEVELS - 1) those g are constants of
the generator: this script computes them (phi by Berlekamp-Massey on one output bit, the powers by square-and-multiply) and
the device reaches block k * STRIDE, k = 1 .. W - 1, in ceil(log2 W) rounds of doubling — round m jumps the 2^m stretches'
first blocks it already has by 2^m stretches — then W waves fill their stretches side by side.

Layout: a polynomial g of degree < 19937 is split as g(t) = sum_j t^(624 j) r_j(t), deg r_j < 624, because A^624 is the
engine's own block step (a "twist", which the device does in 0.35 us) and r_j(A) s is a plain XOR of 624-word windows of
the 1247 words behind s.  MT_JUMP[m][j][20] holds r_j as 640 bits, bit i of r_j = bit (i & 31) of word i >> 5.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "top_down_renderer_amd", "csrc", "tdr_mt_jump.h")
N, M = 624, 397
DEG = 19937
STRIDE = 128      # blocks per stretch: 80 000 words, the noise of ~9 500 particles; a stretch takes one wave 45 us
LEVELS = 10       # up to 2^10 stretches: 131 072 blocks, ~9.7 million particles per call
NCHUNK = 32       # 32 x 624 = 19 968 >= 19 937


def twist(x):
    """One block step of the engine (mersenne_twister_engine::_M_gen_rand) on a numpy uint32[624], in place order."""
    x = x.copy()
    for k in range(N):
        y = (int(x[k]) & 0x80000000) | (int(x[(k + 1) % N]) & 0x7FFFFFFF)
        x[k] = int(x[(k + M) % N]) ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
    return x


def seed_state(seed=5489):
    x = np.zeros(N, np.uint64)
    x[0] = seed
    for i in range(1, N):
        x[i] = (1812433253 * (int(x[i - 1]) ^ (int(x[i - 1]) >> 30)) + i) & 0xFFFFFFFF
    return x.astype(np.uint32)


def stream_bits(nbits):
    """Bit 0 of the untempered word stream x_0, x_1, ...: a non-trivial linear functional of the state sequence."""
    x = seed_state()
    out = []
    while len(out) < nbits:
        x = twist(x)
        out.extend(int(v) & 1 for v in x)
    return out[:nbits]


def berlekamp_massey(bits):
    """Minimal polynomial of a GF(2) sequence; polynomials as Python ints (bit i = coefficient of t^i of the connection
    polynomial C, C_0 = 1).  Returns (C, L)."""
    n = len(bits)
    s = 0                                   # the sequence reversed into an int window, rebuilt incrementally
    C, B, L, m = 1, 1, 0, 1
    # discrepancy d = sum_{i=0..L} C_i s_{k-i}: keep `hist` = sum s_{k-i} << i for i = 0..k (most recent bit at bit 0)
    hist = 0
    for k in range(n):
        hist = (hist << 1) | bits[k]
        d = (C & hist).bit_count() & 1
        if d == 0:
            m += 1
        elif 2 * L <= k:
            T = C
            C ^= B << m
            L = k + 1 - L
            B = T
            m = 1
        else:
            C ^= B << m
            m += 1
    return C, L


def poly_mod(a, phi, deg):
    while a.bit_length() > deg:
        a ^= phi << (a.bit_length() - 1 - deg)
    return a


def poly_sqr_mod(a, phi, deg):
    # squaring over GF(2) spreads the bits: bit i -> bit 2 i
    s = int(bin(a)[2:].replace("0", "00").replace("1", "01"), 2) if a else 0
    return poly_mod(s, phi, deg)


def t_pow_mod(e, phi, deg):
    r = 1
    for bit in bin(e)[2:]:
        r = poly_sqr_mod(r, phi, deg)
        if bit == "1":
            r = poly_mod(r << 1, phi, deg)
    return r


def characteristic_polynomial():
    bits = stream_bits(2 * DEG + 64)
    C, L = berlekamp_massey(bits)
    assert L == DEG, L
    # connection polynomial C(t) = sum C_i t^i with s_k = sum_{i>=1} C_i s_{k-i}; the characteristic polynomial is its
    # reciprocal: phi(t) = t^L C(1/t)
    phi = int(bin(C)[2:].zfill(L + 1)[::-1], 2)
    assert phi.bit_length() == DEG + 1 and phi & 1
    return phi


def chunks_of(g):
    """g -> [NCHUNK][20] uint32: r_j, bit i of r_j = coefficient of t^(624 j + i)."""
    out = np.zeros((NCHUNK, 20), np.uint32)
    for j in range(NCHUNK):
        r = (g >> (N * j)) & ((1 << N) - 1)
        for w in range(20):
            out[j, w] = (r >> (32 * w)) & 0xFFFFFFFF
    return out


def apply_jump(chunks, block):
    """The device's procedure on the host (numpy): block (uint32[624], the raw stream from some position on) -> the block
    `jump` words later.  Windows of block || twist(block), Horner in the block step, word 0's low bits repaired."""
    stream = np.concatenate([block, twist(block)])
    win = np.lib.stride_tricks.sliding_window_view(stream, N)[:N]        # win[i] = stream[i : i + 624]
    R = np.zeros((NCHUNK, N), np.uint32)
    for j in range(NCHUNK):
        sel = np.array([(int(chunks[j, i >> 5]) >> (i & 31)) & 1 for i in range(N)], bool)
        if sel.any():
            R[j] = np.bitwise_xor.reduce(win[sel], axis=0)
    h = R[NCHUNK - 1].copy()
    for j in range(NCHUNK - 2, -1, -1):
        h = twist(h) ^ R[j]
    # word 0: its top bit is exact, the other 31 follow from x_(J+623) = x_(J+396) ^ T(upper(x_(J-1)) | lower(x_J))
    v = int(h[623]) ^ int(h[396])
    lsb = v >> 31
    y = (((v ^ (0x9908B0DF if lsb else 0)) << 1) | lsb) & 0xFFFFFFFF
    h[0] = (int(h[0]) & 0x80000000) | (y & 0x7FFFFFFF)
    return h


def build():
    phi = characteristic_polynomial()
    g = t_pow_mod(STRIDE * N, phi, DEG)
    levels = []
    for m in range(LEVELS):
        levels.append(chunks_of(g))
        g = poly_sqr_mod(g, phi, DEG)
    return np.stack(levels)


def self_check(levels):
    """Level 0 and level 1 against STRIDE / 2 STRIDE honest block steps from an arbitrary state."""
    x = twist(twist(seed_state(20241005)))
    want = x.copy()
    for _ in range(STRIDE):
        want = twist(want)
    assert np.array_equal(apply_jump(levels[0], x), want), "level 0 does not jump STRIDE blocks"
    for _ in range(STRIDE):
        want = twist(want)
    assert np.array_equal(apply_jump(levels[1], x), want), "level 1 does not jump 2 STRIDE blocks"


def header(levels):
    out = ["// tdr_mt_jump.h — GENERATED by tools/gen_mt_jump.py; do not edit.",
           "// Jump-ahead polynomials of std::mt19937: MT_JUMP[m][j] = chunk r_j (640 bits) of t^(MT_JUMP_STRIDE * 624 * 2^m) mod",
           "// the generator's characteristic polynomial, g(t) = sum_j t^(624 j) r_j(t).  See the generator for the mathematics.",
           "#ifndef TDR_MT_JUMP_H_", "#define TDR_MT_JUMP_H_",
           f"#define MT_JUMP_STRIDE {STRIDE}", f"#define MT_JUMP_LEVELS {LEVELS}", f"#define MT_JUMP_CHUNKS {NCHUNK}",
           "__device__ const uint32_t MT_JUMP[MT_JUMP_LEVELS][MT_JUMP_CHUNKS][20] = {"]
    for m in range(levels.shape[0]):
        out.append("  {")
        for j in range(NCHUNK):
            out.append("    {" + ", ".join(f"0x{int(v):08x}u" for v in levels[m, j]) + "},")
        out.append("  },")
    out.append("};")
    out.append("#endif  // TDR_MT_JUMP_H_")
    return "\n".join(out) + "\n"


def main():
    levels = build()
    if "--no-check" not in sys.argv:
        self_check(levels)
    open(OUT, "w").write(header(levels))
    print(OUT)


if __name__ == "__main__":
    main()
