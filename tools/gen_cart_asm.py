#!/usr/bin/env python3
"""Writes top_down_renderer_amd/csrc/tdr_score_cart_asm.h: the hand-scheduled gfx950 sample loop of
score_cart_su_kernel (tdr_score_cart.hip) — the Cartesian integer form's dense kernel for maps with two-dword compact
records (4-6 classes) — as inline-assembly text, in three variants: general / without the clamp of the coordinates into
the map's guard ring / without clamp and known-mask lookups (every cell the wave can reach in the segment is known).
LOADS AND THEIR WAITS LIVE INSIDE THIS ONE TEXT with a planned register file: the compiler sees a single statement with
declared outputs and clobbers and never a load in flight (rounds 3-4 issued loads through separate inline-assembly
statements with hand-counted waits; the compiler moved a copy between a load and its wait twice — DESIGN.md 5.1).

    python3 tools/gen_cart_asm.py        (re-run after editing; the header is committed; tests/test_asm_guard.py checks both)

One statement = one SEGMENT: `nblk` blocks of 4 window rows x the NCOL = 8 window columns of a column group, lane =
particle.  A step = the 4 samples (rows i .. i + 3, column jc); the column loop is unrolled (the column terms AB[jc] of the
rotation are eight operands computed once per column group), the row-block loop is a loop.
Sample (i, j) of getLocalMap (src/top_down_map.cpp:367-389, 429-459):  p = (cs * y_i + AB_j) + centre, cell = round(p),
with y_i = lo_r + float(i) * step_r (Eigen's LinSpaced, two roundings), cs = {cos, sin}(theta), AB_j = {-sin * x_j, cos * x_j}
— the float operations of score_cart_kernel, in its order.

Register plan (fixed registers, all named in the statement's clobber list):
  v8..v15   sample u: v(8+2u) row -> ri, v(9+2u) column -> ci
  v16..v19  sample u: mask word address -> 0 / -1 known;  v28..v31 sample u: mask word
  v20..v23  scratch;  v24..v27 sample u: the 2-byte cell of the bin's class plane
  v32..v39  cs * y_(i+u), u = 0..3 (recomputed at every row block);  v40, v41 scratch of that
  v42..v53  the six 64-bit class accumulators: the statement's a0..a5 operands are TIED to these registers ("+{v[42:43]}" ...),
            the loop addresses them relative to v[40:41] through the GPR index (M0)
  s48..s63 / s80..s95   the one-dword descriptors of columns 0-3 / 4-7 of the row block, 4 rows x 4 columns each
                        (cart_prep_kernel's block layout: count in bits 0-23, class code in bits 24-26; a step's first
                        descriptor: which of its four bins hold a class in bits 28-31; a block's first: bit 27 = any does).
                        Half a block is ONE scalar load, requested half a block ahead into the other set
  s64 first row of the block, s65 row blocks left, s66 byte offset of the block in the descriptor array,
  s67 offset being requested, s76, s77 scratch, s68 the all-known variant's normalisation (wave-uniform)
Arithmetic: EXACT integer sums like the polar loop (tools/gen_su_asm.py): any order, any partition, any kernel gives the
same bits.  Bins holding several classes are EMPTY to this loop (their descriptor code is 0 in the array it reads): the
kernel adds their products from a per-chunk list afterwards; their known bit is counted here like an empty bin's.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_su_asm as su  # noqa: E402  (the register rules and their checker are shared)

OUT = os.path.join(ROOT, "top_down_renderer_amd", "csrc", "tdr_score_cart_asm.h")
NCOL = 8
CLOBBER_V = range(8, 42)
ACC_V = range(42, 54)     # the accumulators: operands of the statement tied to these registers (not clobbers)
CLOBBER_S = range(48, 96)
VARIANTS = (("", True, True), ("_NOCLAMP", False, True), ("_ALLKNOWN", False, False))


def loop_text(clamp, mask):
    L = []
    a = L.append
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")               # drain whatever the compiler still has in flight (see gen_su_asm.py)
    a("s_mov_b32 s64, %[i0]")
    a("s_mov_b32 s65, %[nblk]")
    a("s_mov_b32 s66, %[doff]")
    a("s_load_dwordx16 s[48:63], %[db], s66")        # the first half of the first block
    if not mask:
        a("s_mov_b32 s68, 0")
    a(".Lct_blk%=:")
    a("s_waitcnt lgkmcnt(0)")                        # set A: columns 0-3 of this block
    if not mask:
        a(f"v_add_u32 %[known], {4 * NCOL}, %[known]")   # every sample of the block is a known cell
        a("s_bitcmp1_b32 s48, 27")                   # no bin of the block holds anything: nothing else to do for it
        a("s_cbranch_scc0 .Lct_eblk%=")
    a("s_add_u32 s67, s66, 64")
    a("s_load_dwordx16 s[80:95], %[db], s67")        # set B: columns 4-7, requested behind everything this block waits for first
    # cs * y_(i+u): y = lo_r + float(i + u) * step_r (LinSpaced: one product, one sum — no fused multiply-add)
    for u in range(4):
        a(f"s_add_u32 s76, s64, {u}")
        a("v_cvt_f32_i32 v40, s76")
        a("s_nop 0")
        a("v_mul_f32 v40, v40, %[stepr]")
        a("s_nop 0")
        a("v_add_f32 v40, %[lor], v40")
        a("s_nop 0")
        a(f"v_pk_mul_f32 v[{32 + 2 * u}:{33 + 2 * u}], %[cs], v[40:41] op_sel_hi:[1,0]")   # {cos * y, sin * y}
    for jc in range(NCOL):
        if jc == NCOL // 2:
            a("s_waitcnt lgkmcnt(0)")                # set B (long since requested)
            a("s_add_u32 s67, s66, 128")             # set A is free: the first half of the NEXT block (behind the segment's
            a("s_load_dwordx16 s[48:63], %[db], s67")   # last block lies the group's next block: inside the array)
        step(a, jc, clamp, mask, S=48 if jc < NCOL // 2 else 80)
    a("s_add_u32 s66, s66, 128")
    a("s_add_u32 s64, s64, 4")
    a("s_sub_u32 s65, s65, 1")
    a("s_cmp_lg_u32 s65, 0")
    a("s_cbranch_scc1 .Lct_blk%=")
    if not mask:
        a("s_branch .Lct_end%=")
        a(".Lct_eblk%=:")                            # an empty block of an all-known segment: on to the next one
        a("s_add_u32 s66, s66, 128")
        a("s_load_dwordx16 s[48:63], %[db], s66")
        a("s_add_u32 s64, s64, 4")
        a("s_sub_u32 s65, s65, 1")
        a("s_cmp_lg_u32 s65, 0")
        a("s_cbranch_scc1 .Lct_blk%=")
        a(".Lct_end%=:")
    a("s_waitcnt lgkmcnt(0)")                        # (the request for the block behind the last one)
    if not mask:
        a("v_add_u32 %[norm], s68, %[norm]")
    return [x for x in L if su.PAD or x != "s_nop 0"]


def step(a, jc, clamp, mask, S):
    """One step = rows i .. i + 3 of column jc.  Its four descriptors are s(d0) .. s(d0 + 3), one dword each: count in bits
    0-23, class code (1-6; 0 = nothing for the loop) in bits 24-26; the step's FIRST descriptor also carries, in bits 28-31,
    which of the four bins hold a class (cart_prep_kernel)."""
    tag = f"c{jc}"
    d0 = S + 4 * (jc % (NCOL // 2))

    def coords(us):
        for u in us:
            p = 8 + 2 * u
            a(f"v_pk_add_f32 v[{p}:{p + 1}], v[{32 + 2 * u}:{33 + 2 * u}], %[ab{jc}]")   # rotm * pts (:383-385)
        if len(us) == 1:
            a("s_nop 0")
        for u in us:
            p = 8 + 2 * u
            a(f"v_pk_add_f32 v[{p}:{p + 1}], v[{p}:{p + 1}], %[offv]")                    # + centre (:387-388)
        if len(us) == 1:
            a("s_nop 0")
        if clamp:
            for u in us:
                p = 8 + 2 * u
                a(f"v_med3_f32 v{p}, v{p}, %[rmax], -1.0")                               # clamp into the guard ring
                a(f"v_med3_f32 v{p + 1}, v{p + 1}, %[cmax], -1.0")
        for u in us:
            p = 8 + 2 * u
            a(f"v_pk_add_f32 v[{p}:{p + 1}], v[{p}:{p + 1}], %[half]")                    # round_half_away_clamped (:437)
        if len(us) == 1:
            a("s_nop 0")
        for u in us:
            p = 8 + 2 * u
            a(f"v_cvt_flr_i32_f32 v{p}, v{p}")
            a(f"v_cvt_flr_i32_f32 v{p + 1}, v{p + 1}")

    if mask:
        coords(range(4))
        # word of the wave's staged known mask (LDS) of each cell: ri * krow4 + (ci >> 5) * 4 + kconst
        for u in range(4):
            a(f"v_mad_i32_i24 v{16 + u}, v{8 + 2 * u}, %[krow4], %[kconst]")
        for u in range(4):
            a(f"v_ashrrev_i32 v{20 + u}, 5, v{9 + 2 * u}")
        for u in range(4):
            a(f"v_lshl_add_u32 v{16 + u}, v{20 + u}, 2, v{16 + u}")
        a("s_nop 0")
        for u in range(4):
            a(f"ds_read_b32 v{28 + u}, v{16 + u}")
    # a step without any class: an all-known segment skips it altogether, the others only count their known bits
    a(f"s_lshr_b32 s76, s{d0}, 28")
    a(f"s_cbranch_scc0 .Lct_ae{tag}%=" if mask else f"s_cbranch_scc0 .Lct_be{tag}%=")
    # the 2-byte cell of the class's plane for every bin holding ONE class (plane_offset, tdr_score_dev.h)
    for u in range(4):
        a(f"s_bitcmp1_b32 s{d0}, {28 + u}")
        a(f"s_cbranch_scc0 .Lct_a{u}{tag}%=")
        if not mask:
            coords([u])     # (an empty bin of an all-known segment needs no cell at all)
        a(f"s_bfe_u32 s77, s{d0 + u}, 0x30018")                     # the class code: bits 24-26
        a("s_mul_i32 s77, s77, %[pbytes]")                          # its plane's constant: pbase0 + code * plane bytes
        a("s_add_u32 s77, s77, %[pbase0]")
        a(f"v_ashrrev_i32 v{20 + u}, 3, v{9 + 2 * u}")
        a("s_nop 0")
        a(f"v_mad_i32_i24 v{20 + u}, v{20 + u}, %[pkcol], s77")
        a("s_nop 0")
        a(f"v_lshl_add_u32 v{20 + u}, v{9 + 2 * u}, 1, v{20 + u}")
        a("s_nop 0")
        a(f"v_lshl_add_u32 v{20 + u}, v{8 + 2 * u}, 4, v{20 + u}")
        a("s_nop 0")
        a(f"global_load_ushort v{24 + u}, v{20 + u}, %[crec]")     # (never into its own address register)
        a(f".Lct_a{u}{tag}%=:")
    if mask:
        a(f".Lct_ae{tag}%=:")
    # ---- everything requested above is waited for HERE, inside the text
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if mask:
        for u in range(4):
            a(f"v_bfe_i32 v{16 + u}, v{28 + u}, v{9 + 2 * u}, 1")                           # 0 / -1: the cell's known bit
        a("v_add_u32 v20, v16, v17")
        a("v_add_u32 v21, v18, v19")
        a("s_nop 0")
        a("v_add_u32 v20, v20, v21")
        a("s_nop 0")
        a("v_sub_u32 %[known], %[known], v20")
        a(f"s_lshr_b32 s76, s{d0}, 28")
        a(f"s_cbranch_scc0 .Lct_be{tag}%=")
    for u in range(4):
        a(f"s_bitcmp1_b32 s{d0}, {28 + u}")
        a(f"s_cbranch_scc0 .Lct_b{u}{tag}%=")
        a(f"s_and_b32 s77, s{d0 + u}, 0xffffff")                                             # the bin's count
        if mask:
            a(f"v_and_b32 v20, s77, v{16 + u}")                                              # count x known (:141-142)
        a(f"v_and_b32 v21, 0xffc, v{24 + u}")                                                # the class's dictionary index * 4
        if mask:
            a("v_add_u32 %[norm], %[norm], v20")
        else:   # every cell known: the normalisation is the same for every lane — summed in s68, added at the exit
            a("s_add_u32 s68, s68, s77")
        a("s_nop 0")
        a("ds_read_b32 v21, v21")                                                            # the dictionary sits at LDS address 0
        a(f"s_bfe_u32 s76, s{d0 + u}, 0x30018")                                              # the class code
        a("s_waitcnt lgkmcnt(0)")
        # acc[class] += count * distance (state_particle.cpp:136-139) as integers.  The six 64-bit accumulators sit in FIXED
        # registers v[42:53] (the statement's operands are tied to them) and the class picks its pair by VGPR-relative
        # indexing: destination and addend of the multiply-add are v[40:41] + 2 * code (tools/gpr_idx_probe.hip checks the
        # mode on gfx950) — three scalar instructions and no branch where a tree of compares and branches stood
        a("s_lshl_b32 s76, s76, 1")
        a("s_set_gpr_idx_on s76, 0xc")                                                       # SRC2 | DST relative
        a("v_mad_u64_u32 v[40:41], vcc, s77, v21, v[40:41]")
        a("s_set_gpr_idx_off")
        a(f".Lct_b{u}{tag}%=:")
    a(f".Lct_be{tag}%=:")


def statement_operands():
    return su.statement_operands("tdr_score_cart.hip", "CART_ASM_OPERANDS", "CART_ASM_CLOBBERS")


def check_text(lines, outputs, inputs):
    """The shared register rules (gen_su_asm.check_text) plus this loop's own: every load is waited for INSIDE the text — no
    instruction reads a load's destination between the load and a wait that covers it, and the text ends with nothing in
    flight.  (A linear pass over the text: the loop's back edge and its branches only skip instructions.)"""
    import re
    written = su.check_text(lines, outputs, inputs, CLOBBER_V, CLOBBER_S)
    vm, lgkm = set(), set()                    # fixed destination registers of requests not yet waited for, by counter
    # VGPR-relative indexing: exactly one instruction between an `on` and its `off` — the accumulate, whose destination and
    # addend are the pair below the accumulators (v[40:41] + 2 * code reaches v[42:53], the registers a0..a5 are tied to)
    for k, ln in enumerate(lines):
        if ln.startswith("s_set_gpr_idx_on"):
            assert ln == "s_set_gpr_idx_on s76, 0xc", ln
            assert lines[k + 1] == "v_mad_u64_u32 v[40:41], vcc, s77, v21, v[40:41]" and lines[k + 2] == "s_set_gpr_idx_off", \
                f"only the accumulate may run under the GPR index: {lines[k + 1]}"
            assert lines[k - 1] == "s_lshl_b32 s76, s76, 1", "the index is twice the class code"
    assert ACC_V.start == max(CLOBBER_V) + 1, "the accumulators sit right behind the clobbered registers"
    for ln in lines:
        if ln.endswith(":"):
            continue
        mn, _, rest = ln.partition(" ")
        toks = [re.sub(r"\s+op_sel.*", "", t.strip()) for t in re.split(r",(?![^\[]*\])", rest)] if rest else []
        if mn == "s_waitcnt":
            if "vmcnt(0)" in rest:
                vm.clear()
            if "lgkmcnt(0)" in rest:
                lgkm.clear()
            continue
        has_dest = not mn.startswith(su.NO_DEST)
        sources = [r for t in toks[1 if has_dest else 0:] for r in su.regs_of(t)]
        for r in sources:
            assert r not in vm and r not in lgkm, f"reads the destination of a load before its wait: {ln}"
        dest = set(su.regs_of(toks[0])) if has_dest and toks else set()
        assert not (dest & (vm | lgkm)), f"overwrites the destination of a load in flight: {ln}"
        if mn.startswith("global_load"):
            vm |= dest
        elif mn.startswith(("ds_read", "s_load")):
            lgkm |= dest
    assert not vm and not lgkm, "the text ends with a load in flight"
    return written


def check_all():
    outputs, inputs = statement_operands()
    n = 0
    for _, clamp, mask in VARIANTS:
        lines = loop_text(clamp, mask)
        assert lines[0] == "s_waitcnt vmcnt(0) lgkmcnt(0)"
        written = check_text(lines, outputs, inputs)
        assert written <= outputs
        n += 1
    return n


def main():
    check_all()
    out = ["// tdr_score_cart_asm.h — GENERATED by tools/gen_cart_asm.py; do not edit.",
           "// The sample loop of score_cart_su_kernel (tdr_score_cart.hip) for two-dword compact records (4-6 classes): see the",
           "// generator for the register plan.  Loads and their waits are inside these texts.",
           "#ifndef TDR_SCORE_CART_ASM_H_", "#define TDR_SCORE_CART_ASM_H_", "", f"#define CART_ASM_NCOL {NCOL}", ""]
    for tag, clamp, mask in VARIANTS:
        out.append(f"#define CART_ASM{tag} \\")
        lines = loop_text(clamp, mask)
        for i, ln in enumerate(lines):
            out.append(f'  "{ln}\\n"' + (" \\" if i + 1 < len(lines) else ""))
        out.append("")
    out.append("#define CART_ASM_CLOBBERS \\")
    vregs = ", ".join(f'"v{i}"' for i in CLOBBER_V)
    sregs = ", ".join(f'"s{i}"' for i in CLOBBER_S)
    out.append(f"  {vregs}, \\")
    out.append(f'  {sregs}, "vcc", "scc", "m0", "memory"')
    out.append("#endif  // TDR_SCORE_CART_ASM_H_")
    open(OUT, "w").write("\n".join(out) + "\n")
    print(OUT)


if __name__ == "__main__":
    main()
