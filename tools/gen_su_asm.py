#!/usr/bin/env python3
"""Writes top_down_renderer_amd/csrc/tdr_score_su_asm.h: the hand-scheduled gfx950 inner loop of score_polar_su_kernel
(records of two dwords: 4-6 classes) as inline-assembly text, in six variants — uniform / per-lane scale x {general, without
the clamp of the coordinates into the map's guard ring, without clamp and known-mask lookups (every reachable cell known)}.
The text is generated so that the four samples of a step, the two copies of the loop body and the variants cannot drift apart.

    python3 tools/gen_su_asm.py        (re-run after editing; the header is committed)

Register plan (fixed registers, all named in the statement's clobber list):
  v8..v15   sample u: v(8+2u) row -> ri, v(9+2u) column -> ci
  v16..v19  sample u: mask word address -> 0 / -1 known;  v28..v31 sample u: mask word
  v20..v23  scratch
  v24..v27  sample u: the 2-byte cell of the bin's class plane
  v32..v43  the accumulators of classes 0..5 — 64-bit integers, see below: the statement's a0..a5 operands are TIED to these
            registers ("+{v[32:33]}" ...) and the loop reaches them relative to v[30:31] through the GPR index (M0)
  (the normalisation and the known count are operands of the statement: the register allocator places them)
  s40..s47 / s72..s79   the step's sample offsets {tx, ty} x 4           (even / odd steps: the loop body exists twice and
  s48..s63 / s80..s95   the step's descriptors {code, value, plane constant, flag} x 4 (su_prep_kernel)   requests the next
                        step's scalars into the other set before it works on its own)
  s65 / s66 byte offsets of the step in the offset / descriptor streams, s67 steps left - 1,
            s68 steps until the scan rows wrap - 1;  s69..s71 the same offsets / count for the step behind it
Arithmetic: EXACT.  A scan count is an integer and a dictionary value an integer multiple of 2^-q (tdr_cmap.hip), so a
class's product sum is accumulated as a 64-bit integer — v_mad_u64_u32, one instruction at the cost of the v_fmac_f32 it
replaces (tools/valu_cost.hip) — and does not depend on the order of the additions: every kernel, launch and shard gives
the same bits.
Cost model behind the schedule (tools/valu_cost.hip, tools/ta_cost.hip; per CU): scalar instructions issue 1 per cycle,
plain VOP2 2 cycles per SIMD, VOP3 / packed / conversions 4, a 64-lane gather >= 16 cycles of the L1 address path.
Rounds 3-4 kept dependent vector instructions one instruction apart with s_nop 0; the hardware interlocks those
dependencies, and without the padding (13 of a step's ~100 instructions) every configuration runs the same or 1 % faster
(config 2 4.645 / 4.631 ms, config 4 15.57 / 15.36): `--pad` regenerates the padded text for A/B.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "top_down_renderer_amd", "csrc", "tdr_score_su_asm.h")
EXPERIMENT_LDS_GATHER = False   # tools/exp_lds_gather.sh
PAD = "--pad" in sys.argv   # A/B: an s_nop 0 between dependent vector instructions, as rounds 3-4 had it (measured: no difference)


def loop_text(uscale, clamp, mask=True):
    L = []
    a = L.append
    # hipcc waits for one of its own loads where the VALUE is used; a load whose value the taken path never reads is still
    # in flight here, and hipcc does not wait for it on behalf of the registers this statement clobbers: it would land in
    # the middle of the loop (found as garbage mask offsets / memory faults at full size only)
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    # the loop's scalar state travels through vector operands: the compiler keeps loop-carried values that meet vector
    # code in VGPRs, whatever the constraint says
    a("v_readfirstlane_b32 s65, %[toff]")
    a("v_readfirstlane_b32 s66, %[doff]")
    a("v_readfirstlane_b32 s67, %[nleft]")
    a("v_readfirstlane_b32 s68, %[wleft]")
    a("s_load_dwordx8 s[40:47], %[tb], s65")
    a("s_load_dwordx16 s[48:63], %[db], s66")
    if not mask:
        a("s_mov_b32 s96, 0")                        # the steps' counts (see phase B)
        a("s_mov_b32 s97, s67")                      # steps left at entry: every finished step is four more known samples
    a("s_waitcnt lgkmcnt(0)")
    # The loop body exists twice: the offsets / descriptors of step k + 1 are requested (into the other register set) before
    # step k is worked on, so their scalar-cache latency hides behind it
    step(a, uscale, clamp, mask, T=40, D=48, TN=72, DN=80, tag="e", nxt="o")
    step(a, uscale, clamp, mask, T=72, D=80, TN=40, DN=48, tag="o", nxt="e")
    a(".Lsu_out%=:")
    a("s_waitcnt lgkmcnt(0)")                        # (a request for the step behind the last one may still be in flight)
    if not mask:
        a("s_sub_u32 s97, s97, s67")                 # steps done (s67 = 0xFFFFFFFF after the last one: the difference still counts it)
        a("s_lshl_b32 s97, s97, 2")
        a("v_add_u32 %[norm], s96, %[norm]")
        a("v_add_u32 %[known], s97, %[known]")
    a("v_mov_b32 %[toff], s65")
    a("v_mov_b32 %[doff], s66")
    a("v_mov_b32 %[nleft], s67")
    a("v_mov_b32 %[wleft], s68")
    return [x for x in L if PAD or x != "s_nop 0"]


def step(a, uscale, clamp, mask, T, D, TN, DN, tag, nxt):
    a(f".Lsu_step{tag}%=:")
    a(f"s_bitcmp1_b32 s{D + 3}, 31")                 # a bin with several classes in this step: hand it to the C++ step
    a("s_cbranch_scc1 .Lsu_out%=")
    # the step behind this one: T at byte s69, D at byte s70 (the scan rows wrap: (i + shift) mod nb)
    a("s_add_u32 s69, s65, 32")
    a("s_add_u32 s70, s66, 64")
    a("s_sub_u32 s71, s68, 1")
    a(f"s_cbranch_scc0 .Lsu_nw{tag}%=")
    a("s_mov_b32 s70, 0")
    a("s_mov_b32 s71, %[wrapm1]")
    a(f".Lsu_nw{tag}%=:")
    a(f"s_load_dwordx8 s[{TN}:{TN + 7}], %[tb], s69")
    a(f"s_load_dwordx16 s[{DN}:{DN + 15}], %[db], s70")
    # ---- phase A: cells of the four samples (top_down_map_polar.cpp:28-31)
    def coords(us):
        for u in us:
            p, t = 8 + 2 * u, T + 2 * u
            if uscale:
                a(f"v_pk_add_f32 v[{p}:{p + 1}], %[offv], s[{t}:{t + 1}]")
        if not uscale:
            for u in us:
                p, t = 8 + 2 * u, T + 2 * u
                a(f"v_pk_mul_f32 v[{p}:{p + 1}], %[scale2], s[{t}:{t + 1}]")   # (tab * scale)
            for u in us:
                p = 8 + 2 * u
                a(f"v_pk_mul_f32 v[{p}:{p + 1}], v[{p}:{p + 1}], %[res2]")      # * res
            for u in us:
                p = 8 + 2 * u
                a(f"v_pk_add_f32 v[{p}:{p + 1}], v[{p}:{p + 1}], %[offv]")      # + centre / resolution
        if len(us) == 1:
            a("s_nop 0")
        if clamp:   # (the variants without: every cell the workgroup's windows can reach in this sector lies inside the map)
            for u in us:
                p = 8 + 2 * u
                a(f"v_med3_f32 v{p}, v{p}, %[rmax], -1.0")                       # clamp into the guard ring
                a(f"v_med3_f32 v{p + 1}, v{p + 1}, %[cmax], -1.0")
        for u in us:
            p = 8 + 2 * u
            a(f"v_pk_add_f32 v[{p}:{p + 1}], v[{p}:{p + 1}], %[half]")          # round_half_away_clamped
        if len(us) == 1:
            a("s_nop 0")
        for u in us:
            p = 8 + 2 * u
            a(f"v_cvt_flr_i32_f32 v{p}, v{p}")
            a(f"v_cvt_flr_i32_f32 v{p + 1}, v{p + 1}")
    if mask:
        coords(range(4))
    # (without the mask an empty bin needs no cell at all: the coordinates are computed with the record offset below)
    if mask:
        # word of the staged known mask (LDS) of each cell: ri * krow4 + (ci >> 5) * 4 + kconst
        for u in range(4):
            a(f"v_mad_i32_i24 v{16 + u}, v{8 + 2 * u}, %[krow4], %[kconst]")
        for u in range(4):
            a(f"v_ashrrev_i32 v{20 + u}, 5, v{9 + 2 * u}")
        for u in range(4):
            a(f"v_lshl_add_u32 v{16 + u}, v{20 + u}, 2, v{16 + u}")
        a("s_nop 0")
        for u in range(4):
            a(f"ds_read_b32 v{28 + u}, v{16 + u}")
    # the cell of the class's plane for every non-empty bin (plane_offset, tdr_score_dev.h: (c >> 3) * pkcol + 2 c + 16 r +
    # the plane's constant): 2 bytes — the dictionary index * 4 in bits 2..11.  (Until round 4 this was the dword of the
    # compact record the class lives in: 4 x 4-cell tiles, three times the lines per gather of a wave whose particles lie a
    # few cells apart — and the lines a gather touches are what bounds the kernel.)
    for u in range(4):
        code, ckc = D + 4 * u, D + 2 + 4 * u
        a(f"s_cmp_eq_u32 s{code}, 0")
        a(f"s_cbranch_scc1 .Lsu_a{u}{tag}%=")
        if not mask:
            coords([u])
        a(f"v_ashrrev_i32 v{20 + u}, 3, v{9 + 2 * u}")
        a("s_nop 0")
        a(f"v_mad_i32_i24 v{20 + u}, v{20 + u}, %[pkcol], s{ckc}")
        a("s_nop 0")
        a(f"v_lshl_add_u32 v{20 + u}, v{9 + 2 * u}, 1, v{20 + u}")
        a("s_nop 0")
        a(f"v_lshl_add_u32 v{20 + u}, v{8 + 2 * u}, 4, v{20 + u}")
        a("s_nop 0")
        # (never into its own address register: a load that is replayed reads its address again)
        if EXPERIMENT_LDS_GATHER:
            # TIMING EXPERIMENT ONLY (wrong results): what the loop would cost if the class-plane cell came out of LDS instead
            # of a 64-lane global gather — an upper bound on what staging the planes in LDS could gain
            a(f"v_and_b32 v{20 + u}, 0x3ffe, v{20 + u}")
            a("s_nop 0")
            a(f"ds_read_u16 v{24 + u}, v{20 + u}")
        else:
            a(f"global_load_ushort v{24 + u}, v{20 + u}, %[crec]")
        a(f".Lsu_a{u}{tag}%=:")
    # ---- phase B
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if mask:
        for u in range(4):
            a(f"v_bfe_i32 v{16 + u}, v{28 + u}, v{9 + 2 * u}, 1")               # 0 / -1: the cell's known bit
        a("v_add_u32 v20, v16, v17")
        a("v_add_u32 v21, v18, v19")
        a("s_nop 0")
        a("v_add_u32 v20, v20, v21")
        a("s_nop 0")
        a("v_sub_u32 %[known], %[known], v20")
    # (every cell the sector can reach known: four more known samples per step, counted at the exit)
    for u in range(4):
        code, val = D + 4 * u, D + 1 + 4 * u
        a(f"s_cmp_eq_u32 s{code}, 0")
        a(f"s_cbranch_scc1 .Lsu_b{u}{tag}%=")
        if mask:
            a(f"v_and_b32 v20, s{val}, v{16 + u}")                               # the bin's count x known (:141-142)
        a(f"v_and_b32 v21, 0xffc, v{24 + u}")                                    # the class's dictionary index * 4
        if mask:
            a("v_add_u32 %[norm], %[norm], v20")
        else:   # every cell known: the normalisation is the same for every lane — summed in a scalar, added at the exit
            a(f"s_add_u32 s96, s96, s{val}")
        a("ds_read_b32 v21, v21")                                                # the dictionary sits at LDS address 0
        a("s_waitcnt lgkmcnt(0)")
        # acc[class] += count * distance (state_particle.cpp:136-139), as integers.  The six 64-bit accumulators sit in FIXED
        # registers v[32:43] (the statement's a0..a5 operands are tied to them) and the class picks its pair by VGPR-relative
        # indexing: destination and addend are v[30:31] + 2 * code (tools/gpr_idx_probe.hip checks the mode on gfx950) — three
        # scalar instructions and no branch where a tree of compares and branches stood (round 5).
        # (Tried in round 5: all four bins' dictionary reads in flight together and their products without a branch — an empty
        # bin has a count of zero — so that a step waits for the LDS once: 2.93 against 2.83 ms; the extra reads and
        # multiply-adds cost more than the waits.)
        a(f"s_lshl_b32 s64, s{code}, 1")
        a("s_set_gpr_idx_on s64, 0xc")                                                       # SRC2 | DST relative
        a(f"v_mad_u64_u32 v[30:31], vcc, s{val}, v21, v[30:31]")
        a("s_set_gpr_idx_off")
        a(f".Lsu_b{u}{tag}%=:")
    # ---- next step
    a("s_mov_b32 s65, s69")
    a("s_mov_b32 s66, s70")
    a("s_mov_b32 s68, s71")
    a("s_sub_u32 s67, s67, 1")
    if tag == "e":
        a("s_cbranch_scc1 .Lsu_out%=")               # (falls through into the other copy)
    else:
        a(f"s_cbranch_scc0 .Lsu_step{nxt}%=")


# ---- static guard -------------------------------------------------------------------------------------------------
# The loop names its registers itself; the compiler only knows what the statement's operand and clobber lists say.  Three
# rules keep the two in step (a violation showed up once as memory faults at full size only):
#   1. the text begins by draining the compiler's own loads in flight (s_waitcnt vmcnt(0) lgkmcnt(0));
#   2. every register an instruction WRITES is a fixed register of the clobber list or a named operand the statement
#      declares as an output ("+v" / "=v" in SU_ASM_OPERANDS of tdr_score_su.hip);
#   3. every fixed register an instruction READS is inside the clobber list too (nothing outside the plan is touched), and
#      the named inputs are never written.
CLOBBER_V = range(8, 32)
CLOBBER_S = range(40, 98)
NO_DEST = ("s_waitcnt", "s_nop", "s_branch", "s_cbranch_", "s_cmp_", "s_bitcmp")   # write nothing / SCC only


def statement_operands(source="tdr_score_su.hip", macro="SU_ASM_OPERANDS", clobbers="SU_ASM_CLOBBERS"):
    """(outputs, inputs) named in the operand macro of the kernel source."""
    import re
    src = open(os.path.join(ROOT, "top_down_renderer_amd", "csrc", source)).read()
    blk = src[src.index("#define " + macro):src.index(": " + clobbers)]
    ops = re.findall(r'\[(\w+)\]\s*"([^"]+)"', blk)
    return {n for n, c in ops if c[0] in "+="}, {n for n, c in ops if c[0] not in "+="}


def regs_of(tok):
    """Fixed registers a textual operand names: 'v12' -> [('v', 12)], 's[40:47]' -> s40..s47; named / literal -> []."""
    import re
    m = re.fullmatch(r"([vs])(\d+)", tok)
    if m:
        return [(m.group(1), int(m.group(2)))]
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
    if m:
        return [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    return []


def check_text(lines, outputs, inputs, clobber_v=None, clobber_s=None):
    import re
    clobber_v = CLOBBER_V if clobber_v is None else clobber_v
    clobber_s = CLOBBER_S if clobber_s is None else clobber_s
    assert lines[0] == "s_waitcnt vmcnt(0) lgkmcnt(0)", "the loop must begin by draining the loads in flight"
    named_written = set()
    # VGPR-relative indexing: exactly one instruction between an `on` and its `off` — the accumulate, whose destination and
    # addend name the pair below the accumulators (+ 2 * class code reaches the registers a0..a5 are tied to)
    for k, ln in enumerate(lines):
        if ln.startswith("s_set_gpr_idx_on"):
            assert re.fullmatch(r"s_set_gpr_idx_on s\d+, 0xc", ln), ln
            assert re.fullmatch(r"v_mad_u64_u32 (v\[\d+:\d+\]), vcc, s\d+, v\d+, \1", lines[k + 1]) and \
                lines[k + 2] == "s_set_gpr_idx_off", f"only the accumulate may run under the GPR index: {lines[k + 1]}"
    for ln in lines:
        if ln.endswith(":"):
            continue
        mn, _, rest = ln.partition(" ")
        toks = [t.strip() for t in re.split(r",(?![^\[]*\])", rest)] if rest else []
        writes_dest = not mn.startswith(NO_DEST)
        for k, t in enumerate(toks):
            named = re.fullmatch(r"%\[(\w+)\]", t)
            is_dest = writes_dest and k == 0
            if named:
                assert named.group(1) in outputs | inputs, f"unknown operand in: {ln}"
                if is_dest:
                    assert named.group(1) in outputs, f"writes an input operand: {ln}"
                    named_written.add(named.group(1))
                continue
            for kind, i in regs_of(t):
                ok = i in (clobber_v if kind == "v" else clobber_s)
                assert ok, f"{'writes' if is_dest else 'reads'} {kind}{i}, which is outside the clobber list: {ln}"
    return named_written


def check_all():
    outputs, inputs = statement_operands()
    n = 0
    for uscale in (True, False):
        for clamp, mask in ((True, True), (False, True), (False, False)):
            written = check_text(loop_text(uscale, clamp, mask), outputs, inputs)
            assert written <= outputs
            n += 1
    return n


def main():
    global EXPERIMENT_LDS_GATHER
    EXPERIMENT_LDS_GATHER = "--experiment-lds-gather" in __import__("sys").argv
    check_all()
    out = ["// tdr_score_su_asm.h — GENERATED by tools/gen_su_asm.py; do not edit.",
           "// The inner loop of score_polar_su_kernel for two-dword compact records (4-6 classes), see the generator for the",
           "// register plan and the schedule's cost model.",
           "#ifndef TDR_SCORE_SU_ASM_H_", "#define TDR_SCORE_SU_ASM_H_", ""]
    for uscale in (True, False):
        for clamp, mask, tag in ((True, True, ""), (False, True, "_NOCLAMP"), (False, False, "_ALLKNOWN")):
            out.append(f"#define SU_ASM_{'US' if uscale else 'PS'}{tag} \\")
            lines = loop_text(uscale, clamp, mask)
            for i, ln in enumerate(lines):
                out.append(f'  "{ln}\\n"' + (" \\" if i + 1 < len(lines) else ""))
            out.append("")
    out.append('#define SU_ASM_CLOBBERS                                                                                      \\')
    vregs = ", ".join(f'"v{i}"' for i in CLOBBER_V)
    sregs = ", ".join(f'"s{i}"' for i in CLOBBER_S)
    out.append(f"  {vregs}, \\")
    out.append(f'  {sregs}, "vcc", "m0", "memory"')
    out.append("#endif  // TDR_SCORE_SU_ASM_H_")
    open(OUT, "w").write("\n".join(out) + "\n")
    print(OUT)


if __name__ == "__main__":
    main()
