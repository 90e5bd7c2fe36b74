"""Static guard of the generated gfx950 loops (tools/gen_su_asm.py: the polar shift-uniform kernel's; tools/gen_cart_asm.py:
the Cartesian dense kernel's): what a text writes is what its statement declares, and — for the Cartesian loop, whose loads
all live inside the text — no load's destination is read before a wait that covers it."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gen():
    spec = importlib.util.spec_from_file_location("gen_su_asm", os.path.join(ROOT, "tools", "gen_su_asm.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_every_variant_passes_the_register_rules():
    assert gen().check_all() == 6


def test_committed_header_is_the_generators_output(tmp_path):
    g = gen()
    g.OUT = str(tmp_path / "asm.h")
    g.main()
    committed = open(os.path.join(ROOT, "top_down_renderer_amd", "csrc", "tdr_score_su_asm.h")).read()
    assert open(g.OUT).read() == committed


@pytest.mark.parametrize("bad, why", [
    (["v_mov_b32 v40, v8"], "outside the clobber list"),              # writes a register the compiler may own
    (["v_add_u32 v20, v7, v8"], "outside the clobber list"),          # reads one
    (["s_mov_b32 s20, 0"], "outside the clobber list"),
    (["s_load_dwordx8 s[92:99], %[tb], s65"], "outside the clobber list"),
    (["v_mov_b32 %[offv], v8"], "writes an input operand"),
])
def test_the_guard_sees_a_violation(bad, why):
    g = gen()
    outputs, inputs = g.statement_operands()
    with pytest.raises(AssertionError, match=why):
        g.check_text(["s_waitcnt vmcnt(0) lgkmcnt(0)"] + bad, outputs, inputs)


def test_the_guard_wants_the_drain_first():
    g = gen()
    outputs, inputs = g.statement_operands()
    with pytest.raises(AssertionError, match="draining"):
        g.check_text(g.loop_text(True, True)[1:], outputs, inputs)


# ---- the Cartesian loop (tools/gen_cart_asm.py, csrc/tdr_score_cart_asm.h) ------------------------------------------------
def gen_cart():
    spec = importlib.util.spec_from_file_location("gen_cart_asm", os.path.join(ROOT, "tools", "gen_cart_asm.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_cartesian_variants_pass_the_register_and_wait_rules():
    assert gen_cart().check_all() == 3


def test_committed_cartesian_header_is_the_generators_output(tmp_path):
    g = gen_cart()
    g.OUT = str(tmp_path / "asm.h")
    g.main()
    committed = open(os.path.join(ROOT, "top_down_renderer_amd", "csrc", "tdr_score_cart_asm.h")).read()
    assert open(g.OUT).read() == committed


@pytest.mark.parametrize("bad, why", [
    (["v_mov_b32 v42, v8"], "outside the clobber list"),
    (["s_mov_b32 s47, 0"], "outside the clobber list"),
    (["v_mov_b32 %[ab3], v8"], "writes an input operand"),
    # a load whose value is used before any wait: exactly the hazard the generated text exists to rule out
    (["global_load_ushort v24, v20, %[crec]", "v_and_b32 v21, 0xffc, v24"], "before its wait"),
    (["ds_read_b32 v28, v16", "v_bfe_i32 v16, v28, v9, 1"], "before its wait"),
    (["s_load_dwordx16 s[80:95], %[db], s67", "s_cmp_eq_u32 s80, 0"], "before its wait"),
    (["global_load_ushort v24, v20, %[crec]", "v_mov_b32 v24, v8"], "in flight"),
    (["global_load_ushort v24, v20, %[crec]"], "ends with a load in flight"),
])
def test_the_cartesian_guard_sees_a_violation(bad, why):
    g = gen_cart()
    outputs, inputs = g.statement_operands()
    with pytest.raises(AssertionError, match=why):
        g.check_text(["s_waitcnt vmcnt(0) lgkmcnt(0)"] + bad, outputs, inputs)


def test_no_inline_assembly_load_or_wait_outside_the_generated_texts():
    """Rounds 3-4 issued loads through separate inline-assembly statements with hand-counted s_waitcnt (the compiler moved a
    copy between a load and its wait twice).  The pattern must not come back: in the two kernels' sources, no asm statement
    other than the generated loops' may contain a load or a wait."""
    import re
    for f in ("tdr_score_su.hip", "tdr_score_cart.hip"):
        src = open(os.path.join(ROOT, "top_down_renderer_amd", "csrc", f)).read()
        for m in re.finditer(r'asm\s*(?:volatile)?\s*\(\s*"([^"]*)"', src):
            text = m.group(1)
            assert not re.search(r"global_load|buffer_load|flat_load|ds_read|s_load|s_waitcnt", text), f"{f}: {text}"
