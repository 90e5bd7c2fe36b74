"""Static guard of the generated gfx950 loop (tools/gen_su_asm.py): what the text writes is what the statement declares."""
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
