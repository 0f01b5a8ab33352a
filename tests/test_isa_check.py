"""The build's ISA check (tools/check_isa.py): inline-asm scalar loads that are still in flight must not
be touched by compiler-scheduled code.  Synthetic assembly exercises the analysis; the real kernel is
checked by __graft_entry__.build()."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "tools", "check_isa.py"))
check_isa = importlib.util.module_from_spec(spec)
spec.loader.exec_module(check_isa)


def kernel(body: str) -> str:
    return "k1:                                    ; @k1\n" + body + "\n\ts_endpgm\n.end_amdhsa_kernel\n"


LOAD = "\t;;#ASMSTART\n\ts_load_dwordx4 s[8:11], s[2:3], 0\n\t;;#ASMEND\n"
WAIT = "\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n"


def test_use_after_wait_is_clean():
    asm = kernel(LOAD + "\tv_mov_b32_e32 v0, s4\n" + WAIT + "\tv_mov_b32_e32 v1, s9\n")
    assert check_isa.check(asm) == []


def test_read_and_spill_before_wait_are_flagged():
    asm = kernel(LOAD + "\tv_mov_b32_e32 v1, s9\n\tv_writelane_b32 v5, s10, 3\n" + WAIT)
    problems = check_isa.check(asm)
    assert len(problems) == 2 and "s[9]" in problems[0] and "s[10]" in problems[1]


def test_compiler_waits_do_not_count():
    asm = kernel(LOAD + "\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v1, s[8:9]\n" + WAIT)
    assert len(check_isa.check(asm)) == 1


def test_hazard_reached_through_a_branch():
    # the wait sits on the fall-through path only; the taken branch reaches a use without it
    asm = kernel(LOAD + "\ts_cbranch_scc1 .LBB0_2\n" + WAIT + ".LBB0_2:\n\tv_add_f32_e32 v0, s11, v0\n")
    assert len(check_isa.check(asm)) == 1
    # with the wait ahead of the branch both paths are clean
    asm = kernel(LOAD + WAIT + "\ts_cbranch_scc1 .LBB0_2\n\tv_mov_b32_e32 v2, s8\n.LBB0_2:\n\tv_add_f32_e32 v0, s11, v0\n")
    assert check_isa.check(asm) == []


def test_loop_carried_load():
    # a load issued at the bottom of a loop and waited for at its top: clean; touching it in between: flagged
    loop = ".LBB0_1:\n" + WAIT + "\tv_mov_b32_e32 v0, s8\n" + LOAD + "{extra}\ts_cbranch_scc1 .LBB0_1\n" + WAIT
    assert check_isa.check(kernel(LOAD + loop.format(extra=""))) == []
    assert len(check_isa.check(kernel(LOAD + loop.format(extra="\ts_mov_b32 s20, s11\n")))) == 1


def test_render_kernels_with_scratch_memory_are_refused():
    """A render kernel that spills registers to scratch memory is reported; device functions that are
    called rather than launched, and other kernels, are not."""
    def info(kind, scratch):
        return f"; {kind} info:\n; codeLenInByte = 8\n; NumVgprs: 96\n; ScratchSize: {scratch}\n; Occupancy: 5\n"
    asm = ("\t.amdhsa_kernel _Z13render_kernelILi2ELb0EEv\n\t.end_amdhsa_kernel\n" + info("Kernel", 12) +
           info("Function", 128) +
           "\t.amdhsa_kernel _Z13render_kernelILi3ELb0EEv\n\t.end_amdhsa_kernel\n" + info("Kernel", 0) +
           "\t.amdhsa_kernel _Z16first_hit_kernelILb1EEv\n\t.end_amdhsa_kernel\n" + info("Kernel", 256))
    found = check_isa.scratch_users(asm)
    assert len(found) == 1 and "ILi2ELb0" in found[0] and "12" in found[0]


def test_packed_fp32_with_an_sgpr_pair_and_op_sel_is_refused():
    """The form that returned wrong lanes on the GPU (tools/hazard_pk.hip) is reported wherever it stands; the same
    instruction with VGPR sources, or with an SGPR pair in the plain pairing, is not."""
    asm = ("k1:                                    ; @k1\n"
           "\tv_pk_fma_f32 v[6:7], s[4:5], v[6:7], v[4:5] op_sel:[1,0,1] op_sel_hi:[0,0,0]\n"
           "\tv_pk_mul_f32 v[28:29], s[64:65], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]\n"
           "\tv_pk_fma_f32 v[6:7], v[8:9], v[6:7], v[4:5] op_sel:[1,0,1] op_sel_hi:[0,0,0]\n"
           "\tv_pk_fma_f32 v[6:7], s[4:5], v[6:7], v[4:5]\n"
           "\tv_pk_fma_f32 v[42:43], v[40:41], s[20:21], v[4:5] op_sel_hi:[1,1,1]\n"
           "\tv_pk_add_f32 v[4:5], s[18:19], v[4:5] neg_lo:[0,1] neg_hi:[0,1]\n"
           "\tv_pk_mul_f32 v[38:39], s[10:11], v[6:7] op_sel_hi:[1,0]\n"
           "\ts_endpgm\n.end_amdhsa_kernel\n")
    found = check_isa.packed_sgpr_opsel(asm)
    assert len(found) == 3, found
    assert "op_sel:[1,0,1]" in found[0] and "s[64:65]" in found[1] and "op_sel_hi:[1,0]" in found[2]
