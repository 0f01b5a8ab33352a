"""Debug aid: variants of the library whose device code is edited at assembly level (wait states, register checks)
and assembled again, so that a fault which any change of the source makes disappear can be narrowed down.
How it was used (paths are this tool's own, under /tmp/asmbis):
  hipcc -v <the flags of the build as it was then, SLP vectoriser on> -DRT_MF_PROBE_FULL_LISTS (a probe switch of that time) -save-temps=obj -o lib.so raytracer_amd/csrc/rt_amd.hip > build.log 2>&1
  rebuild.sh = the sub-commands of build.log from the device assembler on (clang -cc1as of the gfx950 .s, lld,
  clang-offload-bundler, the host compile with the new bundle, the link); dev_orig.s = the device assembly as compiled.
Each variant below inserts lines at kernel-local line numbers of render_kernel<5, true> in dev_orig.s (BASE = the
kernel's first line), runs rebuild.sh and copies the library to tools/v_asm_<name>.so; tools/dbg_repeat_libs.sh and
tools/dbg_lane_masks.py run them.  The line numbers belong to that one compile: read them off again before reuse."""
import subprocess, shutil, sys, re
BASE = 33838  # kernel-local line L of render_kernel<5,true> is global line BASE + L (1-based)
src = open('/tmp/asmbis/dev_orig.s').read().split('\n')
def is_valu(line):
    t = line.strip()
    return t.startswith('v_') 
def build(name, edit):
    lines = list(src)
    ins = edit(lines)   # dict: global 1-based line -> ('before'|'after', text)
    out = []
    for i, l in enumerate(lines, 1):
        if i in ins and ins[i][0] == 'before': out.append(ins[i][1])
        out.append(l)
        if i in ins and ins[i][0] == 'after': out.append(ins[i][1])
    open('/tmp/asmbis/rt_amd-hip-amdgcn-amd-amdhsa-gfx950.s', 'w').write('\n'.join(out))
    subprocess.check_call(['bash', '/tmp/asmbis/rebuild.sh'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    shutil.copy('/tmp/asmbis/lib.so', f'/root/repo/tools/v_asm_{name}.so')
    print(name, 'built,', len(ins), 'insertions')
NOP = '\ts_nop 7'
G = lambda L: BASE + L
variants = {
    'orig': lambda ls: {},
    'A_before_selects': lambda ls: {G(L): ('before', NOP) for L in (4641, 4643, 4650)},
    'B_after_selects': lambda ls: {G(L): ('after', NOP) for L in (4641, 4643, 4650)},
    'C_after_rsq': lambda ls: {G(L): ('after', NOP) for L in (4600, 4631)},
    'D_all_pyramid': lambda ls: {G(L): ('after', NOP) for L in range(4520, 4689) if is_valu(ls[G(L) - 1])},
    'E_all_loop': lambda ls: {G(L): ('after', NOP) for L in range(4729, 4816) if is_valu(ls[G(L) - 1])},
}
variants.update({
    'F_exec_full': lambda ls: {G(4520): ('before', '\ts_mov_b64 exec, -1')},
    'G_drain_before_pyramid': lambda ls: {G(4520): ('before', '\ts_waitcnt vmcnt(0) lgkmcnt(0)')},
    'H_drain_in_loop': lambda ls: {G(4729): ('before', '\ts_waitcnt vmcnt(0) lgkmcnt(0)')},
})
def uni(regs):
    return '\n'.join(f'\tv_readfirstlane_b32 s4, v{r}\n\ts_nop 3\n\tv_mov_b32_e32 v{r}, s4' for r in regs)
variants.update({
    'I_uniform_all': lambda ls: {G(4702): ('before', uni([4, 5] + list(range(26, 37))))},
    'J_uniform_plane1': lambda ls: {G(4702): ('before', uni([31, 32, 33]))},
    'K_uniform_others': lambda ls: {G(4702): ('before', uni([4, 5, 26, 27, 28, 29, 30, 34, 35, 36]))},
})
PRE = '''\ts_load_dwordx2 s[100:101], s[0:1], 0x180
\tv_mbcnt_lo_u32_b32 v146, -1, 0
\tv_mbcnt_hi_u32_b32 v146, -1, v146
\tv_lshlrev_b64 v[148:149], v146, 1
\ts_waitcnt lgkmcnt(0)
\tv_mov_b32_e32 v150, s100
\tv_mov_b32_e32 v151, s101
\ts_nop 3'''
def chk(regs, off):
    return '\n'.join(f'''\tv_readfirstlane_b32 s100, v{r}
\ts_nop 4
\tv_cmp_ne_u32_e32 vcc, s100, v{r}
\ts_nop 4
\ts_and_saveexec_b64 s[100:101], vcc
\tglobal_atomic_or_x2 v[150:151], v[148:149], off offset:{off}
\ts_mov_b64 exec, s[100:101]
\ts_waitcnt vmcnt(0)''' for r in regs)
def with_regs(fn):
    # the kernel's descriptor: six more VGPRs (v146-v151) and two more SGPRs (s100, s101) for the checks
    def edit(ls):
        k0 = BASE
        for i in range(k0, len(ls)):
            if '.amdhsa_next_free_vgpr 146' in ls[i]: ls[i] = ls[i].replace('146', '152')
            elif '.amdhsa_next_free_sgpr 100' in ls[i]: ls[i] = ls[i].replace('100', '102')
            elif '.amdhsa_accum_offset 148' in ls[i]: ls[i] = ls[i].replace('148', '152'); break
        return fn(ls)
    return edit
variants.update({
    'P_end': with_regs(lambda ls: {G(4702): ('before', PRE + '\n' + chk([31, 32, 33], 8) + '\n' + chk([4, 5, 26, 27, 28, 29, 30, 34, 35, 36], 16))}),
})
variants.update({
    'Q1': with_regs(lambda ls: {G(4532): ('before', PRE + '\n' + chk([6, 7, 9, 27, 4, 5], 8))}),
    'Q2': with_regs(lambda ls: {G(4553): ('before', PRE + '\n' + chk([6, 7, 8, 9, 32, 34, 26, 28, 30, 36, 4, 5], 8))}),
    'Q3': with_regs(lambda ls: {G(4607): ('before', PRE + '\n' + chk([6, 7, 38, 39, 40, 41, 32, 33], 8))}),
})
variants.update({
    'R1': with_regs(lambda ls: {G(4538): ('after', PRE + '\n' + chk([4, 5], 8) + '\n' + chk([30, 8, 6], 16))}),
    'R2': with_regs(lambda ls: {G(4540): ('after', PRE + '\n' + chk([6, 7], 8) + '\n' + chk([32], 16))}),
    'R3': with_regs(lambda ls: {G(4542): ('after', PRE + '\n' + chk([34], 8) + '\n' + chk([26], 16))}),
    'R4': with_regs(lambda ls: {G(4550): ('after', PRE + '\n' + chk([8, 9], 8) + '\n' + chk([40, 41, 33, 44, 45, 46, 47], 16))}),
})
def rec(reg, tag):
    return f'''\tv_readfirstlane_b32 s100, v{reg}
\ts_nop 4
\tv_cmp_ne_u32_e32 vcc, s100, v{reg}
\ts_nop 4
\ts_cmp_eq_u64 vcc, 0
\ts_cbranch_scc1 .Ldbg_skip_{tag}
\tv_readlane_b32 s101, v{reg}, 63
\ts_nop 4
\tv_mov_b32_e32 v148, 0
\tv_mov_b32_e32 v149, s101
\tv_mov_b32_e32 v147, s100
\ts_mov_b64 vcc, exec
\ts_mov_b64 exec, 1
\tglobal_atomic_umax_x2 v[150:151], v[148:149], off offset:8
\ts_nop 1
\tv_mov_b32_e32 v149, v147
\ts_nop 1
\tglobal_atomic_umax_x2 v[150:151], v[148:149], off offset:16
\ts_mov_b64 exec, vcc
\ts_waitcnt vmcnt(0)
.Ldbg_skip_{tag}:'''
variants.update({
    'S9': with_regs(lambda ls: {G(4550): ('after', PRE + '\n' + rec(9, 's9'))}),
    'S8': with_regs(lambda ls: {G(4550): ('after', PRE + '\n' + rec(8, 's8'))}),
    'S7': with_regs(lambda ls: {G(4540): ('after', PRE + '\n' + rec(7, 's7'))}),
    'S6': with_regs(lambda ls: {G(4540): ('after', PRE + '\n' + rec(6, 's6'))}),
})
for n in (sys.argv[1:] or variants):
    build(n, variants[n])
