#!/usr/bin/env python3
"""Checks the compiled gfx950 code for hazards the compiler does not see.

The scan loop of the render kernel requests sphere words with inline-asm scalar loads and waits for
them later, with compiler-scheduled code in between (that overlap is the point).  To the compiler the
destination registers hold their values as soon as the load statement ends, so under register
pressure it may spill or copy them before the data has arrived.  This script reads the device
assembly, follows every path of each kernel's control-flow graph, and fails if an instruction that
can execute between such a load and the next inline-asm `s_waitcnt lgkmcnt(0)` touches one of the
load's destination registers.

It also refuses a build in which a render kernel spills registers to scratch memory (scratch_users), and one
in which any kernel holds a packed FP32 instruction that takes an SGPR pair as a source and re-routes halves with
op_sel (packed_sgpr_opsel): on gfx950 that form returned wrong low results in lanes 48-63 whenever other waves of
the CU were running matrix instructions (tools/hazard_pk.hip reproduces it).

usage: check_isa.py [file.s]     (without a file: compiles raytracer_amd/csrc/rt_amd.hip to assembly first)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def device_asm() -> str:
    src = os.path.join(ROOT, "raytracer_amd", "csrc", "rt_amd.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "rt_amd.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                               "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
        return open(out).read()


def sregs(text: str) -> set:
    """Scalar registers named in an operand string."""
    regs = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bs(\d+)\b", text))
    return regs


def parse_kernel(body: str):
    """-> list of (label or None, [(line number, text, in inline asm)])  basic blocks in layout order"""
    blocks, cur, in_asm = [], (None, []), False
    for n, raw in enumerate(body.split("\n")):
        st = raw.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        line = raw.split(";")[0].strip()
        if not line or line.startswith("."):
            m = re.match(r"^(\.LBB\w+):", line)
            if not m:
                continue
        m = re.match(r"^([.\w$]+):$", line)
        if m:
            blocks.append(cur)
            cur = (m.group(1), [])
            continue
        cur[1].append((n, line, in_asm))
        if line.startswith(("s_branch", "s_cbranch", "s_endpgm")):
            blocks.append(cur)
            cur = (None, [])
    blocks.append(cur)
    return [b for b in blocks if b[0] is not None or b[1]]


def check(asm: str) -> list:
    """Forward data-flow over each kernel's control-flow graph: the state is the set of scalar
    registers with an inline-asm load in flight; an inline-asm `s_waitcnt lgkmcnt(0)` empties it."""
    problems = []
    for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)\.end_amdhsa_kernel", asm, re.M | re.S):
        name, body = m.group(1), m.group(2)
        blocks = parse_kernel(body)
        index = {lab: i for i, (lab, _) in enumerate(blocks) if lab}
        state_in = [None] * len(blocks)
        state_in[0] = frozenset()
        work = [0]
        seen = set()
        while work:
            i = work.pop()
            pending = set(state_in[i])
            succ = [i + 1] if i + 1 < len(blocks) else []
            for n, line, in_asm in blocks[i][1]:
                if in_asm and line.startswith("s_waitcnt") and "lgkmcnt(0)" in line:
                    # only the waits written next to the loads count: the compiler's own waits are
                    # for its own loads
                    pending.clear()
                    continue
                if in_asm and line.startswith("s_load_dword"):
                    dest = sregs(line.split(",")[0])
                    used = sregs(",".join(line.split(",")[1:]))
                    hit = pending & (used | dest)
                    if hit and (n, "load") not in seen:
                        seen.add((n, "load"))
                        problems.append(f"{name}: line {n}: '{line}' touches s{sorted(hit)} of a load in flight")
                    pending |= dest
                    continue
                hit = pending & sregs(line)
                if hit and n not in seen:
                    seen.add(n)
                    problems.append(f"{name}: line {n}: '{line}' touches s{sorted(hit)} of a load in flight")
                if line.startswith("s_branch"):
                    succ = [index[line.split()[1]]] if line.split()[1] in index else []
                elif line.startswith("s_cbranch"):
                    tgt = line.split()[1]
                    if tgt in index:
                        succ = succ + [index[tgt]]
                elif line.startswith("s_endpgm"):
                    succ = []
            out = frozenset(pending)
            for j in succ:
                merged = out if state_in[j] is None else state_in[j] | out
                if merged != state_in[j]:
                    state_in[j] = merged
                    work.append(j)
    return problems


def scratch_users(asm: str) -> list:
    """Render kernels that would need private scratch memory (register spills to memory).  The runtime allocates
    that memory at launch, which the entry point that promises to allocate nothing cannot have, and a spill
    inside the path loop is a memory access per iteration: a build with one is refused."""
    found, name, in_kernel_info = [], None, False
    for line in asm.splitlines():
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            name = m.group(1)
        if line.startswith("; Kernel info:"):
            in_kernel_info = True   # the block of the kernel named last
        elif line.startswith("; Function info:"):
            in_kernel_info = False  # a device function that is called, not launched
        m = re.match(r";\s*ScratchSize:\s*(\d+)", line)
        if m and in_kernel_info and name and "render_kernel" in name and int(m.group(1)) > 0:
            found.append(f"{name}: ScratchSize {m.group(1)}")
    return found


def packed_sgpr_opsel(asm: str) -> list:
    """Packed FP32 instructions (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32) with an SGPR pair among their sources and an
    op_sel / op_sel_hi that is not the plain pairing (low halves -> low result, high halves -> high result).  Measured
    on MI355X (tools/hazard_pk.hip): in `v_pk_fma_f32 v[d:d+1], s[a:a+1], v[b:b+1], v[c:c+1] op_sel:[1,0,1] op_sel_hi:[0,0,0]`
    the addend of the low result, which op_sel takes from the high half of the third source, reads as 0 in lanes 48-63
    now and then, only while other waves of the CU issue MFMAs, and only when an SGPR pair is among the sources and
    op_sel re-routes the third one; VGPR sources throughout, or the plain pairing, never did.  The compiler's SLP
    vectoriser emits the form; the build switches that pass off, and this check keeps it from coming back unseen (it
    refuses every re-routing beside an SGPR pair, not only the combinations that were seen to fail)."""
    found, name = [], None
    for n, raw in enumerate(asm.splitlines(), 1):
        m = re.match(r"^(\w+):\s*; @\1", raw)
        if m:
            name = m.group(1)
        line = raw.split(";")[0].strip()
        m = re.match(r"v_pk_(fma|mul|add)_f32\S*\s+(.*)", line)
        if not m or not re.search(r"\bs\[\d+:\d+\]", m.group(2)):
            continue
        nsrc = 3 if m.group(1) == "fma" else 2
        sel = re.search(r"\bop_sel:\[([\d,]+)\]", line)
        sel_hi = re.search(r"\bop_sel_hi:\[([\d,]+)\]", line)
        plain = (sel is None or set(sel.group(1).split(",")) == {"0"}) and \
                (sel_hi is None or sel_hi.group(1).split(",")[:nsrc] == ["1"] * nsrc)
        if not plain:
            found.append(f"{name}: line {n}: '{line}'")
    return found


def main() -> int:
    asm = open(sys.argv[1]).read() if len(sys.argv) > 1 else device_asm()
    spilled = scratch_users(asm)
    for s in spilled:
        print("render kernel with scratch memory:", s)
    if spilled and not os.environ.get("RT_ALLOW_SCRATCH"):  # tuning builds (-DRT_STATS ...) may set this
        return 1
    risky = packed_sgpr_opsel(asm)
    for r in risky[:20]:
        print("packed FP32 with an SGPR pair and op_sel:", r)
    if risky:
        print(f"{len(risky)} packed FP32 instruction(s) of the form that returns wrong lanes beside matrix instructions")
        return 1
    problems = check(asm)
    for p in problems[:40]:
        print("ISA hazard:", p)
    if problems:
        print(f"{len(problems)} instruction(s) touch scalar registers whose load has not been waited for")
        return 1
    print("check_isa: no instruction touches an in-flight scalar load")
    return 0


if __name__ == "__main__":
    sys.exit(main())
