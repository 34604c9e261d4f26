"""Lint the compiled kernels for the one thing hipcc may legally do to a hidden (inline-asm) load that breaks it:
touch the destination register between the load and the wait that releases it.

An `asm volatile("buffer_load_dwordx4 %0, ...")` is one opaque instruction to the compiler: it neither counts
the load nor knows that %0 is not written yet when the statement ends (cdna_hip_programming.md 5.7).  Our
kernels tie every such register to its counted `s_waitcnt` with a "+v" operand, which orders the USES - but
the register allocator may still insert a copy (`v_mov`) of the in-flight register ahead of the wait, typically
at a control-flow merge; the wait then names the copy and the kernel reads stale data on the waves whose
load had not landed (this happened: round 2, two waits in an if / else).  This script replays each kernel's
instruction stream in layout order and reports every instruction that reads or writes a VGPR while a hidden
load into it is still outstanding.

    python tools/audit_asm_loads.py [file.s ...]      # no arguments: compiles the instantiation units to .s

Exit status 1 if anything is flagged.  A linear replay cannot follow branches, so loop-carried state is
approximate: it is a lint for the straight-line mistakes, not a proof.

Round 5 added a second rule, on every instruction of the unit: a VGPR written by a v_dot* instruction may be read by another VALU
instruction (or by a v_dot* as its A / B operand) only three wait states later (gfx90a+).  hipcc's hazard recognizer keeps that for
the instructions it schedules but does not look into asm statements: a per-word bf16 scale multiply written as v_dot2_f32_bf16 asm
statements with the conversion right behind them computed wrong values (tests/test_qgemm_gpu.py caught it; common.h mul_scale4 is the
form that keeps the distance by construction)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flute_amd", "csrc")
UNITS = ["inst_stream_b4.hip", "inst_stream_b3.hip", "inst_stream_b2.hip", "inst_block_b4.hip", "inst_block_b2.hip", "inst_block_b3.hip",
         "inst_oneshot_b4_f16.hip", "inst_oneshot_b4_bf16.hip", "inst_oneshot_b2_f16.hip", "inst_oneshot_b2_bf16.hip", "inst_oneshot_b3.hip",
         "inst_oneshot_persist_b4.hip", "inst_oneshot_persist_b2.hip", "inst_oneshot_persist_b3.hip",
         "inst_oneshot_skinny_b4.hip", "inst_oneshot_fast_b4.hip", "inst_oneshot_fastm_b4.hip", "inst_splitk.hip"]

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LOAD = re.compile(r"^\s*(buffer_load_dword\w*|global_load_dword\w*|global_load_lds_\w+)\s+(\S+),")
DSLOAD = re.compile(r"^\s*(ds_read_\w+)\s+(\S+),")
VMCNT = re.compile(r"s_waitcnt\b.*vmcnt\((\d+)\)")
LGKM = re.compile(r"s_waitcnt\b.*lgkmcnt\((\d+)\)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _sgpr_mentions(text, n):
    """(is_mentioned, only_as_destination) of SGPR n in one instruction."""
    ops = text.split(None, 1)
    if len(ops) < 2:
        return False, False
    parts = [o.strip() for o in ops[1].split(",")]

    def has(tok):
        if re.search(r"\bs%d\b" % n, tok):
            return True
        for m in re.finditer(r"s\[(\d+):(\d+)\]", tok):
            if int(m.group(1)) <= n <= int(m.group(2)):
                return True
        return False
    hits = [has(t) for t in parts]
    if not any(hits):
        return False, False
    return True, hits[0] and not any(hits[1:]) and not ops[0].startswith(("s_cmp", "s_cbranch", "s_bitcmp"))


def _pk_f32_used(text, regs_of):
    """Registers a packed-fp32 instruction really touches, operand by operand, or None for any other instruction.
    `v_pk_mul_f32 d, v[a:a+1], v[b:b+1] op_sel_hi:[0,1]` takes BOTH halves of source 0 from v[a] (op_sel picks the
    register of the low result, op_sel_hi of the high one; defaults [0,..] / [1,..]) - hipcc uses this to broadcast a
    scalar held in the low register of a pair whose high register belongs to something else.  Only the unread half of
    THAT source operand is dropped: the destination and the other sources stay in the set even when they name the
    same register."""
    m = re.match(r"v_pk_(?:mul|add|fma)_f32\s+(.*)$", text)
    if not m:
        return None
    body = m.group(1)
    mods = {"op_sel": None, "op_sel_hi": None}
    for name in mods:
        mm = re.search(name + r":\[([01,]+)\]", body)
        if mm:
            mods[name] = [int(x) for x in mm.group(1).split(",")]
    # operands are separated by commas outside brackets (v[4:5] has none, but be strict)
    ops = [o.strip() for o in re.split(r",(?![^\[]*\])", re.sub(r"\s+op_sel(_hi)?:\[[01,]+\]", "", body))]
    used = set(regs_of(ops[0]))                                # destination
    for i, o in enumerate(ops[1:]):                            # sources
        r = set(regs_of(o))
        mm = re.match(r"v\[(\d+):(\d+)\]$", o)
        if mm and int(mm.group(2)) == int(mm.group(1)) + 1:
            lo_sel = mods["op_sel"][i] if mods["op_sel"] and i < len(mods["op_sel"]) else 0
            hi_sel = mods["op_sel_hi"][i] if mods["op_sel_hi"] and i < len(mods["op_sel_hi"]) else 1
            picked = {lo_sel, hi_sel}
            if 0 not in picked:
                r.discard(int(mm.group(1)))
            if 1 not in picked:
                r.discard(int(mm.group(2)))
        used |= r
    return used


def _dead_readfirstlane(lines, i, text):
    """hipcc materialises an UNDEF scalar with `v_readfirstlane_b32 sN, <any vgpr>`: the value is dead when sN is
    overwritten before it is read.  Follows the fall-through path only (a taken branch in between = not proven)."""
    m = re.match(r"v_readfirstlane_b32\s+s(\d+),", text)
    if not m:
        return False
    n = int(m.group(1))
    for raw in lines[i + 1:i + 80]:
        t = raw.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":") or t.startswith(";;"):
            continue
        mentioned, dest_only = _sgpr_mentions(t, n)
        if mentioned:
            return dest_only
        if t.startswith(("s_branch", "s_endpgm", "s_setpc")):
            return False
    return False


DOT = re.compile(r"^(v_dot\w+)\s+(v\d+|v\[\d+:\d+\])\s*,")


def _dot_hazard(recent, text, used_srcs):
    """gfx90a+ (LLVM GCNHazardRecognizer: DotWriteDifferentVALURead = DotWriteSameDotReadSrcAB = 3): the VGPR a v_dot* instruction
    wrote may be read by another VALU instruction - or by a v_dot* as its A / B operand - only three wait states later (as the C
    operand of the next v_dot* at once).  hipcc keeps that for the instructions it schedules, NOT inside asm statements (round 5: a
    v_dot2_f32_bf16 asm statement whose result reached v_cvt_pk_bf16_f32 one instruction later computed wrong values).  `recent`:
    [(wait states since, dst regs)] of the last dots.  Returns the registers read too early."""
    if not text.startswith("v_"):
        return set()
    bad = set()
    m = DOT.match(text)
    for age, dst in recent:
        if age >= 3:
            continue
        if m:                                              # a dot reading an earlier dot's result: allowed as src C (the last source) only
            ops = [o.strip() for o in text[m.end():].split(",")]
            ab = set()
            for o in ops[:2]:
                ab |= regs_of(o)
            bad |= ab & dst
        else:
            bad |= used_srcs & dst
    return bad


SWAP = re.compile(r"^v_permlane(16|32)_swap_b32\s+(v\d+)\s*,\s*(v\d+)")
VALU_DST = re.compile(r"^v_\w+\s+(v\d+|v\[\d+:\d+\])\s*,")


def audit(path):
    findings = []
    all_lines = open(path).read().split("\n")
    kernel = None
    in_asm = False
    vm_fifo = []          # hidden VMEM loads in flight: (line number, set of VGPRs)
    ds_set = []           # hidden LDS loads in flight
    dots = []             # [wait states since issue, dst regs] of recent v_dot* instructions
    valu = []             # [wait states since issue, dst regs] of the last VALU writes (the lane-swap rule)
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";;#") else raw.strip()
        if raw.lstrip().startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if raw.lstrip().startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            kernel = m.group(1)
            vm_fifo, ds_set = [], []
            continue
        if not line.strip() or line.lstrip().startswith(".") or line.rstrip().endswith(":"):
            if re.match(r"^\.Lfunc_end", line):
                vm_fifo, ds_set = [], []
            continue
        text = line.strip()
        if text.startswith("s_endpgm") or text.startswith("s_branch") or text.startswith("s_setpc"):
            # what follows in layout order is not reached by falling through
            vm_fifo, ds_set = [], []
            continue
        mv = VMCNT.search(text)
        if mv:
            n = int(mv.group(1))
            if n < len(vm_fifo):
                vm_fifo = vm_fifo[len(vm_fifo) - n:] if n > 0 else []
        ml = LGKM.search(text)
        if ml:                                                 # LDS returns in order: all but the youngest N are done
            n = int(ml.group(1))
            if n < len(ds_set):
                ds_set = ds_set[len(ds_set) - n:] if n > 0 else []
        # ---- DOT result read too early (any instruction, asm statement or not) ----
        if text.startswith("s_nop"):
            mm = re.match(r"s_nop\s+(\d+)", text)
            for d in dots:
                d[0] += int(mm.group(1)) + 1 if mm else 1
        elif not text.startswith((".", ";")):
            md = DOT.match(text)
            srcs = regs_of(text.split(",", 1)[1]) if "," in text else set()
            early = _dot_hazard([(a, r) for a, r in dots], text, srcs)
            if early:
                findings.append((kernel, ln, text + "   [DOT result read within 3 wait states]", sorted(early)))
            for d in dots:
                d[0] += 1
            if md:
                dots.append([0, regs_of(md.group(2))])
            dots = [d for d in dots if d[0] < 3]
        # ---- a lane swap reading a VGPR a VALU instruction wrote less than two wait states earlier (LLVM's gfx950 rule "VALU write
        # vdst -> v_permlane*_swap read": hipcc pads its own swaps with s_nop 1, not those inside asm statements) ----
        if text.startswith("s_nop"):
            mm = re.match(r"s_nop\s+(\d+)", text)
            for w in valu:
                w[0] += int(mm.group(1)) + 1 if mm else 1
        elif not text.startswith((".", ";")):
            ms = SWAP.match(text)
            if ms:
                ops = regs_of(ms.group(2)) | regs_of(ms.group(3))
                early = set()
                for age, dst in valu:
                    if age < 2:
                        early |= ops & dst
                if early:
                    findings.append((kernel, ln, text + "   [lane swap reads a VALU result within 2 wait states]", sorted(early)))
            for w in valu:
                w[0] += 1
            mw = VALU_DST.match(text)
            if mw and not text.startswith("v_cmp"):
                valu.append([0, regs_of(mw.group(1))])
            if ms:                                            # (the swap writes both operands)
                valu.append([0, regs_of(ms.group(2)) | regs_of(ms.group(3))])
            valu = [w for w in valu if w[0] < 2]
        if text.startswith("s_waitcnt"):
            continue
        used = regs_of(text)
        if in_asm:
            ld = LOAD.match(text)
            dl = DSLOAD.match(text)
            if ld:
                if re.search(r"\blds\s*$", text) or ld.group(1).startswith("global_load_lds_"):   # LDS-DMA: the first operand is the address, no VGPR is written
                    dst, src = set(), regs_of(text[len(ld.group(1)):])
                else:
                    dst = regs_of(ld.group(2))
                    src = regs_of(text[ld.end():])
                busy = set().union(*[r for _, r in vm_fifo]) if vm_fifo else set()
                hit = (src | dst) & busy
                if hit:
                    findings.append((kernel, ln, text, sorted(hit)))
                vm_fifo.append((ln, dst))
                continue
            if dl:
                dst = regs_of(dl.group(2))
                src = regs_of(text[dl.end():])
                busy = set().union(*[r for _, r in vm_fifo]) if vm_fifo else set()
                if src & busy:
                    findings.append((kernel, ln, text, sorted(src & busy)))
                ds_set.append((ln, dst))
                continue
        if not in_asm and re.match(r"(ds_|s_load_|s_buffer_load_)", text):
            ds_set.append((ln, set()))                         # compiler-managed LGKM operation: a slot in the queue only
        busy = set()
        for _, r in vm_fifo:
            busy |= r
        for _, r in ds_set:
            busy |= r
        pk = _pk_f32_used(text, regs_of)
        hit = (used if pk is None else pk) & busy
        if hit and not _dead_readfirstlane(all_lines, ln - 1, text):
            findings.append((kernel, ln, text, sorted(hit)))
    return findings


def compile_unit(src, outdir):
    out = os.path.join(outdir, os.path.basename(src)[:-4] + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-S", "--cuda-device-only",
                    "-o", out, src], check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def main(argv):
    files = argv
    tmp = None
    if not files:
        tmp = tempfile.mkdtemp(prefix="flute_audit_")
        procs = []
        for u in UNITS:
            out = os.path.join(tmp, u[:-4] + ".s")
            procs.append((out, subprocess.Popen(
                ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-S", "--cuda-device-only"]
                + (["-mllvm", "-amdgpu-kernarg-preload-count=14"] if u.startswith("inst_oneshot") else []) + ["-o", out, u],
                cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
        files = []
        for out, pr in procs:
            _, err = pr.communicate()
            if pr.returncode:
                print(err.decode()[-2000:])
                return 2
            files.append(out)
    total = 0
    for f in files:
        fs = audit(f)
        total += len(fs)
        for (k, ln, text, hit) in fs[:40]:
            print(f"{os.path.basename(f)}:{ln}: {k}: `{text}` touches in-flight v{hit}")
        print(f"{os.path.basename(f)}: {len(fs)} finding(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
