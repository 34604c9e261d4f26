"""The decode kernels (streaming ring kernel, one-shot kernel) hide their loads from hipcc (inline asm) and releases them with counted waits;
nothing may touch a destination register in between (tools/audit_asm_loads.py explains the failure this guards
against).  Compiles the kernel instantiation units to assembly (gfx950 cross-compile, no GPU) and lints them."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_instruction_touches_an_in_flight_hidden_load():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm_loads.py")], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("0 finding(s)") == 18, r.stdout[-2000:]


def _audit_text(tmp_path, body):
    import importlib.util
    spec = importlib.util.spec_from_file_location("audit_asm_loads", os.path.join(ROOT, "tools", "audit_asm_loads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    f = tmp_path / "k.s"
    f.write_text("_Z6kernelv:\n" + body + "\ts_endpgm\n")
    return mod.audit(str(f))


def test_audit_rules_on_synthetic_streams(tmp_path):
    """The lint itself: what it must flag and what it must not."""
    load = "\t;;#ASMSTART\n\tbuffer_load_dwordx4 v[4:7], v1, s[0:3], s8 offen\n\t;;#ASMEND\n"
    wait0 = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n"
    # a copy of the in-flight destination before the wait (the round-2 failure) is flagged ...
    assert len(_audit_text(tmp_path, load + "\tv_mov_b32_e32 v20, v4\n" + wait0)) == 1
    # ... the same copy after the wait is not
    assert _audit_text(tmp_path, load + wait0 + "\tv_mov_b32_e32 v20, v4\n") == []
    # counted waits release the OLDEST loads only
    two = load + "\t;;#ASMSTART\n\tbuffer_load_dwordx4 v[8:11], v1, s[0:3], s8 offen\n\t;;#ASMEND\n"
    wait1 = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(1)\n\t;;#ASMEND\n"
    assert _audit_text(tmp_path, two + wait1 + "\tv_add_u32_e32 v20, v4, v5\n") == []
    assert len(_audit_text(tmp_path, two + wait1 + "\tv_add_u32_e32 v20, v8, v5\n")) == 1
    # LDS-DMA writes no VGPR: its first operand is the address and may be reused at once
    dma = "\t;;#ASMSTART\n\tbuffer_load_dwordx4 v3, s[4:7], s9 offen lds\n\t;;#ASMEND\n\tv_mov_b32_e32 v3, 0\n"
    assert _audit_text(tmp_path, dma) == []
    gdma = "\t;;#ASMSTART\n\tglobal_load_lds_dword v[3:4], off\n\t;;#ASMEND\n\tv_mov_b32_e32 v3, 0\n"
    assert _audit_text(tmp_path, gdma) == []
    # hidden LDS reads: in-order return, counted lgkmcnt
    lds = ("\t;;#ASMSTART\n\tds_read_b32 v30, v2\n\t;;#ASMEND\n\t;;#ASMSTART\n\tds_read_b32 v31, v2\n\t;;#ASMEND\n")
    lg1 = "\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(1)\n\t;;#ASMEND\n"
    assert _audit_text(tmp_path, lds + lg1 + "\tv_mov_b32_e32 v40, v30\n") == []
    assert len(_audit_text(tmp_path, lds + lg1 + "\tv_mov_b32_e32 v40, v31\n")) == 1
    # an MFMA reading a fragment whose hidden read is still outstanding (the qgemm_block2.h bug) is flagged
    frag = "\t;;#ASMSTART\n\tds_read_b128 v[30:33], v2 offset:1024\n\t;;#ASMEND\n"
    assert len(_audit_text(tmp_path, frag + "\tv_mfma_f32_16x16x32_f16 v[60:63], v[50:53], v[30:33], v[60:63]\n")) == 1
    # hipcc's undef placeholder: a readfirstlane of any register whose scalar result is overwritten before use
    undef = load + "\tv_readfirstlane_b32 s10, v4\n\ts_add_i32 s10, s46, 64\n" + wait0
    assert _audit_text(tmp_path, undef) == []
    live = load + "\tv_readfirstlane_b32 s10, v4\n\ts_add_i32 s11, s10, 64\n" + wait0
    assert len(_audit_text(tmp_path, live)) == 1
    # packed fp32 with op_sel_hi:[0,..]: both halves of source 0 come from its LOW register - the high one is not read
    bcast = load + "\tv_pk_mul_f32 v[36:37], v[3:4], v[36:37] op_sel_hi:[0,1]\n" + wait0
    assert _audit_text(tmp_path, bcast) == []
    assert len(_audit_text(tmp_path, load + "\tv_pk_mul_f32 v[36:37], v[3:4], v[36:37]\n" + wait0)) == 1
    assert len(_audit_text(tmp_path, load + "\tv_pk_mul_f32 v[36:37], v[4:5], v[36:37] op_sel_hi:[0,1]\n" + wait0)) == 1
    # ... but only as THAT source: the same register named as destination or as another source is still touched
    assert len(_audit_text(tmp_path, load + "\tv_pk_mul_f32 v[4:5], v[3:4], v[36:37] op_sel_hi:[0,1]\n" + wait0)) == 1
    assert len(_audit_text(tmp_path, load + "\tv_pk_mul_f32 v[36:37], v[3:4], v[4:5] op_sel_hi:[0,1]\n" + wait0)) == 1
    # round 5: a v_dot* result read by ANOTHER VALU instruction (or by a v_dot* as its A / B operand) less than three wait states later
    # (gfx90a+; hipcc keeps it for its own instructions, not inside asm statements) is flagged ...
    dot = "\tv_dot2_f32_bf16 v10, v2, v3, 0\n"
    assert len(_audit_text(tmp_path, dot + "\tv_cvt_pk_bf16_f32 v20, v10, v11\n")) == 1
    assert len(_audit_text(tmp_path, dot + "\tv_mov_b32_e32 v30, v1\n\tv_mov_b32_e32 v31, v1\n\tv_cvt_pk_bf16_f32 v20, v10, v11\n")) == 1
    assert len(_audit_text(tmp_path, dot + "\tv_dot2_f32_bf16 v12, v10, v3, 0\n")) == 1
    # ... three instructions (or an s_nop 2) later it is not, nor as the NEXT dot's accumulator
    assert _audit_text(tmp_path, dot + "\tv_mov_b32_e32 v30, v1\n" * 3 + "\tv_cvt_pk_bf16_f32 v20, v10, v11\n") == []
    assert _audit_text(tmp_path, dot + "\ts_nop 2\n\tv_cvt_pk_bf16_f32 v20, v10, v11\n") == []
    assert _audit_text(tmp_path, dot + "\tv_dot2_f32_bf16 v10, v4, v5, v10\n") == []
    # the multiply stage of common.h's Num<BF16>::mul_scale4: eight products, then the conversions in the same order
    blk = "".join(f"\tv_dot2_f32_bf16 v{10 + i}, v{2 + i // 2}, v{6 + i % 2}, 0\n" for i in range(8)) + \
          "".join(f"\tv_cvt_pk_bf16_f32 v{20 + i}, v{10 + 2 * i}, v{11 + 2 * i}\n" for i in range(4))
    assert _audit_text(tmp_path, blk) == []
    # round 6: a lane swap that reads a VGPR written by a VALU instruction less than two wait states earlier (hipcc pads its own
    # swaps, not those inside asm statements) is flagged; with s_nop 1 (or two instructions) in between it is not
    wr = "\tv_add_f32_e32 v10, v2, v3\n"
    assert len(_audit_text(tmp_path, wr + "\tv_permlane16_swap_b32 v10, v11\n")) == 1
    assert len(_audit_text(tmp_path, wr + "\tv_mov_b32_e32 v30, v1\n\tv_permlane32_swap_b32 v11, v10\n")) == 1
    assert _audit_text(tmp_path, wr + "\ts_nop 1\n\tv_permlane16_swap_b32 v10, v11\n") == []
    assert _audit_text(tmp_path, wr + "\tv_mov_b32_e32 v30, v1\n\tv_mov_b32_e32 v31, v1\n\tv_permlane16_swap_b32 v10, v11\n") == []
    assert len(_audit_text(tmp_path, wr + "\ts_nop 1\n\tv_permlane16_swap_b32 v10, v11\n\tv_permlane16_swap_b32 v11, v12\n")) == 1   # back-to-back swaps sharing a register
    assert _audit_text(tmp_path, wr + "\ts_nop 1\n\tv_permlane16_swap_b32 v10, v11\n\tv_permlane16_swap_b32 v12, v13\n") == []

