"""CPU (-m "not gpu"): the hand-waited asm loads of the LDS-patch forward kernel, checked in the compiler's own output.

conv_patch_fwd2_kernel (3x3 / 1x1 forms) hides its weight-fragment loads from hipcc in `asm volatile` statements and waits for them with
hand-counted `s_waitcnt vmcnt(N)` -- beside LDS-DMA the compiler's own bookkeeping is wrong either way (csrc/conv_patch.hip).  The price:
nothing but the register operands of the wait statements keeps the compiler from touching a destination before its data has landed, and a
new compiler release or an innocent edit can break that silently (round 3 saw it once: a `v_mov` of the four destination registers in front
of one arm of a two-armed wait, garbage on every launch of the 1x1 kernel).  tools/asm_load_audit.py replays the hardware's in-order vmcnt
queue over the generated .s and reports any access to an asm load's destination while the load is outstanding; this test runs it on a
fresh device-only compile of the shipped source, after checking the auditor on two hand-written streams."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _auditor():
    spec = importlib.util.spec_from_file_location("asm_load_audit", os.path.join(ROOT, "tools", "asm_load_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


GOOD = """_ZN1x22conv_patch_fwd2_kernelILi1ELi1ELb0EEEv:
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v2, s[4:5]
\t;;#ASMEND
\t;;#ASMSTART
\tglobal_load_dwordx4 v[14:17], v2, s[6:7]
\t;;#ASMEND
\tbuffer_load_dwordx4 v3, s[16:19], s20 offen lds
\tds_read_b128 v[20:23], v5
\t;;#ASMSTART
\ts_waitcnt vmcnt(2)
\t;;#ASMEND
\tv_mfma_f32_32x32x16_bf16 v[30:45], v[10:13], v[20:23], v[30:45]
\t;;#ASMSTART
\ts_waitcnt vmcnt(1)
\t;;#ASMEND
\tv_mfma_f32_32x32x16_bf16 v[30:45], v[14:17], v[20:23], v[30:45]
\ts_endpgm
"""
# the round-3 failure: the destination copied in front of the wait
BAD = GOOD.replace("\tds_read_b128 v[20:23], v5\n", "\tds_read_b128 v[20:23], v5\n\tv_mov_b64_e32 v[50:51], v[10:11]\n")
# a wait that is one operation too lax for the second fragment (the DMA behind it is still counted as outstanding)
LAX = GOOD.replace("s_waitcnt vmcnt(1)", "s_waitcnt vmcnt(2)")


def test_auditor_on_hand_written_streams():
    a = _auditor()
    assert a.audit(GOOD) == (2, [])
    n, f = a.audit(BAD)
    assert n == 2 and len(f) == 1 and "v[10:13]" in f[0]
    n, f = a.audit(LAX)
    assert n == 2 and len(f) == 1 and "v[14:17]" in f[0]


def test_shipped_patch_kernel_passes_the_audit(tmp_path):
    sys.path.insert(0, ROOT)
    from mindtheedge_amd import _build
    out = tmp_path / "conv_patch.s"
    cmd = [_build._hipcc()] + _build.FLAGS + ["-S", "--cuda-device-only", os.path.join(_build.CSRC, "conv_patch.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True)
    checked, findings = _auditor().audit(out.read_text())
    assert checked >= 72, "the 3x3 kernels alone hold 4 x 18 asm loads: the auditor no longer finds them (%d)" % checked
    assert not findings, "\n".join(findings[:10])


def test_nine_tap_wgrad_kernel_keeps_its_hand_waited_lds_reads_in_order(tmp_path):
    """round 5: conv_wgrad9.hip reads its MFMA operands with inline-asm ds_read_b64_tr_b16 (26 per K-step, issued in the load phase and retired
    by a hand-placed lgkmcnt wait in front of the barrier that opens the MFMA phase).  tools/w9_audit.py walks the compiler's .s along the main loop's control flow: no instruction may touch a
    destination while its read is outstanding, and the loop must hold the 26 reads / 36 MFMAs per K-step the source is written for."""
    sys.path.insert(0, ROOT)
    from mindtheedge_amd import _build
    spec = importlib.util.spec_from_file_location("w9_audit", os.path.join(ROOT, "tools", "w9_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / "conv_wgrad9.s"
    cmd = [_build._hipcc()] + _build.FLAGS + ["-S", "--cuda-device-only", os.path.join(_build.CSRC, "conv_wgrad9.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True)
    assert mod.audit(str(out)) == 0
    text = out.read_text()
    # two instantiations, main loop unrolled over the four ring slots: 4 x (26 reads, 36 MFMAs)
    assert text.count("ds_read_b64_tr_b16") == 2 * 4 * 26 and text.count("v_mfma_f32_16x16x32_bf16") == 2 * 4 * 36
    # (a handful of loop-invariant values -- epilogue pointers, the loader's lane offsets, used in its rare end-of-column block -- may live in scratch;
    #  nothing of the K-step path does: the loop body proper holds no scratch access)
    import re
    assert all(int(m) <= 16 for m in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text))
    # a stream the auditor must reject: an MFMA that reads a fragment in front of its wait
    bad = tmp_path / "bad.s"
    bad.write_text("_ZN1x18conv_wgrad9_kernelILi1EEEv: ; x\n\tds_read_b64_tr_b16 v[10:11], v2\n\tds_read_b64_tr_b16 v[12:13], v2 offset:1024\n"
                   "\ts_waitcnt lgkmcnt(1)\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[10:13], v[20:23], v[0:3]\n\ts_endpgm\n")
    assert mod.audit(str(bad)) == 1
