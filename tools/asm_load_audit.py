#!/usr/bin/env python3
"""Audit of the hand-waited asm loads of conv_patch_fwd2_kernel (3x3 / 1x1 forms) in the compiler's .s output.

hipcc does not model an `asm volatile` load: its destination counts as written at the end of the statement, so the register allocator may
copy or read it before the data lands, and the wait that covers it is ours to count.  This walks each kernel's instruction stream in text
order (the tap code of a slice is straight-line; every slice ends with s_waitcnt vmcnt(0), so nothing is outstanding across the loop's back
edge) and keeps the queue of outstanding vector-memory operations exactly as the hardware does -- loads, LDS-DMA and stores retire in issue
order, `s_waitcnt vmcnt(N)` leaves at most the N youngest outstanding -- and reports every instruction that READS the destination of an asm
load still in the queue.  usage: asm_load_audit.py file.s   (exit code 1 on a finding; tests/test_asm_audit.py runs it on a fresh compile)"""
import re
import sys

VMEM = re.compile(r"^\s*(global_load|global_store|buffer_load|buffer_store|global_atomic|buffer_atomic|scratch_load|scratch_store|flat_load|flat_store)")
WAIT = re.compile(r"s_waitcnt.*vmcnt\((\d+)\)")
REG = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = []
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.append((int(m.group(1)), int(m.group(2))))
        else:
            out.append((int(m.group(3)), int(m.group(3))))
    return out


def audit(txt, pattern=r"conv_patch_fwd2_kernelILi[13]"):
    findings, checked = [], 0
    for m in re.finditer(r"^(_ZN\S*" + pattern + r"\S*):.*?s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(0).splitlines()
        queue = []                                     # outstanding vmem operations, oldest first: (lo, hi) of an asm load's destination or None
        in_asm = False
        for ln, l in enumerate(body):
            code = l.split(";")[0]
            if "#ASMSTART" in l:
                in_asm = True
                continue
            if "#ASMEND" in l:
                in_asm = False
                continue
            w = WAIT.search(code)
            if w:
                n = int(w.group(1))
                while len(queue) > n:
                    queue.pop(0)
                continue
            if "s_waitcnt" in code and "vmcnt" not in code:
                continue
            if VMEM.match(code):
                dst = None
                g = re.match(r"\s*global_load_dwordx4 v\[(\d+):(\d+)\]", code)
                if in_asm and g:
                    dst = (int(g.group(1)), int(g.group(2)))
                    checked += 1
                # an instruction's own address / data registers are read at issue: check them against the pending asm destinations
                srcs = regs_of(code.split(",", 1)[1]) if "," in code else []
                if dst is None and not in_asm and re.match(r"\s*(global_load|buffer_load|scratch_load|flat_load)", code) and "lds" not in code:
                    srcs = regs_of(code.split(",", 1)[1]) if "," in code else []
                for lo, hi in srcs:
                    for q in queue:
                        if q is not None and not (hi < q[0] or lo > q[1]):
                            findings.append("%s: line %d reads v[%d:%d] (asm load still outstanding): %s" % (name[-44:], ln, q[0], q[1], code.strip()))
                queue.append(dst)
                continue
            if not code.strip() or code.strip().startswith(".") or code.strip().endswith(":"):
                continue
            # any other instruction: every VGPR it names is read or written -- neither is allowed while the asm load that owns it is in flight
            for lo, hi in regs_of(code):
                for q in queue:
                    if q is not None and not (hi < q[0] or lo > q[1]):
                        findings.append("%s: line %d touches v[%d:%d] before its wait: %s" % (name[-44:], ln, q[0], q[1], code.strip()))
    return checked, findings


if __name__ == "__main__":
    checked, findings = audit(open(sys.argv[1]).read())
    for f in findings[:40]:
        print(f)
    print("asm loads checked: %d, findings: %d" % (checked, len(findings)))
    sys.exit(1 if findings or not checked else 0)
