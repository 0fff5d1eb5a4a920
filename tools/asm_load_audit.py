#!/usr/bin/env python3
"""development aid: audit of the hand-waited asm loads of conv_patch_fwd2_kernel (3x3 / 1x1 forms) in the compiler's .s output.

hipcc does not model an asm load: the destination counts as written at the end of the statement, so the register allocator may copy or
read it before the data lands.  For every `global_load_dwordx4 v[a:b]` inside an ASMSTART/ASMEND pair this walks forward to the first asm
`s_waitcnt vmcnt` statement whose tied operands... (the .s does not name them) -- conservatively: to the next asm s_waitcnt that follows at
least one MFMA-free stretch -- and reports any instruction in between that READS v[a:b].  usage: asm_load_audit.py file.s"""
import re
import sys

txt = open(sys.argv[1]).read()
bad = 0
for m in re.finditer(r"^(_ZN\S*conv_patch_fwd2_kernelILi([13])\S*):.*?s_endpgm", txt, re.S | re.M):
    name, body = m.group(1), m.group(0).splitlines()
    in_asm, loads = False, []                      # (line index, lo, hi)
    for i, l in enumerate(body):
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif in_asm:
            g = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", l)
            if g:
                loads.append((i, int(g.group(1)), int(g.group(2))))
    checked = 0
    for i, lo, hi in loads:
        # the wait for this pair is the first asm s_waitcnt AFTER which an MFMA reads the register; until that MFMA nothing may read it
        j = i + 1
        waited = False
        while j < len(body):
            l = body[j]
            if "s_waitcnt vmcnt" in l and "#ASM" in body[j - 1]:
                waited = True
            regs = [(int(a), int(b)) for a, b in re.findall(r"v\[(\d+):(\d+)\]", l)] + [(int(a), int(a)) for a in re.findall(r"\bv(\d+)\b", l)]
            reads = any(not (b < lo or a > hi) for a, b in regs)
            if reads and "global_load_dwordx4 v[%d:%d]" % (lo, hi) not in l:
                if not waited:
                    print("%s: line %d reads v[%d:%d] before any asm wait: %s" % (name[:60], j, lo, hi, l.strip()))
                    bad += 1
                break
            j += 1
        checked += 1
    print("%s: %d asm loads checked" % (name[-40:], checked))
print("PROBLEMS: %d" % bad)
sys.exit(1 if bad else 0)
