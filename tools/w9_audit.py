#!/usr/bin/env python3
"""development aid (round 5): audit of conv_wgrad9.hip's hand-waited transposing LDS reads in the compiler's .s.
ds_read_b64_tr_b16 is issued through inline asm, so hipcc neither waits for it nor knows when its destination is valid; LDS returns in order, the
source counts them by hand (s_waitcnt lgkmcnt(N)).  This replays the main loop(s) of each kernel: every ds_read destination is 'pending' until an
lgkmcnt(N) statement leaves at most N younger reads outstanding; any other instruction that touches a pending register (a compiler-inserted copy,
an early MFMA) is reported.  usage: w9_audit.py file.s   (exit code 1 on a finding)"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]|v(\d+)", tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def audit(path, budget=4000):
    text = open(path).read()
    bad = 0
    for km in re.finditer(r"^(_ZN\S*conv_wgrad9_kernel\S*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = km.group(1), km.group(2)
        lines = [l.split(";")[0].strip() for l in body.split("\n")]
        lines = [l for l in lines if l]
        label = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
        # walk the control flow as the kernel's main loop runs it: unconditional and BACKWARD conditional branches taken, forward conditional
        # ones not (the loop is laid out rotated: its second half sits in front of its header), until the instruction budget is spent
        pending = []                       # destinations of outstanding reads, in issue order
        nread = nmfma = 0
        pc = steps = 0
        while pc < len(lines) and steps < budget:
            l = lines[pc]
            steps += 1
            pc += 1
            if l.endswith(":"):
                continue
            m = re.match(r"s_(c?)branch\S*\s+(\S+)", l)
            if m:
                tgt = label.get(m.group(2))
                if tgt is not None and (not m.group(1) or tgt < pc):
                    pc = tgt
                continue
            if l.startswith("ds_read_b64_tr_b16"):
                dst = regs(l.split(",")[0])
                src = regs(",".join(l.split(",")[1:]))
                for p in pending:
                    if p & src:
                        print("%s: address register of `%s` is a pending destination" % (name, l)); bad += 1
                pending.append(dst)
                nread += 1
                continue
            m = re.match(r"s_waitcnt .*lgkmcnt\((\d+)\)", l)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n else []
                continue
            if l.startswith("s_") or l.startswith("."):
                continue
            touched = regs(l)
            for p in pending:
                if p & touched:
                    print("%s: `%s` touches v%s while its transposing read is outstanding" % (name, l, sorted(p & touched))); bad += 1
            if l.startswith("v_mfma"):
                nmfma += 1
        print("%s: walked %d instructions, %d transposing reads, %d MFMAs, %d finding(s) so far" % (name, steps, nread, nmfma, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if audit(sys.argv[1]) else 0)
