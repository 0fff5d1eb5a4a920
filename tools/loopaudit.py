#!/usr/bin/env python3
"""development aid (round 6): instruction census of the loops of one kernel in the compiler's .s -- MFMA / VALU / SALU / LDS / vector-memory counts per loop body and
the s_waitcnt sequence, in order.  Finds loads the compiler sank behind the MFMAs they were meant to run under (the waits then count DOWN inside an iteration),
loads inside wave-uniform branches (conservative vmcnt(0/1) at the loop head) and spills.   usage: hipcc -S --cuda-device-only -o k.s file.hip; loopaudit.py k.s <mangled-name regex>"""
import re,sys
s=open(sys.argv[1]).read()
pat=sys.argv[2]
lines=s.split('\n')
starts=[i for i,l in enumerate(lines) if re.match(r"^_Z\S*"+pat+r"\S*:\s", l)]
for st in starts:
    end=next(j for j in range(st,len(lines)) if 's_endpgm' in lines[j])
    body=lines[st:end]
    print(lines[st].split(':')[0][:90], 'lines', len(body), 'scratch', sum('scratch_' in l for l in body))
    labels={l.split(':')[0]:i for i,l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    for i,l in enumerate(body):
        m=re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg=body[labels[m.group(1)]:i]
            nm=sum('v_mfma' in x for x in seg)
            if len(seg)>40:
                waits=[re.sub(r'\s+',' ',x.strip()) for x in seg if 's_waitcnt' in x]
                print('  loop %s..%d len %d mfma %d valu %d salu %d lds %d vmem %d barrier %d' % (m.group(1), i, len(seg), nm,
                      sum(re.match(r"\s+v_(?!mfma)", x) is not None for x in seg), sum(re.match(r"\s+s_(?!waitcnt|barrier|nop)", x) is not None for x in seg),
                      sum(re.match(r"\s+ds_", x) is not None for x in seg), sum(re.match(r"\s+(buffer_|global_|scratch_)", x) is not None for x in seg), sum('s_barrier' in x for x in seg)))
                print('     waits:', ' | '.join(waits[:40]))
