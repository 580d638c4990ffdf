#!/usr/bin/env python3
"""Basic-block instruction histogram of one kernel of the gfx950 ISA (hipcc -S --cuda-device-only):
usage: isa_blocks.py kernels.s <substring of the mangled kernel name> [min block size]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 25
for nm in re.findall(r'^(_Z\w*):', s, re.M):
    if pat not in nm:
        continue
    a = s.index('\n' + nm + ':')
    b = s.index('.Lfunc_end', a)
    blocks, cur = [], ('entry', [])
    for ln in s[a:b].split('\n'):
        t = ln.strip()
        if re.match(r'^\.LBB\d+_\d+:', t):
            blocks.append(cur); cur = (t.split(':')[0], [])
        elif t and not t.startswith(('.', ';', '//')) and not t.endswith(':'):
            cur[1].append(t.split()[0])
    blocks.append(cur)
    allins = [i for _, ins in blocks for i in ins]
    cnt = lambda ins, f: sum(1 for i in ins if f(i))
    print(nm[:90], 'blocks', len(blocks), 'instr', len(allins), 'valu', cnt(allins, lambda i: i.startswith('v_')),
          'salu', cnt(allins, lambda i: i.startswith('s_')), 'ds', cnt(allins, lambda i: i.startswith('ds_')))
    for lab, ins in blocks:
        if len(ins) >= mn:
            print('  %-12s total %4d valu %4d salu %4d ds %3d (bperm %d) dpp %d readlane %d fma64 %d rcp %d gload %d' % (
                lab, len(ins), cnt(ins, lambda i: i.startswith('v_')), cnt(ins, lambda i: i.startswith('s_')),
                cnt(ins, lambda i: i.startswith('ds_')), cnt(ins, lambda i: 'bpermute' in i), cnt(ins, lambda i: 'dpp' in i),
                cnt(ins, lambda i: 'readlane' in i), cnt(ins, lambda i: i.startswith(('v_fma_f64', 'v_fmac_f64')) ),
                cnt(ins, lambda i: i.startswith('v_rcp')), cnt(ins, lambda i: i.startswith(('global_load', 's_load')))))
