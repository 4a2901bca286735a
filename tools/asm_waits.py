#!/usr/bin/env python3
"""Per kernel of a .s file (hipcc -S --cuda-device-only): global/flat loads that are waited for within a few
instructions of their issue (a latency chain: each such load is its own memory round trip)."""
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
near = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kern = None
stats = collections.OrderedDict()
lines = []
for l in src:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        kern = m.group(1); lines = []; stats[kern] = lines
    elif kern is not None:
        t = l.strip()
        if t and not t.startswith(';') and not t.startswith('.'):
            lines.append(t)
        if 's_endpgm' in l:
            kern = None
for k, L in stats.items():
    loads = [i for i, t in enumerate(L) if re.match(r'(global|flat)_load', t)]
    if not loads:
        continue
    tight = 0
    for i in loads:
        for j in range(i + 1, min(i + 1 + near, len(L))):
            if L[j].startswith('s_waitcnt') and 'vmcnt(0)' in L[j]:
                tight += 1
                break
            if re.match(r'(global|flat)_load', L[j]):
                break
    name = re.sub(r'^_ZN2jb\d+', '', k)[:44]
    print(f"{name:46s} instrs {len(L):5d} loads {len(loads):4d} waited-at-once {tight:4d}")
