#!/usr/bin/env python3
"""Developer tool: list the vector-memory instructions, the vmcnt waits and the branches of one kernel in a
hipcc -S listing, numbered by instruction, to see where a loop waits for what.
    python tools/vm_waits.py listing.s k_decode_fastILb1EdLb0EE [first last]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 1 << 30)
st = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l][0]
n = 0
for i in range(st + 1, len(lines)):
    l = lines[i].strip()
    if l.startswith('s_endpgm'): break
    if not l or l.startswith(';') or (l.startswith('.') and not l.startswith('.LBB')): continue
    n += 1
    if lo <= n <= hi and re.match(r'(s_waitcnt vmcnt|s_waitcnt lgkmcnt\(\d+\) *$|buffer_|global_|scratch_|\.LBB|s_branch|s_cbranch)', l) and 'lgkmcnt' not in l:
        print(n, l.split(';')[0].rstrip())
