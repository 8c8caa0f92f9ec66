#!/usr/bin/env python3
"""Developer tool: how serial is a kernel's hottest basic block?  For every VALU instruction of the
largest block, the distance (in issued instructions) back to the producer of its newest source
register.  A wave issues a dependent instruction ~7.4 cycles after its producer but an independent
one after ~5 (tools/chain_probe.hip), so distance-1 pairs are stalls when few waves share a SIMD."""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
blocks, cur = {}, None
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        cur = m.group(1); blocks[cur] = []
    elif cur and re.match(r"^\s+[a-z]", l) and not l.strip().startswith(";"):
        blocks[cur].append(l.strip())
big = max(blocks, key=lambda k: len(blocks[k]))
def regs(tok):
    out = []
    for m in re.finditer(r"\b([vs])\[(\d+):(\d+)\]|\b([vs])(\d+)\b", tok):
        if m.group(1):
            out += ["%s%d" % (m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)]
        else:
            out.append("%s%s" % (m.group(4), m.group(5)))
    return out
last = {}
hist = collections.Counter()
est = 0.0
n = 0
for idx, l in enumerate(blocks[big]):
    op, _, rest = l.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    if not op.startswith("v_"):
        if op.startswith(("ds_read", "buffer_load", "global_load", "flat_load", "scratch_load")) and ops:
            for r in regs(ops[0]): last[r] = idx
        continue
    nd = 2 if ("cmp" in op and not ops[0].startswith("v")) else 1
    dst, src = ops[:1], ops[1:]
    d = min([idx - last[r] for s in src for r in regs(s) if r in last] or [99])
    hist[min(d, 8)] += 1
    est += 7.4 if d == 1 else (6.0 if d == 2 else 5.0)
    n += 1
    for r in regs(dst[0]): last[r] = idx
print("block %s: %d VALU instructions; distance to newest producer:" % (big, n))
for k in sorted(hist): print("   %s%d: %d (%.0f%%)" % (">=" if k == 8 else "", k, hist[k], 100.0 * hist[k] / n))
print("lone-wave issue estimate: %.0f cycles per pass (%.2f per instruction)" % (est, est / n))
