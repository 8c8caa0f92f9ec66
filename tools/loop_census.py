#!/usr/bin/env python3
"""Developer tool: VALU census of EVERY loop of a kernel (isa_cost.py prices only the hottest one).
  tools/loop_census.py file.s 'k422_fusedILb1' [--hist]"""
import collections, json, sys
import isa_cost as I
def mean_cost(asm, pat, min_valu=400):
    """VALU instructions and cycle-weighted cost summed over the big loops of a kernel -> mean cycles
    per instruction (static mix; the loops run comparable trip counts)."""
    blocks = I.parse_kernel(asm, pat)
    succ = I.cfg(blocks)
    nv = cyc = nfull = 0
    for comp in I.sccs(succ):
        if not (len(comp) > 1 or comp[0] in succ[comp[0]]):
            continue
        ops = [op for n in comp for op, _ in blocks[n]]
        v = sum(1 for op in ops if op.startswith("v_"))
        if v >= min_valu:
            nv += v
            cyc += sum(I.cost_of(op) for op in ops)
            nfull += sum(1 for op in ops if op.startswith("v_") and I.cost_of(op) == I.C_FULL)
    mean_cost.nominal = (4.0 * (nv - nfull) + 2.0 * nfull) / nv if nv else 0.0
    return nv, cyc, (cyc / nv if nv else 0.0)


def main():
    asm, pat = sys.argv[1], sys.argv[2]
    if "--mean" in sys.argv:
        nv, cyc, m = mean_cost(asm, pat)
        print(json.dumps({"kernel": pat, "valu_in_loops": nv, "pipe_cycles": cyc, "mean_cycles_per_valu": m,
                          "mean_cycles_per_valu_nominal": mean_cost.nominal}))
        return
    blocks = I.parse_kernel(asm, pat)
    succ = I.cfg(blocks)
    names = list(blocks)
    comps = [c for c in I.sccs(succ) if len(c) > 1 or c[0] in succ[c[0]]]
    for comp in sorted(comps, key=lambda c: min(names.index(n) for n in c)):
        order = [n for n in blocks if n in comp]
        ops = [op for n in order for op, _ in blocks[n]]
        nv = sum(1 for op in ops if op.startswith("v_"))
        if nv < 100:
            continue
        print("%s: %d blocks, VALU %d (fp64 %d), SALU %d, mem %d, VALU cycles %d" % (
            order[0], len(order), nv, sum(1 for op in ops if "_f64" in op),
            sum(1 for op in ops if op.startswith("s_")),
            sum(1 for op in ops if op.startswith(("global_", "buffer_", "ds_", "scratch_", "flat_"))),
            round(sum(I.cost_of(op) for op in ops))))
        if "--hist" in sys.argv:
            print("   ", collections.Counter(I.strip(op) for op in ops).most_common(24))
if __name__ == "__main__":
    main()
