#!/usr/bin/env python3
"""Developer tool: cycle-weighted instruction census of a kernel's hottest loop, from the gfx950
assembly hipcc emits (-S --cuda-device-only).

  tools/isa_cost.py ntscsim.s 'k_decodeILb1ELb1ELj6Ed' [--skip LBB20_310 ...] [--json out.json]

The loop = the strongly connected component of the kernel's basic-block graph with the most
floating-point instructions.  Every VALU instruction is priced with the issue cost measured on MI355X by
tools/valu_rate_probe.hip (slowest-wave figure at 3 waves per SIMD, profiles/r02_valu_rates.txt):
  "full rate"   2.7 cycles per wave64 instruction per SIMD: v_add/sub_u32, and/or/xor, ashr/lshr,
                v_mov, v_add_f32/fma_f32
  "half rate"   4.3 cycles: every fp64 add/mul/fma/trunc/ldexp/convert, and most other integer
                opcodes (lshl, min/max/med3, add3, lshl_add, lshl_or, bfe, perm, mul_lo/hi,
                mad, cndmask, DPP moves, v_cmp)
The sum is the VALU pipe time one wave's pass through the loop costs its SIMD.
"""
import argparse
import collections
import json
import re
import sys

FULL_RATE = {
    "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
    "v_ashrrev_i32", "v_lshrrev_b32", "v_mov_b32", "v_add_f32", "v_sub_f32", "v_fma_f32",
    "v_mul_f32", "v_not_b32", "v_fmac_f32", "v_accvgpr_read_b32", "v_accvgpr_write_b32",
}
C_FULL, C_HALF = 2.7, 4.3


def strip(op):
    return re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)


def cost_of(op):
    op = strip(op)
    if not op.startswith("v_"):
        return 0.0
    if op in ("v_mov_b64", "v_lshl_add_u64", "v_mad_u64_u32", "v_mad_i64_i32"):
        return 2 * C_HALF if op.startswith("v_mad") else C_HALF
    return C_FULL if op in FULL_RATE else C_HALF


def parse_kernel(path, pattern):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*:", l) and re.search(pattern, l):
            start = i
            break
    if start is None:
        raise SystemExit("kernel matching %r not found" % pattern)
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end") or ".amdhsa_kernel" in l:
            break
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\b(.*)", l)
        if m and not l.strip().startswith(";"):
            blocks[cur].append((m.group(1), m.group(2)))
    return blocks


def cfg(blocks):
    names = list(blocks)
    succ = {n: set() for n in names}
    for i, n in enumerate(names):
        fall = True
        for op, rest in blocks[n]:
            if op.startswith("s_cbranch") or op == "s_branch":
                t = rest.strip().split()[0] if rest.strip() else None
                if t in succ:
                    succ[n].add(t)
                if op == "s_branch":
                    fall = False
            if op in ("s_endpgm",):
                fall = False
        if fall and i + 1 < len(names):
            succ[n].add(names[i + 1])
    return succ


def sccs(succ):
    index, low, on, stack, out = {}, {}, set(), [], []
    sys.setrecursionlimit(100000)
    counter = [0]

    def visit(v):
        index[v] = low[v] = counter[0]
        counter[0] += 1
        stack.append(v)
        on.add(v)
        for w in succ[v]:
            if w not in index:
                visit(w)
                low[v] = min(low[v], low[w])
            elif w in on:
                low[v] = min(low[v], index[w])
        if low[v] == index[v]:
            comp = []
            while True:
                w = stack.pop()
                on.discard(w)
                comp.append(w)
                if w == v:
                    break
            out.append(comp)
    for v in succ:
        if v not in index:
            visit(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--skip", nargs="*", default=[], help="blocks of the loop to leave out (alternative paths)")
    ap.add_argument("--steps", type=float, default=4.0, help="pipeline steps one pass of the loop makes")
    ap.add_argument("--json")
    ap.add_argument("--blocks", action="store_true", help="print the per-block sizes of the loop")
    a = ap.parse_args()
    blocks = parse_kernel(a.asm, a.kernel)
    succ = cfg(blocks)
    loop, best = None, -1
    for comp in sccs(succ):
        if len(comp) > 1 or comp[0] in succ[comp[0]]:
            w = sum(1 for n in comp for op, _ in blocks[n] if "_f64" in op or "_f32" in op)
            if w > best:
                loop, best = comp, w
    if loop is None:
        raise SystemExit("no loop found")
    order = [n for n in blocks if n in loop and n not in a.skip]
    hist = collections.Counter()
    for n in order:
        for op, _ in blocks[n]:
            hist[strip(op)] += 1
    if a.blocks:
        for n in order:
            print("  %-14s %4d instructions" % (n, len(blocks[n])))
    valu = {k: v for k, v in hist.items() if k.startswith("v_")}
    n_valu = sum(valu.values())
    n_f64 = sum(v for k, v in valu.items() if "f64" in k)
    n_full = sum(v for k, v in valu.items() if cost_of(k) == C_FULL)
    cyc = sum(cost_of(k) * v for k, v in valu.items())
    other = {k: v for k, v in hist.items() if not k.startswith("v_")}
    res = {
        "kernel": a.kernel, "loop_blocks": order, "steps_per_pass": a.steps,
        "valu_per_step": n_valu / a.steps, "fp64_per_step": n_f64 / a.steps,
        "full_rate_per_step": n_full / a.steps, "half_rate_int_per_step": (n_valu - n_f64 - n_full) / a.steps,
        "valu_pipe_cycles_per_step": cyc / a.steps,
        "mean_cycles_per_valu": cyc / max(1, n_valu),
        # the same mix at the pipe's nominal issue costs (4 cycles half-rate, 2 full-rate: MI355X_MICROARCH.md)
        "mean_cycles_per_valu_nominal": (4.0 * (n_valu - n_full) + 2.0 * n_full) / max(1, n_valu),
        "salu_per_step": sum(v for k, v in other.items() if k.startswith("s_") and not k.startswith("s_waitcnt") and k != "s_nop") / a.steps,
        "lds_per_step": sum(v for k, v in other.items() if k.startswith("ds_")) / a.steps,
        "vmem_per_step": sum(v for k, v in other.items() if k.startswith(("global_", "scratch_", "buffer_"))) / a.steps,
        "waitcnt_per_step": other.get("s_waitcnt", 0) / a.steps,
        "histogram": dict(hist.most_common()),
    }
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)
    print("loop of %s: %d blocks, per pipeline step: %.1f VALU (%.1f fp64, %.1f other half-rate, %.1f full-rate) "
          "= %.0f VALU pipe cycles (%.2f per instruction); %.1f SALU, %.1f LDS, %.1f VMEM, %.1f s_waitcnt" % (
              a.kernel, len(order), res["valu_per_step"], res["fp64_per_step"], res["half_rate_int_per_step"],
              res["full_rate_per_step"], res["valu_pipe_cycles_per_step"], res["mean_cycles_per_valu"],
              res["salu_per_step"], res["lds_per_step"], res["vmem_per_step"], res["waitcnt_per_step"]))
    for k, v in hist.most_common(60):
        print("   %5d  %s" % (v, k))


if __name__ == "__main__":
    main()
