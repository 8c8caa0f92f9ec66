#!/usr/bin/env python3
"""Developer tool (no GPU needed): instruction census of the role kernels' loops, from the built object.
    python tools/pipe_census.py [kernel-substring ...]      (default: the five-role, the S-Video, the default-preset and the 422 role kernels)
Per loop (innermost backward branches with >= 60 VALU instructions): instructions by class.  A lone wavefront gets one
instruction of ANY class through per ~5.3 cycles (profiles/r04_chain_probe.txt), so a role's time per iteration is about
its total count x 5.3 cycles; the steady loops are the long ones (4 positions per iteration; the encoder's: 16)."""
import collections, os, re, subprocess, sys, tempfile
B = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
obj = os.path.join(ROOT, "composite-video-simulator_amd", "csrc", "ntscsim_hip.o")
pats = sys.argv[1:] or ["k_field_pipeIdLb0ELb0E", "k_field_pipeIdLb0ELb1E", "k_field_pipe_tvIdE", "k422_pipeILb1ELi4ELb0E", "k422_short_pipe"]
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "fat"), os.path.join(td, "co")
    subprocess.check_call([B + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([B + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
    dis = subprocess.run([B + "/llvm-objdump", "-d", co, "--symbolize-operands"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().split("\n")
heads = [i for i, l in enumerate(dis) if re.match(r"^[0-9a-f]+ <_Z", l)]
for pat in pats:
    for hi, i in enumerate(heads):
        if pat not in dis[i]:
            continue
        end = heads[hi + 1] if hi + 1 < len(heads) else len(dis)
        lines = dis[i:end]
        name = subprocess.run(["c++filt", re.search(r"<(_Z\w+)>", lines[0]).group(1)], stdout=subprocess.PIPE).stdout.decode().split("(")[0]
        print("==", name, "(%d instructions)" % sum(1 for x in lines if x.startswith("\t")))
        lab = {m.group(1): k for k, l in enumerate(lines) for m in [re.match(r"^[0-9a-f]+ <(L\d+)>:", l)] if m}
        loops = []
        for k, l in enumerate(lines):
            m = re.search(r"s_cbranch\w*\s+(L\d+)|s_branch\s+(L\d+)", l)
            if m:
                t = m.group(1) or m.group(2)
                if t in lab and lab[t] < k:
                    loops.append((lab[t], k))
        # innermost only: drop loops that contain another big loop
        big = []
        for a, b in sorted(loops):
            ops = [x.split()[0] for x in lines[a:b + 1] if x.startswith("\t")]
            if sum(1 for o in ops if o.startswith("v_")) >= 60:
                big.append((a, b, ops))
        for a, b, ops in big:
            if any(a <= a2 and b2 <= b and (a2, b2) != (a, b) for a2, b2, _ in big):
                continue
            c = collections.Counter()
            for o in ops:
                c["v_f64" if o.startswith("v_") and "f64" in o else "v_f32" if o.startswith("v_") and "f32" in o else "v_int" if o.startswith("v_") else
                  "lds" if o.startswith("ds_") else "vmem" if o.split("_")[0] in ("buffer", "global", "flat") else "waitcnt" if o.startswith("s_waitcnt") else "salu"] += 1
            print("   loop at +%-5d %4d instructions: %s" % (a, len(ops), "  ".join("%s %d" % kv for kv in sorted(c.items()))))
        break
