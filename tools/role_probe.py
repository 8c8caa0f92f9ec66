#!/usr/bin/env python3
"""tools/role_probe.py -- VERDICT r05 item 4: what could a stage-pipelined workgroup (encoder wave -> VCR-half wave ->
TV-half wave on the same rows) buy the SYNCHRONOUS one-field call ntscsim_field()?  GPU box only.

Measures the call (720x486 -vhs, pageable host frames, one field per call) in three arrangements:
  one_launch   the shipped chain: setup | encoder | decoder (VCR half + TV half in one kernel)
  two_launch   the decoder as its two halves, one after the other (ntscsim_debug_no_fast_decode bit 1)
  side_by_side NTSCSIM_ROLE_PROBE=1: encoder, VCR half and TV half launched on three streams with NO dependency between
               them (wrong pixels, right instructions): each role is a lone wavefront per 63 rows and runs at the issue
               rate of a lone wavefront, so the call takes max(role) instead of sum(roles) -- the CEILING of any
               three-role pipeline, before it pays a cycle for its hand-offs.
  pipe         the three roles as three wavefronts of ONE workgroup with the hand-offs in place (k_field_pipe,
               csrc/ntsc_pipe.hip; the other arrangements run with NTSCSIM_PIPE=0) -- right pixels
and the float pipeline (NTSCSIM_MODE_FLOAT) in the shipped arrangement.  Prints one line each and a JSON summary."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import sys, time, json, os
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import _libs as L
import ntscsim
from ntscsim import _capi
arr, mode = sys.argv[1], sys.argv[2]
w, h = 720, 486
p = L.make_params(["-vhs"])
sim = ntscsim.FieldSimulator(params=p)
if mode == "float":
    sim.set_mode(_capi.MODE_FLOAT)
if arr not in ("one_launch", "pipe"):
    sim.debug_no_fast_decode(2)
src = L.bars(w, h, 0)
dst = np.zeros((h, w, 4), np.uint8)
for k in range(50):
    sim.field_host(dst, src, (k & 1) ^ 1, k)
n = 600
t0 = time.perf_counter()
for k in range(n):
    sim.field_host(dst, src, (k & 1) ^ 1, k)
dt = time.perf_counter() - t0
print(json.dumps({"arrangement": arr, "mode": mode, "fields_per_s": n / dt, "us_per_call": dt / n * 1e6,
                  "kernels": sim.last_kernels()}))
sim.close()
''' % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "composite-video-simulator_amd"))


def run(arr, mode="exact"):
    env = dict(os.environ)
    if arr == "side_by_side":
        env["NTSCSIM_ROLE_PROBE"] = "1"
    if arr != "pipe":
        env["NTSCSIM_PIPE"] = "0"
    r = subprocess.run([sys.executable, "-c", WORKER, arr, mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    if r.returncode != 0:
        return {"arrangement": arr, "mode": mode, "error": r.stderr.decode()[-400:]}
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


def main():
    out = [run("one_launch"), run("two_launch"), run("side_by_side"), run("pipe"), run("one_launch", "float"), run("pipe", "float")]
    for o in out:
        if "error" in o:
            print(o)
        else:
            print("%-13s %-6s %8.1f fields/s  %7.1f us per call   %s" % (o["arrangement"], o["mode"], o["fields_per_s"],
                                                                        o["us_per_call"], ",".join(o["kernels"])))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
