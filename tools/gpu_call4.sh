#!/bin/bash
O=gpurun_out/c4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python - <<'PY'
import sys, os, json
sys.path.insert(0, "composite-video-simulator_amd"); sys.path.insert(0, ".")
import torch, ntscsim, bench
dev = torch.device("cuda", 0)
for name, fl, w, h, nfr in (("ntsc-vhs 720x486", ["-vhs"], 720, 486, 300), ("pal-vhs 720x576", ["-tvstd", "pal", "-vhs"], 720, 576, 253),
                            ("ntsc-vhs hs-phase .002 (wrap)", ["-vhs", "-vhs-head-switching-phase", "0.002"], 720, 486, 300)):
    kn = []
    v = bench.device_rate(torch, ntscsim, dev, 0, fl, w, h, nfr, 12, 4, kernels=kn)
    print("%-34s %8.0f fields/s  %s" % (name, v, [k for k in kn if "decode" in k]))
PY
