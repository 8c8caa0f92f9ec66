"""Developer probe: accuracy of MODE_FAST32 against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import _libs as L, ntscsim, torch
def run(flags, w, h, n, kind):
    p = L.make_params(flags)
    srcs = [L.noise_frame(w, h, 5 + j) if kind == "noise" else L.bars(w, h, j) for j in range((n + 1) // 2)]
    o = L.OracleStream(p)
    exp = np.zeros((n, h, w, 4), np.uint8)
    for k in range(n): o.field(exp[k], srcs[k // 2], (k & 1) ^ 1, k)
    sim = ntscsim.FieldSimulator(params=p); sim.set_mode(ntscsim._capi.MODE_FAST32)
    src = torch.from_numpy(np.stack(srcs)).cuda()
    dst = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
    sim.fields(src, dst, [(k // 2, k, (k & 1) ^ 1, k) for k in range(n)]); sim.sync()
    got = dst.cpu().numpy().astype(int); e = exp.astype(int)
    d = np.abs(got - e)
    rows = np.zeros_like(d, bool)
    for k in range(n): rows[k, ((k & 1) ^ 1)::2] = True
    d = d[rows].reshape(-1, 4)[:, :3]
    print(flags, w, h, kind, "max", d.max(), "px!=", (d.max(axis=1) > 0).mean(), "px>1", (d.max(axis=1) > 1).mean(), flush=True)
    sim.close()
run([], 720, 486, 4, "bars"); run([], 720, 486, 4, "noise")
run(["-vhs"], 720, 486, 4, "bars"); run(["-vhs"], 720, 486, 4, "noise")
run(["-vhs", "-vhs-speed", "ep"], 720, 480, 2, "noise"); run(["-vhs", "-comp-catv3"], 720, 480, 2, "noise")
run(["-vhs"], 1920, 1080, 2, "noise")
