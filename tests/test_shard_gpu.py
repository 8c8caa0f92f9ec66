"""Multi-rank path with the REAL kernels: world_size 2 and 3 on the box's one GPU (gloo for the
barrier / gather, RCCL refuses several ranks on one device).  Every rank runs the HIP path on its
frame-round-robin share with closed-form rand() positions; the union of the shards must equal the
serial HIP run byte for byte, and the gathered checksums must be those of the serial run.  Also runs
bench.py's own multi-rank branch (strong scaling, and the 8-stream deal of BASELINE configs[3]) and
checks its self-verification."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import _libs as L
import ntscsim

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
W, H, NF = 96, 34, 12
FLAGS = ["-vhs", "-vhs-head-switching-point", "0.105", "-vhs-head-switching-phase", "0.002"]


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, script_args, timeout=300):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_port())] + script_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_hip_shards_equal_the_serial_run(world, tmp_path):
    import torch
    r = _launch(world, [os.path.join(HERE, "_shard_gpu_worker.py"), str(tmp_path), str(W), str(H), str(NF)] + FLAGS)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    # serial HIP run in this process, every field into its own frame
    p = L.make_params(FLAGS)
    sim = ntscsim.FieldSimulator(params=p)
    src = torch.from_numpy(np.stack([L.noise_frame(W, H, 100 + f) for f in range(NF // 2)])).cuda()
    dst = torch.zeros((NF, H, W, 4), dtype=torch.uint8, device="cuda")
    sim.fields(src, dst, [(k // 2, k, (k & 1) ^ 1, k) for k in range(NF)])
    sim.sync()
    serial = dst.cpu().numpy()
    sim.close()
    # ... which is the oracle's serial run
    o = L.OracleStream(p)
    for k in range(NF):
        e = np.zeros((H, W, 4), np.uint8)
        o.field(e, L.noise_frame(W, H, 100 + k // 2), (k & 1) ^ 1, k)
        assert np.array_equal(serial[k], e), k
    seen = []
    for rank in range(world):
        got = np.load(tmp_path / ("rank%d.npy" % rank))
        curs = np.load(tmp_path / ("rank%d_cur.npy" % rank))
        for k, cur in enumerate(curs):
            assert np.array_equal(got[k], serial[cur]), (rank, cur)
            seen.append(int(cur))
    assert sorted(seen) == list(range(NF))
    cs = np.load(tmp_path / "checksums.npy")
    assert cs.tolist() == [int(L.fnv1a(serial[k]) & 0x7FFFFFFFFFFFFFFF) for k in range(NF)]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--scaling", "strong"], ["--streams", "8"]],
                         ids=["weak", "strong", "streams8"])
def test_bench_multirank_branch_verifies_itself(extra):
    r = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "3",
                    "--warmup", "1", "--frames", "12", "--cpu-fields", "0", "--sustain-seconds", "0"] + extra)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["rank_checksums_verified"] is True
    assert len(d["config"]["rank_checksums"]) == 2
    if extra == ["--scaling", "strong"]:
        assert d["scaling"] == "strong" and sum(d["config"]["fields_per_step_per_gpu"]) == 24
    elif extra == []:
        assert d["config"]["fields_per_step_per_gpu"] == [24, 24]
    else:
        assert d["config"]["fields_per_step_per_gpu"] == [96, 96]      # 4 of the 8 streams each
