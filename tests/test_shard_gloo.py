"""Multi-rank path on CPU: world_size-2 gloo.  The product has no CPU compute path, so the field
kernel is stood in for by the oracle (allowed in tests); what is under test is the host logic the
N>1 bench uses: frame-round-robin sharding, closed-form rand() stream positions, and the
checksum gather -- the union of the shards must equal the serial run byte for byte."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _libs as L
from ntscsim import shard

W, H, NF = 64, 18, 12
FLAGS = ["-vhs", "-vhs-head-switching-point", "0.105", "-vhs-head-switching-phase", "0.002"]


def _serial():
    p = L.make_params(FLAGS)
    o = L.OracleStream(p)
    out = []
    for cur in range(NF):
        dst = np.zeros((H, W, 4), np.uint8)
        o.field(dst, L.noise_frame(W, H, 100 + cur // 2), (cur & 1) ^ 1, cur)
        out.append(dst)
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = L.make_params(FLAGS)
    jobs = shard.jobs_for_rank(p, W, H, NF, rank, world)
    res = {}
    for (cur, field, fieldno, pos) in jobs:
        o = L.OracleStream(p)
        o.skip(pos)                      # explicit stream position: any field, any order
        dst = np.zeros((H, W, 4), np.uint8)
        o.field(dst, L.noise_frame(W, H, 100 + cur // 2), field, fieldno)
        res[cur] = dst
    dist.barrier()
    mine = torch.zeros(NF, dtype=torch.int64)
    for cur, a in res.items():
        mine[cur] = int(L.fnv1a(a) & 0x7FFFFFFFFFFFFFFF)
    gathered = [torch.zeros(NF, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if rank == 0:
        q.put(torch.stack(gathered).sum(0).tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_partition_is_a_partition():
    for world in (1, 2, 3, 4, 8):
        seen = sorted(c for r in range(world) for c in shard.shard_fields(37, r, world))
        assert seen == list(range(37))
        # both fields of a frame live on the same rank (they share a destination frame)
        for r in range(world):
            fs = shard.shard_fields(36, r, world)
            assert all(fs[i] // 2 == fs[i + 1] // 2 for i in range(0, len(fs), 2))


def test_rng_positions_closed_form():
    p = L.make_params(["-vhs"])
    o = L.OracleStream(p)
    src = L.noise_frame(96, 33)
    dst = np.zeros_like(src)
    for cur in range(5):
        assert shard.rng_pos_of_field(p, 96, 33, cur) == o.rng_pos
        o.field(dst, src, (cur & 1) ^ 1, cur)


def test_two_ranks_equal_serial():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    exp = [int(L.fnv1a(a) & 0x7FFFFFFFFFFFFFFF) for a in _serial()]
    assert got == exp
