"""Worker of tests/test_shard_gpu.py: one rank of a world_size-N run in which EVERY rank drives the
HIP path on the (single) GPU of the box; ranks meet over gloo for the barrier and the checksum
gather only -- the exchange pattern of the N > 1 bench.  Launched by torch.distributed.run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "composite-video-simulator_amd"))
import _libs as L  # noqa: E402
import ntscsim  # noqa: E402
from ntscsim import shard  # noqa: E402


def main():
    outdir, w, h, nf = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    flags = sys.argv[5:]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = L.make_params(flags)
    jobs = shard.jobs_for_rank(p, w, h, nf, rank, world)
    frames = sorted({cur // 2 for (cur, _, _, _) in jobs})
    src = torch.from_numpy(np.stack([L.noise_frame(w, h, 100 + f) for f in frames])).cuda()
    dst = torch.zeros((len(jobs), h, w, 4), dtype=torch.uint8, device="cuda")
    sim = ntscsim.FieldSimulator(params=p)
    loc = [(frames.index(cur // 2), k, field, fieldno) for k, (cur, field, fieldno, _) in enumerate(jobs)]
    descs = sim.build_descs(src, dst, loc, rng_pos=[j[3] for j in jobs])
    dist.barrier()
    sim.run_descs(descs, w, h)
    sim.sync()
    dist.barrier()
    got = dst.cpu().numpy()
    sim.close()
    mine = torch.zeros(nf, dtype=torch.int64)
    for k, (cur, _, _, _) in enumerate(jobs):
        mine[cur] = int(L.fnv1a(got[k]) & 0x7FFFFFFFFFFFFFFF)
    gathered = [torch.zeros(nf, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, mine)
    np.save(os.path.join(outdir, "rank%d.npy" % rank), got)
    np.save(os.path.join(outdir, "rank%d_cur.npy" % rank), np.array([j[0] for j in jobs]))
    if rank == 0:
        np.save(os.path.join(outdir, "checksums.npy"), torch.stack(gathered).sum(0).numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
