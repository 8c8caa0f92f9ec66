"""Frame-round-robin sharding of the field loop across ranks (one process per GPU).

The reference's loop (ffmpeg_ntsc.cpp:2202-2282) is serial only because composite_layer() pulls
from one process-wide rand() stream.  The number of draws per call is data independent
(ntscsim_rng_calls_per_field), so the stream position of output field `cur` is a closed form and
every field can be computed anywhere, in any order, and still be bit-identical to the serial run.
No data-path collective is needed; ranks only meet for a barrier and a gather of checksums.
"""
from . import calls_per_field


def field_parity(cur):
    """ffmpeg_ntsc.cpp:2229: field = (current & 1) ^ 1."""
    return (cur & 1) ^ 1


def rng_pos_of_field(params, width, height, cur, base=0):
    """rand() stream position at the start of output field `cur` (fields 0..cur-1 came first)."""
    c1 = calls_per_field(params, width, height, 1)   # even `cur` -> field 1
    c0 = calls_per_field(params, width, height, 0)   # odd  `cur` -> field 0
    n_even = (cur + 1) // 2      # number of even indices in [0, cur)
    n_odd = cur // 2
    return base + n_even * c1 + n_odd * c0


def shard_fields(n_fields, rank, world, fields_per_frame=2):
    """Output-field indices owned by `rank`: frames (pairs of fields) dealt round-robin."""
    out = []
    n_frames = (n_fields + fields_per_frame - 1) // fields_per_frame
    for fr in range(rank, n_frames, world):
        for k in range(fields_per_frame):
            cur = fr * fields_per_frame + k
            if cur < n_fields:
                out.append(cur)
    return out


def jobs_for_rank(params, width, height, n_fields, rank, world):
    """[(cur, field, fieldno, rng_pos)] for this rank; cur // 2 is the source frame index."""
    return [(cur, field_parity(cur), cur, rng_pos_of_field(params, width, height, cur))
            for cur in shard_fields(n_fields, rank, world)]
