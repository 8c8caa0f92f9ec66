"""ntscsim_submit() / ntscsim_wait(): the asynchronous host-frame form of the composite_layer() drop-in
(include/ntscsim.h; the call site it replaces is ffmpeg_ntsc.cpp:2229 inside the loop :2202-2282).

The contract under test: a sequence of submits + waits leaves the SAME BYTES in the caller's frames, and the
ctx at the same rand() position, as the same sequence of synchronous ntscsim_field() calls on the same
pointers -- which in turn equals the oracle's composite_layer() (+ the loop's line doubling with
NTSCSIM_DESC_BOB).  Everything goes through ctypes -> C-ABI.
"""
import ctypes as C

import numpy as np
import pytest

import _libs as L
import ntscsim
from ntscsim import _capi


def page_frame(h, w, fill=0):
    """A [h, w, 4] uint8 frame with a mapping of its own (anonymous mmap, page aligned) -- what malloc / posix_memalign
    hand out for frame-sized requests, made deterministic: the engine pins caller memory page-wise, and only memory
    that is an allocation of its own (never blocks inside the brk heap, never frames under 64 KiB)."""
    import mmap
    nbytes = h * w * 4
    m = mmap.mmap(-1, (nbytes + 4095) // 4096 * 4096)
    a = np.frombuffer(m, np.uint8, nbytes).reshape(h, w, 4)
    a[:] = fill
    return a


def oracle_bob(frame, field):
    h, w = frame.shape[:2]
    L.oracle().ntsc_oracle_bob(L._ptr(frame), w * 4, w, h, field)


def reference_loop(p, frames, n_fields, w, h, ring, bob, fill=0x5A):
    """The loop of ffmpeg_ntsc.cpp:2202-2282 on the oracle: field k reads frame k//2, writes ring[k % len(ring)],
    is line-doubled in place (:2233-2257) when `bob`; a snapshot of the ring frame is what the encoder got."""
    o = L.OracleStream(p)
    bufs = [np.full((h, w, 4), fill, np.uint8) for _ in range(ring)]
    snaps = []
    for k in range(n_fields):
        field = (k & 1) ^ 1
        d = bufs[k % ring]
        o.field(d, frames[k // 2], field, k)
        if bob:
            oracle_bob(d, field)
        snaps.append(d.copy())
    return snaps, o.rng_pos


def run_submit_loop(sim, frames, n_fields, w, h, ring, bob, lag, pad=0, same_src=True, fill=0x5A):
    """The same loop with ntscsim_submit(): ONE source buffer that is rewritten for every new frame (the
    reference's in.rgb), a ring of `ring` destination frames, wait + snapshot `lag` fields behind."""
    src = page_frame(h, w + pad)
    bufs = [page_frame(h, w + pad, fill) for _ in range(ring)]
    snaps = [None] * n_fields
    tickets = []
    for k in range(n_fields):
        field = (k & 1) ^ 1
        new = (k & 1) == 0 or not same_src
        if new:
            src[:, :w] = frames[k // 2]
        t = sim.submit(bufs[k % ring][:, :w], src[:, :w], field, k, bob=bob, same_src=not new)
        tickets.append(t)
        src[:, :w] = 0xEE              # the caller may rewrite src as soon as submit returns
        j = k - lag
        if j >= 0:
            sim.wait(tickets[j])
            snaps[j] = bufs[j % ring][:, :w].copy()
            assert (bufs[j % ring][:, w:] == fill).all(), "padding bytes were written"
    for j in range(max(0, n_fields - lag), n_fields):
        sim.wait(tickets[j])
        snaps[j] = bufs[j % ring][:, :w].copy()
    assert tickets == list(range(tickets[0], tickets[0] + n_fields))
    sim.host_unpin()       # the frames are about to be freed: drop the engine's registrations first
    return snaps


@pytest.mark.gpu
@pytest.mark.parametrize("bob", [False, True])
@pytest.mark.parametrize("pin", [False, True])
@pytest.mark.parametrize("h", [32, 33])
def test_submit_wait_equals_the_synchronous_loop(bob, pin, h):
    """Small frames, depth 4, a ring as deep as the lag: every snapshot == the oracle's loop, the rand()
    position == the oracle's; both delivery paths (pinned in place / staging ring)."""
    w, n = (192 if pin else 96), 26    # (pinned in place: frames of at least 64 KiB; h = 33: the line doubling leaves a
    h = h * 3 if pin else h            #  different row alone for each field parity)
    p = L.make_params(["-vhs"], output_height=h)
    frames = [L.noise_frame(w, h, 700 + j) for j in range(n // 2)]
    ring, lag = 12, 9
    exp, exp_pos = reference_loop(p, frames, n, w, h, ring, bob)
    sim = ntscsim.FieldSimulator(params=p)
    sim.submit_configure(depth=4, slots=16, lanes=2, pin=pin, min_pin_bytes=0)
    got = run_submit_loop(sim, frames, n, w, h, ring, bob, lag, pad=8)
    for k in range(n):
        assert np.array_equal(got[k], exp[k]), "field %d" % k
    assert sim.rng_pos == exp_pos
    st = sim.submit_stats()
    assert st["submitted"] == n and st["uploads"] == n // 2
    if pin:
        assert st["delivered_direct"] == n and st["delivered_staged"] == 0 and st["uploads_staged"] == 0
    else:
        assert st["delivered_staged"] == n and st["registrations"] == 0
    # ... and the synchronous call continues the same stream on the same ctx
    o = L.OracleStream(p)
    o.skip(exp_pos)
    a = np.zeros((h, w, 4), np.uint8)
    b = np.zeros((h, w, 4), np.uint8)
    sim.field_host(a, frames[0], 1, n)
    o.field(b, frames[0], 1, n)
    assert np.array_equal(a, b) and sim.rng_pos == o.rng_pos
    sim.close()


@pytest.mark.gpu
def test_submit_at_the_baseline_size_default_options():
    """720x486 -vhs with the default engine (depth 32, pinned caller frames): 150 fields through a ring of
    40 frames, bob on; compared with the oracle on a sample of fields and with the synchronous ntscsim_field()
    loop on all of them."""
    w, h, n = 720, 486, 150
    p = L.make_params(["-vhs"])
    frames = [L.bars(w, h, j) for j in range(n // 2)]
    ring, lag = 40, 36
    sim = ntscsim.FieldSimulator(params=p)
    got = run_submit_loop(sim, frames, n, w, h, ring, True, lag)
    st = sim.submit_stats()
    assert st["delivered_direct"] == n and st["uploads_staged"] == 0, st
    assert "k_field_pipe<double>" in sim.last_kernels()      # launches of <= 64 fields: the three-role workgroup form
    pos_async = sim.rng_pos
    sim.close()
    # the synchronous loop on the product
    sim2 = ntscsim.FieldSimulator(params=p)
    bufs = [np.full((h, w, 4), 0x5A, np.uint8) for _ in range(ring)]
    for k in range(n):
        field = (k & 1) ^ 1
        sim2.field_host(bufs[k % ring], frames[k // 2], field, k)
        oracle_bob(bufs[k % ring], field)
        assert np.array_equal(got[k], bufs[k % ring]), "field %d" % k
    assert sim2.rng_pos == pos_async
    sim2.close()
    # the oracle on a few of them (explicit rand() positions)
    from ntscsim import shard
    for k in (0, 1, 77, n - 1):
        o = L.OracleStream(p)
        o.skip(shard.rng_pos_of_field(p, w, h, k))
        e = got[k].copy()
        field = (k & 1) ^ 1
        rows = e[field::2].copy()
        e[field::2] = 0
        o.field(e, frames[k // 2], field, k)
        assert np.array_equal(e[field::2], rows), "field %d vs oracle" % k


@pytest.mark.gpu
def test_two_fields_in_flight_share_one_destination_frame():
    """Without bob the two fields of a frame write disjoint rows of ONE dst frame, as in the reference with
    `-d 1`; in flight together they must both land, other rows untouched until their own field arrives."""
    w, h, n = 192, 96, 12
    p = L.make_params([])
    frames = [L.noise_frame(w, h, 900 + j) for j in range(n // 2)]
    exp, exp_pos = reference_loop(p, frames, n, w, h, 1, False)
    sim = ntscsim.FieldSimulator(params=p)
    sim.submit_configure(depth=2, slots=8, lanes=2, min_pin_bytes=0)
    src = page_frame(h, w)
    dst = page_frame(h, w, 0x5A)
    for k in range(0, n, 2):
        src[:] = frames[k // 2]
        t0 = sim.submit(dst, src, (k & 1) ^ 1, k)
        t1 = sim.submit(dst, src, ((k + 1) & 1) ^ 1, k + 1, same_src=True)
        sim.wait(t1)
        assert t1 == t0 + 1
        assert np.array_equal(dst, exp[k + 1]), "pair %d" % k
    assert sim.rng_pos == exp_pos
    sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bob", [False, True])
@pytest.mark.parametrize("lanes,pin", [(1, True), (3, True), (3, False)])
def test_fields_in_flight_on_one_frame_land_in_submit_order(bob, lanes, pin):
    """The header's promise: fields in flight that write the same rows of ONE dst frame -- every field
    with line doubling, every second field without -- are delivered in submit order, whatever lane their launch
    ran on: after the last wait the frame is the synchronous loop's (the reference with `-d 1` :2070, :2277)."""
    w, h, n = 192, 96, 30
    p = L.make_params(["-vhs"], output_height=h)
    frames = [L.noise_frame(w, h, 40 + j) for j in range(n // 2)]
    exp, exp_pos = reference_loop(p, frames, n, w, h, 1, bob)
    sim = ntscsim.FieldSimulator(params=p)
    sim.submit_configure(depth=3, slots=32, lanes=lanes, pin=pin, min_pin_bytes=0)
    src = page_frame(h, w)
    dst = page_frame(h, w, 0x5A)
    for k in range(n):
        if k % 2 == 0:
            src[:] = frames[k // 2]
        sim.submit(dst, src, (k & 1) ^ 1, k, bob=bob, same_src=(k % 2 == 1))
        if k == 17:                  # somewhere in the middle the frame is looked at: everything up to here, in order
            sim.wait()
            assert np.array_equal(dst, exp[k]), "after field %d" % k
    sim.wait()
    assert np.array_equal(dst, exp[n - 1])
    assert sim.rng_pos == exp_pos
    sim.host_unpin()
    sim.close()


@pytest.mark.gpu
def test_geometry_change_unaligned_rows_and_interlaced_source():
    """A width whose rows are not 16-byte aligned (generic kernels, 4-byte delivery), then another geometry on
    the same ctx (the engine drains and rebuilds its rings), interlaced source flags passed through."""
    p = L.make_params(["-vhs", "-vhs-speed", "ep"])
    sim = ntscsim.FieldSimulator(params=p)
    sim.submit_configure(depth=3, slots=6, lanes=1)          # (default threshold: these small heap arrays are staged)
    o = L.OracleStream(p)
    k = 0
    for (w, h, inter, tff) in ((50, 21, 0, 0), (96, 32, 1, 1), (50, 21, 1, 0)):
        fr = [L.noise_frame(w, h, 1000 + w + j) for j in range(4)]
        exp = [np.full((h, w, 4), 7, np.uint8) for _ in range(8)]
        got = [np.full((h, w, 4), 7, np.uint8) for _ in range(8)]
        ts = []
        for i in range(8):
            field = (k & 1) ^ 1
            o.field(exp[i], fr[i // 2], field, k, interlaced=inter, tff=tff)
            ts.append(sim.submit(got[i], fr[i // 2].copy(), field, k, interlaced=inter, tff=tff))
            k += 1
        sim.wait()
        for i in range(8):
            assert np.array_equal(got[i], exp[i]), (w, h, i)
        assert sim.rng_pos == o.rng_pos
    sim.close()


@pytest.mark.gpu
def test_submit_errors_consume_nothing_and_tickets_are_checked():
    w, h = 96, 32
    p = L.make_params([])
    sim = ntscsim.FieldSimulator(params=p)
    lib = sim._lib
    a = np.zeros((h, w, 4), np.uint8)
    b = np.zeros((h, w, 4), np.uint8)
    t = C.c_uint64(99)
    args = (a.ctypes.data, w * 4, 0, 0, b.ctypes.data, w * 4, w, h)
    assert lib.ntscsim_submit(sim._h, None, w * 4, 0, 0, b.ctypes.data, w * 4, w, h, 0, 0, 0, C.byref(t)) == _capi.E_ARG
    assert lib.ntscsim_submit(sim._h, a.ctypes.data, w * 4 - 4, 0, 0, b.ctypes.data, w * 4, w, h, 0, 0, 0, C.byref(t)) == _capi.E_SIZE
    assert lib.ntscsim_submit(sim._h, *args, 2, 0, 0, C.byref(t)) == _capi.E_ARG            # field > 1
    assert lib.ntscsim_submit(sim._h, *args, 0, 0, 0x4, C.byref(t)) == _capi.E_ARG          # unknown flag
    assert t.value == 99 and sim.rng_pos == 0 and sim.submit_stats()["submitted"] == 0
    assert lib.ntscsim_wait(sim._h, 1) == _capi.E_ARG                                       # never issued
    assert lib.ntscsim_wait(sim._h, _capi.TICKET_ALL) == _capi.OK
    assert lib.ntscsim_submit(sim._h, *args, 1, 0, 0, C.byref(t)) == _capi.OK and t.value == 1
    assert lib.ntscsim_wait(sim._h, 2) == _capi.E_ARG
    assert lib.ntscsim_flush(sim._h) == _capi.OK
    assert lib.ntscsim_wait(sim._h, 1) == _capi.OK
    assert lib.ntscsim_wait(sim._h, 1) == _capi.OK                                          # again: already done
    o = _capi.SubmitOpts()
    lib.ntscsim_submit_opts_init(C.byref(o))
    assert (o.depth, o.slots, o.lanes, o.pin_caller_buffers) == (32, 256, 3, 1)
    o.slots = o.depth          # < 2 * depth
    assert lib.ntscsim_submit_configure(sim._h, C.byref(o)) == _capi.E_ARG
    sim.close()


@pytest.mark.gpu
def test_ring_full_blocks_and_unpin_releases():
    """More submits than ring slots without a single wait: the engine retires the oldest launches itself (their
    rows land in the caller's frames) and keeps going; ntscsim_host_unpin() drops the registrations."""
    w, h, n = 192, 96, 40
    p = L.make_params(["-vhs"])
    frames = [L.noise_frame(w, h, 1200 + j) for j in range(n // 2)]
    exp, exp_pos = reference_loop(p, frames, n, w, h, n, False)
    sim = ntscsim.FieldSimulator(params=p)
    sim.submit_configure(depth=4, slots=8, lanes=3, min_pin_bytes=0)
    bufs = [page_frame(h, w, 0x5A) for _ in range(n)]
    srcs = []
    for f in frames:
        s_ = page_frame(h, w)
        s_[:] = f
        srcs.append(s_)
    for k in range(n):
        sim.submit(bufs[k], srcs[k // 2], (k & 1) ^ 1, k, same_src=bool(k & 1))
    sim.wait()
    for k in range(n):
        assert np.array_equal(bufs[k], exp[k]), k
    st = sim.submit_stats()
    assert st["ring_full_waits"] > 0 and st["registrations"] > 0
    sim.host_unpin()
    assert sim.submit_stats()["registrations"] == 0
    assert sim.rng_pos == exp_pos
    sim.close()


def test_submit_opts_defaults_without_a_gpu():
    """(CPU) the options POD and its defaults; submit on a NULL ctx is an argument error, not a crash."""
    lib = ntscsim.lib()
    o = _capi.SubmitOpts()
    lib.ntscsim_submit_opts_init(C.byref(o))
    assert o.struct_size == C.sizeof(_capi.SubmitOpts) and o.depth == 32 and o.min_pin_bytes == 256 << 10
    assert lib.ntscsim_submit(None, None, 0, 0, 0, None, 0, 0, 0, 0, 0, 0, None) == _capi.E_ARG
    assert lib.ntscsim_wait(None, 1) == _capi.E_ARG
    assert lib.ntscsim_flush(None) == _capi.E_ARG


# ---- the C++ host of INTEGRATION.md section 1 / 1b: the reference's loop with the call at :2229 replaced ------------
import json          # noqa: E402
import os            # noqa: E402
import subprocess    # noqa: E402

FIELD_LOOP = os.path.join(L.PKG, "field_loop")


def _run_field_loop(mode, extra=()):
    r = subprocess.run([FIELD_LOOP, "-vhs", "--mode", mode, "--fields", "80", "--warmup", "0", "--hash", "1", "--depth", "8",
                        "--rewrite-src", "1", "--ring", "40"] + list(extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("bob", ["0", "1"])
def test_cpp_field_loop_allocators_pinned_pool_and_plain_malloc(bob):
    """VERDICT r05 item 2 for the BGRA tool: frames from ntscsim_host_frame_alloc() and frames carved from a pool that
    was declared with ntscsim_host_pin() take the no-copy path; posix_memalign'ed frames are staged (copy threads) under
    the default policy and pinned in place only under the opt-in glibc policy.  Same frames in the same order (the same
    ring of 40 output frames in both modes: without the line doubling a frame keeps the other field's rows of its
    previous use, and the hash covers whole frames)."""
    a = _run_field_loop("sync", ["--bob", bob])
    for alloc, pin, direct in (("malloc", "1", False), ("pinned", "1", True), ("pool", "1", True), ("malloc", "0", False)):
        b = _run_field_loop("submit", ["--bob", bob, "--alloc", alloc, "--pin", pin])
        assert b["fnv1a"] == a["fnv1a"] != "0000000000000000" and b["rng_pos"] == a["rng_pos"], (alloc, pin)
        assert (b["stats"]["delivered_direct"] > 0) == direct, (alloc, pin, b["stats"])
        assert (b["stats"]["delivered_staged"] > 0) == (not direct), (alloc, pin, b["stats"])


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["declared", "host_alloc", "neither"])
def test_frames_that_are_not_page_aligned_need_a_declaration_or_pinned_memory(how):
    """The pin rules of include/ntscsim.h ("Host buffers") through the Python veneer: frames that start 80 bytes into a
    block (what an allocator's header + alignment gives) are written by the GPU in place when the block was DECLARED with
    ntscsim_host_pin() or IS pinned memory (ntscsim_host_alloc: the runtime is asked), and are staged -- nothing
    registered, no header word read -- when they are neither.  Same frames every way."""
    w, h, n = 192, 96, 12
    p = L.make_params(["-vhs"], output_height=h)
    frames = [L.noise_frame(w, h, 900 + j) for j in range(n // 2)]
    exp, exp_pos = reference_loop(p, frames, n, w, h, n, False)
    sim = ntscsim.FieldSimulator(params=p)
    sim.submit_configure(depth=4, slots=16, lanes=2, min_pin_bytes=0)
    nbytes = h * w * 4
    blocks = []
    if how == "host_alloc":
        views = []
        for _ in range(n + 1):
            a = ntscsim.host_alloc_array((nbytes + 4096,), align_offset=0)
            blocks.append(a)
            views.append(a[80:80 + nbytes].reshape(h, w, 4))
    else:
        import mmap
        m = mmap.mmap(-1, (n + 1) * (nbytes + 4096))
        pool = np.frombuffer(m, np.uint8)
        views = [pool[i * (nbytes + 4096) + 80: i * (nbytes + 4096) + 80 + nbytes].reshape(h, w, 4) for i in range(n + 1)]
        if how == "declared":
            sim.host_pin(pool)
    src, bufs = views[0], views[1:]
    for b in bufs:
        b[:] = 0x5A
    tickets = []
    for k in range(n):
        if (k & 1) == 0:
            src[:] = frames[k // 2]
        tickets.append(sim.submit(bufs[k], src, (k & 1) ^ 1, k, same_src=(k & 1) == 1))
    sim.wait()
    for k in range(n):
        assert np.array_equal(bufs[k], exp[k]), (how, k)
    assert sim.rng_pos == exp_pos
    st = sim.submit_stats()
    if how == "neither":
        assert st["delivered_direct"] == 0 and st["delivered_staged"] == n and st["registrations"] == 0, st
    else:
        assert st["delivered_direct"] == n and st["uploads_staged"] == 0, st
    sim.host_unpin()
    sim.close()
    for a in blocks:
        ntscsim.host_free_array(a)


@pytest.mark.gpu
def test_destroy_with_staged_fields_in_flight_drops_them():
    """ntscsim_destroy() with submitted-but-never-waited fields whose frames are staged: the copy threads drop what they have
    not delivered yet instead of writing into frames the caller may have freed; nothing hangs, a new ctx works."""
    w, h = 192, 96
    p = L.make_params(["-vhs"], output_height=h)
    src = L.noise_frame(w, h, 5)
    for _ in range(3):
        sim = ntscsim.FieldSimulator(params=p)
        sim.submit_configure(depth=4, slots=16, lanes=2, pin=0)
        bufs = [np.zeros((h, w, 4), np.uint8) for _ in range(10)]
        for k in range(10):
            sim.submit(bufs[k], src, (k & 1) ^ 1, k)
        sim.close()
        del bufs
    sim = ntscsim.FieldSimulator(params=p)
    a, b = np.zeros((h, w, 4), np.uint8), np.zeros((h, w, 4), np.uint8)
    sim.field_host(a, src, 1, 0)
    L.OracleStream(p).field(b, src, 1, 0)
    assert np.array_equal(a, b)
    sim.close()
