"""The YUV422P tool's loop on HOST frames: ntscsim_field422() / ntscsim_submit422() + ntscsim_wait()
(include/ntscsim.h; VERDICT r04 "missing" 1).  One call = one iteration of ffmpeg_to_composite.cpp:1783-1800
(render_field :1784 -> black_key_feedback :1787 -> composite_video_process :1790 / :629 -> output_frame's copy
:1177-1236) on the tool's own frames.  Every test replays the same loop with the oracle on byte-identical
buffers (same linesizes, same padding bytes -- the separator reads two bytes behind each luma row, :496) and
compares WHOLE buffers: the persistent frame, the filter frame, every encoder frame, the rand() position."""
import ctypes as C

import numpy as np
import pytest

import _libs as L
from ntscsim import _capi

pytestmark = pytest.mark.gpu

OUT_BOB422, OUT_BOB420, OUT_INT420, OUT_FRAME = 0, 1, 2, 3
F_IL, F_TFF, F_420, F_SECOND, F_NOCOMP = 1, 2, 4, 8, 16


def f422(fr):
    o = _capi.Frame422()
    if fr is not None:
        for k in range(3):
            o.data[k] = fr.buf.ctypes.data + fr.off[k]
            o.linesize[k] = fr.ls[k]
    return o


class PagedYuv(L.Yuv422):
    """A YUV422P frame whose three planes each start on a page of their own inside ONE anonymous mapping -- what
    av_frame_get_buffer's per-plane buffers look like to the engine, made deterministic: planes of >= 64 KiB that start
    on a page boundary are pinned in place (hipHostRegister) and take the DMA upload / kernel delivery paths."""

    def __init__(self, w, h, pad=0, fill=0):
        import mmap
        self.w, self.h = w, h
        self.ls = [w + pad, w // 2 + pad, w // 2 + pad]
        sizes = [(self.ls[i] * h + 64 + 4095) // 4096 * 4096 for i in range(3)]
        self._m = mmap.mmap(-1, sum(sizes))
        self.buf = np.frombuffer(self._m, np.uint8)
        self.buf[:] = fill
        self.off = [0, sizes[0], sizes[0] + sizes[1]]

    def copy(self):
        o = PagedYuv(self.w, self.h, self.ls[0] - self.w)
        o.buf[:] = self.buf
        return o


def paged_noise(w, h, seed, pad=0):
    f = PagedYuv(w, h, pad)
    src = L.yuv_noise(w, h, seed, pad)
    for k in range(3):
        f.plane(k)[:] = src.plane(k)
    return f


class Ctx:
    def __init__(self, params, depth=None):
        self.lib = L.product()
        self.h = C.c_void_p()
        rc = self.lib.ntscsim_create(C.byref(params), 0, C.byref(self.h))
        assert rc == 0, rc
        if depth is not None:
            assert self.lib.ntscsim_submit422_configure(self.h, depth, 0) == 0

    def close(self):
        self.lib.ntscsim_destroy(self.h)

    def loop(self, frame, src, field, fieldno, flags=0, flt=None, out=None, out_mode=0, out_field=None, sh=None):
        it = _capi.Loop422()
        it.struct_size = C.sizeof(_capi.Loop422)
        it.width, it.height = frame.w, frame.h
        it.frame = f422(frame)
        it.src = f422(src)
        it.src_height = (sh if sh is not None else src.h) if src is not None else 0
        it.filter = f422(flt)
        it.out = f422(out)
        it.field, it.flags, it.fieldno = field, flags, fieldno
        it.out_mode = out_mode
        it.out_field = field if out_field is None else out_field
        return it

    def field(self, it):
        rc = self.lib.ntscsim_field422(self.h, C.byref(it))
        assert rc == 0, (rc, self.lib.ntscsim_last_error(self.h))

    def submit(self, it, flags=0):
        t = C.c_uint64(0)
        rc = self.lib.ntscsim_submit422(self.h, C.byref(it), flags, C.byref(t))
        assert rc == 0, (rc, self.lib.ntscsim_last_error(self.h))
        return t.value

    def wait(self, t=_capi.TICKET_ALL):
        rc = self.lib.ntscsim_wait(self.h, t)
        assert rc == 0, (rc, self.lib.ntscsim_last_error(self.h))

    @property
    def rng_pos(self):
        return self.lib.ntscsim_get_rng_pos(self.h)

    def stats(self):
        a = (C.c_uint64 * 8)()
        self.lib.ntscsim_submit422_stats(self.h, a)
        return list(a)

    def unpin(self):
        assert self.lib.ntscsim_host_unpin(self.h, None) == 0

    def last_kernels(self):
        buf = C.create_string_buffer(1024)
        n = self.lib.ntscsim_debug_last_kernels(self.h, buf, len(buf))
        assert n >= 0, n
        return [k for k in buf.value.decode().split(";") if k]


def enc_frame(w, h, mode, fill=0):
    """the encoder's frame: YUV422P, or YUV420P in a Yuv422-shaped buffer of which (h + 1) // 2 chroma rows count"""
    f = L.Yuv422(w, h, pad=0, fill=fill)
    return f


def crows(h, mode):
    return h if mode in (OUT_BOB422, OUT_FRAME) else (h + 1) // 2


MARGIN_Y, MARGIN_C = 24, 14
MASK_LAST_ROW = False


def margin_mask(frame):
    """True where the host API must equal the oracle.  run_loop() drives the oracle in PLANE mode -- the library's own
    contract for the separator's two bytes behind a row: the caller's bytes inside the luma plane, 16 outside it -- and
    compares EVERY byte (MASK_LAST_ROW = False).  Against the MEMORY mode (the reference's literal read past the last
    row of a tight plane, into the neighbouring allocation) the right margin of the frame's LAST row is excluded
    (tests/test_variant422.py)."""
    m = np.ones(frame.buf.shape, bool)
    if frame.ls[0] >= frame.w + 2 or not MASK_LAST_ROW:
        return m
    for i, marg in ((0, MARGIN_Y), (1, MARGIN_C), (2, MARGIN_C)):
        n = frame.ls[i] * frame.h
        pm = m[frame.off[i]:frame.off[i] + n].reshape(frame.h, frame.ls[i])
        width = frame.w if i == 0 else frame.w // 2
        pm[frame.h - 1, max(0, width - marg):width] = False
    return m


def same_frames(a, b, what):
    bad = np.nonzero((a.buf != b.buf) & margin_mask(a))[0]
    assert bad.size == 0, "%s: first mismatch at byte %d of %d (%d differ)" % (what, bad[0], a.buf.size, bad.size)


def same_out(a, b, h, mode, what, tight=False):
    """encoder frames; with tight frame rows the last row's right margin (see margin_mask) is copied into the last
    rows of the encoder frame by the bob / repack, so it is excluded there as well"""
    for k in range(3):
        n = h if k == 0 else crows(h, mode)
        x, y = a.plane(k)[:n].copy(), b.plane(k)[:n].copy()
        if tight and MASK_LAST_ROW:
            width = a.w if k == 0 else a.w // 2
            marg = MARGIN_Y if k == 0 else MARGIN_C
            x[n - 2:, width - marg:width] = 0
            y[n - 2:, width - marg:width] = 0
        assert np.array_equal(x, y), (what, k)


def random_padding(fr, seed):
    """fill everything that is not a pixel with noise: the bytes behind the rows are the caller's"""
    r = np.random.RandomState(seed)
    keep = [fr.pix(k).copy() for k in range(3)]
    fr.buf[:] = r.randint(0, 256, size=fr.buf.size, dtype=np.uint8)
    for k in range(3):
        fr.pix(k)[:] = keep[k]


def sources(n, w, sh, seed=100, is420=False):
    out = []
    for j in range(n):
        out.append(L.yuv_noise(w, sh, seed + j))
    return out


def oracle_iteration(o, p, frame, src, field, vf, flags, flt, out, out_mode, out_field):
    if src is not None:
        L.tocomp_oracle_render_field(frame, src, int(bool(flags & F_420)), int(bool(flags & F_IL)), int(bool(flags & F_TFF)),
                                     int(bool(flags & F_SECOND)), field)
    if flt is not None and p.black_key_level_feedback >= 0:
        L.tocomp_oracle_black_key(frame, flt, field, p.black_key_level_feedback)
    if not (flags & F_NOCOMP):
        o.process(frame, field, vf)
    if out is not None and out_mode == OUT_FRAME:
        for k in range(3):
            out.pix(k)[:] = frame.pix(k)
    elif out is not None:
        L.tocomp_oracle_output_frame(out, frame, out_field, out_mode)


def run_loop(p, w, h, pad, n_src, out_mode, mode, sh=None, src_flags=0, bkey=False, interlaced_out=False, depth=8,
             with_src=True, frame_seed=5, lag=None):
    """The tool's loop (two fields per source frame) through the oracle and through the HOST API (`mode` = "sync" or
    "submit"); returns nothing, asserts whole-buffer equality of everything the loop touches."""
    sh = sh or h
    srcs = sources(n_src, w, sh)
    # ---- expected: the oracle on its own copies of identical buffers
    frame_o = L.yuv_noise(w, h, frame_seed, pad)
    random_padding(frame_o, frame_seed + 1)
    flt_o = L.yuv_noise(w, h, frame_seed + 2, pad) if bkey else None
    frame_g, flt_g = frame_o.copy(), (flt_o.copy() if bkey else None)
    o = L.TocompOracleStream(p, oob=L.OOB_PLANE)
    exp_outs, jobs = [], []
    vf = 0
    for s in srcs:
        for sub in (0, 1):
            field = (vf & 1) ^ 1
            flags = src_flags | (F_SECOND if sub else 0)
            emit = ((vf & 1) == 1) if interlaced_out else True
            of = (((vf - 1) & 1) ^ 1) if interlaced_out else field
            want_out = emit and out_mode is not None
            eo = enc_frame(w, h, out_mode, fill=7) if want_out else None
            oracle_iteration(o, p, frame_o, s if with_src else None, field, vf, flags, flt_o, eo, out_mode, of)
            exp_outs.append(eo)
            jobs.append((s if with_src else None, field, vf, flags, sub, of, want_out))
            vf += 1
    # ---- the host API on the other copies
    ctx = Ctx(p, depth=depth if mode == "submit" else None)
    got_outs, tickets = [], []
    for (s, field, vf_, flags, sub, of, want_out) in jobs:
        go = enc_frame(w, h, out_mode, fill=7) if want_out else None
        got_outs.append(go)
        it = ctx.loop(frame_g, s, field, vf_, flags, flt_g, go, out_mode or 0, of, sh=sh)
        if mode == "sync":
            ctx.field(it)
        else:
            tickets.append(ctx.submit(it, _capi.SUBMIT_SAME_SRC if (sub and s is not None) else 0))
            if lag is not None and len(tickets) > lag:
                ctx.wait(tickets[len(tickets) - 1 - lag])
    if mode == "submit":
        assert tickets == list(range(1, len(jobs) + 1))
        ctx.wait()
    st = ctx.stats()
    assert ctx.rng_pos == o.rng_pos
    ctx.close()
    same_frames(frame_g, frame_o, "frame")
    if bkey:
        same_frames(flt_g, flt_o, "filter frame")
    for i, (a, b) in enumerate(zip(got_outs, exp_outs)):
        if b is not None:
            same_out(a, b, h, out_mode, "encoder frame %d" % i, tight=frame_o.ls[0] < w + 2)
    return st


@pytest.mark.parametrize("mode", ["sync", "submit"])
@pytest.mark.parametrize("flags,out_mode,pad", [
    (["-vhs"], OUT_BOB420, 64),
    (["-vhs"], OUT_BOB422, 16),
    ([], OUT_BOB420, 32),                                  # the tool's default preset
    (["-vhs", "-vhs-speed", "ep"], OUT_BOB422, 2),         # the smallest padding that keeps the read inside the row
])
def test_loop_on_padded_frames_batches_and_equals_the_tool(flags, out_mode, pad, mode):
    w, h = 96, 36
    p = L.make_params_tocomp(flags + ["-width", str(w)], output_height=h)
    st = run_loop(p, w, h, pad, 9, out_mode, mode, depth=8)
    if mode == "submit":
        assert st[3] == 18 and st[4] == 0 and st[1] <= 4      # batched: 18 iterations in a few launches


@pytest.mark.parametrize("flags,form", [
    (["-vhs"], "k422_pipe<true,4>"),
    (["-vhs", "-vhs-speed", "lp"], "k422_pipe<false,5>"),
    (["-vhs", "-vhs-speed", "ep", "-chroma-dropout", "30000"], "k422_pipe<false,6>"),
    (["-vhs", "-vhs-svideo", "1"], "k422_pipe_sv<4>"),
    (["-vhs", "-noise", "0", "-chroma-noise", "40"], "k422_pipe<false,4>"),
    (["-vhs", "-vhs-head-switching-point", "0.85", "-noise", "9"], "k422_pipe<true,4>"),   # a displacement beyond W/10: the gather's
    (["-vhs", "-tvstd", "pal"], "k422_pipe<false,4>"),            # reach is the workgroup's own (up to the whole row)
    ([], "k422_direct_pipe"),                                     # no VCR: sweep A | gather | the decode sweep
])
def test_synchronous_iteration_takes_the_role_form_and_equals_the_tool(flags, form):
    """ntscsim_field422(): the streamed kernels of the -vhs family run as four wavefront ROLES of one workgroup (k422_pipe:
    sweep A | head-switch gather | front | back of the streamed pass -- the one-wave form's own code cut where only bytes
    cross) for short launches of the host-frame engine; whole-buffer equality with the oracle's loop, form asserted by name."""
    w, h = 720, 122
    p = L.make_params_tocomp(flags + ["-width", str(w)], output_height=h)
    srcs = sources(3, w, h)
    frame_o = L.yuv_noise(w, h, 5, 16)
    random_padding(frame_o, 6)
    frame_g = frame_o.copy()
    o = L.TocompOracleStream(p, oob=L.OOB_PLANE)
    ctx = Ctx(p)
    vf = 0
    for s in srcs:
        for sub in (0, 1):
            field = (vf & 1) ^ 1
            fl = F_SECOND if sub else 0
            eo, go = enc_frame(w, h, OUT_BOB422, fill=7), enc_frame(w, h, OUT_BOB422, fill=7)
            oracle_iteration(o, p, frame_o, s, field, vf, fl, None, eo, OUT_BOB422, field)
            ctx.field(ctx.loop(frame_g, s, field, vf, fl, None, go, OUT_BOB422, field, sh=h))
            kern = ctx.last_kernels()
            assert form in kern, kern
            same_frames(frame_g, frame_o, "frame, iteration %d" % vf)
            same_out(go, eo, h, OUT_BOB422, "encoder frame %d" % vf)
            vf += 1
    assert ctx.rng_pos == o.rng_pos
    ctx.close()


@pytest.mark.parametrize("mode", ["sync", "submit"])
def test_tight_rows_run_in_order_on_a_mirror(mode):
    """linesize == width: the separator's two bytes are the first pixels of the NEXT row -- the other field's, as the
    previous iteration left them (:496); one iteration at a time on a device copy with the caller's own linesizes"""
    w, h = 96, 36
    p = L.make_params_tocomp(["-vhs", "-width", str(w)], output_height=h)
    st = run_loop(p, w, h, 0, 5, OUT_BOB420, mode, depth=8)
    assert st[4] == 10 and st[3] == 0 and st[5] == 1          # serial; the frame went up once


@pytest.mark.parametrize("mode,depth,lag", [("sync", 8, None), ("submit", 8, None), ("submit", 3, None), ("submit", 8, 0), ("submit", 5, 2)])
@pytest.mark.parametrize("flags,w,h,out_mode,il", [
    (["-vhs"], 256, 36, OUT_BOB420, False),
    (["-vhs", "-vhs-speed", "ep"], 128, 35, OUT_BOB422, False),
    ([], 160, 36, OUT_BOB420, False),                       # the tool's default preset
    (["-vhs", "-yc-recomb", "2"], 256, 34, OUT_BOB422, False),
    (["-vhs"], 256, 36, OUT_INT420, True),                  # -vi: the repack reads both fields of the shared device frame
    (["-vhs", "-tvstd", "pal"], 192, 40, OUT_BOB420, False),
])
def test_tight_rows_are_batched_with_the_pad_bytes_chained_on_the_device(flags, w, h, out_mode, il, mode, depth, lag):
    """linesize == width at a width of >= 128: the two bytes behind each row are pixels 0, 1 of the NEXT row -- the other
    field's, as the previous iteration left them (:496) -- and that iteration may be in the same launch.  The engine
    runs such a launch twice with the bytes copied in between (pixels 0, 1 of a row never depend on the bytes behind a
    row); whole buffers equal the oracle's in-order loop, whatever the depth and wherever the waits fall."""
    p = L.make_params_tocomp(flags + ["-width", str(w)], output_height=h)
    st = run_loop(p, w, h, 0, 7, out_mode, mode, depth=depth, interlaced_out=il, lag=lag)
    if mode == "submit" and lag is None:
        assert st[3] == 14 and st[4] == 0, st               # batched, none one at a time
        assert st[1] <= 14 // depth + 2
    # linesize = width + 1: the first byte behind a row is the row's own padding (the caller's, noise here), the second
    # one pixel 0 of the next row (tools/fuzz_host422.py found the first build of the chain taking both from the next row)
    run_loop(p, w, h, 1, 5, out_mode, mode, depth=depth, interlaced_out=il, lag=lag)


def test_tight_rows_chain_survives_a_dirty_frame_and_a_one_at_a_time_iteration():
    """the chain of pad bytes breaks where the caller rewrites the frame (DIRTY) or an iteration without a source runs in
    order on the mirror -- the next tight iteration snapshots the caller's own bytes again"""
    w, h = 256, 36
    p = L.make_params_tocomp(["-vhs", "-width", str(w)], output_height=h)
    srcs = sources(6, w, h)
    frame_o = L.yuv_noise(w, h, 9, 0)
    frame_g = frame_o.copy()
    o = L.TocompOracleStream(p, oob=L.OOB_PLANE)
    ctx = Ctx(p, depth=4)
    for vf in range(12):
        field = (vf & 1) ^ 1
        s_ = srcs[vf // 2] if vf != 7 else None              # iteration 7: no source -> in order on the mirror
        if vf == 4:
            ctx.wait()
            for fr in (frame_o, frame_g):
                fr.pix(0)[3:9, 0:40] = 77
        oracle_iteration(o, p, frame_o, s_, field, vf, F_SECOND if vf & 1 else 0, None, None, 0, field)
        ctx.submit(ctx.loop(frame_g, s_, field, vf, F_SECOND if vf & 1 else 0), _capi.SUBMIT422_DIRTY if vf == 4 else 0)
    ctx.wait()
    st = ctx.stats()
    assert ctx.rng_pos == o.rng_pos and st[4] == 1 and st[3] == 11, st
    ctx.close()
    same_frames(frame_g, frame_o, "frame")


@pytest.mark.parametrize("mode", ["sync", "submit"])
def test_black_key_feedback_recurrence(mode):
    w, h = 96, 36
    p = L.make_params_tocomp(["-bkey-feedback", "40", "-width", str(w)], output_height=h)
    st = run_loop(p, w, h, 64, 6, OUT_BOB422, mode, bkey=True, depth=4)
    assert st[4] == 12


@pytest.mark.parametrize("mode", ["sync", "submit"])
@pytest.mark.parametrize("h,out_mode", [(36, OUT_INT420), (34, OUT_INT420), (36, None), (36, OUT_FRAME)])
def test_interlaced_output_pairs(h, out_mode, mode):
    """-vi: output_frame after every PAIR with the previous field's parity (:1792-1793) -- the repack reads both
    fields of the frame; (h = 34: 2 mod 4, the repack's chroma row past the plane is not delivered).  out_mode None =
    -vi -422: the tool encodes the frame itself (:1158), nothing to copy -- or, for a loop that keeps fields in flight,
    NTSCSIM_OUT422_FRAME: a 1:1 copy of the frame as it stood after the pair."""
    w = 96
    p = L.make_params_tocomp(["-vhs", "-width", str(w)], output_height=h)
    st = run_loop(p, w, h, 64, 7, out_mode, mode, interlaced_out=True, depth=6)
    if mode == "submit":
        assert st[3] == 14           # pairs share a device frame: the repack stays on the batched path


def test_a_pair_split_by_a_wait_still_repacks_exactly():
    """the consumer waits one field behind: the second field of a pair arrives after the first was launched"""
    w, h = 96, 36
    p = L.make_params_tocomp(["-vhs", "-width", str(w)], output_height=h)
    st = run_loop(p, w, h, 64, 5, OUT_INT420, "submit", interlaced_out=True, depth=8, lag=0)
    assert st[4] >= 1                # the odd fields fell back to the in-order path


@pytest.mark.parametrize("mode", ["sync", "submit"])
def test_sources_of_other_shapes(mode):
    w, h = 96, 36
    p = L.make_params_tocomp(["-vhs", "-vhs-speed", "lp", "-width", str(w)], output_height=h)
    run_loop(p, w, h, 64, 5, OUT_BOB420, mode, sh=50, src_flags=F_420)
    run_loop(p, w, h, 64, 5, OUT_BOB422, mode, sh=40, src_flags=F_IL | F_TFF)
    run_loop(p, w, h, 64, 4, OUT_BOB422, mode, src_flags=F_NOCOMP)


@pytest.mark.parametrize("mode", ["sync", "submit"])
def test_no_source_processes_the_frame_as_it_stands(mode):
    w, h = 96, 36
    p = L.make_params_tocomp(["-vhs", "-width", str(w)], output_height=h)
    run_loop(p, w, h, 64, 3, OUT_BOB422, mode, with_src=False)


def test_dirty_flag_rereads_the_frame():
    w, h = 96, 36
    p = L.make_params_tocomp(["-vhs", "-width", str(w)], output_height=h)
    frame_o = L.yuv_noise(w, h, 1, 0)
    frame_g = frame_o.copy()
    o = L.TocompOracleStream(p, oob=L.OOB_PLANE)
    ctx = Ctx(p)
    for vf in range(4):
        field = (vf & 1) ^ 1
        if vf == 2:          # the caller draws into its frame between two iterations
            for fr in (frame_o, frame_g):
                fr.pix(0)[10:20, 30:60] = 200
        o.process(frame_o, field, vf)
        t = ctx.submit(ctx.loop(frame_g, None, field, vf), _capi.SUBMIT422_DIRTY if vf == 2 else 0)
        ctx.wait(t)
    ctx.close()
    same_frames(frame_g, frame_o, "frame")


def test_full_size_stream_equals_sync_and_golden_hash():
    """720x480, the tool's default geometry, 40 fields at depth 32 through submit422 == the synchronous calls ==
    the oracle, whole buffers"""
    w, h = 720, 480
    p = L.make_params_tocomp(["-vhs"])
    run_loop(p, w, h, 48, 20, OUT_BOB420, "submit", depth=32)


@pytest.mark.parametrize("mode", ["sync", "submit"])
@pytest.mark.parametrize("flags,out_mode,bkey", [(["-vhs"], OUT_BOB422, False), ([], OUT_BOB422, False),
                                                  (["-bkey-feedback", "40"], OUT_BOB422, True)])
def test_pinned_planes_take_the_dma_and_kernel_delivery_paths(flags, out_mode, bkey, mode):
    """Frames whose planes can be pinned in place (page-aligned, >= 64 KiB: 720x480): frame / filter / encoder rows are
    written by the delivery kernels straight into them -- same
    bytes as the oracle's loop on identical buffers, whole buffers; with ONE persistent frame and 8 iterations per
    launch only the last writer of each row set delivers it."""
    w, h, pad, n_src = 720, 480, 16, 6
    p = L.make_params_tocomp(flags)
    srcs = [paged_noise(w, h, 300 + j) for j in range(n_src)]
    frame_o = paged_noise(w, h, 5, pad)
    random_padding(frame_o, 6)
    flt_o = paged_noise(w, h, 7, pad) if bkey else None
    frame_g, flt_g = frame_o.copy(), (flt_o.copy() if bkey else None)
    o = L.TocompOracleStream(p, oob=L.OOB_PLANE)
    ctx = Ctx(p, depth=8 if mode == "submit" else None)
    exp, got, tickets = [], [], []
    vf = 0
    for s_ in srcs:
        for sub in (0, 1):
            field = (vf & 1) ^ 1
            eo, go = PagedYuv(w, h, 0, 7), PagedYuv(w, h, 0, 7)
            oracle_iteration(o, p, frame_o, s_, field, vf, F_SECOND if sub else 0, flt_o, eo, out_mode, field)
            it = ctx.loop(frame_g, s_, field, vf, F_SECOND if sub else 0, flt_g, go, out_mode, field)
            if mode == "sync":
                ctx.field(it)
            else:
                tickets.append(ctx.submit(it, _capi.SUBMIT_SAME_SRC if sub else 0))
            exp.append(eo)
            got.append(go)
            vf += 1
    ctx.wait()
    st = ctx.stats()
    assert ctx.rng_pos == o.rng_pos
    assert st[6] >= 2 * n_src - (3 if mode == "submit" else 0)    # the delivery kernels wrote into the pinned planes
    # the synchronous call reads pinned source planes where they are (no snapshot, no upload); the asynchronous one snapshots
    assert st[2] == (0 if mode == "sync" else n_src)
    ctx.unpin()
    ctx.close()
    same_frames(frame_g, frame_o, "frame")
    if bkey:
        same_frames(flt_g, flt_o, "filter frame")
    for i, (a, b) in enumerate(zip(got, exp)):
        same_out(a, b, h, out_mode, "encoder frame %d" % i)


@pytest.mark.parametrize("src_flags,sh,spad", [(F_420, 486, 0), (F_IL | F_TFF, 480, 32), (0, 540, 16), (F_420 | F_IL, 480, 0)])
def test_synchronous_call_reads_pinned_sources_of_other_shapes_in_place(src_flags, sh, spad):
    """ntscsim_field422() on pinned source planes of the tool's other input shapes -- 4:2:0 chroma (half the rows),
    interlaced sources, a source taller than the frame, padded source rows: k422_render reads them with the caller's own
    linesizes (no snapshot, no upload: stats[2] stays 0), same bytes as the oracle's loop."""
    w, h, n_src = 720, 480, 3
    p = L.make_params_tocomp(["-vhs"])
    srcs = [paged_noise(w, sh, 500 + j, spad) for j in range(n_src)]
    for j, s_ in enumerate(srcs):
        random_padding(s_, 40 + j)
    frame_o = paged_noise(w, h, 5, 16)
    random_padding(frame_o, 6)
    frame_g = frame_o.copy()
    o = L.TocompOracleStream(p, oob=L.OOB_PLANE)
    ctx = Ctx(p)
    exp, got = [], []
    vf = 0
    for s_ in srcs:
        for sub in (0, 1):
            field = (vf & 1) ^ 1
            fl = src_flags | (F_SECOND if sub else 0)
            eo, go = PagedYuv(w, h, 0, 7), PagedYuv(w, h, 0, 7)
            oracle_iteration(o, p, frame_o, s_, field, vf, fl, None, eo, OUT_BOB422, field)
            ctx.field(ctx.loop(frame_g, s_, field, vf, fl, None, go, OUT_BOB422, field, sh=sh))
            exp.append(eo); got.append(go)
            vf += 1
    st = ctx.stats()
    assert ctx.rng_pos == o.rng_pos
    assert st[2] == 0, st
    ctx.unpin()
    ctx.close()
    same_frames(frame_g, frame_o, "frame")
    for i, (a, b) in enumerate(zip(got, exp)):
        same_out(a, b, h, OUT_BOB422, "encoder frame %d" % i)


def test_error_codes():
    w, h = 96, 36
    p = L.make_params_tocomp(["-vhs", "-width", str(w)], output_height=h)
    ctx = Ctx(p)
    fr = L.yuv_noise(w, h, 1, 8)
    it = ctx.loop(fr, None, 1, 0)
    lib = ctx.lib
    it.struct_size = 8
    assert lib.ntscsim_field422(ctx.h, C.byref(it)) == _capi.E_ARG
    it = ctx.loop(fr, None, 2, 0)
    assert lib.ntscsim_field422(ctx.h, C.byref(it)) == _capi.E_ARG
    it = ctx.loop(fr, None, 1, 0)
    it.width = 95
    assert lib.ntscsim_field422(ctx.h, C.byref(it)) == _capi.E_SIZE
    it = ctx.loop(fr, None, 1, 0)
    it.frame.linesize[0] = 50
    assert lib.ntscsim_field422(ctx.h, C.byref(it)) == _capi.E_SIZE
    it = ctx.loop(fr, L.yuv_noise(w, h, 2), 1, 0)
    it.src_height = 1
    assert lib.ntscsim_field422(ctx.h, C.byref(it)) == _capi.E_SIZE
    it = ctx.loop(fr, None, 1, 0, out=enc_frame(w, h, 0), out_mode=4)
    assert lib.ntscsim_field422(ctx.h, C.byref(it)) == _capi.E_ARG
    assert ctx.rng_pos == 0 and ctx.stats()[0] == 0          # refused calls consume nothing
    assert lib.ntscsim_wait(ctx.h, 5) == _capi.E_ARG         # a ticket never issued
    ctx.close()


# ---- the C++ host of INTEGRATION.md section 5: the tool's loop with its four calls replaced by one ---------------
import json      # noqa: E402
import os        # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402

LOOP = os.path.join(L.PKG, "field_loop422")


def run_cpp(mode, flags, fields=60, depth=8, extra=(), env=None):
    r = subprocess.run([LOOP, "--mode", mode, "--fields", str(fields), "--warmup", "0", "--hash", "1", "--depth", str(depth)] +
                       list(flags) + list(extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr.decode()
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


@pytest.mark.parametrize("flags,extra,batched", [
    (["-vhs"], [], True),                                   # 720 -> linesize 736: padded rows, batches of `depth`
    (["-vhs", "-422"], [], True),
    (["-vhs", "-vi"], [], True),                            # interlaced 4:2:0 repack after every pair
    (["-vhs", "-vi", "-422"], [], True),                    # the frame itself (:1158) as NTSCSIM_OUT422_FRAME
    ([], ["--height", "120"], True),                        # the tool's default preset
    (["-vhs", "-width", "704"], ["--height", "96"], True),  # linesize == width: batched, pad bytes chained on the device
    (["-vhs", "-width", "96"], ["--height", "96"], False),  # ... narrow and tight: in order, on the mirror
    (["-422", "-bkey-feedback", "40"], ["--height", "96"], False),
])
def test_cpp_loop_submit_equals_sync(flags, extra, batched):
    a = run_cpp("sync", flags, extra=extra)
    b = run_cpp("submit", flags, extra=extra)
    assert a["fnv1a"] == b["fnv1a"] != "0000000000000000" and a["rng_pos"] == b["rng_pos"]
    # ... and the staging rings (the path the oracle comparisons above go through) deliver what the pinned paths do
    c = run_cpp("submit", flags, extra=list(extra) + ["--page-frames", "1"], env={"NTSCSIM_SUBMIT422_PIN": "0"})
    assert c["fnv1a"] == b["fnv1a"] and c["rng_pos"] == b["rng_pos"] and c["stats"]["delivered_direct"] == 0
    # planes with a mapping of their own are pinned in place (full-size planes only: the engine leaves planes under 64 KiB alone)
    d = run_cpp("submit", flags, extra=list(extra) + ["--page-frames", "1"])
    assert d["fnv1a"] == b["fnv1a"] and d["rng_pos"] == b["rng_pos"]
    if "--height" not in extra:
        assert d["stats"]["delivered_direct"] > 0, d["stats"]
    # ... and the source snapshot by DMA out of the pinned planes (developer switch) is the same stream of frames
    e2 = run_cpp("submit", flags, extra=list(extra) + ["--page-frames", "1"], env={"NTSCSIM_SUBMIT422_SRCDMA": "1"})
    assert e2["fnv1a"] == b["fnv1a"] and e2["rng_pos"] == b["rng_pos"]
    assert b["stats"]["submitted"] == 60
    if batched:
        assert b["stats"]["batched"] == 60 and b["stats"]["launches"] <= 9, b["stats"]
    else:
        assert b["stats"]["one_at_a_time"] == 60, b["stats"]


def test_many_distinct_frames_with_feedback_outlive_the_mirror_table():
    """more than 64 distinct (frame, filter) pairs through the in-order path -- every call brings two new
    mirrors, so the engine's table (64 entries) is evicted again and again, also between the frame's mirror and the
    filter's of ONE iteration; every iteration must still run on its own two mirrors and equal the oracle."""
    w, h, pad = 96, 36, 0
    p = L.make_params_tocomp(["-bkey-feedback", "40", "-width", str(w)], output_height=h)
    o = L.TocompOracleStream(p, oob=L.OOB_PLANE)
    ctx = Ctx(p, depth=4)
    src = L.yuv_noise(w, h, 77)
    keep = []
    for vf in range(70):
        field = (vf & 1) ^ 1
        frame_o = L.yuv_noise(w, h, 1000 + vf, pad)
        flt_o = L.yuv_noise(w, h, 2000 + vf, pad)
        frame_g, flt_g = frame_o.copy(), flt_o.copy()
        oracle_iteration(o, p, frame_o, src, field, vf, 0, flt_o, None, 0, field)
        if vf % 3 == 0:
            ctx.field(ctx.loop(frame_g, src, field, vf, 0, flt_g))
        else:
            ctx.submit(ctx.loop(frame_g, src, field, vf, 0, flt_g))
        keep.append((frame_g, frame_o, flt_g, flt_o))
    ctx.wait()
    assert ctx.rng_pos == o.rng_pos and ctx.stats()[4] == 70
    ctx.close()
    for i, (a, b, c, d) in enumerate(keep):
        same_frames(a, b, "frame %d" % i)
        same_frames(c, d, "filter frame %d" % i)


def test_synchronous_call_allocates_two_ring_slots_and_a_submit_grows_them():
    """ntscsim_field422() used to allocate all 128 ring slots (GBs at 1080p and above); it now takes two, and
    the rings grow to the configured size on the first real submit.  A geometry whose two slots exceed the ring budget
    (NTSCSIM_SUBMIT422_RING_MB) is refused with NTSCSIM_E_SIZE instead of failing inside hipMalloc."""
    import torch
    w, h = 1920, 1080
    p = L.make_params_tocomp(["-vhs", "-width", str(w)], output_height=h)
    free0 = torch.cuda.mem_get_info()[0]
    ctx = Ctx(p)
    frame = L.yuv_noise(w, h, 3, 32)
    src = L.yuv_noise(w, h, 4)
    ctx.field(ctx.loop(frame, src, 1, 0))
    used_sync = free0 - torch.cuda.mem_get_info()[0]
    assert used_sync < 600 << 20, used_sync           # two slots + the kernels' scratch, not 128 slots (> 2 GB)
    ctx.submit(ctx.loop(frame, src, 0, 1, F_SECOND), _capi.SUBMIT_SAME_SRC)
    ctx.wait()
    used_async = free0 - torch.cuda.mem_get_info()[0]
    assert used_async > used_sync + (1 << 30), (used_sync, used_async)
    ctx.close()


def test_ring_budget_refuses_what_cannot_fit():
    code = r'''
import sys, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, %r)
import _libs as L
from ntscsim import _capi
import test_host422 as T
w, h = 720, 480
p = L.make_params_tocomp(["-vhs"])
ctx = T.Ctx(p)
it = ctx.loop(L.yuv_noise(w, h, 3, 32), L.yuv_noise(w, h, 4), 1, 0)
rc = ctx.lib.ntscsim_field422(ctx.h, C.byref(it))
assert rc == _capi.E_SIZE, rc
assert ctx.rng_pos == 0
ctx.close()
print("refused")
''' % (os.path.dirname(os.path.abspath(__file__)), L.PKG)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, NTSCSIM_SUBMIT422_RING_MB="2"))
    assert r.returncode == 0 and b"refused" in r.stdout, r.stderr.decode()[-1500:]


@pytest.mark.parametrize("flags", [["-vhs"], ["-vhs", "-422"]])
def test_cpp_loop_allocators_pinned_declared_pool_and_plain_malloc(flags):
    """The deterministic pinned paths: frames from ntscsim_host_frame_alloc() (what the AVFrame
    get_buffer helper uses) and frames carved from a pool declared with ntscsim_host_pin() are written by the GPU in
    place; ordinary malloc'ed frames -- whatever glibc's mmap threshold happens to be -- go through the staging rings
    and the engine's copy threads under the default policy; a foreign allocator whose block header happens to look like
    glibc's IS_MMAPPED word is NOT peeked at.  Same frames in the same order every way."""
    a = run_cpp("sync", flags)
    for alloc, direct in (("malloc", False), ("pinned", True), ("pool", True), ("fakehdr", False), ("mmap", True)):
        b = run_cpp("submit", flags, extra=["--alloc", alloc])
        assert b["alloc"] == alloc and b["fnv1a"] == a["fnv1a"] and b["rng_pos"] == a["rng_pos"], (alloc, b)
        assert (b["stats"]["delivered_direct"] > 0) == direct, (alloc, b["stats"])
        assert b["stats"]["batched"] == 60
    # round 5's arrangement stays available behind the opt-in policy
    c = run_cpp("submit", flags, extra=["--alloc", "malloc", "--pin-policy", "2", "--mmap-threshold", "65536"])
    assert c["fnv1a"] == a["fnv1a"] and c["stats"]["delivered_direct"] > 0
