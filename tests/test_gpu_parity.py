"""GPU parity tests: the HIP path, called through the C-ABI (ctypes), against the CPU oracle and
the committed golden vectors.  Bit-exact everywhere -- the whole path is integer pixels in/out
with fp64 filters evaluated in the reference's operation order (tolerance: 0)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import _libs as L
import cases
import ntscsim
from ntscsim import _capi, shard

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "ntsc_golden.npz"))
MANIFEST = json.load(open(os.path.join(HERE, "golden", "ntsc_golden.json")))["cases"]
FULL = json.load(open(os.path.join(HERE, "golden", "ntsc_fullsize_hashes.json")))["cases"]


def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def run_hip(p, srcs, jobs, h, w, il=0, tff=0, bob=False, dst_init=0, per_field_dst=False,
            rng_pos=None, sim=None):
    """jobs: (src index, field, fieldno).  Default: all fields land in ONE dst frame, like the
    reference's 1-frame delay ring.  Returns the dst frames as numpy."""
    torch = torch_mod()
    own = sim is None
    sim = sim or ntscsim.FieldSimulator(params=p)
    src = torch.from_numpy(np.stack(srcs)).cuda()
    nd = len(jobs) if per_field_dst else 1
    dst = torch.full((nd, h, w, 4), dst_init, dtype=torch.uint8, device="cuda")
    j4 = [(si, (k if per_field_dst else 0), field, fieldno) for k, (si, field, fieldno) in enumerate(jobs)]
    if per_field_dst or len(jobs) == 1:
        sim.fields(src, dst, j4, interlaced=il, tff=tff, bob=bob, rng_pos=rng_pos)
    else:
        # same dst frame: fields of opposite parity touch disjoint rows, same parity must be ordered
        # -> one call per field pair keeps the reference's overwrite order
        for i in range(0, len(j4), 2):
            sim.fields(src, dst, j4[i:i + 2], interlaced=il, tff=tff, bob=bob,
                       rng_pos=None if rng_pos is None else rng_pos[i:i + 2])
    sim.sync()
    out = dst.cpu().numpy()
    if own:
        sim.close()
    return out


@pytest.mark.parametrize("m", MANIFEST, ids=[m["name"] for m in MANIFEST])
def test_golden_vectors(m):
    name, w, h, n = m["name"], m["w"], m["h"], m["n"]
    p = L.make_params(m["flags"])
    srcs = [np.ascontiguousarray(a) for a in GOLD["%s__src" % name]]
    jobs = cases.case_jobs(n)
    per = run_hip(p, srcs, jobs, h, w, m["interlaced"], m["tff"], per_field_dst=True)
    for k, (si, field, fieldno) in enumerate(jobs):
        assert np.array_equal(per[k][field::2], GOLD["%s__field%d" % (name, k)]), "field %d" % k
        assert not per[k][1 - field::2].any(), "rows of the other field were written"
    one = run_hip(p, srcs, jobs, h, w, m["interlaced"], m["tff"])
    assert np.array_equal(one[0], GOLD["%s__final" % name])


@pytest.mark.parametrize("c", cases.CASES, ids=[c[0] for c in cases.CASES])
def test_case_matrix_vs_oracle_other_seeds(c):
    name, flags, w, h, n, kind, il, tff = c
    n += 3
    p = L.make_params(flags)
    srcs = [cases.make_source(kind, w, h, j + 11) for j in range((n + 1) // 2)]
    jobs = cases.case_jobs(n)
    o = L.OracleStream(p)
    exp = np.zeros((n, h, w, 4), np.uint8)
    for k, (si, field, fieldno) in enumerate(jobs):
        o.field(exp[k], srcs[si], field, fieldno, il, tff)
    got = run_hip(p, srcs, jobs, h, w, il, tff, per_field_dst=True)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("flags", [[], ["-vhs"], ["-vhs", "-comp-catv2"]])
def test_composite_signal_stage_tap(flags):
    """Stage-level parity: the composite plane between encoder and decoder."""
    torch = torch_mod()
    w, h, n = 96, 34, 4
    p = L.make_params(flags)
    srcs = [L.noise_frame(w, h, 5 + j) for j in range(2)]
    jobs = cases.case_jobs(n)
    o = L.OracleStream(p)
    taps = []
    for (si, field, fieldno) in jobs:
        taps.append(o.field(np.zeros((h, w, 4), np.uint8), srcs[si], field, fieldno,
                            taps=["composite_y"])["composite_y"])
    sim = ntscsim.FieldSimulator(params=p)
    src = torch.from_numpy(np.stack(srcs)).cuda()
    dst = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
    sim.fields(src, dst, [(si, k, f, fn) for k, (si, f, fn) in enumerate(jobs)])
    comp = sim.debug_composite(n, w, h)
    for k, (si, field, fieldno) in enumerate(jobs):
        assert np.array_equal(comp[k, :L.field_rows(h, field)], taps[k])
    sim.close()


@pytest.mark.parametrize("w,h,flags,nf", [(720, 480, [], 4), (720, 480, ["-vhs"], 6), (720, 486, [], 6),
                                          (720, 486, ["-vhs"], 6), (1920, 1080, ["-vhs"], 3),
                                          (1920, 1080, [], 2), (3840, 2160, ["-vhs"], 8)])
def test_full_size_vs_oracle(w, h, flags, nf):
    p = L.make_params(flags)
    srcs = [L.bars(w, h, j) if j % 2 == 0 else L.noise_frame(w, h, 77 + j) for j in range((nf + 1) // 2)]
    jobs = cases.case_jobs(nf)
    o = L.OracleStream(p)
    exp = np.zeros((nf, h, w, 4), np.uint8)
    for k, (si, field, fieldno) in enumerate(jobs):
        o.field(exp[k], srcs[si], field, fieldno)
    got = run_hip(p, srcs, jobs, h, w, per_field_dst=True)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("w,h,nf", [(720, 486, 6), (3840, 2160, 4)])
def test_full_size_ghosting_extension(w, h, nf):
    """BASELINE configs 2 / 5 name "multi-tap ghosting": the extension (absent from the reference, parity
    unpinned -- the oracle's own definition is the spec) at their sizes, four taps, `-vhs`: HIP == oracle,
    and the kernel chain is the preset's with k_ghost in between."""
    torch = torch_mod()
    p = L.make_params(["-vhs"])
    p.ghost_taps = 4
    for k, (d, g) in enumerate(((9, 80), (31, -40), (70, 24), (w // 5, -12))):
        p.ghost_delay[k] = d
        p.ghost_gain[k] = g
    srcs = [L.bars(w, h, j) if j % 2 == 0 else L.noise_frame(w, h, 177 + j) for j in range((nf + 1) // 2)]
    jobs = cases.case_jobs(nf)
    o = L.OracleStream(p)
    exp = np.zeros((nf, h, w, 4), np.uint8)
    for k, (si, field, fieldno) in enumerate(jobs):
        o.field(exp[k], srcs[si], field, fieldno)
    sim = ntscsim.FieldSimulator(params=p)
    got = run_hip(p, srcs, jobs, h, w, per_field_dst=True, sim=sim)
    ran = sim.last_kernels()
    sim.close()
    assert np.array_equal(got, exp)
    assert [k for k in ran if not k.startswith(("k_field", "k_row"))] == \
        ["k_encode_fast<double>", "k_ghost", "k_decode_fast<true,double>"], ran


def _ghost_params(flags, taps):
    p = L.make_params(flags)
    p.ghost_taps = len(taps)
    for k, (d, g) in enumerate(taps):
        p.ghost_delay[k] = d
        p.ghost_gain[k] = g
    return p


_GHOST_FUSED = [((12, 64),), ((12, 64), (31, -32)), ((1, 256), (63, -256)), ((7, 64), (19, -32), (40, 12)),
                ((1, -256), (2, 256), (62, 100), (63, -7)), ((5, 90), (5, 90), (33, -120), (33, 17))]


@pytest.mark.parametrize("flags", [[], ["-vhs"]], ids=["default", "vhs"])
@pytest.mark.parametrize("taps", _GHOST_FUSED, ids=lambda t: "-".join("%dx%d" % dg for dg in t))
def test_ghosting_folded_into_the_encoder(flags, taps):
    """Delays up to 63 samples: the encoder keeps the last 64 raw samples of every scanline in LDS and stores the
    ghosted plane itself (k_encode_fast_gh<RT, 2 | 4>), no k_ghost pass.  HIP == oracle; the same bytes with the
    fold switched off (ntscsim_debug_no_fast_decode bit 8: k_ghost for every delay); in FAST32 mode the two
    forms agree with each other (the ghost sum is integer work in both)."""
    w, h, n = 112, 40, 6        # 112 = 4 + 6 * 16 + 12: chunks, row start and row end all emit through the ring
    p = _ghost_params(flags, taps)
    srcs = [L.noise_frame(w, h, 301 + j) for j in range(3)]
    jobs = cases.case_jobs(n)
    o = L.OracleStream(p)
    exp = np.zeros((n, h, w, 4), np.uint8)
    for k, (si, field, fieldno) in enumerate(jobs):
        o.field(exp[k], srcs[si], field, fieldno)
    form = "2" if len(taps) <= 2 else "4"
    for mode, rt in ((_capi.MODE_EXACT, "double"), (_capi.MODE_FAST32, "float")):
        outs = []
        for fold in (True, False):
            sim = ntscsim.FieldSimulator(params=p)
            sim.set_mode(mode)
            if not fold:
                sim.debug_no_fast_decode(8)
            outs.append(run_hip(p, srcs, jobs, h, w, per_field_dst=True, sim=sim))
            ran = [k for k in sim.last_kernels() if k.startswith(("k_encode", "k_ghost"))]
            sim.close()
            assert ran == (["k_encode_fast_gh<%s,%s>" % (rt, form)] if fold else ["k_encode_fast<%s>" % rt, "k_ghost"]), ran
        assert np.array_equal(outs[0], outs[1]), rt
        if mode == _capi.MODE_EXACT:
            assert np.array_equal(outs[0], exp)


def test_ghosting_fold_preconditions():
    """A delay of 64 or more, an odd scanline phase, pre-emphasis: the pass of its own (k_ghost), same oracle."""
    w, h, n = 96, 32, 4
    for flags, taps in (([], ((12, 64), (64, -32))), (["-comp-phase", "90"], ((12, 64),)),
                        (["-comp-catv"], ((12, 64),))):
        p = _ghost_params(flags, taps)
        srcs = [L.noise_frame(w, h, 61 + j) for j in range(2)]
        jobs = cases.case_jobs(n)
        o = L.OracleStream(p)
        exp = np.zeros((n, h, w, 4), np.uint8)
        for k, (si, field, fieldno) in enumerate(jobs):
            o.field(exp[k], srcs[si], field, fieldno)
        sim = ntscsim.FieldSimulator(params=p)
        got = run_hip(p, srcs, jobs, h, w, per_field_dst=True, sim=sim)
        ran = sim.last_kernels()
        sim.close()
        assert np.array_equal(got, exp), flags
        assert "k_ghost" in ran and not any(k.startswith("k_encode_fast_gh") for k in ran), ran


@pytest.mark.parametrize("w,h,nf", [(720, 486, 6), (3840, 2160, 4)])
def test_full_size_ghosting_folded(w, h, nf):
    """The ghosting extension at BASELINE's sizes with the taps bench_side.py's vhs_ghost2 leg uses (12 and 31
    samples): folded into the encoder, HIP == oracle."""
    p = _ghost_params(["-vhs"], ((12, 64), (31, -32)))
    srcs = [L.bars(w, h, j) if j % 2 == 0 else L.noise_frame(w, h, 177 + j) for j in range((nf + 1) // 2)]
    jobs = cases.case_jobs(nf)
    o = L.OracleStream(p)
    exp = np.zeros((nf, h, w, 4), np.uint8)
    for k, (si, field, fieldno) in enumerate(jobs):
        o.field(exp[k], srcs[si], field, fieldno)
    sim = ntscsim.FieldSimulator(params=p)
    got = run_hip(p, srcs, jobs, h, w, per_field_dst=True, sim=sim)
    ran = sim.last_kernels()
    sim.close()
    assert np.array_equal(got, exp)
    assert [k for k in ran if not k.startswith(("k_field", "k_row"))] == \
        ["k_encode_fast_gh<double,2>", "k_decode_fast<true,double>"], ran


@pytest.mark.parametrize("c", FULL, ids=lambda c: "%dx%d%s" % (c["w"], c["h"], "".join(c["flags"])))
def test_full_size_reference_hashes(c):
    """Hashes recorded from the reference extract at the BASELINE sizes (incl. 3840x2160)."""
    w, h, n = c["w"], c["h"], c["n"]
    p = L.make_params(c["flags"])
    srcs = [L.bars(w, h, j) for j in range((n + 1) // 2)]
    torch = torch_mod()
    sim = ntscsim.FieldSimulator(params=p)
    src = torch.from_numpy(np.stack(srcs)).cuda()
    dst = torch.zeros((1, h, w, 4), dtype=torch.uint8, device="cuda")
    for k in range(n):
        sim.fields(src, dst, [(k // 2, 0, (k & 1) ^ 1, k)])
        sim.sync()
        assert "%016x" % L.fnv1a(dst[0].cpu().numpy()) == c["fnv1a_after_each_field"][k], "field %d" % k
    sim.close()


def test_host_frame_dropin_sequence():
    """ntscsim_field(): the composite_layer() drop-in on host buffers with linesize padding,
    rand() stream carried across calls like the reference's process-wide generator."""
    w, h = 96, 32
    p = L.make_params(["-vhs"])
    sim = ntscsim.FieldSimulator(params=p)
    o = L.OracleStream(p)
    pad = 24
    src_p = np.zeros((h, w + pad, 4), np.uint8)
    dst_p = np.full((h, w + pad, 4), 0x5A, np.uint8)
    exp = np.full((h, w, 4), 0x5A, np.uint8)
    for k in range(5):
        s = L.noise_frame(w, h, 40 + k // 2)
        src_p[:, :w] = s
        field = (k & 1) ^ 1
        sim.field_host(dst_p[:, :w], src_p[:, :w], field, k)
        o.field(exp, s, field, k)
        assert np.array_equal(dst_p[:, :w], exp), "call %d" % k
        assert (dst_p[:, w:] == 0x5A).all(), "padding bytes were written"
        assert sim.rng_pos == o.rng_pos
    sim.close()


_PIPE5, _PIPE3, _PIPE5W, _PIPE5S = "k_field_pipe<double>", "k_field_pipe_tv<double>", "k_field_pipe<double,true>", "k_field_pipe_sv<double>"


@pytest.mark.parametrize("flags,w,h,piped", [
    (["-vhs"], 720, 486, _PIPE5), (["-vhs"], 720, 480, _PIPE5), (["-vhs", "-vhs-speed", "lp"], 360, 243, _PIPE5),
    (["-vhs", "-vhs-speed", "ep"], 1920, 1080, _PIPE5), (["-vhs"], 3840, 2160, _PIPE5),
    (["-vhs"], 16, 4, _PIPE5), (["-vhs"], 20, 9, _PIPE5), (["-vhs"], 36, 130, _PIPE5), (["-vhs"], 100, 7, _PIPE5),
    (["-vhs", "-vhs-head-switching-point", "0.8"], 360, 244, _PIPE5), (["-vhs", "-chroma-dropout", "30000"], 256, 100, _PIPE5),
    (["-vhs", "-vhs-chroma-vblend", "0"], 256, 100, _PIPE5), (["-vhs", "-noise", "40", "-chroma-noise", "70"], 256, 100, _PIPE5),
    (["-vhs"], 33, 17, _PIPE5),                             # (tight host rows: the device frames have aligned rows of their own)
    # the default preset (no VCR): three roles
    ([], 720, 486, _PIPE3), ([], 256, 100, _PIPE3), ([], 16, 4, _PIPE3), ([], 21, 9, _PIPE3), ([], 1920, 1080, _PIPE3),
    (["-noise", "30", "-chroma-dropout", "20000"], 360, 243, _PIPE3), (["-tvstd", "pal"], 720, 576, _PIPE3),
    (["-vhs-head-switching", "1"], 256, 100, _PIPE3),       # the head switch without the VCR: displaced loads in the TV front
    # head-switch displacements beyond W/10 (wrap-around loads; a workgroup whose rows reach back to the row's end runs
    # its encoder first): PAL's default switching point, a large forward and a large backward displacement
    (["-vhs", "-tvstd", "pal"], 720, 576, _PIPE5W),
    (["-vhs", "-vhs-head-switching-phase", "0.001"], 720, 486, _PIPE5W),
    (["-vhs", "-vhs-head-switching-phase", "0.003"], 720, 486, _PIPE5W), (["-vhs", "-vhs-head-switching-phase", "0.003"], 100, 60, _PIPE5W),
    # S-Video out of the VCR: no re-modulation, no second separation, 7 positions less deep
    (["-vhs", "-vhs-svideo", "1"], 256, 100, _PIPE5S), (["-vhs", "-vhs-svideo", "1"], 720, 486, _PIPE5S),
    (["-vhs", "-vhs-svideo", "1", "-vhs-speed", "ep", "-chroma-dropout", "30000"], 360, 243, _PIPE5S),
    (["-vhs", "-vhs-svideo", "1", "-tvstd", "pal"], 720, 576, _PIPE5S), (["-vhs", "-vhs-svideo", "1"], 20, 9, _PIPE5S),
    # scanline phases of either parity: per-lane picks / signs / carrier roles in every role
    (["-vhs", "-comp-phase", "90"], 256, 100, "k_field_pipe_xi<double>"), (["-vhs", "-comp-phase", "270"], 720, 486, "k_field_pipe_xi<double>"),
    (["-vhs", "-comp-phase-offset", "1"], 360, 243, "k_field_pipe_xi<double>"), (["-vhs", "-comp-phase", "90", "-tvstd", "pal"], 720, 576, "k_field_pipe_xi<double>"),
    (["-vhs", "-comp-phase", "90"], 21, 9, "k_field_pipe_xi<double>"),
    (["-vhs", "-comp-phase", "90", "-vhs-svideo", "1"], 256, 100, None),
    # the pre-emphasis presets with the VCR: composite pre-emphasis in the encoder role, 50 / subcarrier_amplitude_back in
    # the first separator (k_encode_fast_pre / k_decode_fast_bk's forms)
    (["-vhs", "-comp-catv"], 256, 100, "k_field_pipe_catv<double>"), (["-vhs", "-comp-catv2"], 720, 486, "k_field_pipe_catv<double>"),
    (["-vhs", "-comp-catv3", "-tvstd", "pal"], 720, 576, "k_field_pipe_catv<double>"), (["-vhs", "-comp-catv4"], 360, 243, "k_field_pipe_catv<double>"),
    (["-vhs", "-comp-catv"], 21, 9, "k_field_pipe_catv<double>"), (["-vhs", "-comp-catv3", "-vhs-speed", "ep"], 1920, 1080, "k_field_pipe_catv<double>"),
    # ... and without it: three roles, the TV front with the amplitude scale and the presets' chroma phase noise
    (["-comp-catv"], 256, 100, "k_field_pipe_tv_catv<double>"), (["-comp-catv2"], 720, 486, "k_field_pipe_tv_catv<double>"),
    (["-comp-catv3", "-tvstd", "pal"], 720, 576, "k_field_pipe_tv_catv<double>"), (["-comp-catv4", "-chroma-phase-noise", "0"], 360, 243, "k_field_pipe_tv_catv<double>"),
    (["-comp-catv", "-chroma-dropout", "20000", "-vhs-head-switching", "1"], 21, 9, "k_field_pipe_tv_catv<double>"),
    (["-comp-catv2"], 1920, 1080, "k_field_pipe_tv_catv<double>"),
    (["-comp-catv", "-chroma-noise", "5"], 256, 100, None),
])
def test_synchronous_call_takes_the_pipelined_form_and_equals_the_oracle(flags, w, h, piped):
    """ntscsim_field(): launches of up to 64 fields from the host-frame entry points run the chain as ROLES of one
    workgroup (csrc/ntsc_pipe.hip): five wavefronts for the -vhs preset family (k_field_pipe: encoder | VCR chroma front |
    VCR chroma back | VCR luma + TV separator | TV output), three for the default preset (k_field_pipe_tv) -- the
    arithmetic of the one-launch kernels cut where only integers cross, handed over through LDS rings -- bit for bit
    the oracle's, call after call (the rand() stream carried like the reference's); other switch sets keep their forms."""
    p = L.make_params(flags, output_height=h)
    sim = ntscsim.FieldSimulator(params=p)
    o = L.OracleStream(p)
    got = np.full((h, w, 4), 0x5A, np.uint8)
    exp = np.full((h, w, 4), 0x5A, np.uint8)
    for k in range(4):
        s = L.noise_frame(w, h, 90 + k // 2)
        field = (k & 1) ^ 1
        sim.field_host(got, s, field, k)
        o.field(exp, s, field, k)
        assert np.array_equal(got, exp), "call %d" % k
        assert sim.rng_pos == o.rng_pos
        kern = sim.last_kernels()
        assert [x for x in kern if x.startswith("k_field_pipe")] == ([piped] if piped else []), kern
    sim.close()


@pytest.mark.parametrize("flags,n,form", [(["-vhs"], 2, "k_field_pipe<double>"), (["-vhs"], 64, "k_field_pipe<double>"), ([], 6, "k_field_pipe_tv<double>"),
                                          (["-vhs", "-comp-catv"], 4, "k_field_pipe_catv<double>"), (["-vhs"], 65, None),
                                          (["-vhs", "-nocolor-subcarrier"], 4, None)])
def test_device_resident_launches_take_the_latency_form_when_asked(flags, n, form):
    """ntscsim_set_launch_form(NTSCSIM_FORM_LATENCY): ntscsim_fields_device() launches of up to 64 device-resident fields
    run as wavefront roles (0.17 ms against 0.42 ms for a launch of up to 32 fields); longer launches and switch sets the
    role kernels do not cover keep the throughput form; the default (NTSCSIM_FORM_THROUGHPUT) never takes them.  Same bytes
    as the oracle in every case."""
    import torch
    w, h = 256, 100
    p = L.make_params(flags)
    srcs = np.stack([L.noise_frame(w, h, 60 + j) for j in range(3)])
    src = torch.from_numpy(srcs).cuda()
    jobs = [((k // 2) % 3, k, (k & 1) ^ 1, k) for k in range(n)]
    o = L.OracleStream(p)
    exp = np.zeros((n, h, w, 4), np.uint8)
    for (si, di, field, fieldno) in jobs:
        o.field(exp[di], srcs[si], field, fieldno)
    for latency in (True, False):
        sim = ntscsim.FieldSimulator(params=p)
        if latency:
            sim.set_launch_form(True)
        dst = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
        sim.fields(src, dst, jobs)
        sim.sync()
        ran = [x for x in sim.last_kernels() if x.startswith("k_field_pipe")]
        assert ran == ([form] if (latency and form) else []), sim.last_kernels()
        assert np.array_equal(dst.cpu().numpy(), exp)
        assert sim.rng_pos == o.rng_pos
        sim.close()


def test_the_setup_kernel_launched_ahead_is_used_only_by_the_call_it_was_made_for():
    """ntscsim_field() launches the NEXT call's setup kernel behind its own work (speculate_setup: same switches and
    geometry, the field parity continuing the pattern, the rand() stream where the call left it); the next call skips its
    own launch only when everything that kernel read is what the call would hand it.  Sequences that break the
    prediction -- the same field twice, a stream position set by the caller, a batch on the device entry point in between,
    another geometry -- equal the oracle's call after call."""
    import torch
    w, h = 256, 100
    p = L.make_params(["-vhs"])
    sim = ntscsim.FieldSimulator(params=p)
    o = L.OracleStream(p)
    got = np.full((h, w, 4), 0x5A, np.uint8)
    exp = np.full((h, w, 4), 0x5A, np.uint8)
    s = L.noise_frame(w, h, 7)

    def call(field, k):
        sim.field_host(got, s, field, k)
        o.field(exp, s, field, k)
        assert np.array_equal(got, exp), "call %d" % k
        assert sim.rng_pos == o.rng_pos

    k = 0
    for field in (1, 0, 1, 0, 0, 0, 1, 1, 0, 1):                # alternating (predicted), repeats (mispredicted, then predicted)
        call(field, k); k += 1
    o.skip(12345); sim.rng_pos = sim.rng_pos + 12345           # the caller moves the stream: the early kernel's draws are stale
    call(1, k); k += 1
    call(0, k); k += 1
    # a batch through the device entry point rewrites the tables between two synchronous calls
    src_t = torch.from_numpy(np.stack([s])).cuda()
    dst_t = torch.zeros((2, h, w, 4), dtype=torch.uint8, device="cuda")
    sim.fields(src_t, dst_t, [(0, 0, 1, k), (0, 1, 0, k + 1)])
    sim.sync()
    e2 = np.zeros((2, h, w, 4), np.uint8)
    o.field(e2[0], s, 1, k); o.field(e2[1], s, 0, k + 1)
    assert np.array_equal(dst_t.cpu().numpy(), e2)
    k += 2
    call(1, k); k += 1
    call(0, k); k += 1
    # another geometry in between (other tables, other DevParams), then back
    s2 = L.noise_frame(64, 40, 9)
    g2 = np.zeros((40, 64, 4), np.uint8); x2 = np.zeros((40, 64, 4), np.uint8)
    sim.field_host(g2, s2, 1, k); o.field(x2, s2, 1, k); k += 1
    assert np.array_equal(g2, x2)
    call(0, k); k += 1
    call(1, k)
    sim.close()


def test_a_launch_of_more_workgroups_than_the_chip_holds_equals_small_launches():
    """700 fields (2,700 workgroups for 2,048 two-wave slots) in ONE launch == the same fields in launches of 50, byte for
    byte: a workgroup that starts after its neighbours have finished must not see anything of theirs (the YUV422P tool's
    in-place kernels did, through the halo row: tests/test_variant422.py; the BGRA tool reads frames it never writes)."""
    import torch
    w, h, n = 720, 486, 700
    p = L.make_params(["-vhs"])
    src = torch.from_numpy(np.stack([L.noise_frame(w, h, 40 + j) for j in range(4)])).cuda()
    jobs = [((k // 2) % 4, k, (k & 1) ^ 1, k) for k in range(n)]
    outs = []
    for batch in (50, n):
        sim = ntscsim.FieldSimulator(params=p)
        dst = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
        for a in range(0, n, batch):
            sim.fields(src, dst, jobs[a:a + batch])
        sim.sync()
        sim.close()
        outs.append(dst)
    assert torch.equal(outs[0], outs[1])


def test_a_hand_off_that_never_comes_is_an_error_not_a_hang():
    """The role kernels poll each other's counts in LDS.  A wait that can never be satisfied (here: the developer switch
    NTSCSIM_PIPE_ORDER leaves the chroma-back role out, so its ring is never drained) gives up after a bounded number of
    rounds, the workgroup's other waits fall through, and the call returns NTSCSIM_E_HIP instead of hanging the GPU."""
    import subprocess, sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, _libs as L, ntscsim\n"
        "p = L.make_params(['-vhs'], output_height=100)\n"
        "sim = ntscsim.FieldSimulator(params=p)\n"
        "s = L.noise_frame(256, 100, 1); d = np.zeros((100, 256, 4), np.uint8)\n"
        "try:\n"
        "    sim.field_host(d, s, 1, 0)\n"
        "    print('NO ERROR')\n"
        "except ntscsim.NtscsimError as e:\n"
        "    print('ERROR', e)\n"
    ) % (os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "composite-video-simulator_amd"))
    env = dict(os.environ, NTSCSIM_PIPE_ORDER="24001")      # wavefront 2 runs a second encoder instead of the chroma-back role
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    out = r.stdout.decode()
    assert "ERROR" in out and "hand-off timed out" in out, (out, r.stderr.decode()[-400:])


@pytest.mark.parametrize("flags,w,h,off", [(["-vhs"], 720, 486, 0), (["-vhs"], 720, 486, 4), (["-vhs"], 100, 37, 0),
                                           ([], 256, 100, 0), (["-vhs", "-vhs-svideo", "1"], 360, 243, 16)])
def test_synchronous_call_writes_pinned_destination_frames_in_place(flags, w, h, off):
    """ntscsim_field() on a destination frame the GPU can address (ntscsim_host_alloc / ntscsim_host_pin; 16-byte aligned
    rows): the kernels store the field's rows into the caller's frame themselves -- no device copy, no download.  Same
    bytes as the oracle, the other field's rows and the bytes around the frame untouched; frames that are pinned but
    not 16-byte aligned (off = 4) go through the device copy."""
    from ntscsim import host_alloc_array, host_free_array
    p = L.make_params(flags, output_height=h)
    sim = ntscsim.FieldSimulator(params=p)
    o = L.OracleStream(p)
    block = host_alloc_array((h * w * 4 + 64 + off,))
    block[:] = 0xA7
    got = block[32 + off:32 + off + h * w * 4].reshape(h, w, 4)
    got[:] = 0x5A
    exp = np.full((h, w, 4), 0x5A, np.uint8)
    for k in range(4):
        s = L.noise_frame(w, h, 40 + k // 2)
        field = (k & 1) ^ 1
        sim.field_host(got, s, field, k)
        o.field(exp, s, field, k)
        assert np.array_equal(got, exp), "call %d" % k
        assert sim.rng_pos == o.rng_pos
    assert (block[:32 + off] == 0xA7).all() and (block[32 + off + h * w * 4:] == 0xA7).all()
    # declared memory (ntscsim_host_pin) takes the same path -- a mapping of its own: pages of the brk heap (what a small
    # numpy array is made of) are refused, a registration there makes the GPU fault sooner or later
    import mmap
    libc = ntscsim.C.CDLL(None)
    libc.sbrk.restype, libc.sbrk.argtypes = ntscsim.C.c_void_p, [ntscsim.C.c_long]
    small = np.zeros((3000,), np.uint8)
    if small.ctypes.data < libc.sbrk(0):          # (a small block of the brk heap, as expected of the allocator)
        with pytest.raises(ntscsim.NtscsimError):
            sim.host_pin(small)
    mm = mmap.mmap(-1, h * w * 4 + 8192)
    frame = np.frombuffer(mm, np.uint8)
    a0 = (-frame.ctypes.data) % 4096
    got2 = frame[a0:a0 + h * w * 4].reshape(h, w, 4)
    sim.host_pin(frame[a0:a0 + ((h * w * 4 + 4095) // 4096) * 4096])
    exp2 = np.zeros((h, w, 4), np.uint8)
    for k in range(4, 6):
        s = L.noise_frame(w, h, 40 + k // 2)
        sim.field_host(got2, s, (k & 1) ^ 1, k)
        o.field(exp2, s, (k & 1) ^ 1, k)
        assert np.array_equal(got2, exp2), "declared, call %d" % k
    sim.close()
    host_free_array(block)


@pytest.mark.parametrize("flags,w,h,pad,off,il,tff,direct", [
    (["-vhs"], 720, 486, 0, 0, 0, 0, True), ([], 256, 100, 64, 16, 1, 1, True), (["-vhs", "-comp-catv"], 100, 7, 16, 0, 1, 0, True),
    (["-vhs"], 20, 9, 0, 0, 0, 0, True), (["-vhs"], 36, 130, 0, 32, 0, 0, True), (["-vhs", "-vhs-svideo", "1"], 333, 65, 8, 0, 0, 0, False),
    (["-vhs"], 256, 100, 0, 4, 0, 0, False), (["-noise", "0"], 64, 40, 0, 0, 0, 0, True)])
def test_synchronous_call_reads_pinned_source_frames_in_place(flags, w, h, pad, off, il, tff, direct):
    """ntscsim_field() on a SOURCE frame the GPU can address (pinned, 16-byte aligned rows): nothing is uploaded -- the encoder
    reads the caller's rows over the link, with the pixels of its guarded steps (the row's first four, the up to 15 behind
    its last whole chunk) requested at the start; widths with every remainder, padded rows, interlaced sources.  Same bytes as
    the oracle; rows that are not 16-byte aligned (pad = 8 at width 333, off = 4) take the upload.  The bytes of the frame are never
    written."""
    from ntscsim import host_alloc_array, host_free_array
    p = L.make_params(flags, output_height=h)
    sim = ntscsim.FieldSimulator(params=p)
    o = L.OracleStream(p)
    rowb = w * 4 + pad
    block = host_alloc_array((rowb * h + 64 + off,))
    block[:] = 0xC3
    base = block[off:off + rowb * h]
    srcv = np.lib.stride_tricks.as_strided(base, shape=(h, w, 4), strides=(rowb, 4, 1))
    got = np.full((h, w, 4), 0x5A, np.uint8)
    exp = np.full((h, w, 4), 0x5A, np.uint8)
    u8p = ntscsim.C.POINTER(ntscsim.C.c_uint8)
    for k in range(4):
        s = L.noise_frame(w, h, 70 + k // 2)
        srcv[:] = s
        keep = block.copy()
        field = (k & 1) ^ 1
        rc = sim._lib.ntscsim_field(sim._h, srcv.ctypes.data_as(u8p), rowb, il, tff, got.ctypes.data_as(u8p), w * 4, w, h, field, k)
        sim._chk(rc, "ntscsim_field")
        o.field(exp, s, field, k, interlaced=il, tff=tff)
        assert np.array_equal(got, exp), "call %d" % k
        assert sim.rng_pos == o.rng_pos
        assert np.array_equal(block, keep), "the source frame was written"
    st = sim.debug_field_stats()
    assert st[0] == 4 and st[1] == (4 if direct else 0), st
    sim.close()
    host_free_array(block)


def test_synchronous_call_on_one_pinned_frame_as_source_and_destination():
    """composite_layer() reads its whole field before it writes a pixel (:1585-1609), so the tool may be handed ONE frame as
    source and destination.  A pinned frame would qualify for being read AND written in place at once -- the encoder would
    meet the decoder's rows: frames that share bytes take the upload (the snapshot), and the result is the oracle's."""
    from ntscsim import host_alloc_array, host_free_array
    w, h = 256, 100
    p = L.make_params(["-vhs"])
    sim = ntscsim.FieldSimulator(params=p)
    o = L.OracleStream(p)
    block = host_alloc_array((h * w * 4,))
    frame = block.reshape(h, w, 4)
    frame[:] = L.noise_frame(w, h, 11)
    exp = frame.copy()
    for k in range(4):
        field = (k & 1) ^ 1
        src_copy = exp.copy()
        o.field(exp, src_copy, field, k)
        sim.field_host(frame, frame, field, k)
        assert np.array_equal(frame, exp), "call %d" % k
    st = sim.debug_field_stats()
    assert st[1] == 0 and st[2] == 4, st
    sim.close()
    host_free_array(block)


@pytest.mark.parametrize("h", [2, 3, 32, 33])
@pytest.mark.parametrize("il,tff", [(0, 0), (1, 0), (1, 1)])
def test_host_frame_dropin_uploads_the_rows_it_reads(h, il, tff):
    """ntscsim_field() sends up only the source rows the field reads -- row min(y + opposite, H - 1) for the field's y
    (ffmpeg_ntsc.cpp:1585-1588, :1599) -- incl. the last row clamped onto H - 1 (odd / even heights, interlaced
    sources of either field order).  The product never reads the rows in between: they are poisoned in the frame it
    gets, the oracle runs on the clean one."""
    w = 64
    p = L.make_params(["-vhs"])
    sim = ntscsim.FieldSimulator(params=p)
    o = L.OracleStream(p)
    dst = np.full((h, w, 4), 0x5A, np.uint8)
    exp = np.full((h, w, 4), 0x5A, np.uint8)
    for k in range(6):
        s = L.noise_frame(w, h, 140 + k)
        field = (k & 1) ^ 1
        o.field(exp, s, field, k, interlaced=il, tff=tff)
        # rows this field does not read: poisoned for the product
        opposite = (1 if tff else 0) if il else 0
        read = {min(y + opposite, h - 1) for y in range(field, h, 2)}
        sp = s.copy()
        for y in range(h):
            if y not in read:
                sp[y] = 0xA7
        sim.field_host(dst, sp, field, k, interlaced=il, tff=tff)
        assert np.array_equal(dst, exp), "call %d" % k
        assert sim.rng_pos == o.rng_pos
    sim.close()


def test_batch_split_and_order_invariance():
    """Size-independent property at the BASELINE size: a 64-field batch == 64 single-field
    calls == two half batches issued in reverse order (explicit rand() positions)."""
    torch = torch_mod()
    w, h, nf = 720, 486, 64
    p = L.make_params(["-vhs"])
    src = torch.from_numpy(np.stack([L.bars(w, h, j) for j in range(nf // 2)])).cuda()
    jobs = [(k // 2, k, (k & 1) ^ 1, k) for k in range(nf)]
    pos = [shard.rng_pos_of_field(p, w, h, k) for k in range(nf)]
    sim = ntscsim.FieldSimulator(params=p)
    a = torch.zeros((nf, h, w, 4), dtype=torch.uint8, device="cuda")
    sim.fields(src, a, jobs, rng_pos=pos)
    b = torch.zeros_like(a)
    sim.fields(src, b, jobs[nf // 2:], rng_pos=pos[nf // 2:])
    sim.fields(src, b, jobs[:nf // 2], rng_pos=pos[:nf // 2])
    c = torch.zeros_like(a)
    sim.rng_pos = 0
    for j in jobs:
        sim.fields(src, c, [j])            # RNG_AUTO continues from the ctx position
    sim.sync()
    assert torch.equal(a, b) and torch.equal(a, c)
    # spot-check three fields of the batch against the oracle
    o = L.OracleStream(p)
    for k in (0, 1, 2):
        e = np.zeros((h, w, 4), np.uint8)
        o.field(e, src[k // 2].cpu().numpy(), (k & 1) ^ 1, k)
        assert np.array_equal(a[k].cpu().numpy(), e)
    sim.close()


def test_bench_clip_checksum_of_checksums():
    """The bench workload itself (600 fields, 720x486, -vhs): every 37th field vs the oracle at
    its closed-form stream position, plus determinism of the whole batch."""
    torch = torch_mod()
    w, h, nf = 720, 486, 600
    p = L.make_params(["-vhs"])
    frames = np.stack([L.bars(w, h, j) for j in range(nf // 2)])
    src = torch.from_numpy(frames).cuda()
    jobs = [(k // 2, k // 2, (k & 1) ^ 1, k) for k in range(nf)]
    pos = [shard.rng_pos_of_field(p, w, h, k) for k in range(nf)]
    sim = ntscsim.FieldSimulator(params=p)
    d1 = torch.zeros((nf // 2, h, w, 4), dtype=torch.uint8, device="cuda")
    d2 = torch.zeros_like(d1)
    sim.fields(src, d1, jobs, rng_pos=pos)
    sim.fields(src, d2, jobs, rng_pos=pos)
    sim.sync()
    assert torch.equal(d1, d2)
    got = d1.cpu().numpy()
    for k in range(0, nf, 37):
        o = L.OracleStream(p)
        o.skip(pos[k])
        e = np.zeros((h, w, 4), np.uint8)
        field = (k & 1) ^ 1
        o.field(e, frames[k // 2], field, k)
        assert np.array_equal(got[k // 2][field::2], e[field::2]), "field %d" % k
    sim.close()


def test_prepared_batch_equals_direct_call():
    """ntscsim_batch_create/run: same bytes as ntscsim_fields_device, re-runnable, and two
    geometries can alternate on one ctx."""
    torch = torch_mod()
    p = L.make_params(["-vhs"])
    sim = ntscsim.FieldSimulator(params=p)
    outs = {}
    plans = {}
    for (w, h) in ((96, 32), (100, 35)):
        n = 6
        src = torch.from_numpy(np.stack([L.noise_frame(w, h, 50 + j) for j in range(3)])).cuda()
        jobs = [(k // 2, k, (k & 1) ^ 1, k) for k in range(n)]
        pos = [shard.rng_pos_of_field(p, w, h, k) for k in range(n)]
        a = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
        b = torch.zeros_like(a)
        sim.fields(src, a, jobs, rng_pos=pos)
        d = sim.build_descs(src, b, jobs, rng_pos=pos)
        plans[(w, h)] = (sim.prepare(d, w, h), d, src, a, b)
    for _ in range(2):
        for key, (pl, d, src, a, b) in plans.items():
            b.zero_()
            sim.run_prepared(pl)
            sim.sync()
            assert torch.equal(a, b), key
    for pl, *_ in plans.values():
        sim.free_prepared(pl)
    sim.close()


def test_warmup_fallback_path_is_exact():
    """Force the noise-accumulator warm-up to be too short so lanes take the serial-replay
    fallback; results must not change."""
    w, h, n = 96, 32, 4
    p = L.make_params(["-vhs", "-noise", "40", "-chroma-noise", "60"])
    srcs = [L.noise_frame(w, h, 9 + j) for j in range(2)]
    jobs = cases.case_jobs(n)
    ref = run_hip(p, srcs, jobs, h, w, per_field_dst=True)
    sim = ntscsim.FieldSimulator(params=p)
    sim.debug_set_warmup(1, 2)
    got = run_hip(p, srcs, jobs, h, w, per_field_dst=True, sim=sim)
    sim.close()
    o = L.OracleStream(p)
    exp = np.zeros((n, h, w, 4), np.uint8)
    for k, (si, field, fieldno) in enumerate(jobs):
        o.field(exp[k], srcs[si], field, fieldno)
    assert np.array_equal(ref, exp) and np.array_equal(got, exp)


@pytest.mark.parametrize("flags,w,h", [([], 96, 32), (["-vhs"], 96, 32), (["-vhs"], 720, 486),
                                       (["-vhs", "-vhs-speed", "lp"], 100, 33)])
def test_preset_and_generic_kernels_agree(flags, w, h):
    """The compile-time specialised kernels and the run-time-option kernels are the same function."""
    n = 4
    p = L.make_params(flags)
    srcs = [L.noise_frame(w, h, 31 + j) for j in range(2)]
    jobs = cases.case_jobs(n)
    a = run_hip(p, srcs, jobs, h, w, per_field_dst=True)
    sim = ntscsim.FieldSimulator(params=p)
    sim.debug_force_generic(True)
    b = run_hip(p, srcs, jobs, h, w, per_field_dst=True, sim=sim)
    sim.close()
    assert np.array_equal(a, b)
    o = L.OracleStream(p)
    e = np.zeros((n, h, w, 4), np.uint8)
    for k, (si, field, fieldno) in enumerate(jobs):
        o.field(e[k], srcs[si], field, fieldno)
    assert np.array_equal(a, e)


@pytest.mark.parametrize("flags,w,h,fast_ok", [
    ([], 96, 32, 1), ([], 720, 486, 1), (["-vhs"], 96, 32, 1), (["-vhs"], 720, 486, 1), (["-vhs"], 101, 35, 1),
    (["-vhs", "-vhs-speed", "lp"], 100, 33, 1), (["-vhs", "-vhs-speed", "ep"], 128, 40, 1),
    (["-vhs", "-vhs-chroma-vblend", "0"], 96, 32, 1),
    (["-vhs", "-chroma-dropout", "50000"], 96, 32, 1), (["-vhs", "-comp-phase-offset", "2"], 96, 32, 1),
    (["-comp-phase", "0", "-comp-phase-offset", "2"], 96, 32, 1),
    # head-switch displacement beyond W/10 samples (PAL: 312.5 lines per field put the default switching
    # point 18 samples into a 96-sample row; the other two by their switches): the wrap-around form
    (["-tvstd", "pal", "-vhs"], 96, 36, 2), (["-tvstd", "pal", "-vhs"], 720, 576, 2),
    (["-vhs", "-vhs-head-switching-point", "0.105", "-vhs-head-switching-phase", "0.0012"], 96, 32, 2),
    (["-vhs", "-vhs-head-switching-point", "0.105", "-vhs-head-switching-phase", "0.002"], 96, 32, 2),
    # scanline phases of either parity: the any-phase forms k_encode_fast_xi + k_decode_fast_xi (per-lane picks, signs
    # and carrier roles); all four phases occur among the rows of these cases
    (["-vhs", "-comp-phase-offset", "1"], 96, 32, 5), (["-vhs", "-comp-phase", "90"], 96, 32, 5),
    (["-vhs", "-comp-phase", "270", "-comp-phase-offset", "3"], 100, 35, 5), (["-vhs", "-comp-phase", "90"], 720, 486, 5),
    (["-vhs", "-comp-phase", "0", "-comp-phase-offset", "3", "-vhs-speed", "lp"], 96, 32, 5),
    (["-tvstd", "pal", "-vhs", "-comp-phase", "90", "-chroma-dropout", "30000"], 96, 36, 5),
    # the FULL output chroma low-pass behind the VCR (I 2 samples back, Q 4): k_decode_fast_fo
    (["-vhs", "-out-composite-lowpass-lite", "0"], 96, 32, 7), (["-vhs", "-out-composite-lowpass-lite", "0"], 720, 486, 7),
    (["-vhs", "-out-composite-lowpass-lite", "0", "-vhs-speed", "lp"], 100, 35, 7),
    (["-vhs", "-out-composite-lowpass-lite", "0", "-vhs-speed", "ep", "-chroma-dropout", "30000"], 64, 20, 7),
    (["-tvstd", "pal", "-vhs", "-out-composite-lowpass-lite", "0"], 96, 36, 7),
    # preconditions of the hand-tuned kernels NOT met -> they must fall back, results unchanged
    (["-comp-phase", "90"], 96, 32, 0), (["-vhs", "-comp-phase", "90", "-nocolor-subcarrier"], 96, 32, 6),
    # the pre-emphasis presets (subcarrier_amplitude_back != 50): k_encode_fast_pre + k_decode_fast_bk
    (["-vhs", "-comp-catv"], 96, 32, 3), (["-vhs", "-comp-catv2"], 96, 32, 3), (["-vhs", "-comp-catv4"], 64, 38, 3),
    (["-vhs", "-comp-catv3", "-vhs-head-switching-point", "0.105", "-vhs-head-switching-phase", "0.002"], 96, 32, 3),
    # S-Video out of the VCR: k_decode_fast_sv (no re-modulation / second separation, 7 stages fewer)
    (["-vhs", "-vhs-svideo", "1"], 96, 32, 4), (["-vhs", "-vhs-svideo", "1", "-vhs-speed", "ep"], 96, 32, 4),
    (["-vhs", "-vhs-svideo", "1", "-vhs-speed", "lp"], 100, 38, 4), (["-tvstd", "pal", "-vhs", "-vhs-svideo", "1"], 96, 36, 4),
])
def test_every_decoder_path_agrees_with_the_oracle(flags, w, h, fast_ok):
    """The hand-tuned kernels (ntsc_encode_fast.hip / ntsc_decode_fast.hip, one- and two-launch VHS
    forms), the template-specialised PRESET kernels and the GENERIC kernels are one function -- and the
    form each mode is meant to exercise is the form that ran (ntscsim_debug_last_kernels)."""
    n = 4
    p = L.make_params(flags)
    vhs = "-vhs" in flags
    if (4 * w) % 16:                               # the hand-tuned kernels need 16-byte aligned rows
        fast_ok = 0
    srcs = [L.noise_frame(w, h, 77 + j) for j in range(2)]
    jobs = cases.case_jobs(n)
    o = L.OracleStream(p)
    e = np.zeros((n, h, w, 4), np.uint8)
    for k, (si, field, fieldno) in enumerate(jobs):
        o.field(e[k], srcs[si], field, fieldno)
    for mode in ("hand-tuned", "two-launch", "template", "generic"):
        sim = ntscsim.FieldSimulator(params=p)
        if mode == "two-launch":
            sim.debug_no_fast_decode(2)
        elif mode == "template":
            sim.debug_no_fast_decode(1)
        elif mode == "generic":
            sim.debug_force_generic(True)
        got = run_hip(p, srcs, jobs, h, w, per_field_dst=True, sim=sim)
        ran = sim.last_kernels()
        sim.close()
        assert np.array_equal(got, e), mode
        dec = [k for k in ran if k.startswith(("k_decode", "k_vcr_front"))]
        if mode == "generic" or (fast_ok == 3 and mode == "template"):
            # (with an amplitude other than 50 the template forms, which fold (c * 50) / 50 away, do not apply)
            assert len(dec) == 1 and dec[0].startswith("k_decode<") and dec[0].endswith(",1u,double>"), (mode, ran)
        elif fast_ok == 3:
            assert dec == ["k_decode_fast_bk<true,double>"], (mode, ran)
            assert "k_encode_fast_pre<double>" in ran, ran
        elif fast_ok == 7 and mode in ("hand-tuned", "two-launch"):
            assert dec == ["k_decode_fast_fo<double>"], (mode, ran)
        elif fast_ok == 7:
            assert len(dec) == 1 and dec[0].startswith("k_decode<true,true,"), (mode, ran)
        elif fast_ok == 6:
            assert len(dec) == 1 and dec[0].startswith("k_decode<"), (mode, ran)
        elif fast_ok == 5 and mode in ("hand-tuned", "two-launch"):
            assert dec == ["k_decode_fast_xi<double>"], (mode, ran)
            assert "k_encode_fast_xi<double>" in ran, ran
        elif fast_ok == 4 and mode == "template":
            assert dec == ["k_decode<true,false,1u,double>"], (mode, ran)
        elif fast_ok == 4:
            assert dec == ["k_decode_fast_sv<double>"], (mode, ran)
        elif mode == "template" or not fast_ok:
            # the template PRESET forms (every case here keeps the presets' filter switches; what the
            # not-fast_ok cases break is only a precondition of the hand-tuned kernels)
            assert dec == ["k_decode<true,true,6u,double>" if vhs else "k_decode<false,false,0u,double>"], (mode, ran)
        elif fast_ok == 2:
            assert dec == ["k_decode_fast<true,double,true>"], (mode, ran)
        elif mode == "two-launch" and vhs:
            assert dec == ["k_vcr_front<double>", "k_decode_fast<false,double>"], (mode, ran)
        else:
            assert dec == ["k_decode_fast<%s,double>" % ("true" if vhs else "false")], (mode, ran)
            assert "k_encode_fast<double>" in ran, ran


def test_bob_line_doubling():
    w, h = 96, 32
    for hh in (h, h + 1):
        p = L.make_params(["-vhs"])
        srcs = [L.noise_frame(w, hh, 3)]
        for field in (0, 1):
            o = L.OracleStream(p)
            e = np.full((hh, w, 4), 0x33, np.uint8)
            o.field(e, srcs[0], field, 0)
            L.oracle().ntsc_oracle_bob(L._ptr(e), w * 4, w, hh, field)
            g = run_hip(p, srcs, [(0, field, 0)], hh, w, bob=True, dst_init=0x33)
            assert np.array_equal(g[0], e), (hh, field)


_BOB = None


@pytest.mark.parametrize("c", [("bob_default_even", [], 96, 32, 4), ("bob_default_odd", [], 96, 33, 4),
                               ("bob_vhs_even", ["-vhs"], 96, 32, 4), ("bob_vhs_odd", ["-vhs"], 100, 35, 4),
                               ("bob_vhs_h2", ["-vhs"], 64, 2, 2), ("bob_vhs_h3", ["-vhs"], 64, 3, 2)],
                         ids=lambda c: c[0])
def test_field_loop_with_bob_reproduces_reference_golden(c):
    """NTSCSIM_DESC_BOB == the reference's field loop text (composite_layer :2229 + the "field
    deinterlace" block :2233-2257), field after field into one frame, against the frames the reference
    extract wrote (tests/golden/ntsc_bob_golden.npz)."""
    global _BOB
    torch = torch_mod()
    if _BOB is None:
        _BOB = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ntsc_bob_golden.npz"))
    name, flags, w, h, n = c
    p = L.make_params(flags)
    sim = ntscsim.FieldSimulator(params=p)
    src = torch.from_numpy(np.ascontiguousarray(_BOB["%s__src" % name])).cuda()
    dst = torch.full((1, h, w, 4), 0x5A, dtype=torch.uint8, device="cuda")
    for k in range(n):
        sim.fields(src, dst, [(k // 2, 0, (k & 1) ^ 1, k)], bob=True)
        sim.sync()
        assert "k_bob" in sim.last_kernels()
        assert np.array_equal(dst[0].cpu().numpy(), _BOB["%s__after%d" % (name, k)]), "field %d" % k
    sim.close()


def test_error_codes():
    torch = torch_mod()
    p = L.make_params([])
    sim = ntscsim.FieldSimulator(params=p)
    lib = L.product()
    src = torch.zeros((1, 32, 96, 4), dtype=torch.uint8, device="cuda")
    d = sim.build_descs(src, src.clone(), [(0, 0, 0, 0)])
    assert lib.ntscsim_fields_device(sim._h, d, 1, 8, 32, None) == _capi.E_SIZE      # width < 16
    d[0].src_linesize = 4 * 96 - 4
    assert lib.ntscsim_fields_device(sim._h, d, 1, 96, 32, None) == _capi.E_SIZE     # :1580
    d[0].src_linesize = 4 * 96
    d[0].field = 2
    assert lib.ntscsim_fields_device(sim._h, d, 1, 96, 32, None) == _capi.E_ARG
    d[0].field = 0
    d[0].src_dev = None
    assert lib.ntscsim_fields_device(sim._h, d, 1, 96, 32, None) == _capi.E_ARG      # :1578
    assert lib.ntscsim_fields_device(sim._h, d, 0, 96, 32, None) == 0
    h = C.c_void_p()
    bad = L.make_params([], video_noise=-2)
    assert lib.ntscsim_create(C.byref(bad), 0, C.byref(h)) == _capi.E_PARAM
    assert lib.ntscsim_create(C.byref(p), 99, C.byref(h)) == _capi.E_NODEV
    sim.close()


def test_unaligned_rows_take_the_scalar_path():
    """src/dst rows that are not 16-byte aligned (odd width, offset base) stay exact."""
    torch = torch_mod()
    w, h = 99, 20
    p = L.make_params(["-vhs"])
    s = L.noise_frame(w, h, 21)
    o = L.OracleStream(p)
    e = np.zeros((h, w, 4), np.uint8)
    o.field(e, s, 1, 0)
    sim = ntscsim.FieldSimulator(params=p)
    big = torch.zeros((1, h, w + 3, 4), dtype=torch.uint8, device="cuda")
    big[0, :, 1:w + 1] = torch.from_numpy(s).cuda()
    out = torch.zeros_like(big)
    src_v = big[:, :, 1:w + 1]
    dst_v = out[:, :, 1:w + 1]
    sim.fields(src_v, dst_v, [(0, 0, 1, 0)])
    sim.sync()
    assert np.array_equal(dst_v[0].cpu().numpy(), e)
    assert not out[0, :, 0].any() and not out[0, :, w + 1:].any()
    sim.close()


@pytest.mark.parametrize("flags,w,h,n", [
    (["-vhs"], 16, 2, 3),            # smallest accepted frame
    (["-vhs"], 18, 3, 3),            # ragged: width not a multiple of 4, odd height
    ([], 17, 5, 2),
    (["-vhs"], 40, 7, 4),            # narrower than the pipeline depth + 16 (no steady state)
    (["-vhs", "-vhs-speed", "ep", "-out-composite-lowpass-lite", "0"], 52, 6, 2),
    (["-vhs"], 4096, 6, 2),          # one very long scanline per lane
    (["-vhs"], 64, 1030, 2),         # more rows than one wave, head switch far inside
])
def test_extreme_geometries(flags, w, h, n):
    p = L.make_params(flags)
    srcs = [L.noise_frame(w, h, 90 + j) for j in range((n + 1) // 2)]
    jobs = cases.case_jobs(n)
    o = L.OracleStream(p)
    exp = np.zeros((n, h, w, 4), np.uint8)
    for k, (si, field, fieldno) in enumerate(jobs):
        o.field(exp[k], srcs[si], field, fieldno)
    got = run_hip(p, srcs, jobs, h, w, per_field_dst=True)
    assert np.array_equal(got, exp)


def test_many_small_fields_in_one_batch():
    """Ragged batch: 257 fields of 48x10 (rows of many fields share a wavefront)."""
    torch = torch_mod()
    w, h, n = 48, 10, 257
    p = L.make_params(["-vhs"])
    frames = np.stack([L.noise_frame(w, h, 500 + j) for j in range((n + 1) // 2)])
    o = L.OracleStream(p)
    exp = np.zeros((n, h, w, 4), np.uint8)
    for k in range(n):
        o.field(exp[k], frames[k // 2], (k & 1) ^ 1, k)
    sim = ntscsim.FieldSimulator(params=p)
    src = torch.from_numpy(frames).cuda()
    dst = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
    sim.fields(src, dst, [(k // 2, k, (k & 1) ^ 1, k) for k in range(n)])
    sim.sync()
    assert np.array_equal(dst.cpu().numpy(), exp)
    assert sim.rng_pos == o.rng_pos
    sim.close()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 130])
def test_setup_kernels_one_launch_or_two(n):
    """Batches of up to 64 fields run the per-field draws and the row states as ONE launch (k_field_row_setup: the
    field setup's lone wavefront overlaps the row states -- the synchronous call's latency), longer ones as two;
    either way the head-switch plane is cleared by the kernel (no fill).  Same bytes as the oracle at every size, the
    form asserted by name, and the two-launch form forced for the short ones (ntscsim_debug_no_fast_decode bit 16).
    A second batch on the same context with head switching at another point must not see the first one's shifts."""
    torch = torch_mod()
    w, h = 48, 18
    frames = np.stack([L.noise_frame(w, h, 900 + j) for j in range((n + 1) // 2)])
    src = torch.from_numpy(frames).cuda()
    jobs = [(k // 2, k, (k & 1) ^ 1, k) for k in range(n)]
    for forced in (False, True):
        sim = None
        for flags in (["-vhs", "-vhs-head-switching-point", "0.12", "-vhs-head-switching-phase", "0.3"], ["-vhs"]):
            p = L.make_params(flags)
            o = L.OracleStream(p)
            exp = np.zeros((n, h, w, 4), np.uint8)
            for k in range(n):
                o.field(exp[k], frames[k // 2], (k & 1) ^ 1, k)
            if sim is not None:
                sim.close()
            sim = ntscsim.FieldSimulator(params=p)
            if forced:
                sim.debug_no_fast_decode(16)
            for rep in range(2):       # the second run reuses the context's planes
                dst = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
                sim.rng_pos = 0
                sim.fields(src, dst, jobs)
                sim.sync()
                assert np.array_equal(dst.cpu().numpy(), exp), (flags, forced, rep)
            ran = [k for k in sim.last_kernels() if k.startswith(("k_field", "k_row"))]
            assert ran == (["k_field_row_setup"] if n <= 64 and not forced else ["k_field_setup", "k_row_states"]), ran
        sim.close()


def test_frames_host_streaming_equals_field_loop():
    """ntscsim_frames_host: pipelined H2D | kernels | D2H over several chunks == the reference's
    field loop (composite_layer + bob into zeroed frames), rand() stream carried across calls."""
    w, h, n = 96, 34, 11
    p = L.make_params(["-vhs"])
    frames = np.stack([L.noise_frame(w, h, 700 + j) for j in range(n)])
    o = L.OracleStream(p)
    exp = np.zeros((2 * n, h, w, 4), np.uint8)
    for cur in range(2 * n):
        field = (cur & 1) ^ 1
        o.field(exp[cur], frames[cur // 2], field, cur)
        L.oracle().ntsc_oracle_bob(L._ptr(exp[cur]), w * 4, w, h, field)
    sim = ntscsim.FieldSimulator(params=p)
    got = np.full((2 * n, h, w, 4), 0xEE, np.uint8)
    sim.frames_host(got[:8], frames[:4], first_fieldno=0, chunk_frames=3)    # 2 chunks
    sim.frames_host(got[8:], frames[4:], first_fieldno=8, chunk_frames=2)    # 4 chunks
    assert np.array_equal(got, exp)
    assert sim.rng_pos == o.rng_pos
    sim.close()


@pytest.mark.parametrize("flags", [[], ["-vhs"], ["-vhs", "-vhs-head-switching-point", "0.105",
                                                  "-vhs-head-switching-phase", "0.002"]])
def test_ghosting_extension(flags):
    """EXTENSION (absent from the reference): multipath ghost taps on the composite
    signal.  No reference to pin against -- the oracle's own definition is the spec ("parity
    unpinned"); with ghost_taps = 0 (default) every other test shows the output is unchanged."""
    w, h, n = 96, 32, 4
    p = L.make_params(flags)
    p.ghost_taps = 3
    for k, (d, g) in enumerate(((7, 64), (19, -32), (40, 12))):
        p.ghost_delay[k] = d
        p.ghost_gain[k] = g
    srcs = [L.noise_frame(w, h, 61 + j) for j in range(2)]
    jobs = cases.case_jobs(n)
    o = L.OracleStream(p)
    exp = np.zeros((n, h, w, 4), np.uint8)
    for k, (si, field, fieldno) in enumerate(jobs):
        o.field(exp[k], srcs[si], field, fieldno)
    got = run_hip(p, srcs, jobs, h, w, per_field_dst=True)
    assert np.array_equal(got, exp)
    # and it does change the picture
    p0 = L.make_params(flags)
    base = run_hip(p0, srcs, jobs, h, w, per_field_dst=True)
    assert not np.array_equal(base, got)
    bad = L.make_params(flags)
    bad.ghost_taps = 1
    bad.ghost_delay[0] = 0
    assert L.product().ntscsim_params_validate(C.byref(bad)) == _capi.E_PARAM
