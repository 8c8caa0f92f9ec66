// params.cpp -- host-side mirror of the reference's L4 configuration layer for the field DSP:
// the global initialisers (ffmpeg_ntsc.cpp:205-214, :756-809), preset_NTSC/preset_PAL (:815-831)
// and parse_argv() (:972-1282), re-expressed as an immutable ntscsim_params snapshot.
// Same switch names, defaults and preset side effects; errors are returned, not printed-and-exit.
#include "ntscsim.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

void preset_ntsc(ntscsim_params *p)   // preset_NTSC() :824-831
{
    p->output_height = 480;
    p->output_width = 720;
    p->tv_standard = NTSCSIM_TV_NTSC;
}

void preset_pal(ntscsim_params *p)    // preset_PAL() :815-822
{
    p->output_height = 576;
    p->output_width = 720;
    p->tv_standard = NTSCSIM_TV_PAL;
}

} // namespace

extern "C" void ntscsim_params_init(ntscsim_params *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->struct_size = (uint32_t)sizeof(*p);
    preset_ntsc(p);                                   // main() :1924
    p->video_scanline_phase_shift = 180;              // :213
    p->video_scanline_phase_shift_offset = 0;         // :214
    p->composite_preemphasis = 0;                     // :756
    p->composite_preemphasis_cut = 1000000;           // :757
    p->vhs_out_sharpen = 1.5;                         // :759
    p->vhs_head_switching = 0;                        // :761
    p->vhs_head_switching_point = 1.0 - ((4.5 + 0.01) / 262.5);   // :762
    p->vhs_head_switching_phase = ((1.0 - 0.01) / 262.5);         // :763
    p->vhs_head_switching_phase_noise = (((1.0 / 500)) / 262.5);  // :764
    p->composite_in_chroma_lowpass = 1;               // :766
    p->composite_out_chroma_lowpass = 1;              // :767
    p->composite_out_chroma_lowpass_lite = 1;         // :768
    p->video_yc_recombine = 0;                        // :770
    p->video_chroma_noise = 0;                        // :772
    p->video_chroma_phase_noise = 0;                  // :773
    p->video_chroma_loss = 0;                         // :774
    p->video_noise = 2;                               // :775
    p->subcarrier_amplitude = 50;                     // :776
    p->subcarrier_amplitude_back = 50;                // :777
    p->emulating_vhs = 0;                             // :791
    p->nocolor_subcarrier = 0;                        // :794
    p->nocolor_subcarrier_after_yc_sep = 0;           // :795
    p->vhs_chroma_vert_blend = 1;                     // :796
    p->vhs_svideo_out = 0;                            // :797
    p->enable_composite_emulation = 1;                // :798
    p->output_vhs_tape_speed = NTSCSIM_VHS_SP;        // :809
    p->black_key_level_feedback = -1;                 // ffmpeg_to_composite.cpp:322
    p->vhs_out_sharpen_chroma = 0.85;                 // ffmpeg_to_composite.cpp:271
}

extern "C" void ntscsim_params_init_to_composite(ntscsim_params *p)
{
    // ffmpeg_to_composite.cpp:267-333: same as ffmpeg_ntsc except the head-switch model
    ntscsim_params_init(p);
    if (!p) return;
    p->vhs_head_switching_phase = 1.0 - ((4.5 + 0.01) / 262.5);           // :274
    p->vhs_head_switching_point = p->vhs_head_switching_phase;            // no such global: mirror
    p->vhs_head_switching_phase_noise = (((1.0 / 300)) / 262.5);          // :275
}

extern "C" void ntscsim_cli_init(ntscsim_cli *c)
{
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->frame_delay = 1;                   // output_avstream_video_frame_delay :225
    c->use_422_colorspace = 0;            // :205
    c->emulating_preemphasis = 1;         // :792
    c->emulating_deemphasis = 1;          // :793
    c->output_vhs_hifi = 1;               // :788
    c->output_audio_hiss_db = -72;        // :778
    c->output_audio_linear_buzz = -42;    // :779
    c->vhs_linear_high_boost = 0.25;      // :782
}

static int parse_argv_impl(ntscsim_params *p, ntscsim_cli *cli, int argc, const char *const *argv,
                           int require_io, bool tocomp);

extern "C" int ntscsim_params_parse_argv(ntscsim_params *p, ntscsim_cli *cli, int argc,
                                         const char *const *argv, int require_io)
{
    return parse_argv_impl(p, cli, argc, argv, require_io, false);
}

extern "C" int ntscsim_params_parse_argv_to_composite(ntscsim_params *p, ntscsim_cli *cli, int argc,
                                                      const char *const *argv, int require_io)
{
    return parse_argv_impl(p, cli, argc, argv, require_io, true);
}

// tocomp = false: ffmpeg_ntsc.cpp parse_argv :972-1282; true: ffmpeg_to_composite.cpp :1325-1639
static int parse_argv_impl(ntscsim_params *p, ntscsim_cli *cli, int argc, const char *const *argv,
                           int require_io, bool tocomp)
{
    ntscsim_cli local_cli;
    if (!p || (argc > 0 && !argv)) return NTSCSIM_E_ARG;
    if (!cli) { cli = &local_cli; ntscsim_cli_init(cli); }

    int i = 1;
    // every value-taking switch in the reference does `argv[i++]` unchecked (a missing value is a
    // NULL dereference there); here a missing value is NTSCSIM_E_FLAG.
    auto next = [&](const char *&out) -> bool {
        if (i >= argc || argv[i] == nullptr) return false;
        out = argv[i++];
        return true;
    };

    while (i < argc) {
        const char *a = argv[i++];
        const char *v = nullptr;
        if (!a) return NTSCSIM_E_ARG;
        if (*a != '-') {
            std::fprintf(stderr, "Unhandled arg '%s'\n", a);          // :1227
            return NTSCSIM_E_FLAG;
        }
        do { a++; } while (*a == '-');                                  // :979-980

        if (!std::strcmp(a, "h") || !std::strcmp(a, "help")) {
            return NTSCSIM_E_HELP;
        } else if (!std::strcmp(a, "comp-phase-offset")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->video_scanline_phase_shift_offset = std::atoi(v);
        } else if (!std::strcmp(a, "comp-phase")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            int ph = std::atoi(v);
            if (!(ph == 0 || ph == 90 || ph == 180 || ph == 270)) {
                std::fprintf(stderr, "Invalid phase\n");               // :992
                return NTSCSIM_E_FLAG;
            }
            p->video_scanline_phase_shift = ph;
        } else if (!std::strcmp(a, "width")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->output_width = (int)std::strtoul(v, nullptr, 0);
            if (p->output_width < 32) return NTSCSIM_E_FLAG;            // :1001
        } else if (!tocomp && !std::strcmp(a, "d")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            unsigned long d = std::strtoul(v, nullptr, 0);
            if (d == 0 || d > 256) {
                std::fprintf(stderr, "Invalid delay\n");               // :1008
                return NTSCSIM_E_FLAG;
            }
            cli->frame_delay = (int)d;
        } else if (!std::strcmp(a, "i")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            if (tocomp) { cli->input_paths[0] = v; cli->n_inputs = 1; }   // single input :1502
            else {
                if (cli->n_inputs >= NTSCSIM_MAX_INPUTS) return NTSCSIM_E_FLAG;
                cli->input_paths[cli->n_inputs++] = v;
            }
        } else if (tocomp && !std::strcmp(a, "bkey-feedback")) {          // :1356
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->black_key_level_feedback = std::atoi(v);
        } else if (tocomp && (!std::strcmp(a, "ss") || !std::strcmp(a, "se") || !std::strcmp(a, "t") ||
                              !std::strcmp(a, "a") || !std::strcmp(a, "v"))) {
            if (!next(v)) return NTSCSIM_E_FLAG;      // media-layer switches :1368-1392, not DSP
        } else if (tocomp && !std::strcmp(a, "vi")) {
            cli->output_video_as_interlaced = 1;                        // :1399-1401
        } else if (tocomp && !std::strcmp(a, "vp")) {
            cli->output_video_as_interlaced = 0;                        // :1402-1404
        } else if (tocomp && (!std::strcmp(a, "an") || !std::strcmp(a, "vn"))) {
            // :1393-1398 stream selection: media layer
        } else if (!std::strcmp(a, "o")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            cli->output_path = v;
        } else if (!std::strcmp(a, "422")) {
            cli->use_422_colorspace = 1;
        } else if (!std::strcmp(a, "420")) {
            cli->use_422_colorspace = 0;
        } else if (!std::strcmp(a, "tvstd")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            if (!std::strcmp(v, "pal")) preset_pal(p);
            else if (!std::strcmp(v, "ntsc")) preset_ntsc(p);
            else {
                std::fprintf(stderr, "Unknown tv std '%s'\n", v);      // :1038
                return NTSCSIM_E_FLAG;
            }
        } else if (!std::strcmp(a, "in-composite-lowpass")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->composite_in_chroma_lowpass = std::atoi(v) > 0;
        } else if (!std::strcmp(a, "out-composite-lowpass")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->composite_out_chroma_lowpass = std::atoi(v) > 0;
        } else if (!std::strcmp(a, "out-composite-lowpass-lite")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->composite_out_chroma_lowpass_lite = std::atoi(v) > 0;
        } else if (!std::strcmp(a, "nocomp")) {
            // :1051-1054 -- recorded; the reference's video path never tests it
            p->enable_composite_emulation = 0;
        } else if (!std::strcmp(a, "vhs-head-switching-point")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            if (tocomp) {                                                 // to_composite :1405-1407
                p->vhs_head_switching_phase = std::atof(v);
                p->vhs_head_switching_point = p->vhs_head_switching_phase;
            }
            else p->vhs_head_switching_point = std::atof(v);
        } else if (!tocomp && !std::strcmp(a, "vhs-head-switching-phase")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->vhs_head_switching_phase = std::atof(v);
        } else if (!std::strcmp(a, "vhs-head-switching-noise-level")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->vhs_head_switching_phase_noise = std::atof(v);
        } else if (!std::strcmp(a, "vhs-head-switching")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->vhs_head_switching = std::atoi(v) > 0;
        } else if (!std::strcmp(a, "vhs-linear-high-boost")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            cli->vhs_linear_high_boost = std::atof(v);
        } else if (!std::strcmp(a, "comp-pre")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->composite_preemphasis = std::atof(v);
        } else if (!std::strcmp(a, "comp-cut")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->composite_preemphasis_cut = std::atof(v);
        } else if (!std::strcmp(a, "comp-catv")) {              // :1077-1081 | to_composite :1424
            p->composite_preemphasis = tocomp ? 1.5 : 7;
            p->composite_preemphasis_cut = tocomp ? 315000000 / 88 / 2 : 315000000 / 88;
            p->video_chroma_phase_noise = 2;
        } else if (!std::strcmp(a, "comp-catv2")) {             // :1082-1086 | :1429
            p->composite_preemphasis = tocomp ? 2.5 : 15;
            p->composite_preemphasis_cut = tocomp ? 315000000 / 88 / 2 : 315000000 / 88;
            p->video_chroma_phase_noise = 4;
        } else if (!std::strcmp(a, "comp-catv3")) {             // :1087-1091 | :1434
            p->composite_preemphasis = tocomp ? 4 : 25;
            p->composite_preemphasis_cut = tocomp ? 315000000 / 88 / 2 : (315000000 * 2) / 88;
            p->video_chroma_phase_noise = 6;
        } else if (!tocomp && !std::strcmp(a, "comp-catv4")) {  // :1092-1096
            p->composite_preemphasis = 40;
            p->composite_preemphasis_cut = (315000000 * 4) / 88;
            p->video_chroma_phase_noise = 6;
        } else if (!std::strcmp(a, "vhs-linear-video-crosstalk")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            cli->output_audio_linear_buzz = std::atof(v);
        } else if (!std::strcmp(a, "chroma-phase-noise")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->video_chroma_phase_noise = std::atoi(v);
        } else if (!std::strcmp(a, "yc-recomb")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->video_yc_recombine = (int)std::atof(v);          // :1105 (double -> int global)
        } else if (!std::strcmp(a, "audio-hiss")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            cli->output_audio_hiss_db = std::atof(v);
        } else if (!std::strcmp(a, "vhs-svideo")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->vhs_svideo_out = std::atoi(v) > 0;
        } else if (!std::strcmp(a, "vhs-chroma-vblend")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->vhs_chroma_vert_blend = std::atoi(v) > 0;
        } else if (!std::strcmp(a, "chroma-noise")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->video_chroma_noise = std::atoi(v);
        } else if (!std::strcmp(a, "noise")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->video_noise = std::atoi(v);
        } else if (!std::strcmp(a, "subcarrier-amp")) {         // :1125-1129 sets both
            if (!next(v)) return NTSCSIM_E_FLAG;
            int x = std::atoi(v);
            p->subcarrier_amplitude = x;
            p->subcarrier_amplitude_back = x;
        } else if (!std::strcmp(a, "nocolor-subcarrier")) {
            p->nocolor_subcarrier = 1;
        } else if (!std::strcmp(a, "nocolor-subcarrier-after-yc-sep")) {
            p->nocolor_subcarrier_after_yc_sep = 1;
        } else if (!std::strcmp(a, "chroma-dropout")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->video_chroma_loss = std::atoi(v);
        } else if (!std::strcmp(a, "vhs")) {                    // :1141-1151
            p->emulating_vhs = 1;
            p->vhs_head_switching = 1;
            cli->emulating_preemphasis = 0;
            cli->emulating_deemphasis = 0;
            cli->output_audio_hiss_db = -70;
            p->video_chroma_phase_noise = 4;
            p->video_chroma_noise = 16;
            p->video_chroma_loss = 4;
            p->video_noise = 4;
        } else if (!std::strcmp(a, "preemphasis")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            cli->emulating_preemphasis = std::atoi(v) > 0;
        } else if (!std::strcmp(a, "deemphasis")) {
            if (!next(v)) return NTSCSIM_E_FLAG;
            cli->emulating_deemphasis = std::atoi(v) > 0;
        } else if (!std::strcmp(a, "vhs-speed")) {              // :1160-1189
            if (!next(v)) return NTSCSIM_E_FLAG;
            p->emulating_vhs = 1;                               // implies VHS, NOT head switching
            if (!std::strcmp(v, "ep")) {
                p->output_vhs_tape_speed = NTSCSIM_VHS_EP;
                p->video_chroma_phase_noise = 6;
                p->video_chroma_noise = 22;
                p->video_chroma_loss = 8;
                p->video_noise = 6;
            } else if (!std::strcmp(v, "lp")) {
                p->output_vhs_tape_speed = NTSCSIM_VHS_LP;
                p->video_chroma_phase_noise = 5;
                p->video_chroma_noise = 19;
                p->video_chroma_loss = 6;
                p->video_noise = 5;
            } else if (!std::strcmp(v, "sp")) {
                p->output_vhs_tape_speed = NTSCSIM_VHS_SP;
                p->video_chroma_phase_noise = 4;
                p->video_chroma_noise = 16;
                p->video_chroma_loss = 4;
                p->video_noise = 4;
            } else {
                std::fprintf(stderr, "Unknown vhs tape speed '%s'\n", v);   // :1186
                return NTSCSIM_E_FLAG;
            }
        } else if (!std::strcmp(a, "vhs-hifi")) {               // :1191-1204
            if (!next(v)) return NTSCSIM_E_FLAG;
            cli->output_vhs_hifi = std::atoi(v) > 0;
            p->emulating_vhs = 1;
            if (cli->output_vhs_hifi) {
                cli->emulating_preemphasis = 1;
                cli->emulating_deemphasis = 1;
                cli->output_audio_hiss_db = -70;
            } else {
                cli->output_audio_hiss_db = -42;
            }
        } else {
            std::fprintf(stderr, "Unknown switch '%s'\n", a);   // :1222
            return NTSCSIM_E_FLAG;
        }
    }

    // post-parse derivation :1264-1265.  `int += double`: evaluated in double, truncated on store.
    if (p->composite_preemphasis != 0) {
        if (tocomp)                                                     // to_composite :1626-1627
            p->subcarrier_amplitude_back =
                (int)(p->subcarrier_amplitude_back + (50 * p->composite_preemphasis) / 4);
        else
            p->subcarrier_amplitude_back =
                (int)(p->subcarrier_amplitude_back +
                      (50 * p->composite_preemphasis * (315000000 / 88)) /
                          (2 * p->composite_preemphasis_cut));
    }

    if (require_io && tocomp) {
        if (cli->n_inputs == 0 || !cli->output_path || !*cli->output_path) {
            std::fprintf(stderr, "You must specify an input and output file (-i and -o).\n");  // :1634
            return NTSCSIM_E_FLAG;
        }
    } else if (require_io) {
        if (!cli->output_path || !*cli->output_path) {
            std::fprintf(stderr, "No output file specified\n");        // :1272
            return NTSCSIM_E_FLAG;
        }
        if (cli->n_inputs == 0) {
            std::fprintf(stderr, "No input files specified\n");        // :1276
            return NTSCSIM_E_FLAG;
        }
    }
    return NTSCSIM_OK;
}

extern "C" int ntscsim_params_validate(const ntscsim_params *p)
{
    if (!p) return NTSCSIM_E_ARG;
    if (p->struct_size != sizeof(ntscsim_params)) return NTSCSIM_E_PARAM;
    const int ph = p->video_scanline_phase_shift;
    // the reference treats any other value as "phase 0" (:1479-1480); the parser rejects them
    (void)ph;
    // negative noise levels make `rand() % (2k+1)` wrap through unsigned conversion in the
    // reference -- undefined enough that we refuse them.
    const int kmax = 1 << 20;
    if (p->video_noise < 0 || p->video_noise > kmax) return NTSCSIM_E_PARAM;
    if (p->video_chroma_noise < 0 || p->video_chroma_noise > kmax) return NTSCSIM_E_PARAM;
    if (p->video_chroma_phase_noise < 0 || p->video_chroma_phase_noise > 4096) return NTSCSIM_E_PARAM;
    if (p->video_chroma_loss < 0) return NTSCSIM_E_PARAM;
    // division by subcarrier_amplitude(_back) at :1545 (0 is a SIGFPE in the reference);
    // I*amplitude at :1487 overflows int beyond a few thousand.  Documented range is 0..100.
    if (p->subcarrier_amplitude < 1 || p->subcarrier_amplitude > 1000) return NTSCSIM_E_PARAM;
    if (p->subcarrier_amplitude_back < 1 || p->subcarrier_amplitude_back > 1000)
        return NTSCSIM_E_PARAM;
    if (p->output_vhs_tape_speed < NTSCSIM_VHS_SP || p->output_vhs_tape_speed > NTSCSIM_VHS_EP)
        return NTSCSIM_E_PARAM;                                          // reference: abort() :1789
    if (p->tv_standard != NTSCSIM_TV_NTSC && p->tv_standard != NTSCSIM_TV_PAL) return NTSCSIM_E_PARAM;
    if (p->ghost_taps < 0 || p->ghost_taps > NTSCSIM_MAX_GHOST_TAPS) return NTSCSIM_E_PARAM;
    for (int k = 0; k < p->ghost_taps; k++) {
        if (p->ghost_delay[k] < 1 || p->ghost_delay[k] > 4096) return NTSCSIM_E_PARAM;
        if (p->ghost_gain[k] < -256 || p->ghost_gain[k] > 256) return NTSCSIM_E_PARAM;
    }
    if (p->vhs_head_switching) {
        // (unsigned)(fmod(x,1.0)*t) with a negative x is UB in the reference
        const double n = p->vhs_head_switching_phase_noise < 0 ? -p->vhs_head_switching_phase_noise
                                                               : p->vhs_head_switching_phase_noise;
        if (!(p->vhs_head_switching_point - n >= 0) || !(p->vhs_head_switching_phase - n >= 0))
            return NTSCSIM_E_PARAM;
    }
    return NTSCSIM_OK;
}

extern "C" uint64_t ntscsim_rng_calls_per_field(const ntscsim_params *p, int width, int height,
                                                unsigned field)
{
    if (!p || width <= 0 || height <= 0 || field > 1 || (unsigned)height <= field) return 0;
    const uint64_t L = (uint64_t)((height - (int)field + 1) / 2);
    const uint64_t W = (uint64_t)width;
    uint64_t n = 0;
    if (p->video_noise != 0) n += W * L;                                           // :1632
    if (p->vhs_head_switching && p->vhs_head_switching_phase_noise != 0) n += 4;   // :1654-1655
    if (p->video_chroma_noise != 0) n += 2 * W * L;                                // :1719
    if (p->video_chroma_phase_noise != 0) n += L;                                  // :1736
    if (p->video_chroma_loss != 0) n += L;                                         // :1891
    return n;
}

extern "C" uint64_t ntscsim_rng_calls_per_field_422(const ntscsim_params *p, int width, int height,
                                                    unsigned field)
{
    // composite_video_process(), ffmpeg_to_composite.cpp: luma noise 1/pixel :660, head switch 4
    // :680, chroma noise 2 per chroma sample (width/2 per row) :747-750, phase noise 1/row :762,
    // dropout 1/row :937
    if (!p || width <= 0 || height <= 0 || field > 1 || (unsigned)height <= field) return 0;
    const uint64_t L = (uint64_t)((height - (int)field + 1) / 2);
    const uint64_t W = (uint64_t)width, W2 = (uint64_t)(width / 2);
    uint64_t n = 0;
    if (p->video_noise != 0) n += W * L;
    if (p->vhs_head_switching && p->vhs_head_switching_phase_noise != 0) n += 4;
    if (p->video_chroma_noise != 0) n += 2 * W2 * L;
    if (p->video_chroma_phase_noise != 0) n += L;
    if (p->video_chroma_loss != 0) n += L;
    return n;
}

extern "C" const char *ntscsim_strerror(int code)
{
    switch (code) {
    case NTSCSIM_OK: return "ok";
    case NTSCSIM_E_ARG: return "bad argument";
    case NTSCSIM_E_SIZE: return "frame size / linesize not acceptable";
    case NTSCSIM_E_NODEV: return "no HIP device (this library has no CPU fallback)";
    case NTSCSIM_E_HIP: return "HIP call failed";
    case NTSCSIM_E_NOMEM: return "out of memory";
    case NTSCSIM_E_PARAM: return "parameter outside the supported domain";
    case NTSCSIM_E_FLAG: return "unknown switch or bad value";
    case NTSCSIM_E_HELP: return "help requested";
    case NTSCSIM_E_INTERNAL: return "internal error";
    default: return "unknown error";
    }
}
