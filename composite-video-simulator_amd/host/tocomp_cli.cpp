// tocomp_cli.cpp -- `ffmpeg_to_composite`-compatible command line host for the 8-bit YUV422P tool.
//
// Mirrors the reference's parse_argv (ffmpeg_to_composite.cpp:1325-1639, through ntscsim_params_parse_argv_to_composite)
// and the loop of do_video_decode_and_render() :1783-1800 around the C ABI of include/ntscsim.h:
//
//     while (video_field < tgt_field) {
//         render_field(out, in, (video_field & 1) ^ 1, video_field, tgt_pts);                    :1784
//         if (black_key_level_feedback >= 0) black_key_feedback(out, filter, ...);               :1787
//         if (enable_composite_emulation) composite_video_process(out, field, video_field);      :1790
//         if (output_video_as_interlaced) { if (video_field & 1) output_frame(out, video_field - 1, ...); }   :1792
//         else output_frame(out, video_field, field);                                            :1796
//         video_field++;
//     }
//
// The media layer (libav* demux / decode / sws_scale to YUV at output_width / encode, :1650-1781, :1131-1176,
// :1237-1250) is NOT rebuilt: frames enter and leave as raw planar YUV.
//
//   tocomp_cli [reference switches] -i <in.yuv | - | bars:N> -o <out.yuv | - | null:>
//              [--src-height H] [--src-420] [--src-interlaced] [--src-tff] [--height H] [--batch FRAMES]
//
// Input: frames of output_width x src_height (default: the output height), planar 4:2:2 (Y, U, V; chroma width/2) or
// with --src-420 planar 4:2:0 (chroma (h+1)/2 rows) -- what the tool's sws_scale hands to render_field (:1707-1719:
// a 4:2:0 source stays 4:2:0, everything else becomes 4:2:2).  Every source frame lasts two fields (pts 2j, duration
// 2 in the tool's field time base: tgt_field = 2j + 2, tgt_pts = 2j, :1660-1690), so frame j renders fields 2j
// (parity (2j & 1) ^ 1 = 1, "bottom field first") and 2j + 1.
// Output, as output_frame() hands to its encoder (:1177-1236): by default one bob frame per FIELD -- YUV420P, or
// YUV422P with -422; with -vi one frame per field PAIR -- the interlaced 4:2:0 repack, or with -422 the processed
// 4:2:2 frame itself (:1158).  Frames are written tightly packed (Y rows, U rows, V rows).
//
// Batching: the tool works in place on ONE persistent frame; here every source frame gets a device frame of its own
// (luma rows padded by 64 bytes: the separator's two-byte read past each row (:496) then never meets the other field),
// its two fields are one batch entry each, and `--batch` source frames go through the kernels per call.  With
// -bkey-feedback the filter frame is a frame-to-frame recurrence (:974-999): fields are then processed one per call,
// in order, on one persistent frame like the tool.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ntscsim.h"

namespace {

#define HIPOK(call)                                                                         \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e__));                \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

struct Planes {                 // one planar frame in one allocation
    uint8_t *p[3] = {nullptr, nullptr, nullptr};
    int ls[3] = {0, 0, 0}, rows[3] = {0, 0, 0};
    size_t bytes = 0;
    void layout(uint8_t *base, int W, int H, bool c420, int pad, int extra_crows = 0)
    {
        ls[0] = W + pad; ls[1] = ls[2] = W / 2 + pad / 2;
        rows[0] = H; rows[1] = rows[2] = (c420 ? (H + 1) / 2 : H) + extra_crows;
        p[0] = base;
        p[1] = p[0] + (size_t)ls[0] * rows[0];
        p[2] = p[1] + (size_t)ls[1] * rows[1];
        bytes = (size_t)ls[0] * rows[0] + 2 * (size_t)ls[1] * rows[1];
    }
    static size_t size(int W, int H, bool c420, int pad, int extra_crows = 0)
    {
        const size_t cr = (c420 ? ((size_t)H + 1) / 2 : (size_t)H) + (size_t)extra_crows;
        return (size_t)(W + pad) * H + 2 * (size_t)(W / 2 + pad / 2) * cr;
    }
};

// 75 % colour bars in BT.601 limited-range YUV, rotated by `rot` luma samples (synthetic source bars:N)
void make_bars(uint8_t *f, int W, int H, bool c420, long rot)
{
    static const uint8_t Y[8] = {180, 162, 131, 112, 84, 65, 35, 16};
    static const uint8_t U[8] = {128, 44, 156, 72, 184, 100, 212, 128};
    static const uint8_t V[8] = {128, 142, 44, 58, 198, 212, 114, 128};
    const int cr = c420 ? (H + 1) / 2 : H;
    uint8_t *y = f, *u = f + (size_t)W * H, *v = u + (size_t)(W / 2) * cr;
    for (int x = 0; x < W; x++) y[x] = Y[(8 * (int)((x + rot) % W)) / W];
    for (int x = 0; x < W / 2; x++) { const int b = (8 * (int)((2 * x + rot) % W)) / W; u[x] = U[b]; v[x] = V[b]; }
    for (int r = 1; r < H; r++) std::memcpy(y + (size_t)r * W, y, (size_t)W);
    for (int r = 1; r < cr; r++) { std::memcpy(u + (size_t)r * (W / 2), u, (size_t)W / 2); std::memcpy(v + (size_t)r * (W / 2), v, (size_t)W / 2); }
}

} // namespace

int main(int argc, char **argv)
{
    int src_h = 0, height_override = 0, batch = 64;
    bool src420 = false, src_interlaced = false, src_tff = false;
    std::vector<const char *> av;
    av.push_back(argv[0]);
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--src-height") && i + 1 < argc) { src_h = std::atoi(argv[++i]); continue; }
        if (!std::strcmp(argv[i], "--height") && i + 1 < argc) { height_override = std::atoi(argv[++i]); continue; }
        if (!std::strcmp(argv[i], "--batch") && i + 1 < argc) { batch = std::atoi(argv[++i]); continue; }
        if (!std::strcmp(argv[i], "--src-420")) { src420 = true; continue; }
        if (!std::strcmp(argv[i], "--src-interlaced")) { src_interlaced = true; continue; }
        if (!std::strcmp(argv[i], "--src-tff")) { src_tff = true; continue; }
        av.push_back(argv[i]);
    }
    ntscsim_params prm;
    ntscsim_cli cli;
    ntscsim_params_init_to_composite(&prm);
    ntscsim_cli_init(&cli);
    int rc = ntscsim_params_parse_argv_to_composite(&prm, &cli, (int)av.size(), av.data(), 1);
    if (rc == NTSCSIM_E_HELP) {
        std::fprintf(stderr, "%s [ffmpeg_to_composite switches] -i <in.yuv | - | bars:N> -o <out.yuv | - | null:>\n"
                             "   [--src-height H] [--src-420] [--src-interlaced] [--src-tff] [--height H] [--batch FRAMES]\n", argv[0]);
        return 1;
    }
    if (rc != NTSCSIM_OK) return 1;
    if (height_override > 0) prm.output_height = height_override;
    const int W = prm.output_width, H = prm.output_height;
    if (src_h <= 0) src_h = H;
    if (batch < 1) batch = 1;
    const bool feedback = prm.black_key_level_feedback >= 0;
    const bool interlaced_out = cli.output_video_as_interlaced != 0, out422 = cli.use_422_colorspace != 0;
    const bool nocomp = prm.enable_composite_emulation == 0;
    if (feedback) batch = 1;                    // a frame-to-frame recurrence: in order, one field per call

    // source
    const std::string ispec = cli.input_paths[0], ospec = cli.output_path;
    long synth = -1;
    FILE *in = nullptr;
    if (!ispec.compare(0, 5, "bars:")) synth = std::atol(ispec.c_str() + 5);
    else if (ispec == "-") in = stdin;
    else if (!(in = std::fopen(ispec.c_str(), "rb"))) { std::fprintf(stderr, "Failed to open %s\n", ispec.c_str()); return 1; }
    FILE *out = nullptr;
    if (ospec == "-") out = stdout;
    else if (ospec != "null:" && !(out = std::fopen(ospec.c_str(), "wb"))) { std::fprintf(stderr, "Failed to open %s\n", ospec.c_str()); return 1; }

    ntscsim_ctx *sim = nullptr;
    rc = ntscsim_create(&prm, 0, &sim);
    if (rc != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_create: %s\n", ntscsim_strerror(rc)); return 1; }

    // host and device frames of one batch
    const size_t in_bytes = Planes::size(W, src_h, src420, 0);
    const int PAD = 64;
    const size_t frm_bytes = (Planes::size(W, H, false, PAD) + 255) / 256 * 256;     // processed frame (4:2:2, padded rows)
    const bool out420 = !out422;
    // what output_frame() emits: -vi -422: the processed frame; -vi: interlaced 4:2:0; else one bob frame per field
    const bool emit_frame_itself = interlaced_out && out422;
    const size_t out_bytes = Planes::size(W, H, out420, 0);
    uint8_t *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_frm = nullptr, *d_flt = nullptr, *d_out = nullptr;
    HIPOK(hipHostMalloc((void **)&h_in, in_bytes * batch, hipHostMallocDefault));
    HIPOK(hipHostMalloc((void **)&h_out, out_bytes * batch * 2, hipHostMallocDefault));
    HIPOK(hipMalloc((void **)&d_in, (in_bytes + 256) * batch));
    HIPOK(hipMalloc((void **)&d_frm, frm_bytes * batch));
    HIPOK(hipMalloc((void **)&d_out, (Planes::size(W, H, false, 0, 1) + 512) * batch * 2));
    HIPOK(hipMemset(d_frm, 0, frm_bytes * batch));          // av_frame_get_buffer + memset 16/128 would differ: see README
    if (feedback) { HIPOK(hipMalloc((void **)&d_flt, frm_bytes)); HIPOK(hipMemset(d_flt, 0, frm_bytes)); }
    // device output frames carry one spare chroma row: the tool's interlaced 4:2:0 repack writes chroma row
    // (y & 1) + ((y & ~3) >> 1) for y = height - 1, which for a height of 2 mod 4 is row (height + 1) / 2 -- one past the
    // plane (:1215-1223; in the tool it lands in the frame's padding).  It exists here and is not written out.
    const int XC = 1;
    const size_t in_stride = (in_bytes + 255) / 256 * 256, out_stride = (Planes::size(W, H, out420, 0, XC) + 255) / 256 * 256;

    hipStream_t st = nullptr;
    HIPOK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long video_field = 0, frames_in = 0, frames_out = 0;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<ntscsim_field422_desc> descs;
    std::vector<ntscsim_out422_desc> odescs;
    // what output_frame() is called with in this batch, in order: output slot k of d_out gets either the bob /
    // interlaced repack of a frame (ntscsim_output422_device) or, with -vi -422, the processed frame itself (:1158)
    struct Emit { int j; unsigned field; };
    std::vector<Emit> emits;
    auto run_emit = [&](const Emit &e, size_t slot, const Planes &frm) -> int {
        if (emit_frame_itself) {
            Planes o; o.layout(d_out + out_stride * slot, W, H, false, 0, XC);
            for (int k = 0; k < 3; k++)
                if (hipMemcpy2DAsync(o.p[k], (size_t)o.ls[k], frm.p[k], (size_t)frm.ls[k], k ? (size_t)W / 2 : (size_t)W, (size_t)H,
                                     hipMemcpyDeviceToDevice, st) != hipSuccess) return NTSCSIM_E_HIP;
            return NTSCSIM_OK;
        }
        Planes o; o.layout(d_out + out_stride * slot, W, H, out420, 0, XC);
        ntscsim_out422_desc od;
        std::memset(&od, 0, sizeof(od));
        for (int k = 0; k < 3; k++) {
            od.frame_dev[k] = frm.p[k]; od.frame_linesize[k] = frm.ls[k];
            od.bob_dev[k] = o.p[k]; od.bob_linesize[k] = o.ls[k];
        }
        od.field = e.field;
        od.mode = out422 ? NTSCSIM_OUT422_BOB422 : (interlaced_out ? NTSCSIM_OUT422_INTERLACED420 : NTSCSIM_OUT422_BOB420);
        odescs.push_back(od);
        return NTSCSIM_OK;
    };
    bool eof = false;
    while (!eof) {
        // ---- read a batch of source frames
        int nf = 0;
        for (; nf < batch; nf++) {
            uint8_t *dstf = h_in + in_bytes * (size_t)nf;
            if (synth >= 0) {
                if ((long)frames_in >= synth) { eof = true; break; }
                make_bars(dstf, W, src_h, src420, (long)frames_in);
            } else {
                const size_t got = std::fread(dstf, 1, in_bytes, in);
                if (got != in_bytes) {
                    if (got) std::fprintf(stderr, "\n%s: truncated final frame (%zu of %zu bytes) dropped\n", ispec.c_str(), got, in_bytes);
                    eof = true;
                    break;
                }
            }
            frames_in++;
        }
        if (nf == 0) break;
        for (int j = 0; j < nf; j++)
            HIPOK(hipMemcpyAsync(d_in + in_stride * (size_t)j, h_in + in_bytes * (size_t)j, in_bytes, hipMemcpyHostToDevice, st));
        // ---- the loop :1783-1800, two fields per source frame
        descs.clear(); odescs.clear(); emits.clear();
        for (int j = 0; j < nf; j++) {
            Planes src, frm, flt;
            src.layout(d_in + in_stride * (size_t)j, W, src_h, src420, 0);
            frm.layout(d_frm + frm_bytes * (size_t)(feedback ? 0 : j), W, H, false, PAD);
            if (feedback) flt.layout(d_flt, W, H, false, PAD);
            for (int sub = 0; sub < 2; sub++) {
                const unsigned field = (unsigned)((video_field & 1ull) ^ 1ull);               // :1784
                ntscsim_field422_desc d;
                std::memset(&d, 0, sizeof(d));
                for (int k = 0; k < 3; k++) {
                    d.dst_dev[k] = frm.p[k]; d.dst_linesize[k] = frm.ls[k];
                    d.src_dev[k] = src.p[k]; d.src_linesize[k] = src.ls[k];
                    if (feedback) { d.flt_dev[k] = flt.p[k]; d.flt_linesize[k] = flt.ls[k]; }
                }
                d.src_height = src_h;
                d.field = field;
                d.fieldno = video_field;
                d.rng_pos = NTSCSIM_RNG_AUTO;
                d.flags = (src_interlaced ? NTSCSIM_422_INTERLACED : 0u) | (src_tff ? NTSCSIM_422_TFF : 0u) |
                          (src420 ? NTSCSIM_422_SRC420 : 0u) | (sub ? NTSCSIM_422_SECOND : 0u) |    // field_number - src_pts >= 1 :1035
                          (nocomp ? NTSCSIM_422_NOCOMP : 0u);
                descs.push_back(d);
                // output_frame :1792-1797 (-vi: after the pair, with video_field - 1 and ITS parity)
                const bool emit = interlaced_out ? (video_field & 1ull) != 0 : true;
                const Emit e{j, interlaced_out ? (unsigned)(((video_field - 1ull) & 1ull) ^ 1ull) : field};
                video_field++;
                if (feedback) {
                    // a recurrence through the filter frame: this field now, its output before the next one rewrites the frame
                    rc = ntscsim_fields422_device(sim, &descs.back(), 1, W, H, st);
                    if (rc != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_fields422_device: %s (%s)\n", ntscsim_strerror(rc), ntscsim_last_error(sim)); return 1; }
                    if (emit) {
                        odescs.clear();
                        rc = run_emit(e, emits.size(), frm);
                        if (rc == NTSCSIM_OK && !odescs.empty()) rc = ntscsim_output422_device(sim, odescs.data(), 1, W, H, st);
                        if (rc != NTSCSIM_OK) { std::fprintf(stderr, "output_frame: %s (%s)\n", ntscsim_strerror(rc), ntscsim_last_error(sim)); return 1; }
                        emits.push_back(e);
                    }
                } else if (emit) emits.push_back(e);
            }
        }
        if (!feedback) {
            rc = ntscsim_fields422_device(sim, descs.data(), (int)descs.size(), W, H, st);
            if (rc != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_fields422_device: %s (%s)\n", ntscsim_strerror(rc), ntscsim_last_error(sim)); return 1; }
            for (size_t k = 0; k < emits.size(); k++) {
                Planes frm; frm.layout(d_frm + frm_bytes * (size_t)emits[k].j, W, H, false, PAD);
                rc = run_emit(emits[k], k, frm);
                if (rc != NTSCSIM_OK) { std::fprintf(stderr, "output_frame: %s\n", ntscsim_strerror(rc)); return 1; }
            }
            if (!odescs.empty()) {
                rc = ntscsim_output422_device(sim, odescs.data(), (int)odescs.size(), W, H, st);
                if (rc != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_output422_device: %s (%s)\n", ntscsim_strerror(rc), ntscsim_last_error(sim)); return 1; }
            }
        }
        // ---- download and write what output_frame() would have encoded
        for (size_t k = 0; k < emits.size(); k++) {
            Planes o, hpl;
            o.layout(d_out + out_stride * k, W, H, out420 && !emit_frame_itself, 0, XC);
            hpl.layout(h_out + out_bytes * k, W, H, out420 && !emit_frame_itself, 0);
            for (int q = 0; q < 3; q++)
                HIPOK(hipMemcpyAsync(hpl.p[q], o.p[q], (size_t)hpl.ls[q] * hpl.rows[q], hipMemcpyDeviceToHost, st));
        }
        HIPOK(hipStreamSynchronize(st));
        for (size_t k = 0; k < emits.size(); k++) {
            if (out && std::fwrite(h_out + out_bytes * k, 1, out_bytes, out) != out_bytes) { std::fprintf(stderr, "write failed\n"); return 1; }
            frames_out++;
        }
        std::fprintf(stderr, "\rOutput field %llu ", video_field);                                     // :1156
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "\n%llu fields from %llu frames, %llu frames written in %.3f s (%.1f fields/s incl. host I/O)\n",
                 video_field, frames_in, frames_out, dt, dt > 0 ? video_field / dt : 0.0);
    if (out && out != stdout) std::fclose(out);
    ntscsim_destroy(sim);
    (void)hipFree(d_in); (void)hipFree(d_frm); (void)hipFree(d_out); if (d_flt) (void)hipFree(d_flt);
    (void)hipHostFree(h_in); (void)hipHostFree(h_out);
    return 0;
}
