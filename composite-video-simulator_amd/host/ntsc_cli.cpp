// ntsc_cli.cpp -- `ffmpeg_ntsc`-compatible command line host for the GPU field simulator.
//
// Mirrors the reference's L4 + L2 layers (ffmpeg_ntsc.cpp: parse_argv :972-1282, the field loop
// of main() :2146-2283) around the C ABI of include/ntscsim.h.  The L3 media layer (libav* 3.x
// demux/decode/sws_scale/encode, :229-714, :1940-2023) is NOT rebuilt: frames enter and leave as
// raw BGRA (`ffmpeg -i in.mp4 -vf scale=720:480 -pix_fmt bgra -f rawvideo - | ntsc_cli -i - ...`).
//
//   ntsc_cli [reference switches] -i <in.bgra | - | bars:N | noise:N> -o <out.bgra | - | null:>
//
// Output: one bob-deinterlaced BGRA frame per FIELD (59.94 Hz for NTSC), exactly what the
// reference hands to its encoder (:2233-2280).  Field `current` uses source frame current/2,
// field parity (current&1)^1 and fieldno = current (:2229), the frame-delay ring of `-d` frames
// (:2070-2092, :2277) is honoured (it only affects the one row bob does not overwrite).
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ntscsim.h"

namespace {

void usage(const char *arg0)
{
    // same switch list as help() ffmpeg_ntsc.cpp:833-887 (media/audio-only ones are accepted
    // and ignored where the reference's parser accepts them)
    std::fprintf(stderr,
        "%s [options]\n"
        " -i <raw BGRA file | - | bars:N | noise:N>   (more than one: layered, last one wins)\n"
        " -o <raw BGRA file | - | null:>\n"
        " -d <n>                        Video delay buffer (n frames)\n"
        " -width <n>  -tvstd <pal|ntsc>\n"
        " -vhs  -vhs-hifi <0|1>  -vhs-speed <ep|lp|sp>  -vhs-svideo <0|1>  -vhs-chroma-vblend <0|1>\n"
        " -vhs-head-switching <0|1>  -vhs-head-switching-point <x>  -vhs-head-switching-phase <x>\n"
        " -vhs-head-switching-noise-level <x>\n"
        " -noise <0..100>  -chroma-noise <0..100>  -chroma-phase-noise <x>  -chroma-dropout <x>\n"
        " -subcarrier-amp <0...100>  -nocolor-subcarrier  -nocolor-subcarrier-after-yc-sep\n"
        " -comp-pre <s>  -comp-cut <f>  -comp-catv  -comp-catv2  -comp-catv3  -comp-catv4\n"
        " -comp-phase <0|90|180|270>  -comp-phase-offset <n>\n"
        " -in-composite-lowpass <n>  -out-composite-lowpass <n>  -out-composite-lowpass-lite <n>\n"
        " -yc-recomb <n>  -nocomp  -422  -420\n"
        " (audio only, accepted: -preemphasis -deemphasis -audio-hiss -vhs-linear-video-crosstalk\n"
        "  -vhs-linear-high-boost)\n"
        " extra (not in the reference): --batch <fields per GPU batch, default 512> --height <n>\n"
        "                               --gpus <n> | --devices <a,b,...>   (frames dealt round-robin in blocks of 32 over\n"
        "                               n contexts, one per GPU; an ordinal may repeat; default: one context on GPU 0)\n"
        "                               --ghost <delay px>:<gain/256>  (multipath ghost tap, up to 4)\n",
        arg0);
}

struct Source {
    std::string spec;
    FILE *fp = nullptr;
    long synth_frames = -1;   // >= 0: synthetic
    int synth_kind = 0;       // 0 bars, 1 noise
    long next = 0;
    bool eof = false;
};

void make_bars(uint8_t *f, int W, int H, long rot)
{
    static const uint32_t table[8] = {0xC0C0C0, 0xC0C000, 0x00C0C0, 0x00C000,
                                      0xC000C0, 0xC00000, 0x0000C0, 0x000000};
    uint32_t *row0 = reinterpret_cast<uint32_t *>(f);
    for (int x = 0; x < W; x++) row0[x] = table[(8 * (int)((x + rot) % W)) / W];
    for (int y = 1; y < H; y++) std::memcpy(f + (size_t)y * W * 4, f, (size_t)W * 4);   // every row alike
}

void make_noise(uint8_t *f, int W, int H, uint32_t seed)
{
    uint32_t s = seed ? seed : 0x1234567u;
    uint32_t *p = reinterpret_cast<uint32_t *>(f);
    for (size_t i = 0; i < (size_t)W * H; i++) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        p[i] = s & 0xFFFFFFu;
    }
}

bool open_source(Source &s)
{
    if (!s.spec.compare(0, 5, "bars:")) { s.synth_frames = std::atol(s.spec.c_str() + 5); s.synth_kind = 0; return true; }
    if (!s.spec.compare(0, 6, "noise:")) { s.synth_frames = std::atol(s.spec.c_str() + 6); s.synth_kind = 1; return true; }
    if (s.spec == "-") { s.fp = stdin; return true; }
    s.fp = std::fopen(s.spec.c_str(), "rb");
    return s.fp != nullptr;
}

// false at end of input (a trailing partial frame is reported and dropped)
bool read_frame(Source &s, uint8_t *dst, int W, int H)
{
    if (s.eof) return false;
    if (s.synth_frames >= 0) {
        if (s.next >= s.synth_frames) { s.eof = true; return false; }
        if (s.synth_kind == 0) make_bars(dst, W, H, s.next);
        else make_noise(dst, W, H, 0x1234567u + (uint32_t)s.next);
        s.next++;
        return true;
    }
    const size_t n = (size_t)W * H * 4;
    const size_t got = std::fread(dst, 1, n, s.fp);
    if (got != n) {
        if (got != 0)
            std::fprintf(stderr, "\n%s: truncated final frame (%zu of %zu bytes) dropped\n", s.spec.c_str(), got, n);
        s.eof = true;
        return false;
    }
    s.next++;
    return true;
}

#define HIPOK(call)                                                                         \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e__));                \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

} // namespace

int main(int argc, char **argv)
{
    // pull out the two switches the reference does not have, pass the rest to the mirror parser
    int batch_fields = 512, height_override = 0;   // 512: the up / kernels / down pipeline of a batch has room to overlap
    int ghost_n = 0, ghost_d[4] = {0, 0, 0, 0}, ghost_g[4] = {0, 0, 0, 0};
    int gpus = 1;
    std::vector<int> devices;
    std::vector<const char *> av;
    av.push_back(argv[0]);
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--batch") && i + 1 < argc) { batch_fields = std::atoi(argv[++i]); continue; }
        if (!std::strcmp(argv[i], "--height") && i + 1 < argc) { height_override = std::atoi(argv[++i]); continue; }
        if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) { gpus = std::atoi(argv[++i]); continue; }
        if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) {
            for (const char *q = argv[++i]; *q;) {
                devices.push_back(std::atoi(q));
                while (*q && *q != ',') q++;
                if (*q == ',') q++;
            }
            continue;
        }
        if (!std::strcmp(argv[i], "--ghost") && i + 1 < argc) {       // extension: <delay>:<gain/256>
            int dd = 0, gg = 0;
            if (std::sscanf(argv[++i], "%d:%d", &dd, &gg) != 2 || ghost_n >= 4) { std::fprintf(stderr, "bad --ghost\n"); return 1; }
            ghost_d[ghost_n] = dd; ghost_g[ghost_n] = gg; ghost_n++;
            continue;
        }
        av.push_back(argv[i]);
    }
    ntscsim_params prm;
    ntscsim_cli cli;
    ntscsim_params_init(&prm);          // preset_NTSC(), main() :1924
    ntscsim_cli_init(&cli);
    int rc = ntscsim_params_parse_argv(&prm, &cli, (int)av.size(), av.data(), 1);   // :1925
    if (rc == NTSCSIM_E_HELP) { usage(argv[0]); return 1; }
    if (rc != NTSCSIM_OK) return 1;
    if (height_override > 0) prm.output_height = height_override;
    prm.ghost_taps = ghost_n;
    for (int k = 0; k < ghost_n; k++) { prm.ghost_delay[k] = ghost_d[k]; prm.ghost_gain[k] = ghost_g[k]; }
    if (batch_fields < 2) batch_fields = 2;
    batch_fields &= ~1;
    const int W = prm.output_width, H = prm.output_height;
    const size_t fbytes = (size_t)W * H * 4;
    // as the reference prints after parsing (:1268-1269)
    std::fprintf(stderr, "VHS head switching point: %.6f\n", prm.vhs_head_switching_phase);
    std::fprintf(stderr, "VHS head switching noise: %.6f\n", prm.vhs_head_switching_phase_noise);

    std::vector<Source> inputs((size_t)cli.n_inputs);
    for (int i = 0; i < cli.n_inputs; i++) {
        inputs[(size_t)i].spec = cli.input_paths[i];
        if (!open_source(inputs[(size_t)i])) {
            std::fprintf(stderr, "Failed to open %s\n", cli.input_paths[i]);
            return 1;
        }
    }
    FILE *out = nullptr;
    const std::string ospec = cli.output_path;
    if (ospec == "-") out = stdout;
    else if (ospec != "null:") {
        out = std::fopen(ospec.c_str(), "wb");
        if (!out) { std::fprintf(stderr, "Failed to open %s\n", ospec.c_str()); return 1; }
    }

    // one context per GPU (north_star: "partition the input stream frame-round-robin"): ntscsim_pool_*; the
    // layered path below (several -i) composites on the pool's first context
    if (gpus < 1 || gpus > 64) { std::fprintf(stderr, "--gpus 1..64\n"); return 1; }
    if (devices.empty()) { int nd = 0; (void)hipGetDeviceCount(&nd); for (int g = 0; g < gpus; g++) devices.push_back(nd > 0 ? g % nd : 0); }
    ntscsim_pool *pool = nullptr;
    rc = ntscsim_pool_create(&prm, devices.data(), (int)devices.size(), &pool);
    if (rc != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_pool_create: %s\n", ntscsim_strerror(rc)); return 1; }
    ntscsim_ctx *sim = ntscsim_pool_ctx(pool, 0);
    HIPOK(hipSetDevice(devices[0]));

    const int nframes_batch = batch_fields / 2;
    uint8_t *d_src = nullptr, *d_dst = nullptr, *h_srcs[2] = {nullptr, nullptr}, *h_dst = nullptr;
    HIPOK(hipMalloc((void **)&d_src, fbytes * nframes_batch));
    HIPOK(hipMalloc((void **)&d_dst, fbytes * batch_fields));
    for (int i = 0; i < 2; i++) HIPOK(hipHostMalloc((void **)&h_srcs[i], fbytes * nframes_batch, hipHostMallocPortable));
    HIPOK(hipHostMalloc((void **)&h_dst, fbytes * batch_fields, hipHostMallocPortable));
    HIPOK(hipMemset(d_dst, 0, fbytes * batch_fields));   // ring frames start zeroed (:2088)
    {
        // one-off initialisation outside the clock: code objects, streams, chunk slots
        int nw = 32 * ntscsim_pool_size(pool);           // one block per context
        if (nw > nframes_batch) nw = nframes_batch;
        std::memset(h_srcs[0], 0, fbytes * nw);
        (void)ntscsim_pool_frames_host(pool, h_srcs[0], fbytes, W * 4, nw, h_dst, fbytes, W * 4, W, H, 0, NTSCSIM_DESC_BOB, 32);
        ntscsim_pool_set_rng_pos(pool, 0);
        ntscsim_set_rng_pos(sim, 0);
    }

    // frame-delay ring (:2070-2092): only its row H-1 can survive composite_layer + bob
    const int delay = cli.frame_delay;
    std::vector<std::vector<uint8_t>> ring_last((size_t)delay, std::vector<uint8_t>((size_t)W * 4, 0));
    size_t ring_idx = 0;

    std::vector<ntscsim_field_desc> descs((size_t)batch_fields);
    std::vector<uint8_t> scratch(fbytes), top_last(fbytes, 0);
    unsigned long long current = 0;     // output field counter (:2140)
    unsigned long long total_fields = 0;
    const auto t0 = std::chrono::steady_clock::now();
    const bool layered = inputs.size() > 1;
    // ---- reader thread: fills the next batch of source frames while the GPU works on the current
    //      one.  Every layer is composited in every field and the LAST input overwrites the layers below
    //      it (:2203-2230), so only its frames are kept.  The run goes on until EVERY input has ended
    //      (:2149-2153, :2283): an input that has ended keeps compositing the last frame it delivered
    //      (:2192-2197, :2218-2226) -- a lower layer goes on drawing from rand(), the top layer goes on
    //      showing its last frame while a longer lower layer is still running.
    struct Slot { int nf = 0; bool ready = false, last = false; } slot[2];
    bool stop = false;                  // set on every way out of main(): lets the reader leave
    std::mutex mu;
    std::condition_variable cv;
    std::thread reader([&] {
        bool eof = false;
        for (int b = 0; !eof; b ^= 1) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !slot[b].ready || stop; }); if (stop) return; }
            uint8_t *buf = h_srcs[b];
            int nf = 0;
            Source &top = inputs.back();
            if (top.synth_frames >= 0 && !layered) {
                // synthetic source: frames are independent, fill the batch with a few threads
                const long left = top.synth_frames - top.next;
                nf = (int)(left < nframes_batch ? left : nframes_batch);
                const long first = top.next;
                const int nt = nf >= 8 ? 4 : 1;
                std::vector<std::thread> th;
                for (int t = 0; t < nt; t++)
                    th.emplace_back([&, t] {
                        for (int i = t; i < nf; i += nt) {
                            if (top.synth_kind == 0) make_bars(buf + fbytes * i, W, H, first + i);
                            else make_noise(buf + fbytes * i, W, H, 0x1234567u + (uint32_t)(first + i));
                        }
                    });
                for (auto &x : th) x.join();
                top.next += nf;
                if (top.next >= top.synth_frames) { top.eof = true; eof = true; }
            } else {
                for (; nf < nframes_batch; nf++) {
                    bool any = false;
                    for (size_t li = 0; li + 1 < inputs.size(); li++)
                        any = read_frame(inputs[li], scratch.data(), W, H) || any;
                    uint8_t *dstf = buf + fbytes * nf;
                    if (read_frame(top, dstf, W, H)) { any = true; if (layered) std::memcpy(top_last.data(), dstf, fbytes); }
                    else std::memcpy(dstf, top_last.data(), fbytes);       // ended: its last frame (black if it never had one)
                    if (!any) { eof = true; break; }                       // every input has ended
                }
            }
            { std::lock_guard<std::mutex> lk(mu); slot[b].nf = nf; slot[b].last = eof; slot[b].ready = true; }
            cv.notify_all();
        }
    });
    // On the normal way out the reader has finished and is joined.  On an error return it may be sitting in
    // fread() on a pipe nobody writes to any more, holding references into main()'s frame: the process ends
    // right there (std::_Exit) instead of unwinding under it.
    struct Joiner {
        std::thread &t; std::mutex &m; std::condition_variable &c; bool &stop; bool done = false;
        ~Joiner()
        {
            { std::lock_guard<std::mutex> lk(m); stop = true; }
            c.notify_all();
            if (!t.joinable()) return;
            if (done) t.join();
            else {
                // error return: the reader may be blocked in fread() on a pipe nobody writes to any more and it
                // references main()'s stack -- never unwind under it: leave the process here
                std::fflush(nullptr);
                std::_Exit(1);
            }
        }
    } joiner{reader, mu, cv, stop};
    for (int b = 0;; b ^= 1) {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return slot[b].ready; }); }
        const int nf = slot[b].nf;
        const bool last_batch = slot[b].last;
        uint8_t *h_src = h_srcs[b];
        auto release = [&] { { std::lock_guard<std::mutex> lk(mu); slot[b].ready = false; } cv.notify_all(); };
        if (nf == 0) { release(); break; }
        const int nfields = nf * 2;
        if (!layered) {
            // one input: the library's own field loop over host frames (H2D | kernels | D2H pipelined
            // in chunks; field = (current & 1) ^ 1, fieldno = current, sequential rand() stream)
            // (with more than one context: blocks of 32 frames dealt round-robin, each context its own pipeline)
            rc = ntscsim_pool_frames_host(pool, h_src, fbytes, W * 4, nf, h_dst, fbytes, W * 4, W, H, current,
                                          NTSCSIM_DESC_BOB, 32);
            if (rc != NTSCSIM_OK) {
                std::fprintf(stderr, "ntscsim_pool_frames_host: %s (%s)\n", ntscsim_strerror(rc), ntscsim_pool_last_error(pool));
                return 1;
            }
        } else {
            HIPOK(hipMemcpy(d_src, h_src, fbytes * nf, hipMemcpyHostToDevice));
            const uint64_t layers = (uint64_t)inputs.size();
            uint64_t pos = ntscsim_get_rng_pos(sim);
            for (int k = 0; k < nfields; k++) {
                const unsigned long long cur = current + (unsigned)k;
                const unsigned field = (unsigned)((cur & 1) ^ 1);                       // :2229
                const uint64_t calls = ntscsim_rng_calls_per_field(&prm, W, H, field);
                pos += calls * (layers - 1);          // draws made by the layers underneath
                ntscsim_field_desc &d = descs[(size_t)k];
                std::memset(&d, 0, sizeof(d));
                d.src_dev = d_src + fbytes * (size_t)(k / 2);
                d.dst_dev = d_dst + fbytes * (size_t)k;
                d.src_linesize = W * 4; d.dst_linesize = W * 4;
                d.field = field;
                d.flags = NTSCSIM_DESC_BOB;                                             // :2233-2257
                d.fieldno = cur;
                d.rng_pos = pos;
                pos += calls;
            }
            rc = ntscsim_fields_device(sim, descs.data(), nfields, W, H, nullptr);
            if (rc != NTSCSIM_OK) {
                std::fprintf(stderr, "ntscsim_fields_device: %s (%s)\n", ntscsim_strerror(rc), ntscsim_last_error(sim));
                return 1;
            }
            rc = ntscsim_sync(sim);
            if (rc != NTSCSIM_OK) return 1;
            ntscsim_set_rng_pos(sim, pos);
            HIPOK(hipMemcpy(h_dst, d_dst, fbytes * nfields, hipMemcpyDeviceToHost));
            // the device frames are reused by the next batch: clear what bob did not overwrite
            HIPOK(hipMemset(d_dst, 0, fbytes * batch_fields));
        }
        release();                          // the reader may refill this slot now
        for (int k = 0; k < nfields; k++) {
            uint8_t *f = h_dst + fbytes * (size_t)k;
            const unsigned field = (unsigned)(((current + (unsigned)k) & 1) ^ 1);
            uint8_t *last = f + (size_t)(H - 1) * W * 4;
            // the row neither composite_layer nor bob writes keeps the ring frame's content
            const bool stale = ((H & 1) == 0 && field == 0) || ((H & 1) == 1 && field == 1);
            std::vector<uint8_t> &slot = ring_last[ring_idx];
            if (stale) std::memcpy(last, slot.data(), (size_t)W * 4);
            std::memcpy(slot.data(), last, (size_t)W * 4);
            ring_idx = (ring_idx + 1) % (size_t)delay;                              // :2277
            if (out && std::fwrite(f, 1, fbytes, out) != fbytes) { std::fprintf(stderr, "write failed\n"); return 1; }
        }
        current += (unsigned)nfields;
        total_fields += (unsigned)nfields;
        std::fprintf(stderr, "\rOutput field %llu ", current);                      // :1361
        if (last_batch) break;
    }
    joiner.done = true;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "\n%llu fields in %.3f s (%.1f fields/s incl. host I/O) on %d context(s)\n", total_fields, dt,
                 dt > 0 ? total_fields / dt : 0.0, ntscsim_pool_size(pool));
    if (out && out != stdout) std::fclose(out);
    ntscsim_pool_destroy(pool);
    (void)hipFree(d_src); (void)hipFree(d_dst); (void)hipHostFree(h_srcs[0]); (void)hipHostFree(h_srcs[1]); (void)hipHostFree(h_dst);
    return 0;
}
