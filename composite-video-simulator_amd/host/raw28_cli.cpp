// raw28_cli.cpp -- `ffmpeg_raw28ntsc`-compatible command line host for the GPU raw-composite decoder.
//
// Mirrors the reference's switch parser (ffmpeg_raw28ntsc.cpp parse_argv :442-520) and its field
// loop (main() :1006-1038) around ntscsim_raw28_stream_push().  The media layer (libav* encode of the
// rendered frames, :1032-1046) is NOT rebuilt: the capture is read from a file of 8-bit samples and
// the frames leave as raw BGRA, `width x 262` each, one per field --
//
//   raw28_cli [reference switches] -i <capture.u8 | -> -o <frames.bgra | - | null:> [--max-fields N]
//             [--chunk-bytes N] [--ring-fields N]        (input piece and output ring of the stream)
//   ... | ffmpeg -f rawvideo -pix_fmt bgra -s 1820x262 -r 60000/1001 -i - out.mkv
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ntscsim.h"

int main(int argc, char **argv)
{
    ntscsim_raw28_opts o;
    ntscsim_raw28_opts_init(&o);
    std::string in, out;
    long max_fields = -1;
    size_t chunk_bytes = 64u << 20;       // samples read (and pushed to the decoder) at a time
    long ring_fields = 64;                // device frames per push: the output ring
    // our own switches first; everything else goes through the mirror of the reference's parser
    std::vector<const char *> rest;
    rest.push_back(argv[0]);
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--max-fields") && i + 1 < argc) { max_fields = std::atol(argv[++i]); continue; }
        if (!std::strcmp(argv[i], "--chunk-bytes") && i + 1 < argc) { chunk_bytes = (size_t)std::atoll(argv[++i]); continue; }
        if (!std::strcmp(argv[i], "--ring-fields") && i + 1 < argc) { ring_fields = std::atol(argv[++i]); continue; }
        if ((!std::strcmp(argv[i], "-i") || !std::strcmp(argv[i], "--i")) && i + 1 < argc) in = argv[i + 1];
        if ((!std::strcmp(argv[i], "-o") || !std::strcmp(argv[i], "--o")) && i + 1 < argc) out = argv[i + 1];
        rest.push_back(argv[i]);
    }
    const int rc = ntscsim_raw28_parse_argv(&o, (int)rest.size(), rest.data(), 1);
    if (rc == NTSCSIM_E_HELP || rc != NTSCSIM_OK || chunk_bytes < 1 || chunk_bytes >= 0xF0000000ull || ring_fields < 1) {
        std::fprintf(stderr, "%s [options]\n -i <capture of 8-bit samples | ->\n -o <raw BGRA frames | - | null:>\n"
                             " -s <rate>                     ntsc28, 40mhz, or samples per second\n"
                             " -marksig -noequ -nowequ -nosig -nosc -showsc\n"
                             " extra (not in the reference): --max-fields <n>  --chunk-bytes <n>  --ring-fields <n>\n", argv[0]);
        return 1;
    }
    if (out.empty()) { std::fprintf(stderr, "No output file specified\n"); return 1; }      // :510-513
    if (in.empty()) { std::fprintf(stderr, "No input file specified\n"); return 1; }        // :514-517

    int W = 0, H = 0, len = 0;
    if (ntscsim_raw28_geometry(&o, &W, &H, &len) != NTSCSIM_OK) { std::fprintf(stderr, "unsupported sample rate\n"); return 1; }
    std::fprintf(stderr, "Raw render to:          %d\n", len);                               // :883

    FILE *fi = in == "-" ? stdin : std::fopen(in.c_str(), "rb");
    if (!fi) { std::fprintf(stderr, "Failed to open src\n"); return 1; }                    // :894-897
    FILE *fo = nullptr;
    if (out == "-") fo = stdout;
    else if (out != "null:") { fo = std::fopen(out.c_str(), "wb"); if (!fo) { std::fprintf(stderr, "Failed to open %s\n", out.c_str()); return 1; } }

    ntscsim_raw28 *dec = nullptr;
    if (ntscsim_raw28_create(&o, 0, &dec) != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_raw28_create failed\n"); return 1; }
    // The capture is read in pieces and handed to the decoder as a stream, like the tool's own 2048-line
    // window over its input (:264-357): a capture of any length, or a pipe, with a bounded ring of frames.
    const size_t fbytes = (size_t)W * 4 * (size_t)H;
    uint8_t *d_frames = nullptr;
    if (hipMalloc((void **)&d_frames, fbytes * (size_t)ring_fields) != hipSuccess) {
        std::fprintf(stderr, "out of device memory for %ld frames\n", ring_fields);
        return 1;
    }
    std::vector<uint8_t> piece(chunk_bytes), host(fbytes);
    const auto t0 = std::chrono::steady_clock::now();
    long total = 0;
    bool eof = false, stop = false;
    if (ntscsim_raw28_stream_reset(dec) != NTSCSIM_OK) return 1;
    while (!stop) {
        size_t got = 0;
        if (!eof) {
            while (got < chunk_bytes) {
                const size_t r = std::fread(piece.data() + got, 1, chunk_bytes - got, fi);
                if (r == 0) { eof = true; break; }
                got += r;
            }
        }
        // every push hands back at most one ring of fields; empty pushes drain what a push left behind
        for (bool first = true;; first = false) {
            long room = ring_fields;
            if (max_fields >= 0 && max_fields - total < room) room = max_fields - total;
            if (room <= 0) { stop = true; break; }
            int nf = 0;
            const int drc = ntscsim_raw28_stream_push(dec, first ? piece.data() : nullptr, first ? got : 0, 0, eof ? 1 : 0,
                                                      d_frames, fbytes, W * 4, (int)room, &nf);
            if (drc != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_raw28_stream_push: %s (%s)\n", ntscsim_strerror(drc), ntscsim_raw28_last_error(dec)); return 1; }
            for (int f = 0; f < nf; f++) {
                if (fo) {
                    if (hipMemcpy(host.data(), d_frames + (size_t)f * fbytes, fbytes, hipMemcpyDeviceToHost) != hipSuccess) { std::fprintf(stderr, "download failed\n"); return 1; }
                    if (std::fwrite(host.data(), 1, fbytes, fo) != fbytes) { std::fprintf(stderr, "write failed\n"); return 1; }
                }
                std::fprintf(stderr, "\rOutput field %ld ", total + f);                     // :538
            }
            total += nf;
            if (nf < room) break;                   // nothing more to give until more samples arrive
        }
        if (eof) stop = true;
    }
    if (fi != stdin) std::fclose(fi);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "\n%ld fields of %dx%d in %.3f s (%.1f fields/s incl. input, upload and output)\n", total, W, H, dt, dt > 0 ? total / dt : 0.0);
    if (fo && fo != stdout) std::fclose(fo);
    (void)hipFree(d_frames);
    ntscsim_raw28_destroy(dec);
    return 0;
}
