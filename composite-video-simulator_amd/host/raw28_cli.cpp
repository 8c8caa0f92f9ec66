// raw28_cli.cpp -- `ffmpeg_raw28ntsc`-compatible command line host for the GPU raw-composite decoder.
//
// Mirrors the reference's switch parser (ffmpeg_raw28ntsc.cpp parse_argv :442-520) and its field
// loop (main() :1006-1038) around ntscsim_raw28_decode().  The media layer (libav* encode of the
// rendered frames, :1032-1046) is NOT rebuilt: the capture is read from a file of 8-bit samples and
// the frames leave as raw BGRA, `width x 262` each, one per field --
//
//   raw28_cli [reference switches] -i <capture.u8 | -> -o <frames.bgra | - | null:> [--max-fields N]
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
    // our own switches first; everything else goes through the mirror of the reference's parser
    std::vector<const char *> rest;
    rest.push_back(argv[0]);
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--max-fields") && i + 1 < argc) { max_fields = std::atol(argv[++i]); continue; }
        if ((!std::strcmp(argv[i], "-i") || !std::strcmp(argv[i], "--i")) && i + 1 < argc) in = argv[i + 1];
        if ((!std::strcmp(argv[i], "-o") || !std::strcmp(argv[i], "--o")) && i + 1 < argc) out = argv[i + 1];
        rest.push_back(argv[i]);
    }
    const int rc = ntscsim_raw28_parse_argv(&o, (int)rest.size(), rest.data(), 1);
    if (rc == NTSCSIM_E_HELP || rc != NTSCSIM_OK) {
        std::fprintf(stderr, "%s [options]\n -i <capture of 8-bit samples | ->\n -o <raw BGRA frames | - | null:>\n"
                             " -s <rate>                     ntsc28, 40mhz, or samples per second\n"
                             " -marksig -noequ -nowequ -nosig -nosc -showsc\n"
                             " extra (not in the reference): --max-fields <n>\n", argv[0]);
        return 1;
    }
    if (out.empty()) { std::fprintf(stderr, "No output file specified\n"); return 1; }      // :510-513
    if (in.empty()) { std::fprintf(stderr, "No input file specified\n"); return 1; }        // :514-517

    int W = 0, H = 0, len = 0;
    if (ntscsim_raw28_geometry(&o, &W, &H, &len) != NTSCSIM_OK) { std::fprintf(stderr, "unsupported sample rate\n"); return 1; }
    std::fprintf(stderr, "Raw render to:          %d\n", len);                               // :933

    FILE *fi = in == "-" ? stdin : std::fopen(in.c_str(), "rb");
    if (!fi) { std::fprintf(stderr, "Failed to open src\n"); return 1; }                    // :948-951
    std::vector<uint8_t> cap;
    {
        uint8_t buf[1 << 16];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof(buf), fi)) > 0) cap.insert(cap.end(), buf, buf + n);
        if (fi != stdin) std::fclose(fi);
    }
    FILE *fo = nullptr;
    if (out == "-") fo = stdout;
    else if (out != "null:") { fo = std::fopen(out.c_str(), "wb"); if (!fo) { std::fprintf(stderr, "Failed to open %s\n", out.c_str()); return 1; } }

    ntscsim_raw28 *dec = nullptr;
    if (ntscsim_raw28_create(&o, 0, &dec) != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_raw28_create failed\n"); return 1; }
    const size_t fbytes = (size_t)W * 4 * (size_t)H;
    long cap_fields = (long)(cap.size() / ((size_t)len * 240)) + 2;     // the tool consumes >= 240 scanlines per field (:836)
    if (max_fields >= 0 && max_fields < cap_fields) cap_fields = max_fields;
    uint8_t *d_frames = nullptr;
    if (cap_fields > 0 && hipMalloc((void **)&d_frames, fbytes * (size_t)cap_fields) != hipSuccess) {
        std::fprintf(stderr, "out of device memory for %ld frames\n", cap_fields);
        return 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int nf = 0;
    const int drc = cap_fields > 0 ? ntscsim_raw28_decode(dec, cap.data(), cap.size(), d_frames, fbytes, W * 4, (int)cap_fields, &nf) : NTSCSIM_OK;
    if (drc != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_raw28_decode: %s (%s)\n", ntscsim_strerror(drc), ntscsim_raw28_last_error(dec)); return 1; }
    std::vector<uint8_t> host(fbytes);
    for (int f = 0; f < nf; f++) {
        if (fo) {
            if (hipMemcpy(host.data(), d_frames + (size_t)f * fbytes, fbytes, hipMemcpyDeviceToHost) != hipSuccess) { std::fprintf(stderr, "download failed\n"); return 1; }
            if (std::fwrite(host.data(), 1, fbytes, fo) != fbytes) { std::fprintf(stderr, "write failed\n"); return 1; }
        }
        std::fprintf(stderr, "\rOutput field %d ", f);                                      // :538
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "\n%d fields of %dx%d in %.3f s (%.1f fields/s incl. upload and output)\n", nf, W, H, dt, dt > 0 ? nf / dt : 0.0);
    if (fo && fo != stdout) std::fclose(fo);
    (void)hipFree(d_frames);
    ntscsim_raw28_destroy(dec);
    return 0;
}
