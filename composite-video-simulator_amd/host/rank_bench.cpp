// rank_bench.cpp -- the multi-GPU form of BASELINE's metric with a C++ host: one process per GPU, the clip dealt
// frame-round-robin, RCCL (rccl.h, directly -- no torch, no Python) for the barrier around the timed region, the MAX of
// the elapsed times and the all-gather of every rank's {checksum, fields, elapsed} (north_star: "host stays C++ ...
// RCCL over xGMI only for the barrier/gather"; VERDICT r04 "missing" 4).
//
//   rank_bench [reference switches] [--spawn N | (RANK / WORLD_SIZE / LOCAL_RANK from the environment, e.g. torchrun)]
//              [--frames F] [--steps K] [--warmup W] [--scaling weak|strong] [--verify 0|1] [--width W --height H]
//              [--size WxH] [--streams S] [--one-device 0|1] [--inflight Q]
//
// BASELINE's two multi-GPU configurations have a leg each:
//   configs[3]  --streams 8                  eight independent 300-frame 720x486 streams, stream s on rank s % N (every
//                                            stream its own fieldno / rand() sequence from 0, like eight runs of the tool)
//   configs[4]  --size 3840x2160 --frames 600   the 60 fps long-form clip (600 frames = 10 s), frame-round-robin
//
// Every rank owns the frames g = rank, rank + N, ... of the clip (weak: the clip has N x F frames, strong: F), keeps
// its source and destination frames in ITS GPU's HBM and runs ONE prepared batch per step (ntscsim_batch_run: the
// kernel chain only); like bench.py it keeps Q steps in flight (default 4: Q contexts with a stream, scratch and a
// destination clip each, the steps rotate over them -- a lone 600-field launch leaves a second round of one wave per
// SIMD).  Frame g produces fields 2g and 2g + 1; the rand() position of field k is a closed form
// (the draws of a composite_layer() call do not depend on the pixels, ffmpeg_ntsc.cpp:1632-1764):
//     pos(k) = (k / 2) * (draws(parity 1) + draws(parity 0)) + (k & 1) * draws(parity 1)
// so the union of the ranks' outputs is the one-GPU run byte for byte, and no data moves between GPUs.
// Rank 0 prints one JSON line: whole-job fields/s (fields of all ranks / MAX elapsed), per-rank checksums (sum of
// all destination bytes), and with --verify 1 whether rank 0 could reproduce every rank's checksum on its own GPU.
// The ncclUniqueId travels through a file ($NTSCSIM_RCCL_ID_FILE, default /tmp/ntscsim_rccl_id.<MASTER_PORT|ppid>): rank 0
// creates it exclusively (O_EXCL | O_NOFOLLOW under a temporary name, then rename) with a run nonce in front -- the
// launcher's pid, MASTER_PORT and TORCHELASTIC_RUN_ID hashed -- and the other ranks only accept a file that carries
// THIS run's nonce, so a file left behind by an earlier run on the same port is never taken for the id.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <thread>
#include <vector>

#include "ntscsim.h"

#define HIPOK(call)                                                                                   \
    do { hipError_t e__ = (call); if (e__ != hipSuccess) {                                            \
        std::fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, hipGetErrorString(e__)); return 1; } } while (0)
#define NCCLOK(call)                                                                                  \
    do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) {                                         \
        std::fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, ncclGetErrorString(r__)); return 1; } } while (0)

static int g_rank = 0;

__global__ void k_bytesum(const uint32_t *__restrict__ p, size_t nwords, unsigned long long *out)
{
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t w = p[i];
        s += (w & 255u) + ((w >> 8) & 255u) + ((w >> 16) & 255u) + (w >> 24);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

static void make_bars(uint8_t *frame, int W, int H, long rot)          // 75 % bars rotated by `rot`
{
    static const uint32_t table[8] = {0xC0C0C0, 0xC0C000, 0x00C0C0, 0x00C000, 0xC000C0, 0xC00000, 0x0000C0, 0x000000};
    uint32_t *row0 = reinterpret_cast<uint32_t *>(frame);
    for (int x = 0; x < W; x++) row0[x] = table[(8 * (int)((x + rot) % W)) / W];
    for (int y = 1; y < H; y++) std::memcpy(frame + (size_t)y * W * 4, frame, (size_t)W * 4);
}

struct Share {                       // one rank's frames, resident on one GPU
    ntscsim_ctx *sim = nullptr;
    ntscsim_batch *batch = nullptr;
    uint8_t *src = nullptr, *dst = nullptr;
    unsigned long long *sum_dev = nullptr;
    size_t nframes = 0, fbytes = 0;
};

// `streams` > 0: BASELINE configs[3] -- this rank owns the streams s = rank, rank + world, ... (< streams), each a clip of
// `total_frames` frames of its own (bars rotated by 37 s + j, fieldno and rand() position from 0); otherwise the frames
// g = rank, rank + world, ... of ONE clip of `total_frames` frames
static int build_share(Share &S, const ntscsim_params &prm, int device, int W, int H, int rank, int world, long total_frames,
                       int streams = 0)
{
    HIPOK(hipSetDevice(device));
    int rc = ntscsim_create(&prm, device, &S.sim);
    if (rc != NTSCSIM_OK) { std::fprintf(stderr, "[rank %d] ntscsim_create: %s\n", g_rank, ntscsim_strerror(rc)); return 1; }
    S.fbytes = (size_t)W * 4 * H;
    std::vector<long> mine, rot;        // global frame number within its clip (-> fieldno, rand() position); bar rotation
    if (streams > 0) {
        for (int sidx = rank; sidx < streams; sidx += world)
            for (long j = 0; j < total_frames; j++) { mine.push_back(j); rot.push_back(37L * sidx + j); }
    } else
        for (long g = rank; g < total_frames; g += world) { mine.push_back(g); rot.push_back(g); }
    S.nframes = mine.size();
    HIPOK(hipMalloc((void **)&S.src, S.fbytes * (S.nframes ? S.nframes : 1)));
    HIPOK(hipMalloc((void **)&S.dst, S.fbytes * (S.nframes ? S.nframes : 1)));
    HIPOK(hipMemset(S.dst, 0, S.fbytes * (S.nframes ? S.nframes : 1)));
    HIPOK(hipMalloc((void **)&S.sum_dev, sizeof(unsigned long long)));
    std::vector<uint8_t> host(S.fbytes);
    const uint64_t c1 = ntscsim_rng_calls_per_field(&prm, W, H, 1), c0 = ntscsim_rng_calls_per_field(&prm, W, H, 0);
    std::vector<ntscsim_field_desc> descs;
    for (size_t j = 0; j < S.nframes; j++) {
        make_bars(host.data(), W, H, rot[j]);
        HIPOK(hipMemcpy(S.src + S.fbytes * j, host.data(), S.fbytes, hipMemcpyHostToDevice));
        for (int sub = 0; sub < 2; sub++) {
            const uint64_t k = 2 * (uint64_t)mine[j] + (uint64_t)sub;
            ntscsim_field_desc d;
            std::memset(&d, 0, sizeof(d));
            d.src_dev = S.src + S.fbytes * j; d.dst_dev = S.dst + S.fbytes * j;
            d.src_linesize = d.dst_linesize = W * 4;
            d.field = (unsigned)((k & 1) ^ 1);                           // ffmpeg_ntsc.cpp:2229
            d.fieldno = k;
            d.rng_pos = (k / 2) * (c1 + c0) + (k & 1) * c1;              // closed form (see the head of this file)
            descs.push_back(d);
        }
    }
    if (!descs.empty()) {
        rc = ntscsim_batch_create(S.sim, descs.data(), (int)descs.size(), W, H, &S.batch);
        if (rc != NTSCSIM_OK) { std::fprintf(stderr, "[rank %d] ntscsim_batch_create: %s (%s)\n", g_rank, ntscsim_strerror(rc), ntscsim_last_error(S.sim)); return 1; }
    }
    return 0;
}

static int run_share(Share &S) { return S.batch ? ntscsim_batch_run(S.batch, nullptr) : NTSCSIM_OK; }

static int checksum_share(Share &S, unsigned long long *out)
{
    HIPOK(hipMemset(S.sum_dev, 0, sizeof(unsigned long long)));
    if (S.nframes) hipLaunchKernelGGL(k_bytesum, dim3(2048), dim3(256), 0, 0, (const uint32_t *)S.dst, S.fbytes * S.nframes / 4, S.sum_dev);
    HIPOK(hipGetLastError());
    HIPOK(hipMemcpy(out, S.sum_dev, sizeof(*out), hipMemcpyDeviceToHost));
    return 0;
}

static void free_share(Share &S)
{
    if (S.batch) ntscsim_batch_destroy(S.batch);
    if (S.sim) ntscsim_destroy(S.sim);
    if (S.src) (void)hipFree(S.src);
    if (S.dst) (void)hipFree(S.dst);
    if (S.sum_dev) (void)hipFree(S.sum_dev);
    S = Share();
}

static int rank_main(int rank, int world, int local_rank, int argc, char **argv, const std::string &id_file)
{
    g_rank = rank;
    long frames = 300, steps = 20, warmup = 5;
    int verify = 1, width = 0, height = 486, one_device = 0, inflight = 4, streams = 0;
    std::string scaling = "weak";
    std::vector<const char *> av;
    av.push_back(argv[0]);
    for (int i = 1; i < argc; i++) {
        auto opt = [&](const char *name) { return !std::strcmp(argv[i], name) && i + 1 < argc; };
        if (opt("--spawn")) { ++i; continue; }
        if (opt("--frames")) { frames = std::atol(argv[++i]); continue; }
        if (opt("--steps")) { steps = std::atol(argv[++i]); continue; }
        if (opt("--warmup")) { warmup = std::atol(argv[++i]); continue; }
        if (opt("--scaling")) { scaling = argv[++i]; continue; }
        if (opt("--verify")) { verify = std::atoi(argv[++i]); continue; }
        if (opt("--width")) { width = std::atoi(argv[++i]); continue; }
        if (opt("--height")) { height = std::atoi(argv[++i]); continue; }
        if (opt("--size")) { if (std::sscanf(argv[++i], "%dx%d", &width, &height) != 2) { std::fprintf(stderr, "--size WxH\n"); return 1; } continue; }
        if (opt("--streams")) { streams = std::atoi(argv[++i]); continue; }
        if (opt("--one-device")) { one_device = std::atoi(argv[++i]); continue; }
        if (opt("--inflight")) { inflight = std::atoi(argv[++i]); continue; }
        av.push_back(argv[i]);
    }
    ntscsim_params prm;
    ntscsim_params_init(&prm);
    prm.output_height = height;
    if (av.size() == 1) av.push_back("-vhs");                            // BASELINE configs[1]
    int rc = ntscsim_params_parse_argv(&prm, nullptr, (int)av.size(), av.data(), 0);
    if (rc != NTSCSIM_OK) return 1;
    if (width > 0) prm.output_width = width;
    const int W = prm.output_width, H = prm.output_height;
    int ndev = 0;
    HIPOK(hipGetDeviceCount(&ndev));
    const int device = one_device ? 0 : local_rank % (ndev > 0 ? ndev : 1);
    HIPOK(hipSetDevice(device));

    // ---- the communicator: rank 0 makes the id, everybody reads it from the file
    struct IdRecord { unsigned long long magic, nonce; ncclUniqueId id; } recid;
    unsigned long long nonce = 1469598103934665603ull;
    {
        auto mix = [&](const char *t) { for (; t && *t; t++) { nonce ^= (unsigned char)*t; nonce *= 1099511628211ull; } };
        mix(std::to_string((long)getppid()).c_str());            // torchrun's agent / the --spawn parent: the same for every local rank
        mix(std::getenv("MASTER_PORT"));
        mix(std::getenv("TORCHELASTIC_RUN_ID"));
        mix(std::getenv("NTSCSIM_RCCL_NONCE"));                  // (several nodes: give every rank the same value)
    }
    ncclUniqueId id;
    if (rank == 0) {
        NCCLOK(ncclGetUniqueId(&id));
        recid.magic = 0x4e54534352434c31ull; recid.nonce = nonce; recid.id = id;
        const std::string tmp = id_file + ".tmp." + std::to_string((long)getpid());
        (void)unlink(tmp.c_str());
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
        if (fd < 0 || write(fd, &recid, sizeof(recid)) != (ssize_t)sizeof(recid)) { std::fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 1; }
        close(fd);
        if (std::rename(tmp.c_str(), id_file.c_str()) != 0) { std::fprintf(stderr, "cannot publish %s\n", id_file.c_str()); return 1; }
    } else {
        bool got = false;
        for (int tries = 0; tries < 6000 && !got; tries++) {
            const int fd = open(id_file.c_str(), O_RDONLY | O_NOFOLLOW);
            if (fd >= 0) {
                // (a file of another run -- an earlier one on the same port that died -- carries another nonce: keep waiting)
                got = read(fd, &recid, sizeof(recid)) == (ssize_t)sizeof(recid) && recid.magic == 0x4e54534352434c31ull && recid.nonce == nonce;
                close(fd);
            }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (!got) { std::fprintf(stderr, "[rank %d] no id file %s of this run\n", rank, id_file.c_str()); return 1; }
        id = recid.id;
    }
    ncclComm_t comm;
    NCCLOK(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t cs;
    HIPOK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    unsigned long long *xbuf = nullptr;                 // [3] mine | [3 * world] gathered | [1] barrier token
    HIPOK(hipMalloc((void **)&xbuf, sizeof(unsigned long long) * (size_t)(3 + 3 * world + 2)));
    HIPOK(hipMemset(xbuf, 0, sizeof(unsigned long long) * (size_t)(3 + 3 * world + 2)));
    unsigned long long *mine_dev = xbuf, *all_dev = xbuf + 3, *tok = xbuf + 3 + 3 * world;
    auto barrier = [&]() -> int {                       // an all-reduce IS a barrier: nobody leaves before everybody arrived
        NCCLOK(ncclAllReduce(tok, tok + 1, 1, ncclUint64, ncclSum, comm, cs));
        HIPOK(hipStreamSynchronize(cs));
        return 0;
    };

    const long total_frames = streams > 0 ? frames : (scaling == "strong" ? frames : frames * world);
    if (inflight < 1) inflight = 1;
    if (inflight > 16) inflight = 16;
    std::vector<Share> Q((size_t)inflight);
    for (auto &q : Q) if (build_share(q, prm, device, W, H, rank, world, total_frames, streams)) return 1;
    Share &S = Q[0];
    for (auto &q : Q) { rc = run_share(q); if (rc != NTSCSIM_OK) return 1; }          // first-call allocations of every context
    for (long i = 0; i < warmup; i++) { rc = run_share(Q[(size_t)(i % inflight)]); if (rc != NTSCSIM_OK) return 1; }
    HIPOK(hipDeviceSynchronize());
    if (barrier()) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (long i = 0; i < steps; i++) {
        rc = run_share(Q[(size_t)(i % inflight)]);
        if (rc != NTSCSIM_OK) { std::fprintf(stderr, "[rank %d] step %ld: %s (%s)\n", rank, i, ntscsim_strerror(rc), ntscsim_last_error(S.sim)); return 1; }
    }
    HIPOK(hipDeviceSynchronize());
    const double mine_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (barrier()) return 1;
    const double wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    // ---- MAX of the elapsed times (all-reduce), then everybody's {checksum, fields per step, elapsed} (all-gather)
    double *dmax = reinterpret_cast<double *>(tok);
    HIPOK(hipMemcpy(dmax, &wall_s, sizeof(double), hipMemcpyHostToDevice));
    NCCLOK(ncclAllReduce(dmax, dmax + 1, 1, ncclDouble, ncclMax, comm, cs));
    HIPOK(hipStreamSynchronize(cs));
    double max_s = 0;
    HIPOK(hipMemcpy(&max_s, dmax + 1, sizeof(double), hipMemcpyDeviceToHost));
    unsigned long long rec[3] = {0, 2ull * S.nframes, 0};
    if (checksum_share(S, &rec[0])) return 1;
    std::memcpy(&rec[2], &mine_s, sizeof(double));
    HIPOK(hipMemcpy(mine_dev, rec, sizeof(rec), hipMemcpyHostToDevice));
    NCCLOK(ncclAllGather(mine_dev, all_dev, 3, ncclUint64, comm, cs));
    HIPOK(hipStreamSynchronize(cs));
    std::vector<unsigned long long> all((size_t)3 * world);
    HIPOK(hipMemcpy(all.data(), all_dev, sizeof(unsigned long long) * all.size(), hipMemcpyDeviceToHost));

    int ret = 0;
    if (rank == 0) {
        unsigned long long fields_per_step = 0;
        for (int r = 0; r < world; r++) fields_per_step += all[(size_t)3 * r + 1];
        // rank 0 re-runs every rank's share on its own GPU: the gathered checksums are not taken on trust
        int verified = -1;
        if (verify) {
            verified = 1;
            for (int r = 0; r < world; r++) {
                unsigned long long cs_r = 0;
                if (r == 0) { if (checksum_share(S, &cs_r)) return 1; }
                else {
                    Share T;
                    if (build_share(T, prm, device, W, H, r, world, total_frames, streams)) return 1;
                    if (run_share(T) != NTSCSIM_OK) return 1;
                    HIPOK(hipDeviceSynchronize());
                    if (checksum_share(T, &cs_r)) return 1;
                    free_share(T);
                }
                if (cs_r != all[(size_t)3 * r]) verified = 0;
            }
        }
        std::printf("{\"metric\": \"frames/sec (output frames = fields; %dx%d, C++ host, one process per GPU, RCCL barrier / max / all-gather)\", "
                    "\"value\": %.1f, \"unit\": \"frames/s\", \"n_gpus\": %d, \"steps\": %ld, \"warmup\": %ld, \"ms_per_step\": %.4f, "
                    "\"higher_is_better\": true, \"scaling\": \"%s\", \"streams\": %d, \"dtype\": \"f64\", \"data\": \"synthetic\", \"steps_in_flight\": %d, "
                    "\"fields_per_step\": %llu, \"collectives\": [\"ncclAllReduce(sum) x2 as barriers\", \"ncclAllReduce(max) of the elapsed time\", "
                    "\"ncclAllGather of {checksum, fields, elapsed}\"], \"rank_checksums_verified\": %s, \"ranks\": [",
                    W, H, max_s > 0 ? (double)fields_per_step * steps / max_s : 0.0, world, steps, warmup, max_s / steps * 1e3,
                    streams > 0 ? "weak" : scaling.c_str(), streams, inflight, fields_per_step, verified < 0 ? "null" : (verified ? "true" : "false"));
        for (int r = 0; r < world; r++) {
            double el; std::memcpy(&el, &all[(size_t)3 * r + 2], sizeof(double));
            std::printf("%s{\"rank\": %d, \"checksum\": %llu, \"fields_per_step\": %llu, \"seconds\": %.6f}", r ? ", " : "", r,
                        all[(size_t)3 * r], all[(size_t)3 * r + 1], el);
        }
        std::printf("]}\n");
        std::fflush(stdout);
        if (verified == 0) ret = 2;
        std::remove(id_file.c_str());
    }
    for (auto &q : Q) free_share(q);
    (void)hipFree(xbuf);
    (void)hipStreamDestroy(cs);
    ncclCommDestroy(comm);
    return ret;
}

int main(int argc, char **argv)
{
    int spawn = 0;
    for (int i = 1; i + 1 < argc; i++) if (!std::strcmp(argv[i], "--spawn")) spawn = std::atoi(argv[i + 1]);
    const char *port = std::getenv("MASTER_PORT");
    std::string id_file = std::getenv("NTSCSIM_RCCL_ID_FILE") ? std::getenv("NTSCSIM_RCCL_ID_FILE")
                          : std::string("/tmp/ntscsim_rccl_id.") + (port ? port : std::to_string((long)(spawn > 0 ? getpid() : getppid())));
    if (spawn > 0) {
        // self-launch: N ranks forked BEFORE anything touches HIP or RCCL (neither survives a fork)
        std::remove(id_file.c_str());
        std::vector<pid_t> kids;
        for (int r = 0; r < spawn; r++) {
            const pid_t p = fork();
            if (p < 0) { std::perror("fork"); return 1; }
            if (p == 0) _exit(rank_main(r, spawn, r, argc, argv, id_file));
            kids.push_back(p);
        }
        int bad = 0;
        for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1; }
        return bad;
    }
    const int rank = std::getenv("RANK") ? std::atoi(std::getenv("RANK")) : 0;
    const int world = std::getenv("WORLD_SIZE") ? std::atoi(std::getenv("WORLD_SIZE")) : 1;
    const int local = std::getenv("LOCAL_RANK") ? std::atoi(std::getenv("LOCAL_RANK")) : rank;
    return rank_main(rank, world, local, argc, argv, id_file);
}
