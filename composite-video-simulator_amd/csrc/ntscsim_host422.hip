// ntscsim_host422.hip -- the YUV422P tool's loop on HOST frames: ntscsim_field422() / ntscsim_submit422()
// (include/ntscsim.h; SURVEY.md 8(b), VERDICT r04 "missing" 1).  Included by ntscsim_hip.hip (one translation unit).
//
// What it replaces: one iteration of do_video_decode_and_render()'s loop, ffmpeg_to_composite.cpp:1783-1800 --
//     render_field(frame, input_frame, field, video_field, tgt_pts);                          :1784
//     if (black_key_level_feedback >= 0) black_key_feedback(frame, filter_frame, field, ..);  :1787
//     if (enable_composite_emulation) composite_video_process(frame, field, video_field);     :1790
//     output_frame(frame, ...)  -- its pixel work, the bob / repack copy :1177-1236           :1793-1796
// -- on the tool's own AVFrames (host memory), synchronously or with `depth` iterations in flight.
//
// The tool works IN PLACE on ONE persistent frame.  Two ways of running an iteration on the device:
//   FAST    the iteration renders its field from a source (render_field rewrites every row of the field, :1076-1128),
//           the luma rows carry two bytes of padding (linesize >= width + 2) and there is no black-key feedback:
//           what the iteration computes then depends on the frame only through the two bytes behind each of its
//           rows that the Y/C separator reads (:496) -- padding nobody writes.  The field gets a device frame of
//           its own (shared with the other field of its pair when that one is still pending), those bytes are
//           snapshotted at submit, and up to `depth` iterations run as ONE batch of ntscsim_fields422_device().
//   SERIAL  everything else (no source, tight rows, black-key feedback -- a frame-to-frame recurrence :974-999 --,
//           an interlaced repack whose other field did not come through the same device frame): the iteration
//           runs alone, in order, on a device MIRROR of the caller's frame that has the caller's own linesizes
//           (so the separator's read past a tight row meets the same bytes as in the tool).
// Both deliver the rows of `field` of the frame (and of the filter frame), and the encoder frame when one was
// given, back into the caller's memory at ntscsim_wait(), in submit order.  Everything runs on the ctx's stream;
// sources are snapshotted into a pinned staging ring by the submitting thread and uploaded on a copy stream.
#include <deque>

namespace {

struct CopyRec422 {               // rows src -> dst on the device side of the link (k422_rows)
    const uint8_t *src;
    uint8_t *dst;
    int32_t sp, dp, rowbytes, nrows;
};
struct PadRec422 {                // the two bytes behind every row of the field (k422_pad)
    uint8_t *y;                   // luma plane of the device frame
    const uint8_t *pad;           // [nrows][2], pinned host memory: the bytes as snapshotted at submit
    const uint8_t *chain;         // TIGHT rows (linesize == width): luma plane of the device frame that holds the OTHER field as
                                  // the previous iteration left it -- the two bytes behind row y are pixels 0, 1 of row y + 1
                                  // (16 behind the frame's last row); NULL: take `pad`
    int32_t ls, W, field, nrows;
    int32_t chain_ls, H;
    int32_t host_ls, _pad;        // the caller's linesize: byte W + j of a row is the row's own padding while W + j < host_ls
                                  // (nobody writes it: the snapshot), pixel W + j - host_ls of the NEXT row otherwise
};

// copy nrows rows of rowbytes bytes, this workgroup the rows k0, k0 + step, ...; records with nrows == 0 are skipped
__device__ __forceinline__ void rows422(const CopyRec422 r, int k0, int step)
{
    for (int k = k0; k < r.nrows; k += step) {
        const uint8_t *s = r.src + (size_t)k * (size_t)r.sp;
        uint8_t *d = r.dst + (size_t)k * (size_t)r.dp;
        if (!(((uintptr_t)s | (uintptr_t)d | (uintptr_t)r.rowbytes) & 3)) {
            const uint32_t *s4 = reinterpret_cast<const uint32_t *>(s);
            uint32_t *d4 = reinterpret_cast<uint32_t *>(d);
            for (int i = threadIdx.x; i < r.rowbytes / 4; i += blockDim.x) d4[i] = s4[i];
        } else
            for (int i = threadIdx.x; i < r.rowbytes; i += blockDim.x) d[i] = s[i];
    }
}

// grid (row chunks, records)
__global__ void k422_rows(const CopyRec422 *__restrict__ recs)
{
    rows422(recs[blockIdx.y], (int)blockIdx.x, (int)gridDim.x);
}

// output_frame's copy AND the rows of the field in one launch -- grid (H + 16 * 6, iterations): the first H workgroups of an
// iteration take a row of k422_output each, the others are k422_rows' (16 row chunks x 6 records)
__global__ void k422_deliver(ntscsim::DevParams P, const ntscsim::Out422Dev *__restrict__ outs, int al4, const CopyRec422 *__restrict__ recs)
{
    const unsigned H = (unsigned)P.H;
    if (blockIdx.x < H) ntscsim::output422_row(P, outs[blockIdx.y], blockIdx.x, al4);
    else {
        const unsigned j = blockIdx.x - H;
        rows422(recs[6u * blockIdx.y + (j >> 4)], (int)(j & 15u), 16);
    }
}

__global__ void k422_pad(const PadRec422 *__restrict__ recs)
{
    const PadRec422 r = recs[blockIdx.x];
    for (int k = threadIdx.x; k < r.nrows; k += blockDim.x) {
        const int y = r.field + 2 * k;
        uint8_t *row = r.y + (size_t)r.ls * (size_t)y;
        uint8_t a = r.pad[2 * k], b = r.pad[2 * k + 1];
        if (r.chain) {
            const uint8_t *nx = r.chain + (size_t)r.chain_ls * (size_t)(y + 1);
            if (r.W >= r.host_ls) a = y + 1 < r.H ? nx[r.W - r.host_ls] : (uint8_t)16;
            if (r.W + 1 >= r.host_ls) b = y + 1 < r.H ? nx[r.W + 1 - r.host_ls] : (uint8_t)16;
        }
        row[r.W] = a;
        row[r.W + 1] = b;
    }
}

} // namespace

struct Host422Engine {
    int depth = 32, nslots = 128;     // configured: iterations per launch, iterations that may be in flight
    int ring_want = 0;                // the request the rings were sized for (ring may be smaller: the byte budget)
    int ring = 0;                     // slots the rings were allocated with (<= nslots: lazily, and capped by a byte budget)
    // geometry of the rings
    int W = 0, H = 0, L = 0;
    int lsd[3] = {0, 0, 0};
    size_t foff[3] = {0, 0, 0}, fbytes = 0;             // device frame of a FAST iteration
    size_t dn_frm = 0, dn_flt = 0, dn_out = 0, dbytes = 0;   // delivery record of one iteration (device and staging)
    size_t sbytes = 0;                                  // capacity of one source slot
    DevBuf<uint8_t> dfrm, dsrc, ddn;
    uint8_t *hsrc = nullptr, *hdn = nullptr;            // pinned staging
    PadRec422 *prec = nullptr;                          // pinned, device-visible, one per slot
    CopyRec422 *crec = nullptr;                         // 6 per slot
    Out422Dev *orec = nullptr;                          // 1 per slot
    uint8_t *pads = nullptr;                            // [nslots][2 * L]
    bool launch_for_wait = false;                       // h422_launch called from h422_wait_ticket: delivery on the ctx's stream
    hipStream_t s_up = nullptr, s_dn = nullptr;
    hipEvent_t ev_k = nullptr;
    // caller frames pinned in place (hipHostRegister, cached): the rows of frame / filter / encoder frame are written by
    // the delivery kernels straight into them -- no staging copy on the caller's thread at ntscsim_wait().
    // NTSCSIM_SUBMIT422_PIN=0: everything through the staging rings.
    PinCache pins;

    struct Mirror {
        const uint8_t *host[3]; int ls[3]; int H;
        DevBuf<uint8_t> dev; size_t off[3];
        bool stale = true;
    };
    std::vector<Mirror *> mirrors;

    struct Item {
        uint64_t ticket = 0;
        int slot = 0, fslot = 0, sslot = -1;
        bool serial = false;
        bool tight = false;               // batched although linesize[0] < width + 2: the pad bytes are chained on the device
        int chain_fslot = -1;             // device frame that holds the other field as the previous iteration left it (-1: none, snapshot)
        uint64_t chain_ticket = 0;        // ... and the ticket of the iteration that wrote it
        ntscsim_loop422 it;
        Mirror *mfrm = nullptr, *mflt = nullptr;
        uint64_t rng_pos = 0;
        uint8_t *frm_dev[3] = {nullptr, nullptr, nullptr};    // device-visible addresses of the caller's planes when pinned
        uint8_t *flt_dev[3] = {nullptr, nullptr, nullptr};
        uint8_t *out_dev[3] = {nullptr, nullptr, nullptr};
        uint8_t *src_dev[3] = {nullptr, nullptr, nullptr};    // ntscsim_field422(): pinned source planes, read in place (no snapshot)
        // set at launch -- how each of the three results reaches the caller: 0 through the staging record (copied at
        // ntscsim_wait), 1 written by the delivery kernels into the pinned frame, 2 not at all (a later iteration of the same
        // launch writes the same rows of the same pinned frame)
        int frm_how = 0, flt_how = 0, out_how = 0;
    };
    std::vector<Item> pending;
    // TIGHT rows: who wrote each field of a caller frame last, and into which device frame (see h422_submit)
    struct Writer { const uint8_t *frame; uint64_t ticket[2]; int fslot[2]; };
    std::vector<Writer> writers;
    struct Batch {
        uint64_t first = 0, last = 0;
        hipEvent_t done = nullptr;
        std::vector<Item> items;
        int rc = NTSCSIM_OK;
        bool launched_ok = false;
        bool posted = false;              // staged results: handed to the copy threads (Delivery), id = `last`
    };
    std::deque<Batch> inflight;
    Delivery dlv;                         // staging ring -> caller frames, off the caller's thread
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev_up = nullptr;
    uint64_t next_ticket = 1, done_ticket = 0;
    int src_cur = -1;
    uint64_t src_ring_pos = 0;
    std::vector<uint64_t> src_last_ticket;
    uint64_t stats_two_pass = 0;           // launches that ran twice (TIGHT rows chained inside the launch)
    uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // [0] submitted [1] launches [2] uploads [3] FAST [4] SERIAL [5] mirror uploads [6] iterations with a result written by the delivery kernels into pinned caller planes
};

static int h422_wait_ticket(ntscsim_ctx *c, uint64_t ticket);
static int h422_launch(ntscsim_ctx *c);

static Host422Engine *h422_get(ntscsim_ctx *c)
{
    if (!c->h422) {
        c->h422 = new (std::nothrow) Host422Engine();
        if (c->h422) {
            const char *ev = std::getenv("NTSCSIM_SUBMIT422_PIN");      // developer A/B: 0 / 1 / 2 = the pin policy
            c->h422->pins.policy = (ev && ev[0] >= '0' && ev[0] <= '2') ? ev[0] - '0' : c->pin_policy;
            c->h422->pins.min_bytes = 64u << 10;        // (the floor of pin_lookup: a 4:2:0 chroma plane of 720x480 is 86 KiB)
        }
    }
    return c->h422;
}

// all three planes of a caller frame pinned in place?  (`rows_c` chroma rows; one plane that cannot be pinned sends
// the whole frame through the staging ring)
static bool h422_pin_frame(ntscsim_ctx *c, Host422Engine *e, const ntscsim_frame422 &f, int W, int H, int rows_c, uint8_t *dev[3])
{
    if (!f.data[0]) return false;
    e->pins.declared = c->declared;
    for (int k = 0; k < 3; k++) {
        const size_t rb = k ? (size_t)W / 2 : (size_t)W, rows = k ? (size_t)rows_c : (size_t)H;
        dev[k] = pin_lookup(e->pins, f.data[k], (size_t)f.linesize[k] * (rows - 1) + rb);
        if (!dev[k]) { dev[0] = dev[1] = dev[2] = nullptr; return false; }
    }
    return true;
}

static void h422_release_rings(Host422Engine *e)
{
    e->dfrm.release(); e->dsrc.release(); e->ddn.release();
    if (e->hsrc) (void)hipHostFree(e->hsrc);
    if (e->hdn) (void)hipHostFree(e->hdn);
    if (e->prec) (void)hipHostFree(e->prec);
    if (e->crec) (void)hipHostFree(e->crec);
    if (e->orec) (void)hipHostFree(e->orec);
    if (e->pads) (void)hipHostFree(e->pads);
    e->hsrc = e->hdn = e->pads = nullptr; e->prec = nullptr; e->crec = nullptr; e->orec = nullptr;
    e->W = e->H = 0; e->sbytes = 0; e->ring = 0; e->ring_want = 0;
    e->src_cur = -1;
    e->src_last_ticket.clear();
}

static void host422_engine_destroy(ntscsim_ctx *c)
{
    Host422Engine *e = c->h422;
    if (!e) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    e->dlv.stop(true);
    for (auto &b : e->inflight) if (b.done) (void)hipEventDestroy(b.done);
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    for (auto *m : e->mirrors) { m->dev.release(); delete m; }
    h422_release_rings(e);
    pin_release_all(e->pins);
    if (e->s_up) (void)hipStreamDestroy(e->s_up);
    if (e->s_dn) (void)hipStreamDestroy(e->s_dn);
    if (e->ev_up) (void)hipEventDestroy(e->ev_up);
    if (e->ev_k) (void)hipEventDestroy(e->ev_k);
    delete e;
    c->h422 = nullptr;
}

static size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// bytes the rings may take, device and pinned host each (NTSCSIM_SUBMIT422_RING_MB overrides the default of 8 GiB)
static size_t h422_ring_budget()
{
    static const size_t b = [] {
        const char *ev = std::getenv("NTSCSIM_SUBMIT422_RING_MB");
        const long long mb = ev ? std::atoll(ev) : 0;
        return mb > 0 ? (size_t)mb << 20 : (size_t)8 << 30;
    }();
    return b;
}

// rings for this geometry; `src_need` = bytes of the largest source frame seen so far; `want` = slots the caller
// needs (the synchronous call: 2; a submit: the configured `nslots`).  The rings grow lazily -- a tool that only ever
// calls ntscsim_field422() never pays for 128 slots -- and never past the byte budget: a geometry of which not even
// two slots fit is refused with NTSCSIM_E_SIZE, a large one runs with fewer iterations in flight.
static int h422_ensure_rings(ntscsim_ctx *c, Host422Engine *e, int W, int H, size_t src_need, int want)
{
    if (want > e->nslots) want = e->nslots;
    if (e->W == W && e->H == H && e->sbytes >= src_need && e->ring > 0 && e->ring_want >= want) return NTSCSIM_OK;
    int rc = h422_wait_ticket(c, NTSCSIM_TICKET_ALL);
    if (rc != NTSCSIM_OK) return rc;
    const size_t keep_src = e->sbytes > src_need ? e->sbytes : src_need;
    if (e->W == W && e->H == H && e->ring_want > want) want = e->ring_want;     // (a larger source keeps the slots it had)
    h422_release_rings(e);
    const int L = (H + 1) / 2, W2 = W / 2;
    e->lsd[0] = (W + 16 + 63) / 64 * 64;
    e->lsd[1] = e->lsd[2] = (W2 + 8 + 31) / 32 * 32;
    e->foff[0] = 0;
    e->foff[1] = up256((size_t)e->lsd[0] * H);
    e->foff[2] = e->foff[1] + up256((size_t)e->lsd[1] * H);
    e->fbytes = e->foff[2] + up256((size_t)e->lsd[2] * H);
    // delivery record: [frame rows of the field: Y L x W, U, V L x W/2][encoder frame: Y H x W, U, V `h422_out_chroma_alloc`
    // rows x W/2 -- 4:2:2 has H chroma rows, 4:2:0 (H + 1) / 2 and the repack's one spare row :1215-1223][filter rows]
    // (in this order: a launch downloads the head of every record -- as far as its iterations filled them -- with ONE
    //  pitched copy; the filter rows, which only the feedback path has, come last)
    e->dn_frm = 0;
    e->dn_out = up256((size_t)L * W * 2);
    e->dn_flt = e->dn_out + up256((size_t)H * W + 2 * (size_t)(H + 1) * W2);
    e->dbytes = e->dn_flt + up256((size_t)L * W * 2);
    e->sbytes = up256(keep_src);
    int ns = want;
    {
        const size_t per_dev = e->fbytes + e->sbytes + e->dbytes, per_host = e->sbytes + e->dbytes + (size_t)2 * L;
        const size_t per = per_dev > per_host ? per_dev : per_host;
        const size_t fit = h422_ring_budget() / per;
        if (fit < 2) { c->err = "ntscsim_submit422: two ring slots of this geometry exceed the ring budget"; e->ring = 0; return NTSCSIM_E_SIZE; }
        if ((size_t)ns > fit) ns = (int)fit;
    }
    e->ring = 0;
    HIPCHK(c, e->dfrm.ensure(e->fbytes * (size_t)ns));
    HIPCHK(c, e->dsrc.ensure(e->sbytes * (size_t)ns));
    HIPCHK(c, e->ddn.ensure(e->dbytes * (size_t)ns));
    // (device frames start zeroed: rows of the other field that no iteration ever rendered read as 0 in an
    //  interlaced repack's source -- a case the FAST path does not take, see h422_classify)
    HIPCHK(c, hipMemsetAsync(e->dfrm.p, 0, e->fbytes * (size_t)ns, c->stream));
    HIPCHK(c, hipHostMalloc((void **)&e->hsrc, e->sbytes * (size_t)ns, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&e->hdn, e->dbytes * (size_t)ns, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&e->prec, sizeof(PadRec422) * (size_t)ns, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&e->crec, sizeof(CopyRec422) * 6 * (size_t)ns, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&e->orec, sizeof(Out422Dev) * (size_t)ns, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&e->pads, (size_t)2 * L * (size_t)ns, hipHostMallocDefault));
    if (!e->s_up) HIPCHK(c, hipStreamCreateWithFlags(&e->s_up, hipStreamNonBlocking));
    if (!e->s_dn) HIPCHK(c, hipStreamCreateWithFlags(&e->s_dn, hipStreamNonBlocking));
    if (!e->ev_up) HIPCHK(c, hipEventCreateWithFlags(&e->ev_up, hipEventDisableTiming));
    if (!e->ev_k) HIPCHK(c, hipEventCreateWithFlags(&e->ev_k, hipEventDisableTiming));
    e->W = W; e->H = H; e->L = L; e->ring = ns; e->ring_want = want;
    e->src_last_ticket.assign((size_t)ns, 0);
    e->src_ring_pos = 0;
    e->src_cur = -1;
    return NTSCSIM_OK;
}

static int h422_field_rows(int H, unsigned field) { return H > (int)field ? (H - (int)field + 1) / 2 : 0; }

// chroma rows of the encoder frame that output_frame() writes inside the plane (:1177-1236)
static int h422_out_chroma_rows(int H, uint32_t mode) { return (mode == NTSCSIM_OUT422_BOB422 || mode == NTSCSIM_OUT422_FRAME) ? H : (H + 1) / 2; }

// ... and the rows the record keeps per chroma plane (the interlaced repack writes one row past a 4:2:0 plane for a
// height of 2 mod 4, :1215-1223: it has room here and is not delivered)
static int h422_out_chroma_alloc(int H, uint32_t mode) { return h422_out_chroma_rows(H, mode) + 1; }
static size_t h422_out_bytes(int W, int H, uint32_t mode) { return (size_t)W * H + 2 * (size_t)(W / 2) * (size_t)h422_out_chroma_alloc(H, mode); }

// (`keep`: a mirror the caller already holds for the same iteration -- it survives the eviction below)
static Host422Engine::Mirror *h422_mirror(ntscsim_ctx *c, Host422Engine *e, const ntscsim_frame422 &f, int H,
                                          Host422Engine::Mirror *keep = nullptr)
{
    for (auto *m : e->mirrors)
        if (m->host[0] == f.data[0] && m->host[1] == f.data[1] && m->host[2] == f.data[2] && m->ls[0] == f.linesize[0] &&
            m->ls[1] == f.linesize[1] && m->ls[2] == f.linesize[2] && m->H == H) return m;
    if (e->mirrors.size() >= 64) {           // (a tool has one frame and one filter frame; keep the table bounded)
        if (h422_wait_ticket(c, NTSCSIM_TICKET_ALL) != NTSCSIM_OK) return nullptr;
        (void)hipStreamSynchronize(c->stream);
        for (auto *m : e->mirrors) if (m != keep) { m->dev.release(); delete m; }
        e->mirrors.clear();
        if (keep) e->mirrors.push_back(keep);
    }
    auto *m = new (std::nothrow) Host422Engine::Mirror();
    if (!m) return nullptr;
    for (int k = 0; k < 3; k++) { m->host[k] = f.data[k]; m->ls[k] = f.linesize[k]; }
    m->H = H;
    m->off[0] = 0;
    m->off[1] = up256((size_t)m->ls[0] * H + 16);
    m->off[2] = m->off[1] + up256((size_t)m->ls[1] * H + 16);
    if (m->dev.ensure(m->off[2] + up256((size_t)m->ls[2] * H + 16)) != hipSuccess) {
        c->err = "hipMalloc failed (frame mirror)";
        delete m;
        return nullptr;
    }
    m->stale = true;
    e->mirrors.push_back(m);
    return m;
}

// bring a mirror up to date with the caller's frame: everything in flight is delivered first
static int h422_refresh_mirror(ntscsim_ctx *c, Host422Engine *e, Host422Engine::Mirror *m, int W)
{
    if (!m->stale) return NTSCSIM_OK;
    int rc = h422_wait_ticket(c, NTSCSIM_TICKET_ALL);
    if (rc != NTSCSIM_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 3; k++) {
        // the plane as the tool's contract has it: height * linesize bytes, of which the last row's are only
        // touched as far as the separator reads (width + 2) / the pixels reach
        const size_t need = k == 0 ? (size_t)W + 2 : (size_t)W / 2;
        const size_t last = need < (size_t)m->ls[k] ? need : (size_t)m->ls[k];
        const size_t bytes = (size_t)m->ls[k] * (size_t)(m->H - 1) + last;
        for (size_t off = 0; off < bytes; off += (1u << 20)) {
            const size_t n = bytes - off < (1u << 20) ? bytes - off : (1u << 20);
            const hipError_t er = upload_table(m->dev.p + m->off[k] + off, m->host[k] + off, n);
            if (er != hipSuccess) { c->err = std::string("mirror upload: ") + hipGetErrorString(er); return NTSCSIM_E_HIP; }
        }
    }
    m->stale = false;
    e->stats[5]++;
    return NTSCSIM_OK;
}

// what the copy threads do for one launch: the staged results of its iterations, staging record -> caller planes
static void h422_delivery_ops(const Host422Engine *e, const Host422Engine::Batch &b, std::vector<CopyOp> &ops)
{
    const int W = e->W, H = e->H, W2 = W / 2;
    for (const auto &it : b.items) {
        const uint8_t *st = e->hdn + e->dbytes * (size_t)it.slot;
        const int n = h422_field_rows(H, it.it.field);
        auto rows_out = [&](const ntscsim_frame422 &f, const uint8_t *s) {
            for (int k = 0; k < 3; k++) {
                const size_t rb = k ? (size_t)W2 : (size_t)W;
                if (n > 0) ops.push_back({f.data[k] + (size_t)f.linesize[k] * it.it.field, s, 2 * (size_t)f.linesize[k], rb, rb, n});
                s += rb * (size_t)n;
            }
        };
        if (it.frm_how == 0) rows_out(it.it.frame, st + e->dn_frm);
        if (it.serial && it.mflt && it.flt_how == 0) rows_out(it.it.filter, st + e->dn_flt);
        if (it.it.out.data[0] && it.out_how == 0) {
            const uint8_t *s = st + e->dn_out;
            const int ch = h422_out_chroma_rows(H, it.it.out_mode);
            for (int k = 0; k < 3; k++) {
                const size_t rb = k ? (size_t)W2 : (size_t)W;
                const int nr = k ? ch : H;
                // (luma in two halves: the ops of a launch are the unit the copy threads share out)
                if (k == 0 && nr >= 64) {
                    const int h0 = nr / 2;
                    ops.push_back({it.it.out.data[0], s, (size_t)it.it.out.linesize[0], rb, rb, h0});
                    ops.push_back({it.it.out.data[0] + (size_t)it.it.out.linesize[0] * h0, s + rb * (size_t)h0, (size_t)it.it.out.linesize[0], rb, rb, nr - h0});
                } else
                    ops.push_back({it.it.out.data[k], s, (size_t)it.it.out.linesize[k], rb, rb, nr});
                s += rb * (size_t)(k ? h422_out_chroma_alloc(H, it.it.out_mode) : H);
            }
        }
    }
}

static int h422_retire_front(ntscsim_ctx *c, Host422Engine *e)
{
    Host422Engine::Batch &b = e->inflight.front();
    int rc = b.rc;
    if (b.launched_ok) {
        bool ok;
        if (b.posted) ok = e->dlv.wait(b.last);          // (the copy threads synchronised on the event)
        else ok = hipEventSynchronize(b.done) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); c->err = "submit422: a launch failed on the device (hipEventSynchronize)"; rc = NTSCSIM_E_HIP; }
    }
    e->done_ticket = b.last;
    if (b.done) e->ev_pool.push_back(b.done);
    e->inflight.pop_front();
    return rc;
}

static int h422_wait_ticket(ntscsim_ctx *c, uint64_t ticket)
{
    Host422Engine *e = c->h422;
    if (!e) return ticket == NTSCSIM_TICKET_ALL ? NTSCSIM_OK : NTSCSIM_E_ARG;
    if (ticket == NTSCSIM_TICKET_ALL) ticket = e->next_ticket - 1;
    if (ticket == 0) return NTSCSIM_OK;
    if (ticket >= e->next_ticket) return NTSCSIM_E_ARG;
    int rc = NTSCSIM_OK;
    if (!e->pending.empty() && ticket >= e->pending.front().ticket) {
        static const bool own_stream = std::getenv("NTSCSIM_SUBMIT422_DLVSTREAM") && std::getenv("NTSCSIM_SUBMIT422_DLVSTREAM")[0] == '1';   // developer A/B
        e->launch_for_wait = !own_stream;
        const int r = h422_launch(c);
        e->launch_for_wait = false;
        if (r != NTSCSIM_OK) rc = r;
    }
    while (!e->inflight.empty() && e->inflight.front().first <= ticket) {
        const int r = h422_retire_front(c, e);
        if (r != NTSCSIM_OK && rc == NTSCSIM_OK) rc = r;
    }
    return rc;
}

// Enqueue the pending iterations (all FAST, or one SERIAL) as one launch on the ctx's stream.
static int h422_launch(ntscsim_ctx *c)
{
    Host422Engine *e = c->h422;
    if (!e || e->pending.empty()) return NTSCSIM_OK;
    Host422Engine::Batch b;
    b.first = e->pending.front().ticket;
    b.last = e->pending.back().ticket;
    b.items.swap(e->pending);
    e->pending.clear();
    const int n = (int)b.items.size();
    auto finish = [&](int rc) {
        b.rc = rc;
        e->inflight.push_back(std::move(b));
        return rc;
    };
    if (!e->ev_pool.empty()) { b.done = e->ev_pool.back(); e->ev_pool.pop_back(); }
    else if (hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) {
        c->err = "hipEventCreate failed";
        return finish(NTSCSIM_E_HIP);
    }
    const int W = e->W, H = e->H, W2 = W / 2;
    hipStream_t st = c->stream;
    // uploads of this launch's sources have been enqueued on the copy stream
    // (the ctx's stream waits for them in front of its first kernel that reads pixels -- launch422: the records and the
    //  per-field / per-row draws run beside the upload; the event is recorded below, behind the padding bytes)
    hipError_t er = hipSuccess;
    static const bool early_wait = std::getenv("NTSCSIM_SUBMIT422_EARLYWAIT") && std::getenv("NTSCSIM_SUBMIT422_EARLYWAIT")[0] == '1';   // developer A/B

    // Delivery by the kernels into pinned caller frames.  The iterations of a launch run concurrently, so of several that
    // write the same rows of the same caller frame (ONE persistent frame: every second field) only the LAST delivers them
    // -- the frame then holds what the in-order loop leaves after this launch; launches are stream-ordered among themselves.
    for (int i = 0; i < n; i++) {
        Host422Engine::Item &it = b.items[(size_t)i];
        it.frm_how = it.frm_dev[0] ? 1 : 0; it.flt_how = it.flt_dev[0] ? 1 : 0; it.out_how = it.out_dev[0] ? 1 : 0;
    }
    for (int i = 0; i < n; i++) {
        Host422Engine::Item &a = b.items[(size_t)i];
        for (int j = i + 1; j < n; j++) {
            const Host422Engine::Item &z = b.items[(size_t)j];
            // (pinned or staged alike: a frame is one or the other as a whole)
            if (a.frm_how != 2 && z.it.frame.data[0] == a.it.frame.data[0] && z.it.field == a.it.field) a.frm_how = 2;
            if (a.flt_how != 2 && a.it.filter.data[0] && z.it.filter.data[0] == a.it.filter.data[0] && z.it.field == a.it.field) a.flt_how = 2;
            if (a.out_how != 2 && a.it.out.data[0] && z.it.out.data[0] == a.it.out.data[0]) a.out_how = 2;
        }
    }
    for (int i = 0; i < n; i++) {
        const Host422Engine::Item &it = b.items[(size_t)i];
        if (it.frm_how == 1 || it.out_how == 1 || it.flt_how == 1) e->stats[6]++;
    }
    std::vector<ntscsim_field422_desc> descs((size_t)n);
    bool any_pad = false, any_out = false, any_chain = false;
    // TIGHT rows whose other field is written by an iteration of THIS launch: the bytes behind their rows -- pixels 0, 1 of
    // the neighbouring rows -- exist only after that iteration has run.  Nothing an iteration computes for pixels 0, 1
    // of a row depends on the bytes behind ANY row (the separator's read :496 reaches the last positions of a row only;
    // every later stage is causal in x up to the filters' few samples of delay), so: run the launch once, copy the bytes,
    // run it again -- the second pass sees exactly what the in-order loop sees.  The GPU has the time (a launch is
    // ~0.1 ms of kernels against ~1 ms of link traffic).
    bool two_pass = false;
    bool al4 = (W2 & 3) == 0;
    size_t dn_need = e->dn_out;                      // bytes at the head of every delivery record that this launch fills
    bool any_staged = false, flt_staged = false;     // anything for the staging ring at all?
    for (int i = 0; i < n; i++) {
        const Host422Engine::Item &it = b.items[(size_t)i];
        const ntscsim_loop422 &L = it.it;
        ntscsim_field422_desc &d = descs[(size_t)i];
        std::memset(&d, 0, sizeof(d));
        uint8_t *frm[3]; int fls[3];
        if (it.serial) for (int k = 0; k < 3; k++) { frm[k] = it.mfrm->dev.p + it.mfrm->off[k]; fls[k] = it.mfrm->ls[k]; }
        else for (int k = 0; k < 3; k++) { frm[k] = e->dfrm.p + e->fbytes * (size_t)it.fslot + e->foff[k]; fls[k] = e->lsd[k]; }
        const bool c420 = (L.flags & NTSCSIM_422_SRC420) != 0;
        const size_t src_crows = c420 ? ((size_t)L.src_height + 1) / 2 : (size_t)L.src_height;
        for (int k = 0; k < 3; k++) {
            al4 = al4 && !(((uintptr_t)frm[k] | (uintptr_t)fls[k]) & 3);
            d.dst_dev[k] = frm[k]; d.dst_linesize[k] = fls[k];
            if (it.sslot >= 0) {
                uint8_t *s = e->dsrc.p + e->sbytes * (size_t)it.sslot;
                d.src_dev[k] = k == 0 ? s : s + (size_t)W * L.src_height + (k == 2 ? (size_t)W2 * src_crows : 0);
                d.src_linesize[k] = k ? W2 : W;
            } else if (it.src_dev[0]) { d.src_dev[k] = it.src_dev[k]; d.src_linesize[k] = L.src.linesize[k]; }
            if (it.mflt) { d.flt_dev[k] = it.mflt->dev.p + it.mflt->off[k]; d.flt_linesize[k] = it.mflt->ls[k]; }
        }
        d.src_height = L.src_height;
        d.field = L.field;
        d.flags = L.flags;
        d.fieldno = L.fieldno;
        d.rng_pos = it.rng_pos;
        const int nr = h422_field_rows(H, L.field);
        // the separator's two bytes behind each row of the field (FAST: the device frame's rows are wider than the
        // caller's; the bytes were snapshotted at submit)
        PadRec422 &p = e->prec[it.slot];
        p.y = frm[0]; p.pad = e->pads + (size_t)2 * e->L * (size_t)it.slot; p.ls = fls[0]; p.W = W;
        p.chain = (it.tight && it.chain_fslot >= 0) ? e->dfrm.p + e->fbytes * (size_t)it.chain_fslot + e->foff[0] : nullptr;
        p.chain_ls = e->lsd[0]; p.H = H; p.host_ls = L.frame.linesize[0]; p._pad = 0;
        if (it.tight && it.chain_fslot >= 0 && it.chain_ticket >= b.first) two_pass = true;
        any_chain = any_chain || p.chain != nullptr;
        p.field = (int32_t)L.field; p.nrows = it.serial ? 0 : nr;
        any_pad = any_pad || !it.serial;
        // delivery: the field's rows of the frame (and of the filter frame) -> the delivery record
        uint8_t *dn = e->ddn.p + e->dbytes * (size_t)it.slot;
        CopyRec422 *cr = e->crec + 6 * (size_t)it.slot;
        size_t o = e->dn_frm;
        for (int k = 0; k < 3; k++) {
            const int rb = k ? W2 : W;
            const uint8_t *from = frm[k] + (size_t)fls[k] * L.field;
            if (it.frm_how == 1) cr[k] = {from, it.frm_dev[k] + (size_t)L.frame.linesize[k] * L.field, 2 * fls[k], 2 * L.frame.linesize[k], rb, nr};
            else if (it.frm_how == 0) { cr[k] = {from, dn + o, 2 * fls[k], rb, rb, nr}; any_staged = true; }
            else cr[k] = {nullptr, nullptr, 0, 0, 0, 0};
            o += (size_t)rb * nr;
        }
        o = e->dn_flt;
        for (int k = 0; k < 3; k++) {
            const int rb = k ? W2 : W;
            if (it.mflt && it.flt_how != 2) {
                const uint8_t *from = it.mflt->dev.p + it.mflt->off[k] + (size_t)it.mflt->ls[k] * L.field;
                if (it.flt_how == 1) cr[3 + k] = {from, it.flt_dev[k] + (size_t)L.filter.linesize[k] * L.field, 2 * it.mflt->ls[k], 2 * L.filter.linesize[k], rb, nr};
                else { cr[3 + k] = {from, dn + o, 2 * it.mflt->ls[k], rb, rb, nr}; flt_staged = true; }
            } else cr[3 + k] = {nullptr, nullptr, 0, 0, 0, 0};
            o += (size_t)rb * nr;
        }
        // output_frame's copy, straight into the delivery record
        Out422Dev &orc = e->orec[it.slot];
        std::memset(&orc, 0, sizeof(orc));
        orc.mode = 0xFFFFFFFFu;            // = none
        if (L.out.data[0] && it.out_how != 2) {
            any_out = true;
            for (int k = 0; k < 3; k++) {
                orc.frame[k] = frm[k]; orc.frame_ls[k] = fls[k];
                if (it.out_how == 1) { orc.bob[k] = it.out_dev[k]; orc.bob_ls[k] = L.out.linesize[k]; }
                else {
                    orc.bob[k] = dn + e->dn_out + (k == 0 ? 0 : (size_t)W * H + (k == 2 ? (size_t)W2 * (size_t)h422_out_chroma_alloc(H, L.out_mode) : 0));
                    orc.bob_ls[k] = k ? W2 : W;
                }
                al4 = al4 && !(((uintptr_t)orc.bob[k] | (uintptr_t)orc.bob_ls[k]) & 3);
            }
            orc.field = L.out_field; orc.mode = L.out_mode;
            // (the caller's planes end where they end: the repack's row past a 4:2:0 plane is not written there)
            orc.crows = it.out_how == 1 ? (uint32_t)h422_out_chroma_rows(H, L.out_mode) : 0u;
            if (it.out_how == 0) {
                any_staged = true;
                const size_t end = e->dn_out + h422_out_bytes(W, H, L.out_mode);
                if (end > dn_need) dn_need = end;
            }
        }
    }
    // slot runs: tickets are consecutive, slots = ticket mod nslots -> at most two runs
    const int s0 = b.items.front().slot;
    const int run0 = s0 + n <= e->ring ? n : e->ring - s0;
    // The padding bytes come out of the submit-time snapshot (pinned host memory): on the UPLOAD stream, behind the source
    // and beside the draws -- unless a record chains on a device frame that kernels of the ctx's stream write (TIGHT rows).
    const bool pad_up = any_pad && !any_chain && !early_wait;
    if (pad_up) {
        hipLaunchKernelGGL(k422_pad, dim3((unsigned)run0), dim3(256), 0, e->s_up, e->prec + s0);
        if (run0 < n) hipLaunchKernelGGL(k422_pad, dim3((unsigned)(n - run0)), dim3(256), 0, e->s_up, e->prec);
    }
    er = hipEventRecord(e->ev_up, e->s_up);
    if (er == hipSuccess && early_wait) er = hipStreamWaitEvent(st, e->ev_up, 0);
    if (er != hipSuccess) { c->err = std::string("submit422 launch: ") + hipGetErrorString(er); return finish(NTSCSIM_E_HIP); }
    if (any_pad && !pad_up) {
        hipLaunchKernelGGL(k422_pad, dim3((unsigned)run0), dim3(256), 0, st, e->prec + s0);
        if (run0 < n) hipLaunchKernelGGL(k422_pad, dim3((unsigned)(n - run0)), dim3(256), 0, st, e->prec);
    }
    const uint64_t keep_pos = c->rng_pos;           // the stream advanced at submit; descriptors carry positions
    c->latency_form = true;             // (short launches take the role form of the streamed kernels: k422_pipe)
    c->wait422_ev = early_wait ? nullptr : e->ev_up;
    int rc = ntscsim_fields422_device(c, descs.data(), n, W, H, st);
    c->latency_form = false;
    c->rng_pos = keep_pos;
    if (c->wait422_ev) {                // (the call failed before its launches: nothing later on this stream may overtake the upload)
        c->wait422_ev = nullptr;
        (void)hipStreamWaitEvent(st, e->ev_up, 0);
    }
    if (rc != NTSCSIM_OK) return finish(rc);
    if (two_pass) {
        hipLaunchKernelGGL(k422_pad, dim3((unsigned)run0), dim3(256), 0, st, e->prec + s0);
        if (run0 < n) hipLaunchKernelGGL(k422_pad, dim3((unsigned)(n - run0)), dim3(256), 0, st, e->prec);
        c->latency_form = true;
        rc = ntscsim_fields422_device(c, descs.data(), n, W, H, st);
        c->latency_form = false;
        c->rng_pos = keep_pos;
        if (rc != NTSCSIM_OK) return finish(rc);
        e->stats_two_pass++;
    }
    // Delivery -- output_frame's copy, the rows of the field, the download of what is staged -- runs on a stream of its own
    // behind the batch's kernels: the NEXT launch's kernels do not wait for it (batched iterations own their device frames
    // and delivery records).  An iteration on a mirror is different: the next one modifies the rows this one delivers from,
    // so the ctx's stream waits for its delivery (below).
    // (a launch made because the caller is about to wait for it -- the synchronous call, a wait on a pending ticket --
    //  delivers on the ctx's stream itself: nothing comes behind it, and the hop between streams costs ~10 us)
    hipStream_t sd = e->launch_for_wait ? st : e->s_dn;
    if (sd != st) {
        er = hipEventRecord(e->ev_k, st);
        if (er == hipSuccess) er = hipStreamWaitEvent(sd, e->ev_k, 0);
        if (er != hipSuccess) { c->err = std::string("submit422 launch: ") + hipGetErrorString(er); return finish(NTSCSIM_E_HIP); }
    }
    DevParams D;
    std::memset(&D, 0, sizeof(D));
    D.W = W; D.H = H; D.variant = 1;
    static const bool one_deliver = !(std::getenv("NTSCSIM_SUBMIT422_DELIVER1") && std::getenv("NTSCSIM_SUBMIT422_DELIVER1")[0] == '0');   // developer A/B
    if (any_out && one_deliver) {
        // (the two copies read the same device frame and write different caller frames: one launch, side by side)
        hipLaunchKernelGGL(k422_deliver, dim3((unsigned)H + 96u, (unsigned)run0), dim3(256), 0, sd, D, e->orec + s0, al4 ? 1 : 0, e->crec + 6 * (size_t)s0);
        if (run0 < n) hipLaunchKernelGGL(k422_deliver, dim3((unsigned)H + 96u, (unsigned)(n - run0)), dim3(256), 0, sd, D, e->orec, al4 ? 1 : 0, e->crec);
    } else {
    if (any_out) {
        hipLaunchKernelGGL(k422_output, dim3((unsigned)H, (unsigned)run0), dim3(256), 0, sd, D, e->orec + s0, al4 ? 1 : 0);
        if (run0 < n) hipLaunchKernelGGL(k422_output, dim3((unsigned)H, (unsigned)(n - run0)), dim3(256), 0, sd, D, e->orec, al4 ? 1 : 0);
    }
    hipLaunchKernelGGL(k422_rows, dim3(16, (unsigned)(6 * run0)), dim3(256), 0, sd, e->crec + 6 * (size_t)s0);
    if (run0 < n) hipLaunchKernelGGL(k422_rows, dim3(16, (unsigned)(6 * (n - run0))), dim3(256), 0, sd, e->crec);
    }
    er = hipGetLastError();
    if (er != hipSuccess) { c->err = std::string("submit422 kernels: ") + hipGetErrorString(er); return finish(NTSCSIM_E_HIP); }
    if (flt_staged) { dn_need = e->dbytes; any_staged = true; }
    dn_need = (dn_need + 255) / 256 * 256;
    if (!any_staged) dn_need = 0;                    // everything went straight into pinned caller frames
    if (dn_need)
        er = hipMemcpy2DAsync(e->hdn + e->dbytes * (size_t)s0, e->dbytes, e->ddn.p + e->dbytes * (size_t)s0, e->dbytes, dn_need,
                              (size_t)run0, hipMemcpyDeviceToHost, sd);
    if (er == hipSuccess && run0 < n && dn_need)
        er = hipMemcpy2DAsync(e->hdn, e->dbytes, e->ddn.p, e->dbytes, dn_need, (size_t)(n - run0), hipMemcpyDeviceToHost, sd);
    if (er == hipSuccess) er = hipEventRecord(b.done, sd);
    if (er == hipSuccess && b.items.front().serial && sd != st) er = hipStreamWaitEvent(st, b.done, 0);
    if (er != hipSuccess) { c->err = std::string("submit422 D2H: ") + hipGetErrorString(er); return finish(NTSCSIM_E_HIP); }
    b.launched_ok = true;
    e->stats[1]++;
    // (the synchronous call: the next call's setup kernel behind the event this one waits for -- ntscsim_hip.hip)
    if (e->launch_for_wait && n == 1 && sd == st) speculate_setup(c, st);
    if (any_staged) {
        std::vector<CopyOp> ops;
        h422_delivery_ops(e, b, ops);
        e->dlv.post(c->device, b.done, std::move(ops), b.last);
        b.posted = true;
    }
    return finish(NTSCSIM_OK);
}

static int h422_validate(const ntscsim_ctx *c, const ntscsim_loop422 *L)
{
    if (!c || !L || L->struct_size != sizeof(ntscsim_loop422)) return NTSCSIM_E_ARG;
    const int W = L->width, H = L->height;
    if (W < 16 || (W & 1) || H < 2 || W > 16384 || H > 16384) return NTSCSIM_E_SIZE;
    if (L->field > 1 || L->out_field > 1) return NTSCSIM_E_ARG;
    if (L->flags & ~(NTSCSIM_422_INTERLACED | NTSCSIM_422_TFF | NTSCSIM_422_SRC420 | NTSCSIM_422_SECOND | NTSCSIM_422_NOCOMP))
        return NTSCSIM_E_ARG;
    auto planes = [&](const ntscsim_frame422 &f, int rows_c_min) -> int {
        (void)rows_c_min;
        for (int k = 0; k < 3; k++) {
            if (!f.data[k]) return NTSCSIM_E_ARG;
            if (f.linesize[k] < (k ? W / 2 : W)) return NTSCSIM_E_SIZE;
        }
        return NTSCSIM_OK;
    };
    int rc = planes(L->frame, H);
    if (rc != NTSCSIM_OK) return rc;
    if (L->src.data[0]) {
        rc = planes(L->src, 0);
        if (rc != NTSCSIM_OK) return rc;
        if (L->src_height < ((L->flags & NTSCSIM_422_INTERLACED) ? 4 : 2) || L->src_height > 16384) return NTSCSIM_E_SIZE;
    }
    if (L->filter.data[0]) { rc = planes(L->filter, H); if (rc != NTSCSIM_OK) return rc; }
    if (L->out.data[0]) {
        rc = planes(L->out, 0);
        if (rc != NTSCSIM_OK) return rc;
        if (L->out_mode > NTSCSIM_OUT422_FRAME) return NTSCSIM_E_ARG;
    }
    return NTSCSIM_OK;
}

static int h422_submit(ntscsim_ctx *c, const ntscsim_loop422 *L, uint32_t flags, uint64_t *ticket, bool sync_call)
{
    int rc = h422_validate(c, L);
    if (rc != NTSCSIM_OK) return rc;
    if (flags & ~(NTSCSIM_SUBMIT_SAME_SRC | NTSCSIM_SUBMIT422_DIRTY)) return NTSCSIM_E_ARG;
    Host422Engine *e = h422_get(c);
    if (!e) return NTSCSIM_E_NOMEM;
    HIPCHK(c, hipSetDevice(c->device));
    const int W = L->width, H = L->height, W2 = W / 2;
    const bool have_src = L->src.data[0] != nullptr;
    const bool c420 = (L->flags & NTSCSIM_422_SRC420) != 0;
    const size_t src_crows = c420 ? ((size_t)L->src_height + 1) / 2 : (size_t)L->src_height;
    const size_t src_bytes = have_src ? (size_t)W * L->src_height + 2 * (size_t)W2 * src_crows : 0;
    rc = h422_ensure_rings(c, e, W, H, src_bytes, sync_call ? 2 : e->nslots);
    if (rc != NTSCSIM_OK) return rc;

    if (flags & NTSCSIM_SUBMIT422_DIRTY) {
        // the caller wrote into frame / filter: deliver what is in flight (it lands in those frames), read them again
        rc = h422_wait_ticket(c, NTSCSIM_TICKET_ALL);
        if (rc != NTSCSIM_OK) return rc;
        for (auto *m : e->mirrors)
            if (m->host[0] == L->frame.data[0] || m->host[0] == L->filter.data[0]) m->stale = true;
    }
    // ring space: ticket t uses slot t mod nslots; its previous user and that one's pair partner must have retired
    const uint64_t t = e->next_ticket;
    if (t + 1 > (uint64_t)e->ring && e->done_ticket < t + 1 - (uint64_t)e->ring) {
        e->stats[7]++;
        rc = h422_wait_ticket(c, t + 1 - (uint64_t)e->ring);
        if (rc != NTSCSIM_OK) return rc;
    }
    const bool bkey = c->prm.black_key_level_feedback >= 0 && L->filter.data[0] != nullptr;
    // ---- FAST or SERIAL (see the head of this file)
    // TIGHT rows (linesize[0] < width + 2: av_frame_get_buffer(f, 32) gives them to every width that is a multiple of 32):
    // batched like padded ones, the two bytes behind each row chained on the device (h422_launch).  Narrow frames keep
    // the in-order path: the argument that pixels 0, 1 of a row never depend on the bytes behind a row wants the filter
    // delays (a dozen samples per Y/C pass) to be small against the width.
    const bool tight = L->frame.linesize[0] < W + 2;
    const bool chain_ok = tight && have_src && !bkey && W >= 128 && !(std::getenv("NTSCSIM_SUBMIT422_NOCHAIN"));
    bool serial = !have_src || bkey || (tight && !chain_ok);
    int fslot = (int)(t % (uint64_t)e->ring);
    bool paired = false;
    if (!serial && !e->pending.empty()) {
        const Host422Engine::Item &pv = e->pending.back();
        if (!pv.serial && pv.ticket + 1 == t && pv.it.field != L->field && pv.fslot == pv.slot &&
            pv.it.frame.data[0] == L->frame.data[0] && pv.it.frame.data[1] == L->frame.data[1] &&
            pv.it.frame.data[2] == L->frame.data[2] && pv.it.frame.linesize[0] == L->frame.linesize[0]) {
            fslot = pv.fslot;
            paired = true;
        }
    }
    // the interlaced repack reads BOTH fields of the frame (:1202-1223): only exact on a device frame that holds
    // the other field as well
    // (likewise a bob of the OTHER field's rows: -vi hands output_frame the previous field's parity :1792-1793)
    if (!serial && L->out.data[0] && (L->out_mode == NTSCSIM_OUT422_INTERLACED420 || L->out_mode == NTSCSIM_OUT422_FRAME || L->out_field != L->field) && !paired)
        serial = true;

    Host422Engine::Item it;
    it.ticket = t;
    it.slot = (int)(t % (uint64_t)e->ring);
    it.fslot = fslot;
    it.serial = serial;
    it.tight = tight && !serial;
    it.it = *L;
    {
        // who wrote each field of this caller frame last (TIGHT rows chain on it; any iteration updates it)
        Host422Engine::Writer *wr = nullptr;
        for (auto &x : e->writers) if (x.frame == L->frame.data[0]) { wr = &x; break; }
        if (!wr) {
            if (e->writers.size() >= 64) e->writers.erase(e->writers.begin());
            e->writers.push_back({L->frame.data[0], {0, 0}, {-1, -1}});
            wr = &e->writers.back();
        }
        if (flags & NTSCSIM_SUBMIT422_DIRTY) { wr->ticket[0] = wr->ticket[1] = 0; }
        const unsigned other = L->field ^ 1u;
        if (it.tight) {
            // the other field's rows as the in-order loop has them now: written by iteration wr->ticket[other] into device
            // frame wr->fslot[other] -- still there while fewer than ring - 2 iterations have passed -- or, when this
            // engine has not written them (first iterations, after a one-at-a-time iteration or a DIRTY), the caller's
            // own bytes, snapshotted below
            if (wr->ticket[other] != 0 && t - wr->ticket[other] + 2 < (uint64_t)e->ring) { it.chain_fslot = wr->fslot[other]; it.chain_ticket = wr->ticket[other]; }
            else if (e->next_ticket - 1 > e->done_ticket) {
                // snapshot of the caller's bytes: whatever is in flight lands in them first (start of a stream, after a
                // one-at-a-time iteration; with alternating fields the chain never breaks afterwards)
                rc = h422_wait_ticket(c, NTSCSIM_TICKET_ALL);
                if (rc != NTSCSIM_OK) return rc;
            }
        }
        if (serial) { wr->ticket[0] = wr->ticket[1] = 0; }      // the frame moves on in the caller's memory (mirror path)
        else { wr->ticket[L->field] = t; wr->fslot[L->field] = fslot; }
    }
    // pinned caller frames: results are written in place by the delivery kernels
    (void)h422_pin_frame(c, e, L->frame, W, H, H, it.frm_dev);
    if (bkey) (void)h422_pin_frame(c, e, L->filter, W, H, H, it.flt_dev);
    if (L->out.data[0]) (void)h422_pin_frame(c, e, L->out, W, H, h422_out_chroma_rows(H, L->out_mode), it.out_dev);
    if (serial) {
        // in order, alone: what is pending goes first
        rc = h422_launch(c);
        if (rc != NTSCSIM_OK) return rc;
        it.mfrm = h422_mirror(c, e, L->frame, H);
        if (!it.mfrm) return NTSCSIM_E_NOMEM;
        rc = h422_refresh_mirror(c, e, it.mfrm, W);
        if (rc != NTSCSIM_OK) return rc;
        if (bkey) {
            it.mflt = h422_mirror(c, e, L->filter, H, it.mfrm);
            if (!it.mflt) return NTSCSIM_E_NOMEM;
            rc = h422_refresh_mirror(c, e, it.mflt, W);
            if (rc != NTSCSIM_OK) return rc;
        }
    } else {
        // the caller's frame moves on without its mirror
        for (auto *m : e->mirrors)
            if (m->host[0] == L->frame.data[0]) m->stale = true;
        uint8_t *pad = e->pads + (size_t)2 * e->L * (size_t)it.slot;
        const int nr = h422_field_rows(H, L->field);
        for (int k = 0; k < nr; k++) {
            const int y = (int)L->field + 2 * k;
            const uint8_t *row = L->frame.data[0] + (size_t)L->frame.linesize[0] * (size_t)y;
            // (behind the last row of a TIGHT plane lies nothing of the caller's: the contract's value 16 -- byte by byte: with
            //  linesize = width + 1 the first of the two is still the row's own padding)
            const size_t plane = (size_t)L->frame.linesize[0] * (size_t)H, at = (size_t)L->frame.linesize[0] * (size_t)y + (size_t)W;
            pad[2 * k] = (!it.tight || at < plane) ? row[W] : 16;
            pad[2 * k + 1] = (!it.tight || at + 1 < plane) ? row[W + 1] : 16;
        }
    }
    // ---- source snapshot
    // (the SYNCHRONOUS call returns when its iteration is done: pinned source planes need no snapshot -- k422_render reads
    //  them where they are, 0.7 MB over the link instead of a memcpy on the calling thread + a DMA of the same bytes)
    static const bool sync_direct = !(std::getenv("NTSCSIM_FIELD422_SRCDIRECT") && std::getenv("NTSCSIM_FIELD422_SRCDIRECT")[0] == '0');   // developer A/B
    // (never planes that share bytes with a frame the iteration writes: the snapshot is what such a caller relies on)
    auto shares = [&](const ntscsim_frame422 &f, int rows_y, int rows_c) {
        if (!f.data[0]) return false;
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                const uintptr_t s0 = (uintptr_t)L->src.data[a], s1 = s0 + (size_t)L->src.linesize[a] * (a ? src_crows : (size_t)L->src_height);
                const uintptr_t f0 = (uintptr_t)f.data[b], f1 = f0 + (size_t)f.linesize[b] * (size_t)(b ? rows_c : rows_y);
                if (s0 < f1 && f0 < s1) return true;
            }
        return false;
    };
    const bool src_apart = have_src && !shares(L->frame, H, H) && !shares(L->filter, H, H) && !shares(L->out, H, H);
    if (have_src && sync_call && sync_direct && src_apart && h422_pin_frame(c, e, L->src, W, L->src_height, (int)src_crows, it.src_dev)) {
        e->src_cur = -1;                  // (no device copy of this source: a later SAME_SRC submit takes its own)
    } else
    if (have_src) {
        for (int k = 0; k < 3; k++) it.src_dev[k] = nullptr;
        int sslot = e->src_cur;
        if (!(flags & NTSCSIM_SUBMIT_SAME_SRC) || sslot < 0) {
            sslot = (int)(e->src_ring_pos % (uint64_t)e->ring);
            const uint64_t last = e->src_last_ticket[(size_t)sslot];
            if (last > e->done_ticket) {
                rc = h422_wait_ticket(c, last);
                if (rc != NTSCSIM_OK) return rc;
            }
            uint8_t *sdev[3];
            // The source is SNAPSHOTTED: by a memcpy into the pinned staging ring (default: 25-30 us per 720x480 frame on the
            // calling thread, nothing to wait for), or -- NTSCSIM_SUBMIT422_SRCDMA=1, developer A/B -- by DMA straight out of
            // pinned caller planes, which the call then has to WAIT for (three plane copies + a stream synchronisation:
            // 46-50 us measured, and the engine's launches are no faster for it: profiles/r05_host422_loop_probe.txt)
            static const bool src_dma = std::getenv("NTSCSIM_SUBMIT422_SRCDMA") && std::getenv("NTSCSIM_SUBMIT422_SRCDMA")[0] == '1';
            if (src_dma && h422_pin_frame(c, e, L->src, W, L->src_height, (int)src_crows, sdev)) {
                uint8_t *o = e->dsrc.p + e->sbytes * (size_t)sslot;
                for (int k = 0; k < 3; k++) {
                    const size_t rb = k ? (size_t)W2 : (size_t)W, nr = k ? src_crows : (size_t)L->src_height;
                    if ((size_t)L->src.linesize[k] == rb)
                        HIPCHK(c, hipMemcpyAsync(o, L->src.data[k], rb * nr, hipMemcpyHostToDevice, e->s_up));
                    else
                        HIPCHK(c, hipMemcpy2DAsync(o, rb, L->src.data[k], (size_t)L->src.linesize[k], rb, nr, hipMemcpyHostToDevice, e->s_up));
                    o += rb * nr;
                }
                HIPCHK(c, hipStreamSynchronize(e->s_up));
            } else {
            uint8_t *hs = e->hsrc + e->sbytes * (size_t)sslot;
            uint8_t *o = hs;
            for (int k = 0; k < 3; k++) {
                const size_t rb = k ? (size_t)W2 : (size_t)W, nr = k ? src_crows : (size_t)L->src_height;
                if ((size_t)L->src.linesize[k] == rb) std::memcpy(o, L->src.data[k], rb * nr);
                else for (size_t r = 0; r < nr; r++) std::memcpy(o + rb * r, L->src.data[k] + (size_t)L->src.linesize[k] * r, rb);
                o += rb * nr;
            }
            HIPCHK(c, hipMemcpyAsync(e->dsrc.p + e->sbytes * (size_t)sslot, hs, src_bytes, hipMemcpyHostToDevice, e->s_up));
            }
            e->src_ring_pos++;
            e->src_cur = sslot;
            e->stats[2]++;
        }
        it.sslot = sslot;
        e->src_last_ticket[(size_t)sslot] = t;
    }
    it.rng_pos = c->rng_pos;
    if (!(L->flags & NTSCSIM_422_NOCOMP)) c->rng_pos += ntscsim_rng_calls_per_field_422(&c->prm, W, H, L->field);
    e->pending.push_back(it);
    e->next_ticket++;
    e->stats[0]++;
    e->stats[serial ? 4 : 3]++;
    if (ticket) *ticket = t;
    if (serial) {
        e->launch_for_wait = sync_call;     // (ntscsim_field422(): the caller waits for this launch next)
        rc = h422_launch(c);
        e->launch_for_wait = false;
        return rc;
    }
    // a launch at `depth` iterations -- but not between the two fields of a pair (the second one shares the first
    // one's device frame while that is still pending)
    // (rings smaller than configured -- the byte budget -- hold fewer iterations: a launch and its successor must fit)
    const size_t np = e->pending.size();
    const int depth = e->depth <= (e->ring - 2) / 2 ? e->depth : ((e->ring - 2) / 2 > 0 ? (e->ring - 2) / 2 : 1);
    if ((int)np >= depth && (L->field == 0 || (int)np > depth)) {
        e->launch_for_wait = sync_call;     // (ntscsim_field422(): the caller waits for this launch next)
        rc = h422_launch(c);
        e->launch_for_wait = false;
        return rc;
    }
    return NTSCSIM_OK;
}

extern "C" int ntscsim_submit422(ntscsim_ctx *c, const ntscsim_loop422 *L, uint32_t flags, uint64_t *ticket)
{
    return h422_submit(c, L, flags, ticket, false);
}

extern "C" int ntscsim_field422(ntscsim_ctx *c, const ntscsim_loop422 *L)
{
    uint64_t t = 0;
    int rc = h422_submit(c, L, 0, &t, true);
    if (rc == NTSCSIM_OK) rc = h422_wait_ticket(c, t);
    // (failed: something that reads the caller's planes may still be in flight -- they are the caller's again when this returns)
    if (rc != NTSCSIM_OK && c && c->h422 && c->h422->s_up) {
        (void)hipStreamSynchronize(c->h422->s_up);
        (void)hipStreamSynchronize(c->stream);       // (k422_render may have been queued on pinned source planes)
    }
    return rc;
}

extern "C" int ntscsim_submit422_configure(ntscsim_ctx *c, int depth, int slots)
{
    if (!c || depth < 1 || depth > 4096) return NTSCSIM_E_ARG;
    if (slots == 0) slots = 4 * depth;
    if (slots < 2 * depth + 2 || slots > 65536) return NTSCSIM_E_ARG;
    Host422Engine *e = h422_get(c);
    if (!e) return NTSCSIM_E_NOMEM;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = h422_wait_ticket(c, NTSCSIM_TICKET_ALL);
    if (rc != NTSCSIM_OK) return rc;
    h422_release_rings(e);
    e->depth = depth; e->nslots = slots;
    return NTSCSIM_OK;
}

extern "C" void ntscsim_submit422_stats(const ntscsim_ctx *c, uint64_t out[8])
{
    if (!out) return;
    for (int i = 0; i < 8; i++) out[i] = (c && c->h422) ? c->h422->stats[i] : 0;
}

// ntscsim_host_unpin() for the frames this engine has pinned: everything in flight is delivered first
static int h422_host_unpin(ntscsim_ctx *c, const void *base)
{
    Host422Engine *e = c->h422;
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = h422_wait_ticket(c, NTSCSIM_TICKET_ALL);
    if (e->s_up) HIPCHK(c, hipStreamSynchronize(e->s_up));
    if (e->s_dn) HIPCHK(c, hipStreamSynchronize(e->s_dn));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!base) pin_release_all(e->pins);
    else (void)pin_release(e->pins, base);
    e->src_cur = -1;
    return rc;
}

static bool h422_pins_overlap(ntscsim_ctx *c, uintptr_t p0, uintptr_t p1)
{
    if (!c->h422) return false;
    for (auto &r : c->h422->pins.regs) if (p0 < r.p1 && r.p0 < p1) return true;
    return false;
}

static void h422_set_pin_policy(ntscsim_ctx *c, int policy)
{
    if (c->h422) c->h422->pins.policy = policy;
}
