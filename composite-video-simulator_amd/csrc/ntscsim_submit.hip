// ntscsim_submit.hip -- the asynchronous host-frame form of the drop-in: ntscsim_submit() / ntscsim_wait()
// (include/ntscsim.h).  Included by ntscsim_hip.hip (one translation unit).
//
// What it replaces: the call `composite_layer(ring[idx], in.rgb, in, (current&1)^1, current)` at
// ffmpeg_ntsc.cpp:2229 when the caller keeps its loop (:2202-2282) and its AVFrames, but lets `depth`
// fields be in flight.  Structure:
//
//   submit   src frame --DMA (s_up)--> device source ring          (once per frame: NTSCSIM_SUBMIT_SAME_SRC)
//            field recorded in `pending`; at `depth` fields: launch
//   launch   one lane (child ctx: own stream + scratch) runs the ordinary batched kernel chain
//            (ntscsim_fields_device) on the pending fields, then k_deliver writes every field's rows from the
//            device destination ring straight into the caller's pinned frames (or one linear D2H into the
//            staging ring for frames that are not pinned); an event closes the launch
//   wait     synchronise on the launch's event, copy staged rows out, retire the launch
//
// Lanes exist because a launch of 32 fields is ~125 wavefronts on a chip with 2,048 slots and takes the same
// ~0.5 ms as one of 600: three launches side by side hide that latency.  Fields carry explicit rand()
// positions, so it does not matter which lane runs which launch.
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <unistd.h>

namespace {

struct DeliverRec {
    const uint8_t *dev;       // device frame (row 0)
    uint8_t *host;            // device-visible address of the caller's frame (row 0)
    int32_t dev_pitch, host_pitch;
    int32_t row0, row_step, nrows;
    int32_t bob;              // 0: row y <- device row y;  1 + field: line doubling, row y <- the field's row beside it
};

// One workgroup per (row chunk, field): copies rows row0, row0+row_step, ... of a device frame into the
// caller's (pinned, device-mapped) frame.  Stores go over the host link; 16-byte stores when every address is
// 16-byte aligned.  With `bob` the loop's line doubling (ffmpeg_ntsc.cpp:2233-2257) happens on the way out: field 1
// copies odd row y onto y - 1, field 0 copies row y + 1 onto odd row y -- destination row y takes the field's row
// y | 1 (field 1) or (y + 1) & ~1 (field 0); the one row without a partner is not in [row0, row0 + nrows).
__global__ void k_deliver(const DeliverRec *__restrict__ recs, int row_bytes, int vec16)
{
    const DeliverRec r = recs[blockIdx.y];
    for (int k = blockIdx.x; k < r.nrows; k += gridDim.x) {
        const size_t y = (size_t)r.row0 + (size_t)k * r.row_step;
        const size_t ys = r.bob == 0 ? y : (r.bob == 2 ? (y | 1) : ((y + 1) & ~(size_t)1));
        const uint8_t *s = r.dev + ys * (size_t)r.dev_pitch;
        uint8_t *d = r.host + y * (size_t)r.host_pitch;
        if (vec16) {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(s);
            uint4 *d4 = reinterpret_cast<uint4 *>(d);
            for (int i = threadIdx.x; i < row_bytes / 16; i += blockDim.x) d4[i] = s4[i];
        } else {
            const uint32_t *s1 = reinterpret_cast<const uint32_t *>(s);
            uint32_t *d1 = reinterpret_cast<uint32_t *>(d);
            for (int i = threadIdx.x; i < row_bytes / 4; i += blockDim.x) d1[i] = s1[i];
        }
    }
}

} // namespace

// ---- delivery of STAGED results (frames that are not pinned): the rows travel device -> pinned staging ring by DMA and
// from there into the caller's frames by memcpy.  That memcpy used to run on the caller's thread inside ntscsim_wait()
// (1.1 MB per 720x480 4:2:2 iteration: the whole budget of a 30k fields/s loop); it now runs on threads of the engine as
// soon as the launch's event has fired, i.e. usually long before the caller asks.  One lead thread takes the launches in
// order (hipEventSynchronize, then the copies, split over itself and `helpers` more threads); ntscsim_wait() only waits
// for the launch's id.  Copies of one launch never overlap each other (the engines drop all but the last writer of a
// row at launch time), launches are delivered one after the other: the caller's frames end up as the in-order loop
// leaves them.
struct CopyOp { uint8_t *dst; const uint8_t *src; size_t dstep, sstep, rb; int rows; };

class Delivery {
public:
    ~Delivery() { stop(); }
    // everything posted before is delivered in post order; `done` must have been recorded
    void post(int device, hipEvent_t done, std::vector<CopyOp> &&ops, uint64_t id)
    {
        std::unique_lock<std::mutex> lk(m_);
        if (!started_) start(device);
        q_.push_back(Job{done, std::move(ops), id});
        posted_ = id;
        cv_.notify_all();
    }
    // true once launch `id` is in the caller's frames; false: its event failed (the rows are lost)
    bool wait(uint64_t id)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return delivered_ >= id; });
        return failed_.empty() || std::find(failed_.begin(), failed_.end(), id) == failed_.end();
    }
    void drain() { std::unique_lock<std::mutex> lk(m_); cv_done_.wait(lk, [&] { return delivered_ >= posted_; }); }
    // cancel: what has not been delivered yet is dropped, not copied (ntscsim_destroy() with fields in flight: the caller
    // never waited for them, its frames may be gone)
    void stop(bool cancel = false)
    {
        {
            std::unique_lock<std::mutex> lk(m_);
            if (!started_) return;
            cancel_ = cancel;
            cv_done_.wait(lk, [&] { return delivered_ >= posted_; });
            quit_ = true;
            cv_.notify_all();
        }
        for (auto &t : threads_) t.join();
        threads_.clear();
        started_ = false; quit_ = false; cancel_ = false;
    }

private:
    struct Job { hipEvent_t done; std::vector<CopyOp> ops; uint64_t id; };
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    std::deque<Job> q_;
    std::vector<std::thread> threads_;
    std::vector<uint64_t> failed_;
    uint64_t posted_ = 0, delivered_ = 0;
    bool started_ = false, quit_ = false, cancel_ = false;
    // the launch being copied: helpers pull ops by index
    const std::vector<CopyOp> *cur_ = nullptr;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0;
    int busy_ = 0;

    static void run_op(const CopyOp &o)
    {
        if (o.dstep == o.rb && o.sstep == o.rb) { std::memcpy(o.dst, o.src, o.rb * (size_t)o.rows); return; }
        for (int r = 0; r < o.rows; r++) std::memcpy(o.dst + o.dstep * (size_t)r, o.src + o.sstep * (size_t)r, o.rb);
    }
    void pull()
    {
        const std::vector<CopyOp> &ops = *cur_;
        for (size_t i = next_.fetch_add(1); i < ops.size(); i = next_.fetch_add(1)) run_op(ops[i]);
    }
    void helper()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [&] { return quit_ || (cur_ && gen_ != seen); });
            if (quit_) return;
            seen = gen_;
            lk.unlock();
            pull();
            lk.lock();
            if (--busy_ == 0) cv_done_.notify_all();
        }
    }
    void lead(int device)
    {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [&] { return quit_ || !q_.empty(); });
            if (q_.empty()) return;          // quit_ and nothing left
            Job j = std::move(q_.front());
            q_.pop_front();
            lk.unlock();
            const bool ok = hipEventSynchronize(j.done) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            lk.lock();
            if (ok && !cancel_ && !j.ops.empty()) {
                cur_ = &j.ops; next_.store(0); gen_++;
                busy_ = (int)threads_.size() - 1;
                cv_.notify_all();
                lk.unlock();
                pull();
                lk.lock();
                cv_done_.wait(lk, [&] { return busy_ == 0; });
                cur_ = nullptr;
            }
            if (!ok) failed_.push_back(j.id);
            delivered_ = j.id;
            cv_done_.notify_all();
        }
    }
    void start(int device)          // m_ held
    {
        const char *ev = std::getenv("NTSCSIM_COPY_THREADS");
        int n = ev ? std::atoi(ev) : 4;
        if (n < 1) n = 1;
        if (n > 16) n = 16;
        threads_.emplace_back([this, device] { lead(device); });
        for (int i = 1; i < n; i++) threads_.emplace_back([this] { helper(); });
        started_ = true;
    }
};

// registrations of caller memory (hipHostRegister in place), cached: a tool recycles a handful of frame buffers
// policy: 0 = never (everything is staged); 1 (default) = memory that IS pinned already -- declared with
// ntscsim_host_pin(), allocated with ntscsim_host_alloc() / hipHostMalloc, registered by the caller -- plus buffers
// that start on a page boundary; 2 = additionally blocks that carry glibc's header of a chunk with a mapping of its
// own (a peek at the allocator's internals: opt-in, glibc only)
struct PinCache {
    struct Reg { uintptr_t p0, p1; uint8_t *dev; bool owned; };
    std::vector<Reg> regs;
    int policy = 1;
    size_t min_bytes = 256u << 10;
    const PinCache *declared = nullptr;      // the ctx's explicit registrations (ntscsim_host_pin)
};
static void pin_release_all(PinCache &pc)      // everything that uses the registrations must have completed
{
    for (auto &r : pc.regs)
        if (r.owned) (void)hipHostUnregister((void *)r.p0);
    pc.regs.clear();
    (void)hipGetLastError();
}
static bool pin_release(PinCache &pc, const void *base)
{
    const uintptr_t a = (uintptr_t)base;
    for (size_t i = 0; i < pc.regs.size(); i++)
        if (a >= pc.regs[i].p0 && a < pc.regs[i].p1) {
            if (pc.regs[i].owned) (void)hipHostUnregister((void *)pc.regs[i].p0);
            pc.regs.erase(pc.regs.begin() + (long)i);
            return true;
        }
    return false;
}

static void declared_pins_destroy(ntscsim_ctx *c)
{
    if (!c->declared) return;
    pin_release_all(*c->declared);
    delete c->declared;
    c->declared = nullptr;
}

struct SubmitEngine {
    ntscsim_submit_opts o;
    bool configured = false;
    int W = 0, H = 0;
    size_t pitch = 0, fbytes = 0;
    int nslots = 0;
    DevBuf<uint8_t> dsrc, ddst;
    uint8_t *hsrc = nullptr, *hdst = nullptr;         // pinned staging rings (lazy, nslots frames each)
    DeliverRec *recs = nullptr;                       // pinned, device-visible: nslots records
    hipStream_t s_up = nullptr;
    hipEvent_t ev_up = nullptr;
    std::vector<ntscsim_ctx *> lanes;
    unsigned lane_next = 0;

    struct Item {
        uint64_t ticket;
        int src_slot, dst_slot;
        uint8_t *host_dst;        // caller's frame
        uint8_t *host_dst_dev;    // its device-visible address when pinned, else NULL (staged)
        bool direct;              // the decoder writes its rows straight into the caller's pinned frame (no ring, no k_deliver)
        int dst_ls;
        unsigned field;
        uint32_t flags;
        uint64_t fieldno, rng_pos;
    };
    std::vector<Item> pending;
    std::vector<uint64_t> pending_deps;   // first tickets of launches in flight that the pending one must follow (shared dst rows)
    // NTSCSIM_SUBMIT_DIRECT=0: A/B switch back to the device destination ring + k_deliver for every field
    bool direct_ok = !(std::getenv("NTSCSIM_SUBMIT_DIRECT") && std::getenv("NTSCSIM_SUBMIT_DIRECT")[0] == '0');
    struct Batch {
        uint64_t first = 0, last = 0;
        hipEvent_t up = nullptr, done = nullptr;
        hipEvent_t t0 = nullptr, t1 = nullptr;    // NTSCSIM_SUBMIT_TIMING: GPU time stamps around the launch
        unsigned lane = 0;
        std::vector<Item> items;
        int rc = NTSCSIM_OK;
        bool launched_ok = false;
        bool posted = false;              // staged fields: handed to the copy threads (Delivery), id = `last`
    };
    std::deque<Batch> inflight;
    Delivery dlv;                         // staging ring -> caller frames, off the caller's thread
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    uint64_t next_ticket = 1;         // next to issue
    uint64_t done_ticket = 0;         // everything <= this has been delivered
    // source ring
    int src_cur = -1;                 // slot holding the frame of the previous submit
    uint64_t src_ring_pos = 0;
    std::vector<uint64_t> src_last_ticket;    // last ticket that reads the slot
    PinCache pins;                    // registrations of caller memory
    uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int sticky_rc = NTSCSIM_OK;
    bool timing = std::getenv("NTSCSIM_SUBMIT_TIMING") != nullptr;      // developer switch: print every launch's GPU span
    hipEvent_t tref = nullptr;
};

extern "C" void ntscsim_submit_opts_init(ntscsim_submit_opts *o)
{
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->struct_size = (uint32_t)sizeof(*o);
    o->depth = 32;
    o->slots = 256;
    o->lanes = 3;
    o->pin_caller_buffers = 1;
    o->min_pin_bytes = 256u << 10;
}

static int sub_wait_ticket(ntscsim_ctx *c, uint64_t ticket);
static int sub_launch(ntscsim_ctx *c);

static SubmitEngine *sub_get(ntscsim_ctx *c)
{
    if (!c->sub) {
        c->sub = new (std::nothrow) SubmitEngine();
        if (c->sub) { ntscsim_submit_opts_init(&c->sub->o); c->sub->o.pin_caller_buffers = c->pin_policy; }
    }
    return c->sub;
}

static void sub_release_geometry(SubmitEngine *e)
{
    e->dsrc.release(); e->ddst.release();
    if (e->hsrc) (void)hipHostFree(e->hsrc);
    if (e->hdst) (void)hipHostFree(e->hdst);
    if (e->recs) (void)hipHostFree(e->recs);
    e->hsrc = e->hdst = nullptr; e->recs = nullptr;
    e->W = e->H = 0; e->nslots = 0;
    e->src_cur = -1;
    e->src_last_ticket.clear();
}

// everything in flight must have been waited for
static void sub_unpin_all(SubmitEngine *e) { pin_release_all(e->pins); }

static void submit_engine_destroy(ntscsim_ctx *c)
{
    SubmitEngine *e = c->sub;
    if (!e) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    e->dlv.stop(true);
    for (auto &b : e->inflight) { if (b.up) (void)hipEventDestroy(b.up); if (b.done) (void)hipEventDestroy(b.done); }
    for (auto &p : e->ev_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (ntscsim_ctx *l : e->lanes) ntscsim_destroy(l);
    sub_release_geometry(e);
    sub_unpin_all(e);
    if (e->s_up) (void)hipStreamDestroy(e->s_up);
    if (e->ev_up) (void)hipEventDestroy(e->ev_up);
    delete e;
    c->sub = nullptr;
}

extern "C" int ntscsim_submit_configure(ntscsim_ctx *c, const ntscsim_submit_opts *o)
{
    if (!c || !o || o->struct_size != sizeof(ntscsim_submit_opts)) return NTSCSIM_E_ARG;
    if (o->depth < 1 || o->depth > 4096 || o->lanes < 1 || o->lanes > 8) return NTSCSIM_E_ARG;
    if (o->slots != 0 && o->slots < 2 * o->depth) return NTSCSIM_E_ARG;
    if (o->slots > 65536) return NTSCSIM_E_ARG;
    SubmitEngine *e = sub_get(c);
    if (!e) return NTSCSIM_E_NOMEM;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = sub_wait_ticket(c, NTSCSIM_TICKET_ALL);
    if (rc != NTSCSIM_OK) return rc;
    e->o = *o;
    if (e->o.slots == 0) e->o.slots = 8 * e->o.depth;
    while ((int)e->lanes.size() > e->o.lanes) { ntscsim_destroy(e->lanes.back()); e->lanes.pop_back(); }
    sub_release_geometry(e);          // the rings are sized by `slots`
    e->configured = true;
    return NTSCSIM_OK;
}

static int sub_ensure_geometry(ntscsim_ctx *c, SubmitEngine *e, int W, int H)
{
    if (e->W == W && e->H == H && e->nslots == e->o.slots) return NTSCSIM_OK;
    // another geometry: drain, then rebuild the rings
    int rc = sub_wait_ticket(c, NTSCSIM_TICKET_ALL);
    if (rc != NTSCSIM_OK) return rc;
    sub_release_geometry(e);
    const size_t pitch = (((size_t)W * 4 + 15) / 16) * 16;
    const size_t fbytes = ((pitch * (size_t)H + 255) / 256) * 256;
    const int ns = e->o.slots;
    HIPCHK(c, e->dsrc.ensure(fbytes * (size_t)ns));
    HIPCHK(c, e->ddst.ensure(fbytes * (size_t)ns));
    HIPCHK(c, hipHostMalloc((void **)&e->recs, sizeof(DeliverRec) * (size_t)ns, hipHostMallocDefault));
    if (!e->s_up) HIPCHK(c, hipStreamCreateWithFlags(&e->s_up, hipStreamNonBlocking));
    if (!e->ev_up) HIPCHK(c, hipEventCreateWithFlags(&e->ev_up, hipEventDisableTiming));
    e->W = W; e->H = H; e->pitch = pitch; e->fbytes = fbytes; e->nslots = ns;
    e->src_last_ticket.assign((size_t)ns, 0);
    e->src_ring_pos = 0;
    e->src_cur = -1;
    return NTSCSIM_OK;
}

static int sub_ensure_staging(ntscsim_ctx *c, SubmitEngine *e, bool src)
{
    uint8_t **p = src ? &e->hsrc : &e->hdst;
    if (*p) return NTSCSIM_OK;
    HIPCHK(c, hipHostMalloc((void **)p, e->fbytes * (size_t)e->nslots, hipHostMallocDefault));
    return NTSCSIM_OK;
}

// Device-visible address of caller memory [p, p+span), pinning it in place on first sight.  NULL = not pinned
// (policy, too small, shares a page with another registration, registration refused): the staging ring is used.
static uint8_t *pin_lookup(PinCache &pc, const void *p, size_t span)
{
    if (pc.policy <= 0 || span == 0) return nullptr;
    const uintptr_t a0 = (uintptr_t)p, a1 = a0 + span;
    // (1) declared by the caller, or seen before
    if (pc.declared)
        for (auto &r : pc.declared->regs)
            if (a0 >= r.p0 && a1 <= r.p1) return r.dev + (a0 - r.p0);
    for (auto &r : pc.regs)
        if (a0 >= r.p0 && a1 <= r.p1) return r.dev + (a0 - r.p0);
    if (pc.regs.size() >= 1024) return nullptr;
    // (2) pinned already -- ntscsim_host_alloc() / hipHostMalloc / the caller's own hipHostRegister: the runtime knows, no
    // guess about who owns the pages is needed.  First and last byte must belong to one registration (same offset).
    {
        hipPointerAttribute_t at0, at1;
        const bool ok0 = hipPointerGetAttributes(&at0, (const void *)a0) == hipSuccess && at0.type == hipMemoryTypeHost && at0.devicePointer;
        if (!ok0) (void)hipGetLastError();
        else {
            const bool ok1 = hipPointerGetAttributes(&at1, (const void *)(a1 - 1)) == hipSuccess && at1.type == hipMemoryTypeHost &&
                             at1.devicePointer == (void *)((uint8_t *)at0.devicePointer + (span - 1));
            if (!ok1) { (void)hipGetLastError(); return nullptr; }      // pinned in part: leave it alone, stage
            pc.regs.push_back({a0, a1, (uint8_t *)at0.devicePointer, false});
            return (uint8_t *)at0.devicePointer;
        }
    }
    if (span < pc.min_bytes) return nullptr;
    // (3) pin in place.  Registration is page-wise, so the buffer's first and last page get pinned whole.  That is only
    // harmless when nothing else lives in them: a foreign heap block that starts in a pinned page and runs on into
    // pageable memory can no longer be the source of a hipMemcpy.  Never memory of the brk heap (the allocator trims
    // and recycles those pages under a registration), never small blocks.
    if (span < (64u << 10) || a0 < (uintptr_t)sbrk(0)) return nullptr;
    if (a0 & 4095u) {
        // not on a page boundary: policy 1 stages it.  Policy 2 accepts a block that starts within the allocator's
        // header + alignment padding of a page AND carries glibc's IS_MMAPPED bit (bit 1 of the size word in front
        // of the pointer malloc / posix_memalign returned) with a chunk size that covers the span -- an allocation
        // with a mapping of its own, whose edge pages are its own.
#if defined(__GLIBC__)
        if (pc.policy < 2 || (a0 & 4095u) > 128u) return nullptr;
        size_t hdr;
        std::memcpy(&hdr, (const void *)(a0 - sizeof(size_t)), sizeof(hdr));
        if (!(hdr & 2u) || (hdr & ~(size_t)7) < span) return nullptr;
#else
        return nullptr;
#endif
    }
    const uintptr_t PG = 4096;
    const uintptr_t p0 = a0 & ~(PG - 1), p1 = (a1 + PG - 1) & ~(PG - 1);
    for (auto &r : pc.regs)
        if (p0 < r.p1 && r.p0 < p1) return nullptr;          // partial overlap with a live registration
    if (pc.declared)
        for (auto &r : pc.declared->regs)
            if (p0 < r.p1 && r.p0 < p1) return nullptr;
    hipError_t er = hipHostRegister((void *)p0, p1 - p0, hipHostRegisterDefault);
    (void)hipGetLastError();
    if (er != hipSuccess) return nullptr;
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, (void *)p0, 0) != hipSuccess || !dev) {
        (void)hipGetLastError();
        (void)hipHostUnregister((void *)p0);
        return nullptr;
    }
    pc.regs.push_back({p0, p1, (uint8_t *)dev, true});
    return (uint8_t *)dev + (a0 - p0);
}

// ntscsim_field(): the device-visible address of a caller frame that is the GPU's to address ALREADY -- declared through
// ntscsim_host_pin(), or pinned memory the runtime knows (first and last byte of one registration) -- else NULL.  Registers
// nothing and remembers nothing: the synchronous call must not outlive the caller's allocation with a cache entry.
static uint8_t *pinned_device_ptr(ntscsim_ctx *c, const void *p, size_t span)
{
    if (!span) return nullptr;
    const uintptr_t a0 = (uintptr_t)p, a1 = a0 + span;
    if (c->declared)
        for (auto &r : c->declared->regs)
            if (a0 >= r.p0 && a1 <= r.p1) return r.dev + (a0 - r.p0);
    hipPointerAttribute_t at0, at1;
    if (hipPointerGetAttributes(&at0, (const void *)a0) != hipSuccess || at0.type != hipMemoryTypeHost || !at0.devicePointer) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (hipPointerGetAttributes(&at1, (const void *)(a1 - 1)) != hipSuccess || at1.type != hipMemoryTypeHost ||
        at1.devicePointer != (void *)((uint8_t *)at0.devicePointer + (span - 1))) {
        (void)hipGetLastError();
        return nullptr;
    }
    return (uint8_t *)at0.devicePointer;
}

static uint8_t *sub_pinned(ntscsim_ctx *c, SubmitEngine *e, const void *p, size_t span)
{
    e->pins.policy = e->o.pin_caller_buffers < 0 ? 0 : (e->o.pin_caller_buffers > 2 ? 2 : e->o.pin_caller_buffers);
    e->pins.min_bytes = e->o.min_pin_bytes;
    e->pins.declared = c->declared;
    uint8_t *d = pin_lookup(e->pins, p, span);
    e->stats[6] = e->pins.regs.size();
    return d;
}

static int h422_host_unpin(ntscsim_ctx *c, const void *base);      // ntscsim_host422.hip
static bool h422_pins_overlap(ntscsim_ctx *c, uintptr_t p0, uintptr_t p1);
static void h422_set_pin_policy(ntscsim_ctx *c, int policy);
extern "C" int ntscsim_host_unpin(ntscsim_ctx *c, const void *base)
{
    if (!c) return NTSCSIM_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = NTSCSIM_OK;
    if (c->h422) rc = h422_host_unpin(c, base);
    if (SubmitEngine *e = c->sub) {
        const int r = sub_wait_ticket(c, NTSCSIM_TICKET_ALL);
        if (rc == NTSCSIM_OK) rc = r;
        HIPCHK(c, hipStreamSynchronize(e->s_up ? e->s_up : c->stream));
        if (!base) sub_unpin_all(e);
        else (void)pin_release(e->pins, base);
        e->stats[6] = e->pins.regs.size();
        e->src_cur = -1;
    }
    if (c->declared) {           // ... and the caller's own declaration (ntscsim_host_pin) last: nothing uses it any more
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!base) pin_release_all(*c->declared);
        else (void)pin_release(*c->declared, base);
    }
    return rc;
}

// ---- explicit contract: the caller says which memory is its own to pin ----------
extern "C" int ntscsim_host_pin(ntscsim_ctx *c, const void *base, size_t len)
{
    if (!c || !base || len == 0) return NTSCSIM_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->declared) c->declared = new (std::nothrow) PinCache();
    if (!c->declared) return NTSCSIM_E_NOMEM;
    const uintptr_t PG = 4096, a0 = (uintptr_t)base, a1 = a0 + len;
    const uintptr_t p0 = a0 & ~(PG - 1), p1 = (a1 + PG - 1) & ~(PG - 1);
    // Never memory of the brk heap (small malloc blocks): the allocator trims and recycles those pages under a registration,
    // and the GPU then faults on them -- in this call or in an unrelated later one (seen: sporadic aborts of the process,
    // rocr's VMFaultHandler).  What can be declared is an allocation of its own: a mapping, a block above the mmap threshold,
    // a pool.  (The engines' own pinning has always refused these addresses: pin_lookup.)
    if (p0 < (uintptr_t)sbrk(0)) {
        c->err = "ntscsim_host_pin: the range lies in the brk heap (a small malloc block); declare an allocation of its own";
        return NTSCSIM_E_ARG;
    }
    for (auto &r : c->declared->regs)
        if (p0 >= r.p0 && p1 <= r.p1) return NTSCSIM_OK;                 // declared before
    // registrations the engines made on their own for these pages go first (everything in flight is delivered)
    auto overlaps = [&](const PinCache &pc) { for (auto &r : pc.regs) if (p0 < r.p1 && r.p0 < p1) return true; return false; };
    if ((c->sub && overlaps(c->sub->pins)) || (c->h422 && h422_pins_overlap(c, p0, p1)) || overlaps(*c->declared)) {
        int rc = NTSCSIM_OK;
        for (uintptr_t a = p0; a < p1 && rc == NTSCSIM_OK; a += PG) {
            bool hit = false;
            if (c->sub) for (auto &r : c->sub->pins.regs) if (a >= r.p0 && a < r.p1) hit = true;
            if (c->h422 && h422_pins_overlap(c, a, a + 1)) hit = true;
            for (auto &r : c->declared->regs) if (a >= r.p0 && a < r.p1) hit = true;
            if (hit) rc = ntscsim_host_unpin(c, (const void *)a);
        }
        if (rc != NTSCSIM_OK) return rc;
    }
    hipError_t er = hipHostRegister((void *)p0, p1 - p0, hipHostRegisterDefault);
    bool owned = true;
    if (er == hipErrorHostMemoryAlreadyRegistered) { owned = false; er = hipSuccess; }      // e.g. ntscsim_host_alloc() memory
    (void)hipGetLastError();
    if (er != hipSuccess) { c->err = std::string("hipHostRegister: ") + hipGetErrorString(er); return NTSCSIM_E_HIP; }
    void *dev = nullptr, *dev_last = nullptr;
    if (hipHostGetDevicePointer(&dev, (void *)a0, 0) != hipSuccess || !dev ||
        hipHostGetDevicePointer(&dev_last, (void *)(a1 - 1), 0) != hipSuccess || dev_last != (uint8_t *)dev + (len - 1)) {
        (void)hipGetLastError();
        if (owned) (void)hipHostUnregister((void *)p0);
        c->err = "ntscsim_host_pin: the range is not one registration";
        return NTSCSIM_E_HIP;
    }
    if (owned) c->declared->regs.push_back({p0, p1, (uint8_t *)dev - (a0 - p0), true});
    else c->declared->regs.push_back({a0, a1, (uint8_t *)dev, false});
    return NTSCSIM_OK;
}

extern "C" void *ntscsim_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0) return nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

extern "C" void ntscsim_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

// Planes like av_frame_get_buffer(frame, align) lays them out -- linesize = row bytes rounded up to `align`, every
// plane start aligned, 32 spare rows' worth of padding dropped in favour of `align` + 64 spare bytes per plane -- in
// ONE pinned allocation.
extern "C" int ntscsim_host_frame_alloc(int n_planes, const int *row_bytes, const int *rows, int align,
                                        uint8_t **data, int *linesize, void **base, size_t *bytes)
{
    if (n_planes < 1 || n_planes > 8 || !row_bytes || !rows || !data || !linesize || !base) return NTSCSIM_E_ARG;
    if (align < 1) align = 1;
    if (align & (align - 1)) return NTSCSIM_E_ARG;
    const size_t A = (size_t)(align < 64 ? 64 : align);
    size_t off[8], total = 0;
    for (int k = 0; k < n_planes; k++) {
        if (row_bytes[k] < 1 || rows[k] < 1) return NTSCSIM_E_ARG;
        linesize[k] = (int)(((size_t)row_bytes[k] + (size_t)align - 1) / (size_t)align * (size_t)align);
        off[k] = total;
        total += ((size_t)linesize[k] * (size_t)rows[k] + 64 + A - 1) / A * A;
    }
    uint8_t *p = (uint8_t *)ntscsim_host_alloc(total);
    if (!p) return NTSCSIM_E_NOMEM;
    for (int k = 0; k < n_planes; k++) data[k] = p + off[k];
    *base = p;
    if (bytes) *bytes = total;
    return NTSCSIM_OK;
}

extern "C" int ntscsim_set_pin_policy(ntscsim_ctx *c, int policy)
{
    if (!c || policy < 0 || policy > 2) return NTSCSIM_E_ARG;
    c->pin_policy = policy;
    if (c->sub) c->sub->o.pin_caller_buffers = policy;
    h422_set_pin_policy(c, policy);
    return NTSCSIM_OK;
}

extern "C" void ntscsim_submit_stats(const ntscsim_ctx *c, uint64_t out[8])
{
    if (!out) return;
    for (int i = 0; i < 8; i++) out[i] = (c && c->sub) ? c->sub->stats[i] : 0;
}

// Do two fields in flight write the same bytes of a caller frame?  (rows: 0 / 1 = the rows of that parity, 2 = every
// row: line doubling.)  The two fields of one frame do not; anything else that overlaps is ordered by the engine:
// the header promises delivery in submit order.
static bool sub_dst_conflict(const uint8_t *a, int a_ls, unsigned a_rows, const uint8_t *b, int b_ls, unsigned b_rows,
                             int W, int H)
{
    const uintptr_t a0 = (uintptr_t)a, a1 = a0 + (size_t)a_ls * (size_t)(H - 1) + (size_t)W * 4;
    const uintptr_t b0 = (uintptr_t)b, b1 = b0 + (size_t)b_ls * (size_t)(H - 1) + (size_t)W * 4;
    if (a1 <= b0 || b1 <= a0) return false;
    if (a_rows == 2u || b_rows == 2u) return true;
    if (a0 == b0 && a_ls == b_ls) return a_rows == b_rows;
    return true;          // overlapping views that are not the same frame: assume the worst
}
static unsigned sub_item_rows(uint32_t flags, unsigned field) { return (flags & NTSCSIM_DESC_BOB) ? 2u : (field & 1u); }

// rows the synchronous call + (optionally) the loop's line doubling write: first row, step, count
static void sub_rows(int H, unsigned field, bool bob, int &row0, int &step, int &n)
{
    if (!bob) { row0 = (int)field; step = 2; n = (H - (int)field + 1) / 2; return; }
    // ffmpeg_ntsc.cpp:2233-2257: field 1 copies odd row y onto y-1; field 0 copies row y+1 onto odd row y while
    // y+1 < H -- every row except the last one when it has no partner
    row0 = 0; step = 1;
    const bool last_untouched = field ? (H & 1) != 0 : (H & 1) == 0;
    n = last_untouched ? H - 1 : H;
}

// Retire the oldest launch: wait for it, hand staged rows to the caller, recycle its events.
static int sub_retire_front(ntscsim_ctx *c, SubmitEngine *e)
{
    SubmitEngine::Batch &b = e->inflight.front();
    int rc = b.rc;
    if (b.launched_ok) {
        bool ok;
        if (b.posted) ok = e->dlv.wait(b.last);          // (the copy threads synchronised on the event and moved the rows)
        else ok = hipEventSynchronize(b.done) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); c->err = "submit: a launch failed on the device (hipEventSynchronize)"; rc = NTSCSIM_E_HIP; }
    }
    if (e->timing && b.t0 && b.t1) {
        float a = 0, z = 0;
        (void)hipEventSynchronize(b.t1);
        (void)hipEventElapsedTime(&a, e->tref, b.t0); (void)hipEventElapsedTime(&z, e->tref, b.t1);
        std::fprintf(stderr, "launch tickets %llu..%llu lane %u: GPU %.3f .. %.3f ms (%.3f)\n", (unsigned long long)b.first,
                     (unsigned long long)b.last, b.lane, a, z, z - a);
        (void)hipEventDestroy(b.t0); (void)hipEventDestroy(b.t1);
    }
    e->done_ticket = b.last;
    if (b.up && b.done) e->ev_pool.push_back({b.up, b.done});
    e->inflight.pop_front();
    if (rc != NTSCSIM_OK && e->sticky_rc == NTSCSIM_OK) e->sticky_rc = rc;
    return rc;
}

static int sub_wait_ticket(ntscsim_ctx *c, uint64_t ticket)
{
    SubmitEngine *e = c->sub;
    if (!e) return ticket == NTSCSIM_TICKET_ALL ? NTSCSIM_OK : NTSCSIM_E_ARG;
    if (ticket == NTSCSIM_TICKET_ALL) ticket = e->next_ticket - 1;
    if (ticket == 0) return NTSCSIM_OK;
    if (ticket >= e->next_ticket) return NTSCSIM_E_ARG;
    int rc = NTSCSIM_OK;
    if (!e->pending.empty() && ticket >= e->pending.front().ticket) {
        const int r = sub_launch(c);
        if (r != NTSCSIM_OK) rc = r;
    }
    while (!e->inflight.empty() && e->inflight.front().first <= ticket) {
        const int r = sub_retire_front(c, e);
        if (r != NTSCSIM_OK && rc == NTSCSIM_OK) rc = r;
    }
    return rc;
}

// Enqueue the pending fields as one launch on the next lane.
static int sub_launch(ntscsim_ctx *c)
{
    SubmitEngine *e = c->sub;
    if (!e || e->pending.empty()) return NTSCSIM_OK;
    SubmitEngine::Batch b;
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
    if (!e->ev_pool.empty()) { b.up = e->ev_pool.back().first; b.done = e->ev_pool.back().second; e->ev_pool.pop_back(); }
    else if (hipEventCreateWithFlags(&b.up, hipEventDisableTiming) != hipSuccess ||
             hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) {
        c->err = "hipEventCreate failed";
        return finish(NTSCSIM_E_HIP);
    }
    // the lane: a child ctx with the parent's parameters and switches
    const unsigned li = e->lane_next++ % (unsigned)e->o.lanes;
    while (e->lanes.size() <= li) {
        ntscsim_ctx *l = nullptr;
        const int rc = ntscsim_create(&c->prm, c->device, &l);
        if (rc != NTSCSIM_OK) return finish(rc);
        e->lanes.push_back(l);
    }
    ntscsim_ctx *lane = e->lanes[li];
    lane->mode = c->mode; lane->force_generic = c->force_generic; lane->no_fast_decode = c->no_fast_decode;
    lane->split_vhs = c->split_vhs;
    if (lane->warm_override[0] != c->warm_override[0] || lane->warm_override[1] != c->warm_override[1])
        ntscsim_debug_set_warmup(lane, c->warm_override[0], c->warm_override[1]);

    std::vector<ntscsim_field_desc> descs((size_t)n);
    bool any_staged = false, any_direct = false;     // any_direct: pinned frames served by k_deliver (line doubling, unaligned)
    int n_direct = 0;                                // pinned frames the decoder writes itself
    for (int i = 0; i < n; i++) {
        const SubmitEngine::Item &it = b.items[(size_t)i];
        ntscsim_field_desc &d = descs[(size_t)i];
        std::memset(&d, 0, sizeof(d));
        d.src_dev = e->dsrc.p + e->fbytes * (size_t)it.src_slot;
        d.src_linesize = (int)e->pitch;
        if (it.direct) { d.dst_dev = it.host_dst_dev; d.dst_linesize = it.dst_ls; }      // rows leave the decoder over the link
        else { d.dst_dev = e->ddst.p + e->fbytes * (size_t)it.dst_slot; d.dst_linesize = (int)e->pitch; }
        d.field = it.field;
        // (no NTSCSIM_DESC_BOB: the line doubling happens on the way out -- k_deliver's row map, or the host copy of a
        //  staged frame -- so the device frame only ever holds the field's own rows)
        d.flags = it.flags & (NTSCSIM_DESC_INTERLACED | NTSCSIM_DESC_TFF);
        d.fieldno = it.fieldno;
        d.rng_pos = it.rng_pos;
        if (it.direct) n_direct++;
        else if (it.host_dst_dev) any_direct = true;
        else any_staged = true;
    }
    hipError_t er = hipEventRecord(b.up, e->s_up);
    if (er == hipSuccess) er = hipStreamWaitEvent(lane->stream, b.up, 0);
    // launches in flight that write the same rows of the same caller frame come first (delivery in submit order)
    for (uint64_t dep : e->pending_deps)
        for (const auto &ob : e->inflight)
            if (ob.first == dep && ob.launched_ok && ob.lane != li && er == hipSuccess)
                er = hipStreamWaitEvent(lane->stream, ob.done, 0);
    e->pending_deps.clear();
    b.lane = li;
    if (e->timing) {
        if (!e->tref) { (void)hipEventCreate(&e->tref); (void)hipEventRecord(e->tref, lane->stream); }
        (void)hipEventCreate(&b.t0); (void)hipEventCreate(&b.t1);
        (void)hipEventRecord(b.t0, lane->stream);
    }
    if (er != hipSuccess) { c->err = std::string("submit launch: ") + hipGetErrorString(er); return finish(NTSCSIM_E_HIP); }
    lane->latency_form = true;         // (short launches take the three-role workgroup form: ntsc_pipe.hip)
    int rc = ntscsim_fields_device(lane, descs.data(), n, e->W, e->H, lane->stream);
    lane->latency_form = false;
    if (rc != NTSCSIM_OK) { c->err = lane->err; return finish(rc); }
    c->kernels = lane->kernels;
    // delivery
    if (any_direct) {
        // records live in the slot-indexed pinned array (a slot's record is rewritten only after its launch
        // retired); slots are handed out in ring order, so the pinned frames of a launch are a few runs of
        // consecutive records: one k_deliver per run
        bool vec16 = ((e->W * 4) & 15) == 0;
        for (const auto &it : b.items) {
            if (!it.host_dst_dev || it.direct) continue;
            DeliverRec &r = e->recs[it.dst_slot];
            r.dev = e->ddst.p + e->fbytes * (size_t)it.dst_slot;
            r.host = it.host_dst_dev;
            r.dev_pitch = (int32_t)e->pitch; r.host_pitch = it.dst_ls;
            sub_rows(e->H, it.field, (it.flags & NTSCSIM_DESC_BOB) != 0, r.row0, r.row_step, r.nrows);
            r.bob = (it.flags & NTSCSIM_DESC_BOB) ? 1 + (int32_t)it.field : 0;
            vec16 = vec16 && !(((uintptr_t)it.host_dst_dev | (uintptr_t)it.dst_ls) & 15);
        }
        size_t i = 0;
        int nd = 0;
        auto via_kernel = [](const SubmitEngine::Item &x) { return x.host_dst_dev && !x.direct; };
        while (i < b.items.size()) {
            if (!via_kernel(b.items[i])) { i++; continue; }
            size_t j = i + 1;
            while (j < b.items.size() && via_kernel(b.items[j]) && b.items[j].dst_slot == b.items[j - 1].dst_slot + 1) j++;
            hipLaunchKernelGGL(k_deliver, dim3(16, (unsigned)(j - i)), dim3(256), 0, lane->stream,
                               e->recs + b.items[i].dst_slot, e->W * 4, vec16 ? 1 : 0);
            nd += (int)(j - i);
            i = j;
        }
        er = hipGetLastError();
        if (er != hipSuccess) { c->err = std::string("k_deliver: ") + hipGetErrorString(er); return finish(NTSCSIM_E_HIP); }
        e->stats[4] += (uint64_t)nd;
    }
    e->stats[4] += (uint64_t)n_direct;
    if (any_staged) {
        rc = sub_ensure_staging(c, e, false);
        if (rc != NTSCSIM_OK) return finish(rc);
        // whole frames, runs of consecutive slots as one linear copy
        size_t i = 0;
        while (i < b.items.size()) {
            if (b.items[i].host_dst_dev) { i++; continue; }
            size_t j = i + 1;
            while (j < b.items.size() && !b.items[j].host_dst_dev && b.items[j].dst_slot == b.items[j - 1].dst_slot + 1) j++;
            const size_t off = e->fbytes * (size_t)b.items[i].dst_slot;
            er = hipMemcpyAsync(e->hdst + off, e->ddst.p + off, e->fbytes * (j - i), hipMemcpyDeviceToHost, lane->stream);
            if (er != hipSuccess) { c->err = std::string("submit D2H: ") + hipGetErrorString(er); return finish(NTSCSIM_E_HIP); }
            e->stats[5] += (uint64_t)(j - i);
            i = j;
        }
    }
    if (e->timing) (void)hipEventRecord(b.t1, lane->stream);
    er = hipEventRecord(b.done, lane->stream);
    if (er != hipSuccess) { c->err = std::string("hipEventRecord: ") + hipGetErrorString(er); return finish(NTSCSIM_E_HIP); }
    b.launched_ok = true;
    e->stats[1]++;
    if (any_staged) {
        // the copy threads move the staged rows into the caller's frames as soon as the event fires.  (Fields of one launch
        // never write the same rows: ntscsim_submit() launches before it accepts a field that clashes with a pending one.)
        std::vector<CopyOp> ops;
        const size_t rb = (size_t)e->W * 4;
        for (const auto &it : b.items) {
            if (it.host_dst_dev) continue;
            int row0, step, nr;
            sub_rows(e->H, it.field, (it.flags & NTSCSIM_DESC_BOB) != 0, row0, step, nr);
            const uint8_t *s = e->hdst + e->fbytes * (size_t)it.dst_slot;
            if (!(it.flags & NTSCSIM_DESC_BOB)) {
                // rows field, field + 2, ...: in two halves (the ops of a launch are what the threads share out)
                const int h0 = nr / 2;
                if (h0 > 0) ops.push_back({it.host_dst + (size_t)row0 * (size_t)it.dst_ls, s + (size_t)row0 * e->pitch, 2 * (size_t)it.dst_ls, 2 * e->pitch, rb, h0});
                if (nr - h0 > 0) ops.push_back({it.host_dst + (size_t)(row0 + 2 * h0) * (size_t)it.dst_ls, s + (size_t)(row0 + 2 * h0) * e->pitch, 2 * (size_t)it.dst_ls, 2 * e->pitch, rb, nr - h0});
            } else {
                // line doubling :2233-2257: destination row y takes the field's row beside it -- two strided passes
                // (even destination rows, odd destination rows), each reading every second source row
                for (int par = 0; par < 2; par++) {
                    const int cnt = (nr - par + 1) / 2;           // destination rows par, par + 2, ... < nr
                    if (cnt <= 0) continue;
                    const size_t y0 = (size_t)par;
                    const size_t ys0 = it.field ? (y0 | 1) : ((y0 + 1) & ~(size_t)1);
                    ops.push_back({it.host_dst + y0 * (size_t)it.dst_ls, s + ys0 * e->pitch, 2 * (size_t)it.dst_ls, 2 * e->pitch, rb, cnt});
                }
            }
        }
        e->dlv.post(c->device, b.done, std::move(ops), b.last);
        b.posted = true;
    }
    return finish(NTSCSIM_OK);
}

extern "C" int ntscsim_flush(ntscsim_ctx *c)
{
    if (!c) return NTSCSIM_E_ARG;
    if (!c->sub && !c->h422) return NTSCSIM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = c->sub ? sub_launch(c) : NTSCSIM_OK;
    if (c->h422) { const int r = h422_launch(c); if (rc == NTSCSIM_OK) rc = r; }
    return rc;
}

extern "C" int ntscsim_wait(ntscsim_ctx *c, uint64_t ticket)
{
    if (!c) return NTSCSIM_E_ARG;
    // a ctx serves one of the two tools: tickets of ntscsim_submit422() (ntscsim_host422.hip) are waited for here too
    if (c->h422) {
        HIPCHK(c, hipSetDevice(c->device));
        if (!c->sub) return h422_wait_ticket(c, ticket);
        if (ticket != NTSCSIM_TICKET_ALL) return NTSCSIM_E_ARG;      // two ticket sequences on one ctx: only "all" is unambiguous
        const int r = h422_wait_ticket(c, ticket);
        const int r2 = sub_wait_ticket(c, ticket);
        return r != NTSCSIM_OK ? r : r2;
    }
    if (!c->sub) return ticket == NTSCSIM_TICKET_ALL ? NTSCSIM_OK : NTSCSIM_E_ARG;
    if (ticket != NTSCSIM_TICKET_ALL && ticket <= c->sub->done_ticket && ticket != 0) return NTSCSIM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    return sub_wait_ticket(c, ticket);
}

extern "C" int ntscsim_submit(ntscsim_ctx *c, const uint8_t *src, int src_ls, int src_interlaced, int src_tff,
                              uint8_t *dst, int dst_ls, int W, int H, unsigned field, uint64_t fieldno,
                              uint32_t flags, uint64_t *ticket)
{
    if (!c || !src || !dst) return NTSCSIM_E_ARG;                   // :1578-1579
    if (src_ls < 4 * W || dst_ls < 4 * W) return NTSCSIM_E_SIZE;     // :1580-1581
    if (field > 1) return NTSCSIM_E_ARG;
    if (W < 16 || H < 2 || W > 16384 || H > 16384) return NTSCSIM_E_SIZE;
    if (((uintptr_t)src | (uintptr_t)dst | (uintptr_t)src_ls | (uintptr_t)dst_ls) & 3) return NTSCSIM_E_ARG;
    if (flags & ~(NTSCSIM_DESC_BOB | NTSCSIM_SUBMIT_SAME_SRC | NTSCSIM_SUBMIT_SRC_STABLE)) return NTSCSIM_E_ARG;
    SubmitEngine *e = sub_get(c);
    if (!e) return NTSCSIM_E_NOMEM;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = sub_ensure_geometry(c, e, W, H);
    if (rc != NTSCSIM_OK) return rc;

    // ring space: ticket t uses destination slot t mod nslots; the launch that held it must have retired
    const uint64_t t = e->next_ticket;
    if (t > (uint64_t)e->nslots && e->done_ticket < t - (uint64_t)e->nslots) {
        e->stats[7]++;
        rc = sub_wait_ticket(c, t - (uint64_t)e->nslots);
        if (rc != NTSCSIM_OK) return rc;
    }
    const size_t rb = (size_t)W * 4;
    // source frame
    int sslot = e->src_cur;
    if (!(flags & NTSCSIM_SUBMIT_SAME_SRC) || sslot < 0) {
        sslot = (int)(e->src_ring_pos % (uint64_t)e->nslots);
        const uint64_t last = e->src_last_ticket[(size_t)sslot];
        if (last > e->done_ticket) {
            // (cannot happen while sources <= fields in flight <= nslots, kept as a guard)
            rc = sub_wait_ticket(c, last);
            if (rc != NTSCSIM_OK) return rc;
        }
        uint8_t *dslot = e->dsrc.p + e->fbytes * (size_t)sslot;
        const size_t span = (size_t)src_ls * (size_t)(H - 1) + rb;
        const bool pinned = sub_pinned(c, e, src, span) != nullptr;
        const uint8_t *from = src;
        size_t from_ls = (size_t)src_ls;
        if (!pinned) {
            rc = sub_ensure_staging(c, e, true);
            if (rc != NTSCSIM_OK) return rc;
            uint8_t *hs = e->hsrc + e->fbytes * (size_t)sslot;
            if ((size_t)src_ls == e->pitch) std::memcpy(hs, src, span);
            else for (int y = 0; y < H; y++) std::memcpy(hs + e->pitch * (size_t)y, src + (size_t)src_ls * (size_t)y, rb);
            from = hs; from_ls = e->pitch;
            e->stats[3]++;
        }
        if (from_ls == e->pitch)
            HIPCHK(c, hipMemcpyAsync(dslot, from, e->pitch * (size_t)(H - 1) + rb, hipMemcpyHostToDevice, e->s_up));
        else
            HIPCHK(c, hipMemcpy2DAsync(dslot, e->pitch, from, from_ls, rb, (size_t)H, hipMemcpyHostToDevice, e->s_up));
        // the caller may rewrite src as soon as we return: the DMA must have read it
        if (pinned && !(flags & NTSCSIM_SUBMIT_SRC_STABLE)) HIPCHK(c, hipStreamSynchronize(e->s_up));
        e->src_ring_pos++;
        e->src_cur = sslot;
        e->stats[2]++;
    }
    // destination
    int row0, step, nrows;
    sub_rows(H, field, (flags & NTSCSIM_DESC_BOB) != 0, row0, step, nrows);
    const size_t dspan = (size_t)dst_ls * (size_t)(H - 1) + rb;
    SubmitEngine::Item it;
    it.ticket = t;
    it.src_slot = sslot;
    it.dst_slot = (int)(t % (uint64_t)e->nslots);
    it.host_dst = dst;
    it.host_dst_dev = sub_pinned(c, e, dst, dspan);
    it.dst_ls = dst_ls;
    // the decoder's 64-byte row bursts go straight into a pinned, 16-byte aligned frame; line doubling (every row written
    // twice) and unaligned frames keep the device ring + k_deliver
    it.direct = e->direct_ok && it.host_dst_dev && !(flags & NTSCSIM_DESC_BOB) &&
                !(((uintptr_t)it.host_dst_dev | (uintptr_t)dst_ls) & 15);
    {
        // delivery in submit order for fields that write the same rows of the same frame: the pending launch goes out
        // first, and the next one waits for every launch in flight that holds such a field
        const unsigned rows = sub_item_rows(flags, field);
        bool clash = false;
        for (const auto &pi : e->pending)
            if (sub_dst_conflict(dst, dst_ls, rows, pi.host_dst, pi.dst_ls, sub_item_rows(pi.flags, pi.field), W, H)) { clash = true; break; }
        if (clash) {
            rc = sub_launch(c);
            if (rc != NTSCSIM_OK) return rc;
        }
        for (const auto &ob : e->inflight) {
            bool hit = false;
            for (const auto &oi : ob.items)
                if (sub_dst_conflict(dst, dst_ls, rows, oi.host_dst, oi.dst_ls, sub_item_rows(oi.flags, oi.field), W, H)) { hit = true; break; }
            if (hit && std::find(e->pending_deps.begin(), e->pending_deps.end(), ob.first) == e->pending_deps.end())
                e->pending_deps.push_back(ob.first);
        }
    }
    it.field = field;
    it.flags = (flags & NTSCSIM_DESC_BOB) | (src_interlaced ? NTSCSIM_DESC_INTERLACED : 0u) | (src_tff ? NTSCSIM_DESC_TFF : 0u);
    it.fieldno = fieldno;
    it.rng_pos = c->rng_pos;
    c->rng_pos += ntscsim_rng_calls_per_field(&c->prm, W, H, field);
    e->src_last_ticket[(size_t)sslot] = t;
    e->pending.push_back(it);
    e->next_ticket++;
    e->stats[0]++;
    if (ticket) *ticket = t;
    if ((int)e->pending.size() >= e->o.depth) return sub_launch(c);
    return NTSCSIM_OK;
}
