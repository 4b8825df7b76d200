// The deferred launch graph (round 6) — included by gft_api.hip inside its anonymous namespace, after Buf / LazyOp.
//
// A Genfer program is a memoised recursion over input points (generating_function.rs:186-222, 609-627): statement L of a
// chain of `if`s is evaluated at the d + 1 distinct points of its depth d, the evaluations of one depth are independent of
// each other and have the same shapes — but the reference's evaluator visits them depth-first, so on one in-order stream
// every 5 us kernel waits for its predecessor while 255 of 256 CUs idle.  Rounds 3-5 recorded single operations and let a
// launch carry one or two "riders".  This is the general form: the operations the interpreter issues by the thousand —
// observation chains (with the consumer's Add as epilogue), two-chain Adds, nested Adds, proven linear Horner loops — are
// RECORDED on their result buffer (`Buf::lazy`, as before) and nothing is launched until somebody needs a value.  Then the
// recordings the value depends on are LEVELLED — level = longest path from tensors that are in memory — and every level is
// issued as one launch per kernel kind and launch geometry: a batch (gft_kernels.hpp ObsItem: blockIdx.y = item).  The
// evaluator, the ABI and the per-element operation order are untouched; a recording nobody reads is never launched.
//
//   * Output buffers of recordings are allocated when their level is issued (`Buf::p == nullptr` until then) and a
//     recording releases its inputs when its level has been issued, so the pool holds a few levels, not the program.
//   * The items of a level travel through an ARGUMENT ARENA: a pinned host ring mirrored in device memory, one
//     host-to-device copy per level, segments recycled behind events.
//   * "batch_dag" / GFT_BATCH = 0: every recording is launched as rounds 3-5 did (A/B, bisecting, the verification matrix).

struct DagRec {  // what the scheduler needs from a recording (implemented per kind in Ops<E>)
    virtual ~DagRec() {}
    // device buffers the launch reads (at most MAX_DEPS); recordings among them are predecessors in the graph
    static constexpr int MAX_DEPS = 4;
    struct Deps {
        Buf* b[MAX_DEPS];
        int n = 0;
        void push_back(Buf* x) {
            if (n < MAX_DEPS) b[n++] = x;
        }
        Buf** begin() { return b; }
        Buf** end() { return b + n; }
    };
    virtual void deps(Deps& out) = 0;
    // brings the inputs into memory (they are: their levels have been issued), allocates the output, appends the launch to
    // the current level's groups.  false: not a batch item after all — the caller launches it with LazyOp::run
    virtual bool emit(Buf* self) = 0;
};

// ---- argument arena -------------------------------------------------------------------------------------------------------
struct ArgArena {
    static constexpr size_t SEG = (size_t)8 << 20, NSEG = 8;
    unsigned char* h = nullptr;   // pinned host ring
    unsigned char* d = nullptr;   // its device mirror
    size_t seg = 0, used = 0;     // current segment, bytes used in it
    bool entered = false;
    hipEvent_t ev[NSEG] = {};
    bool pending[NSEG] = {};      // an event was recorded behind the last launch that reads the segment
    void init() {
        if (h) return;
        HIP_OK(hipHostMalloc((void**)&h, SEG * NSEG, hipHostMallocDefault));
        HIP_OK(hipMalloc((void**)&d, SEG * NSEG));
        for (auto& e : ev) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    void release() {
        if (!h) return;
        (void)hipHostFree(h);
        (void)hipFree(d);
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
        *this = ArgArena();
    }
    // room for `bytes` (<= SEG) contiguous bytes; returns the offset.  Entering a segment waits for the launches that read
    // its previous contents (almost always long finished); leaving one records the event behind those that read this one.
    size_t reserve(size_t bytes) {
        init();
        bytes = (bytes + 255) / 256 * 256;
        if (bytes > SEG) throw Error("internal: argument batch larger than an arena segment");
        if (!entered || used + bytes > SEG) {
            if (entered) {  // leave the current segment
                hipEvent_t e = ev[seg];
                hipStream_t st = R.stream;
                enqueue_task([e, st] { lq_note((hipEventRecord)(e, st), nullptr, "hipEventRecord (argument arena)"); });
                pending[seg] = true;
                seg = (seg + 1) % NSEG;
            }
            entered = true;
            used = 0;
            if (pending[seg]) {
                launch_drain();
                HIP_OK((hipEventSynchronize)(ev[seg]));
                pending[seg] = false;
            }
        }
        const size_t off = seg * SEG + used;
        used += bytes;
        return off;
    }
    // the host bytes at [off, off + bytes) go to the device mirror, in stream order
    void upload(size_t off, size_t bytes) {
        unsigned char* dst = d + off;
        const unsigned char* src = h + off;
        hipStream_t st = R.stream;
        enqueue_task([dst, src, bytes, st] {
            lq_note((hipMemcpyAsync)(dst, src, bytes, hipMemcpyHostToDevice, st), nullptr, "hipMemcpyAsync (argument arena)");
        });
    }
};
static ArgArena g_arena;

// ---- the groups of the level under construction ---------------------------------------------------------------------------
struct BatchGroup {
    // launches n items at device address `items`; n == 1 may come with the HOST copy (`host_item`) for a kernel-argument launch
    void (*launch)(const BatchGroup& g, const void* dev_items, const void* host_items);
    unsigned item_bytes = 0, n = 0;
    unsigned gx = 0, threads = 0;
    size_t lds = 0;
    int variant = 0;
    std::vector<unsigned char> bytes;
};
struct DagLevelCtx {
    std::vector<BatchGroup> groups;
    DagLevelCtx* prev = nullptr;
};
static DagLevelCtx* g_level = nullptr;  // innermost level under construction (nullptr: nothing is being scheduled)
static size_t g_dag_stats[5] = {0, 0, 0, 0, 0};  // {graph executions, recordings issued through them, batch launches, items in them, microseconds of the calling thread inside run_dag}

// room for one more item of the group (launch, geometry): the caller fills it in place (zeroed)
template <class IT>
static IT* batch_alloc(void (*launch)(const BatchGroup&, const void*, const void*), unsigned gx, unsigned threads, size_t lds, int variant) {
    static_assert(sizeof(IT) % 16 == 0, "batch items are copied in 16-byte pieces");
    if (!g_level) throw Error("internal: batch item outside a graph execution");
    BatchGroup* g = nullptr;
    for (auto& c : g_level->groups)
        if (c.launch == launch && c.gx == gx && c.threads == threads && c.lds == lds && c.variant == variant && c.item_bytes == sizeof(IT) &&
            (size_t)(c.n + 1) * sizeof(IT) <= ArgArena::SEG / 2 && c.n < 65535u) {
            g = &c;
            break;
        }
    if (!g) {
        g_level->groups.emplace_back();
        g = &g_level->groups.back();
        g->launch = launch;
        g->item_bytes = (unsigned)sizeof(IT);
        g->gx = gx;
        g->threads = threads;
        g->lds = lds;
        g->variant = variant;
        g->bytes.reserve(sizeof(IT) * 16);
    }
    const size_t at = g->bytes.size();
    g->bytes.resize(at + sizeof(IT));  // (value-initialised: zero)
    g->n++;
    return reinterpret_cast<IT*>(g->bytes.data() + at);
}
// issues the groups: one upload for all of them, one launch each
static void batch_flush(DagLevelCtx& L) {
    size_t total = 0;
    for (auto& g : L.groups)
        if (g.n > 1) total += (g.bytes.size() + 255) / 256 * 256;
    size_t off = 0;
    if (total) {
        // (a level larger than a segment goes up in several pieces)
        if (total <= ArgArena::SEG) {
            off = g_arena.reserve(total);
            size_t at = off;
            for (auto& g : L.groups)
                if (g.n > 1) {
                    std::memcpy(g_arena.h + at, g.bytes.data(), g.bytes.size());
                    at += (g.bytes.size() + 255) / 256 * 256;
                }
            g_arena.upload(off, total);
        }
    }
    size_t at = off;
    for (auto& g : L.groups) {
        if (g.n == 1) {
            g.launch(g, nullptr, g.bytes.data());
        } else if (g.n > 1) {
            size_t here;
            if (total <= ArgArena::SEG) {
                here = at;
                at += (g.bytes.size() + 255) / 256 * 256;
            } else {
                here = g_arena.reserve(g.bytes.size());
                std::memcpy(g_arena.h + here, g.bytes.data(), g.bytes.size());
                g_arena.upload(here, g.bytes.size());
            }
            g.launch(g, g_arena.d + here, g.bytes.data());
            g_dag_stats[2]++;
            g_dag_stats[3] += g.n;
        }
    }
    L.groups.clear();
}

// Executes the recordings `root` depends on, and `root`'s own, level by level.
static void run_dag(Buf* root) {
    static unsigned epoch = 0;
    ++epoch;
    struct Clock {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        bool outer = g_level == nullptr;
        ~Clock() {
            if (outer) g_dag_stats[4] += (size_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        }
    } clock;
    // 1. closure in post-order (iterative: the graph of a program is thousands of levels deep)
    struct Frame {
        Buf* b;
        DagRec::Deps deps;
        int next = 0;
    };
    std::vector<Buf*> order;
    std::vector<Frame> stack;
    auto open = [&](Buf* b) {
        b->dag_mark = epoch;
        b->dag_level = 0;
        stack.emplace_back();
        stack.back().b = b;
        b->lazy->rec->deps(stack.back().deps);
    };
    open(root);
    while (!stack.empty()) {
        Frame& f = stack.back();
        if (f.next < f.deps.n) {
            Buf* d = f.deps.b[f.next++];
            if (!d || d->host || !d->lazy) continue;         // in memory
            if (!d->lazy->rec) {                              // a recording of the old kind: launched where it stands
                force_buf(d);
                continue;
            }
            if (d->dag_mark == epoch) continue;               // seen (the graph is acyclic: its level is final when we return here)
            open(d);
            continue;
        }
        int lvl = 0;
        for (Buf* d : f.deps)
            if (d && !d->host && d->lazy && d->lazy->rec && d->dag_mark == epoch) lvl = std::max(lvl, d->dag_level + 1);
        f.b->dag_level = lvl;
        order.push_back(f.b);
        stack.pop_back();
    }
    // 2. levels
    int maxl = 0;
    for (Buf* b : order) maxl = std::max(maxl, b->dag_level);
    std::vector<std::vector<Buf*>> levels((size_t)maxl + 1);
    for (Buf* b : order) levels[(size_t)b->dag_level].push_back(b);
    g_dag_stats[0]++;
    g_dag_stats[1] += order.size();
    DagLevelCtx ctx;
    for (auto& nodes : levels) {
        // the recordings of this level stay alive (and with them their inputs: no pool block of an input is reused by an
        // output of the same level) until the level's launches have been issued
        std::vector<Rc<LazyOp>> keep;
        keep.reserve(nodes.size());
        ctx.prev = g_level;
        g_level = &ctx;
        try {
            for (Buf* b : nodes) {
                if (!b->lazy) continue;  // (launched by a nested execution)
                Rc<LazyOp> op = b->lazy;
                keep.push_back(op);
                if (!op->rec->emit(b)) {
                    ensure_alloc(b);
                    op->run(b);
                }
            }
            batch_flush(ctx);
        } catch (...) {
            g_level = ctx.prev;
            ctx.groups.clear();
            // (what was not launched is still a recording)
            throw;
        }
        g_level = ctx.prev;
        for (Buf* b : nodes) {
            b->lazy = nullptr;
        }
    }
}
