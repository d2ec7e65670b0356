// Handle types behind the C ABI (include/g16_mi355x.h): the context (streams, events, arena, timers, the prepared halves of the next
// call), device-resident key / circuit / distributed-map handles, and small host utilities (error text, clock, optional roctx ranges).
// One translation unit (api.hip) includes this and the four implementation headers next to it.
#pragma once
#include "internal.hpp"
#include "msm_common.hpp"
#include "fp30.hpp"
#include "fixed_base.hpp"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <future>
#include <mutex>
#include <thread>
#include <new>
#include <type_traits>
#include <dlfcn.h>

namespace g16 {

static thread_local std::string g_last_error;

void set_last_error(const char* what, hipError_t e, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s:%d: %s -> %s", file, line, what, hipGetErrorString(e));
    g_last_error = buf;
}

int Arena::alloc(size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    for (auto& c : chunks) {
        if (c.cap - c.used >= bytes) {
            *out = c.p + c.used;
            c.used += bytes;
            return G16_OK;
        }
    }
    Chunk c;
    c.cap = bytes > min_chunk ? bytes : min_chunk;
    c.used = 0;
    c.p = nullptr;
    G16_HIP_TRY(hipMalloc((void**)&c.p, c.cap));
    c.used = bytes;
    *out = c.p;
    chunks.push_back(c);
    return G16_OK;
}
void Arena::release() {
    for (auto& c : chunks) (void)hipFree(c.p);
    chunks.clear();
}

// roctx ranges with the reference's timer names (prover.rs:36,62,89,99,111,119: start_timer! / end_timer!), visible in
// `rocprofv3 --marker-trace`.  The library is resolved at run time and only when G16_ROCTX=1, so the product has no link-time
// dependency on a profiler.  The work under a range is ENQUEUED inside it (the GPU runs asynchronously); the phase durations
// themselves come from HIP events (g16_timings).
struct Roctx {
    typedef int (*push_t)(const char*);
    typedef int (*pop_t)();
    push_t push = nullptr;
    pop_t pop = nullptr;
    Roctx() {
        const char* e = getenv("G16_ROCTX");
        if (!e || atoi(e) == 0) return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<push_t>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<pop_t>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
    static const Roctx& get() { static const Roctx r; return r; }
};
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(Roctx::get().push != nullptr) { if (on) Roctx::get().push(name); }
    ~RoctxRange() { if (on) Roctx::get().pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace g16

using namespace g16;

struct g16_ctx {
    int curve;
    int device;
    hipStream_t stream;   // witness map, the five bucket passes back to back, the batched G1 reduction
    hipStream_t stream2;  // witness digit/sort pass beside the witness map; the G2 reduction of a whole-key proof
    hipStream_t stream3;  // h's digit/sort pass, underneath the first bucket pass
    hipStream_t stream_h2d = nullptr;  // a host assignment's upload, in pieces (ZUpload): the mat-vec's row blocks follow piece by piece
    hipEvent_t ev_up[8] = {};
    hipStream_t stream_wm = nullptr;   // g16_dwm_stage_async: the distributed witness map's stages (and the caller's exchanges between them)
    hipEvent_t ev_dwm = nullptr;
    hipEvent_t ev_heavy[4] = {};   // G1 MSM k's heavy-bucket combine (side stream) done
    hipEvent_t ev_edge[8] = {};    // timestamps on stream 1 at the boundaries of the bucket passes (see prove_partial)
    // g16_prove_partial_prepare: the witness digit/sort pass of the NEXT g16_prove_partial[_h] over (pk, z), already enqueued on
    // stream 2 (its buffers live in the arena, which that call then must not reset)
    struct Prepared {
        bool valid = false;
        const g16_pk* pk = nullptr;
        uint64_t key_id = 0;      // g16_pk::id: a freed key's ADDRESS may be reused by the next load, its id never is
        uint64_t a_start = 0, a_count = 0;   // the slice of z the sort covers and its window size, as the key had them then
        int c_z = 0;
        const uint64_t* z = nullptr;
        uint64_t n_assign = 0;
        ScalarSort sort_z;
    } prep;
    // forget a prepared sort of key `id` (the key is being freed or re-labelled as another rank's shard)
    void drop_prepared_sort(uint64_t id) {
        if (prep.valid && prep.key_id == id) { (void)hipStreamSynchronize(stream2); prep.valid = false; }
    }
    // a prepared sort that is being DROPPED (the next call is not the prove_partial it was made for, or a failed call left it
    // behind) may still be running on stream 2 inside arena buffers: wait for it before the arena is handed out again
    void reset_arena() {
        if (prep.valid) (void)hipStreamSynchronize(stream2);
        arena.reset();
        prep.valid = false;
    }
    hipStream_t red[5];   // per MSM: the heavy-bucket combine of G1 MSM k right after its pass ([0..3]); the G2 reduction of a sharded
                          // proof ([4]: short passes, it must not queue behind the witness sort on stream 2)
    Arena arena;
    g16_timings tm;
    EventTimer t_wm, t_prep_h, t_prep_z, t_bucket[5], t_ntt[2];
    hipEvent_t ev_z = nullptr, ev_h = nullptr, ev_wm = nullptr, ev_done[5] = {};
    // g16_prove_finalize_prepare: the (r, s)-only half of the host glue, running on a host thread (api.hip: FinalizePrep)
    struct FinPrep {
        bool valid = false;
        std::future<void> fut;
        std::shared_ptr<void> data;
        uint64_t key_id = 0;   // g16_pk::id of the key it was prepared for (ids are never reused, addresses are)
        uint64_t r[4] = {}, s[4] = {};
        bool matches(uint64_t id, const uint64_t* r_, const uint64_t* s_) const {
            return valid && key_id == id && memcmp(r, r_, 32) == 0 && memcmp(s, s_, 32) == 0;
        }
        void drop() {   // wait for a running thread and forget its result (the thread holds its own reference to the key's host half)
            if (fut.valid()) fut.wait();
            valid = false;
            data.reset();
            key_id = 0;
        }
    } finprep;
    void* pinned = nullptr;  // window sums land here (hipHostMalloc)
    size_t pinned_bytes = 0;
    // g16_ctx_create_multi: a multi-device context owns one full context per device and no device state of its own
    std::vector<g16_ctx*> subs;
    std::vector<char> peer;   // [i * n + j]: device i reaches device j's memory directly (hipDeviceCanAccessPeer + enabled); see g16_ctx_peer_access
};

// Error exits of the entry points that launch on several streams: kernels still in flight reference arena memory that the
// next call resets and reuses, so an early return first drains every stream of the ctx.
struct DrainOnError {
    g16_ctx* ctx;
    bool armed = true;
    explicit DrainOnError(g16_ctx* c) : ctx(c) {}
    void dismiss() { armed = false; }
    ~DrainOnError() {
        if (!armed) return;
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamSynchronize(ctx->stream3);
        (void)hipStreamSynchronize(ctx->stream_wm);
        (void)hipStreamSynchronize(ctx->stream_h2d);
        for (int i = 0; i < 5; ++i) (void)hipStreamSynchronize(ctx->red[i]);
    }
};

struct g16_dwm;
// multi-device context, device i: its side of the distributed witness map (g16_prove runs the four stages on every device's
// host thread and moves the chunks between the devices with peer copies -- the single-process form of the all-to-all)
struct DwmSlot {
    g16_dwm* dwm = nullptr;
    uint64_t *work[3] = {nullptr, nullptr, nullptr}, *recv[3] = {nullptr, nullptr, nullptr}, *h_local = nullptr;   // M Fr each
    mutable uint64_t* h_full = nullptr;  // domain_size Fr: every device's block of h, back to back (the all-gather of a bucket-space key);
                                         // allocated by the first g16_prove with such a key (base-range keys never pay for it)
    uint64_t* z_dev = nullptr;   // num_variables Fr: a host assignment is uploaded once per proof and device
};

struct g16_circuit {
    int curve;
    g16_ctx* ctx;
    void* dc;  // DeviceCircuit<C>*
    uint64_t domain_size;
    std::vector<g16_circuit*> subs;   // multi-device context: the circuit replicated on every device (dc == nullptr)
    std::vector<DwmSlot> dist;        // multi-device context whose device count admits the distributed witness map
    uint64_t num_variables = 0;
};

// The host-side half of a key: the eight fixed points the glue of prover.rs:76-131 reads and the multiples of delta_g1 / delta_g2
// built for it (fixed_base.hpp).  Held by shared_ptr: the (r, s)-only half of the glue runs on a host thread of whichever context
// prepared it (g16_prove_finalize_prepare) and keeps this alive on its own, so g16_pk_free never pulls it from under a thread of
// ANOTHER context that shares the key (throughput mode) -- the device arrays go at once, this goes with its last reader.
template <class C>
struct KeyGlue {
    typedef typename C::G1A G1A;
    typedef typename C::G2A G2A;
    G1A alpha_g1, beta_g1, delta_g1, a_query0, b_g1_query0;
    G2A beta_g2, delta_g2, b_g2_query0;
    // Built by the SECOND finalize over the key (~35 ms of host work once, ~0.5 ms saved per later proof): a key that proves once never pays.
    FixedBaseTable<typename C::G1X> delta1_tab;
    FixedBaseTable<typename C::G2X> delta2_tab;
    std::mutex tab_mu;
    int finalize_calls = 0;
    // set (release) only AFTER both tables are complete; readers that are not inside tab_mu look at this flag (acquire), never at the
    // tables' own emptiness: a key may serve two contexts at once, and a reader must not walk a table another thread is still filling
    std::atomic<bool> tabs_ready{false};
};

template <class C>
struct DevicePk {
    typedef typename C::G1A G1A;
    typedef typename C::G2A G2A;
    std::shared_ptr<KeyGlue<C>> glue = std::make_shared<KeyGlue<C>>();
    G1A *a = nullptr, *b_g1 = nullptr, *h = nullptr, *l = nullptr;
    G2A* b_g2 = nullptr;
    uint64_t a_start = 0, a_count = 0, b_g1_start = 0, b_g1_count = 0, b_g2_start = 0, b_g2_count = 0;
    uint64_t h_start = 0, h_count = 0, l_start = 0, l_count = 0;
    // window size of the precomputed window tables (msm.hip, merged windows): every query array then holds W rows of
    // `count` points, row j = 2^(cj) * query.  0 = no tables (plain bases, per-window buckets).  a, b_g1, b_g2 and l share
    // the witness sort and therefore one window size; h has its own.
    int c_z = 0, c_h = 0;
    // bucket-space shard (g16_pk_load_bucket_shard): the key is held WHOLE and this rank owns the buckets b mod shard_n == shard_r of
    // every MSM (MsmPlan::shard_n); 1 = the key (or base-range shard) is proved over all its buckets
    int shard_n = 1, shard_r = 0;
    // why the key is held the way it is (g16_pk_info): 0 window tables as planned; 1 plain bases by request (G16_MSM_PRECOMP=0);
    // 2 plain bases because a query is too long for merged entries; 3 plain bases because the tables did not fit (allocation failed
    // or G16_PK_TABLE_BUDGET_MB) -- a slower prover (c <= 16, more windows), which the caller can now see
    int table_fallback = 0;
    uint64_t table_bytes = 0;   // device bytes of the five query arrays as held
};

// every key handle gets a fresh id: a prepared finalize half is matched by id, never by the handle's address (a freed key's
// address can be handed out again to the next g16_pk_load)
inline uint64_t next_key_id() { static std::atomic<uint64_t> n{1}; return n.fetch_add(1); }

struct g16_pk {
    int curve;
    g16_ctx* ctx;
    void* dp;  // DevicePk<C>*
    std::vector<g16_pk*> subs;        // multi-device context: shard i of the key on device i (dp == nullptr)
    uint64_t dist_n = 0;              // != 0: the h shards are gathered in the block order of the distributed witness map over a
                                      // domain of dist_n points (h_query holds dist_n - 1 bases, generator.rs:168)
    bool bucket_mode = false;         // multi-device key cut in BUCKET space: every device holds the whole key's tables (h_query in the
                                      // order the all-gathered blocks arrive in) and owns the buckets b mod n_dev == device index
    uint64_t id = next_key_id();
};

struct g16_dwm {
    int curve;
    g16_ctx* ctx;
    const g16_circuit* circuit;
    void* dw;   // DistWm<C>*
    int rank, world;
    uint64_t local_size;
};


namespace {

template <class T>
T load_pod(const uint64_t* p) {
    T t;
    memcpy(&t, p, sizeof(T));
    return t;
}
// partial sums cross the ABI as raw XYZZ limbs (g16_partial)
template <class X>
void store_xyzz(uint64_t* dst, const X& p) { memcpy(dst, &p, sizeof(X)); }
template <class X>
X load_xyzz(const uint64_t* src) { X p; memcpy(&p, src, sizeof(X)); return p; }

}  // namespace
