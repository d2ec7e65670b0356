// g16_ctx_create_multi support: one host thread per device, error aggregation, the host barrier of the in-library distributed
// witness map, and the admissibility rule of that map.
#pragma once
#include "api_types.hpp"

static void g16_dwm_free_impl(g16_dwm* d) {
    (void)hipSetDevice(d->ctx->device);
    if (d->curve == G16_BLS12_381) dwm_destroy<Bls12_381>(static_cast<DistWm<Bls12_381>*>(d->dw));
    else dwm_destroy<Bn254>(static_cast<DistWm<Bn254>*>(d->dw));
    d->dw = nullptr;
}

// run fn(i) for i < n on n host threads.  Status: the first REAL failure by device index -- a thread that only gave up because
// a sibling failed returns SIBLING_FAILED, which never masks the sibling's own code -- and that thread's error text becomes the
// caller's g16_last_error() (g_last_error is thread_local).  Every thread is created before any of them runs fn (a start gate):
// if thread creation fails part-way nobody has entered a barrier that expects n participants.
// serial = true runs them one after the other on the calling thread.
static constexpr int SIBLING_FAILED = -1;
template <class Fn>
static int for_each_device(int n, Fn fn, bool serial = false) {
    if (serial) {
        for (int i = 0; i < n; ++i) {
            const int rc = fn(i);
            if (rc) return rc;
        }
        return G16_OK;
    }
    std::vector<int> rc((size_t)n, G16_OK);
    std::vector<std::string> msg((size_t)n);
    std::mutex mu;
    std::condition_variable cv;
    int gate = 0;   // 0: wait, 1: go, -1: abort
    auto body = [&](int i) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return gate != 0; });
            if (gate < 0) return;
        }
        rc[(size_t)i] = fn(i);
        if (rc[(size_t)i]) msg[(size_t)i] = g_last_error;
    };
    std::vector<std::thread> th;
    try {
        for (int i = 1; i < n; ++i) th.emplace_back(body, i);
    } catch (...) {
        { std::lock_guard<std::mutex> lk(mu); gate = -1; }
        cv.notify_all();
        for (auto& t : th) t.join();
        g_last_error = "could not start one host thread per device";
        return G16_ERR_INTERNAL;
    }
    { std::lock_guard<std::mutex> lk(mu); gate = 1; }
    cv.notify_all();
    rc[0] = fn(0);
    if (rc[0]) msg[0] = g_last_error;
    for (auto& t : th) t.join();
    int pick = -1;
    for (int i = 0; i < n && pick < 0; ++i) if (rc[(size_t)i] != G16_OK && rc[(size_t)i] != SIBLING_FAILED) pick = i;
    if (pick >= 0) { g_last_error = msg[(size_t)pick]; return rc[(size_t)pick]; }
    for (int v : rc) if (v) return G16_ERR_INTERNAL;   // only markers: cannot happen (a marker needs a failed sibling)
    return G16_OK;
}

// reusable barrier for the per-device host threads of one call (C++17: no std::barrier)
struct HostBarrier {
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0;
    uint64_t phase = 0;
    explicit HostBarrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t my = phase;
        if (++waiting == n) { waiting = 0; ++phase; cv.notify_all(); return; }
        cv.wait(lk, [&] { return phase != my; });
    }
};

// the distributed witness map runs over `world` ranks when world is a power of two in [2, 16] and world^2 divides the domain
static bool dist_wm_admissible(int world, uint64_t domain) {
    if (world < 2 || world > 16 || (world & (world - 1)) != 0) return false;
    if (domain == 0 || (domain & (domain - 1)) != 0) return false;
    return domain % ((uint64_t)world * (uint64_t)world) == 0;
}

// A context may list one physical device several times (tests: N shards on the one GPU of the box).  The LOAD paths of such a
// context run one after the other: n concurrent window-table builds on sibling queues of ONE device aborted inside the HIP runtime
// in the full test suite of round 2 (never in isolation, never root-caused; the builders were since rewritten without their 9-17 KB
// of scratch per lane and a concurrent run of the suite passed in round 3, but an abort cannot be caught and retried, so the safe
// order is the default).  Distinct devices -- the case that matters -- always load concurrently.  G16_MULTI_CONCURRENT_LOAD=1
// loads a repeated device concurrently too.
static bool serial_loads(const g16_ctx* ctx) {
    const char* e = getenv("G16_MULTI_CONCURRENT_LOAD");
    if (e && atoi(e) != 0) return false;
    for (size_t a = 0; a < ctx->subs.size(); ++a)
        for (size_t b = a + 1; b < ctx->subs.size(); ++b)
            if (ctx->subs[a]->device == ctx->subs[b]->device) return true;
    return false;
}

// static multiply-add counts of the bucket kernels' arithmetic, from the tables the kernels themselves are generated from
template <class B30>
