// CPU test of webdgs_amd/csrc/alloc_cache.h with a mock backend: two device ordinals, per-device epochs, the preference for safe blocks,
// the stamp a block freed DURING a synchronisation keeps.  Built and run by tests/test_alloc_cache.py (g++; no GPU, no HIP).
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "../../webdgs_amd/csrc/alloc_cache.h"

struct Mock {
    struct State {
        int mallocs = 0, frees = 0;
        std::vector<int> syncs;                 // device of every synchronisation, in order
        std::function<void(int)> during_sync;   // runs "while the device is being waited for"
        size_t budget = (size_t)1 << 40;
    };
    State* s;
    void* malloc(int device, size_t bytes) {
        (void)device;
        if (bytes > s->budget) return nullptr;
        s->budget -= bytes;
        s->mallocs++;
        return std::malloc(16);
    }
    void free(void* p) { s->frees++; std::free(p); }
    bool sync(int device) {
        s->syncs.push_back(device);
        if (s->during_sync) s->during_sync(device);
        return true;
    }
};

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) { std::printf("FAILED line %d: %s\n", __LINE__, #cond); return 1; } \
    } while (0)

int main() {
    const size_t LIMIT = (size_t)1 << 30;
    {   // size classes: eighth-of-a-power-of-two steps
        CHECK(wdgs::alloc_size_class(1) == 4096 && wdgs::alloc_size_class(4097) == 4608 && wdgs::alloc_size_class(1000000) == 1048576);
        CHECK(wdgs::alloc_size_class(1048577) == 1048576 + 131072);
    }
    {   // a block freed on device 0 is handed out again only behind a synchronisation of device 0 -- not of device 1
        Mock::State st;
        wdgs::AllocCache<Mock> c(Mock{&st});
        void* a0 = c.alloc(0, 10000, true);
        void* a1 = c.alloc(1, 10000, true);
        CHECK(a0 && a1 && st.mallocs == 2);
        CHECK(c.free(a0, LIMIT) && c.free(a1, LIMIT));
        CHECK(c.held(0) == wdgs::alloc_size_class(10000) && c.held(1) == c.held(0));
        void* b1 = c.alloc(1, 10000, true);                      // same class, device 1: its one block is fresh -> wait for device 1
        CHECK(b1 == a1 && st.syncs == std::vector<int>{1});
        void* b0 = c.alloc(0, 10000, true);                      // device 0's block is STILL fresh: device 1's wait said nothing about it
        CHECK(b0 == a0 && (st.syncs == std::vector<int>{1, 0}));
        CHECK(st.mallocs == 2 && c.synchronisations() == 2);
        c.free(b0, LIMIT); c.free(b1, LIMIT);
        c.release(0); c.release(1);
        CHECK(st.frees == 2);
    }
    {   // one synchronisation covers every block freed before it; a safe block is preferred over a fresh one; a recording never waits
        Mock::State st;
        wdgs::AllocCache<Mock> c(Mock{&st});
        void* p[3];
        for (auto& q : p) q = c.alloc(0, 5000, true);
        c.free(p[0], LIMIT); c.free(p[1], LIMIT);
        CHECK(c.alloc(0, 5000, false) != p[0] && st.mallocs == 4 && st.syncs.empty());   // recording: the fresh blocks stay where they are
        void* r = c.alloc(0, 5000, true);                       // waits once ...
        CHECK((r == p[0] || r == p[1]) && st.syncs.size() == 1);
        c.free(p[2], LIMIT);                                     // ... p[2] is freed AFTER that wait: fresh
        void* s2 = c.alloc(0, 5000, true);                       // the other old block is safe by the same wait: taken, no second wait
        CHECK((s2 == p[0] || s2 == p[1]) && s2 != r && st.syncs.size() == 1);
        void* s3 = c.alloc(0, 5000, false);                      // only p[2] is left, fresh, and we may not wait: a new block
        CHECK(s3 != p[2] && st.mallocs == 5);
        void* s4 = c.alloc(0, 5000, true);
        CHECK(s4 == p[2] && st.syncs.size() == 2);
    }
    {   // a block freed by another thread WHILE the device is being waited for keeps its stamp: the wait that was already running does not cover it
        Mock::State st;
        wdgs::AllocCache<Mock> c(Mock{&st});
        void* a = c.alloc(0, 7000, true);
        void* b = c.alloc(0, 7000, true);
        c.free(a, LIMIT);
        bool freed_during = false;
        st.during_sync = [&](int) { if (!freed_during) { freed_during = true; c.free(b, LIMIT); } };
        void* r = c.alloc(0, 7000, true);                       // waits (a is fresh); b is freed inside the wait
        CHECK(r == a && st.syncs.size() == 1 && freed_during);
        st.during_sync = nullptr;
        void* r2 = c.alloc(0, 7000, true);                      // b must NOT pass as safe: a second wait
        CHECK(r2 == b && st.syncs.size() == 2);
    }
    {   // the cache's holdings are given back when the backend runs out of memory, and blocks beyond the limit are not kept
        Mock::State st;
        st.budget = 3 * wdgs::alloc_size_class(100000);
        wdgs::AllocCache<Mock> c(Mock{&st});
        void* a = c.alloc(0, 100000, true);
        void* b = c.alloc(0, 100000, true);
        c.free(a, LIMIT); c.free(b, LIMIT);
        st.budget = wdgs::alloc_size_class(300000) - 1;          // not enough until the two cached blocks go back
        struct Refund { Mock::State* s; } refund{&st};
        (void)refund;
        const int frees_before = st.frees;
        void* big = c.alloc(0, 300000, true);
        CHECK(big == nullptr && st.frees == frees_before + 2 && c.held(0) == 0);   // released, and the mock's budget does not grow back: still a clean failure
        void* x = c.alloc(0, 64, true);
        CHECK(x && !c.free(x, 0) );                              // limit 0: not kept -- the caller frees it
        std::free(x);
        int unknown;
        CHECK(!c.free(&unknown, LIMIT));                         // not one of ours
    }
    std::printf("alloc_cache: ok\n");
    return 0;
}
