// CPU stress test of the claim / publish / read protocol of csrc/ehm_midtable.h: the header is
// compiled for the host with the device builtins mapped onto GCC atomics and fences (relaxed
// accesses + release / acquire fences, the same shape as on the device), threads stand in for
// wavefronts.  Checks: every request gets the value of ITS key, every key is solved exactly once
// (unless the neighbourhood was full), nobody hangs.
//   g++ -O2 -std=c++17 -pthread tests/native/midtable_stress.cpp -o /tmp/midtable_stress && /tmp/midtable_stress
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#define __device__
#define __HIP_MEMORY_SCOPE_AGENT 0
template <class T> static inline T __hip_atomic_load(T* p, int order, int) {
    uint64_t b = __atomic_load_n(reinterpret_cast<uint64_t*>(p), order);
    T v; memcpy(&v, &b, 8); return v;
}
template <class T> static inline void __hip_atomic_store(T* p, T v, int order, int) {
    uint64_t b; memcpy(&b, &v, 8);
    __atomic_store_n(reinterpret_cast<uint64_t*>(p), b, order);
}
static inline bool __hip_atomic_compare_exchange_strong(unsigned long long* p, unsigned long long* e,
                                                        unsigned long long d, int so, int fo, int) {
    return __atomic_compare_exchange_n(p, e, d, false, so, fo);
}
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(order)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) std::this_thread::yield()
static thread_local unsigned long long g_ballot_or = 0;
#define __builtin_amdgcn_ballot_w64(pred) ((g_ballot_or |= (pred) ? 1ull : 0ull), g_ballot_or)
static inline long long __double_as_longlong(double v) { long long b; memcpy(&b, &v, 8); return b; }
static inline long long wall_clock64() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(
        std::chrono::steady_clock::now().time_since_epoch()).count() / 10;      // 100 MHz ticks
}
#include "../../explicit_hybrid_mpc_amd/csrc/ehm_midtable.h"
using namespace ehm;

static double value_of(const double* k, int p) {
    double v = 1.0;
    for (int i = 0; i < p; ++i) v = v * 1.000001 + k[i] * (i + 1);
    return v;
}

int main() {
    const int p = 4, n_u = 2, n_keys = 3000, n_threads = 8, per_thread = 400000;
    const unsigned slots = 8192;
    std::vector<unsigned long long> state(slots, 0ull);
    std::vector<double> data((size_t)slots * MT_DOUBLES, 0.0);
    MidTable M{state.data(), data.data(), slots - 1};
    std::vector<double> keys((size_t)n_keys * p);
    std::mt19937_64 rng(1);
    for (auto& k : keys) k = (double)(int)(rng() % 64) / 64.0 - 0.5;   // a lattice: duplicates likely
    std::atomic<long long> owns{0}, hits{0}, nones{0}, wrong{0}, collisions{0};
    const long long t_start = wall_clock64();
    auto worker = [&](int tid) {
        std::mt19937_64 r(100 + tid);
        for (int it = 0; it < per_thread; ++it) {
            const double* mid = &keys[(size_t)(r() % n_keys) * p];
            unsigned idx;
            const unsigned long long tag = mt_tag(mid, p, M.mask, &idx);
            int slot = 0;
            const int res = mt_claim(M, tag, idx, t_start, 60LL * 100000000LL, &slot);
            double J;
            if (res == MT_HIT) {
                double e[64];
                bool same = true;
                g_ballot_or = 0;
                for (int lane = 0; lane < 64; ++lane) e[lane] = mt_read(M, slot, lane, mid, p, &same);
                if (!same) { ++collisions; J = value_of(mid, p); }
                else {
                    J = e[8];
                    if (e[10] != mid[0] * 2 || e[18] != mid[1] * 3 || (int)e[9] != (7 | (1 << 8) | (11 << 16)))
                        ++wrong;
                    ++hits;
                }
            } else {
                J = value_of(mid, p);
                if ((r() & 7) == 0) std::this_thread::yield();          // a solve takes a while
                if (res == MT_OWN) {
                    const double u0[2] = {mid[0] * 2, 0.0}, grad[4] = {mid[1] * 3, 0, 0, 0};
                    for (int lane = 63; lane >= 0; --lane)              // lane 0 flips the state: last
                        mt_publish(M, slot, tag, lane, mid, p, J, 7, 1, 11, u0, n_u, grad);
                    ++owns;
                } else {
                    ++nones;
                }
            }
            if (J != value_of(mid, p)) ++wrong;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(worker, t);
    for (auto& t : th) t.join();
    // distinct keys by value
    std::vector<std::vector<double>> seen;
    long long distinct = 0;
    {
        std::vector<unsigned long long> tags;
        for (int k = 0; k < n_keys; ++k) {
            unsigned idx;
            tags.push_back(mt_tag(&keys[(size_t)k * p], p, M.mask, &idx));
        }
        std::sort(tags.begin(), tags.end());
        distinct = std::unique(tags.begin(), tags.end()) - tags.begin();
    }
    printf("requests %lld: owns %lld (distinct keys %lld), hits %lld, without table %lld, key mismatches %lld, wrong %lld\n",
           (long long)n_threads * per_thread, owns.load(), distinct, hits.load(), nones.load(),
           collisions.load(), wrong.load());
    const bool ok = wrong == 0 && owns <= distinct && (nones > 0 || owns == distinct) &&
                    owns + hits + nones + collisions == (long long)n_threads * per_thread;
    printf(ok ? "OK\n" : "FAILED\n");
    return ok ? 0 : 1;
}
