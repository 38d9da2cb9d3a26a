// ehm_midtable.h -- table of midpoint optima shared by the wavefronts of the persistent frontier
// kernel (single-commutation runs).
//
// A split solves P_theta_delta at the midpoint of the split edge (lib/worker.py:406-407).  The
// simplices around an edge all bisect it at the same point and, with one commutation, solve the
// same LP there: 7.6 splits per distinct midpoint on a 21 k-node tree of the bench instance
// (tests/study_midpoint_sharing.py).  The first wavefront to ask for a midpoint claims a slot,
// solves and publishes (optimum, first input, cost gradient, solver status); the others take the
// published entry.  An LP's optimum does not depend on who solves it: the tree is the same.
//
// Open addressing in HBM, one 64-bit state word + MT_DOUBLES doubles per slot:
//     state = 0                     empty
//     state = tag | 1               claimed, being solved (tag = 62-bit hash of the coordinates)
//     state = tag | 3               published
// An entry never changes once published and nothing is ever removed, so the only wait is for a
// claimed slot with the asker's own tag -- its owner is running the solve and holds nothing it
// could wait for itself: no cycle, no deadlock, whatever the residency of the grid.  The payload
// is written through to the device coherence point (agent-scope atomic stores) and the state word
// follows after s_waitcnt, exactly like the node records and queue slots of the kernel (ehm_k2.hip);
// a reader that has seen "published" invalidates its non-coherent lines (acquire fence) and then
// compares the full key: a tag collision (2^-62 per pair) degrades to "solve it yourself".
#pragma once

namespace ehm {

constexpr int MT_DOUBLES = 32;   // [0,8) key | [8] J | [9] status | conv << 8 | iters << 16
                                 // | [10,18) first input | [18,26) dJ/dtheta | [26,32) unused
constexpr int MT_PROBES = 16;
enum { MT_NONE = 0, MT_OWN = 1, MT_HIT = 2, MT_BUSY = 3 };

struct MidTable {
    unsigned long long* state;   // nullptr: no table (every other engine, option share_midpoints 0)
    double* data;
    unsigned int mask;           // slots - 1 (a power of two)
};

// Hash of the midpoint's coordinates (bit patterns): slot index and tag.  Uniform over the wave.
__device__ inline unsigned long long mt_tag(const double* mid, int p, unsigned int mask,
                                            unsigned int* idx) {
    unsigned long long h = 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < p; ++i) {
        h ^= (unsigned long long)__double_as_longlong(mid[i]);
        h *= 0xbf58476d1ce4e5b9ull;
        h ^= h >> 29;
    }
    *idx = (unsigned int)(h >> 33) & mask;
    return (h | 4ull) & ~3ull;          // never 0, low two bits free for the phase
}

// The same table keyed by (point, kind): the multi-commutation engine asks for P_theta_delta and
// its phase-one form at the same parameter with the same commutation from every simplex around an
// edge or a vertex (entry[26] holds the kind and is compared like the coordinates).
__device__ inline unsigned long long pt_tag(const double* th, int p, unsigned int kind,
                                            unsigned int mask, unsigned int* idx) {
    unsigned long long h = 0x9e3779b97f4a7c15ull ^ ((unsigned long long)kind * 0xd6e8feb86659fd93ull);
    for (int i = 0; i < p; ++i) {
        h ^= (unsigned long long)__double_as_longlong(th[i]);
        h *= 0xbf58476d1ce4e5b9ull;
        h ^= h >> 29;
    }
    *idx = (unsigned int)(h >> 33) & mask;
    return (h | 4ull) & ~3ull;
}

// ONE lane: claim or find the slot of `tag`.  MT_OWN: *slot is ours, mt_publish must follow;
// MT_HIT: *slot holds a published entry with this tag; MT_NONE: neighbourhood full or the wait
// ran into the watchdog -- solve without the table; MT_BUSY (only with no_wait): the entry is
// being solved by another wavefront right now.
__device__ inline int mt_claim(const MidTable& M, unsigned long long tag, unsigned int idx,
                               long long t_start, long long watchdog_ticks, int* slot,
                               long long* waited = nullptr, bool no_wait = false) {
    for (int probe = 0; probe < MT_PROBES; ++probe, idx = (idx + 1u) & M.mask) {
        unsigned long long s = __hip_atomic_load(&M.state[idx], __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
        if (s == 0ull) {
            // (relaxed: the claim publishes nothing -- the owner's payload follows with mt_publish,
            // and a loser re-reads the word it lost to)
            unsigned long long expect = 0ull;
            if (__hip_atomic_compare_exchange_strong(&M.state[idx], &expect, tag | 1ull,
                                                     __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
                *slot = (int)idx;
                return MT_OWN;
            }
            s = expect;                 // somebody else took it: look at what it holds now
        }
        if ((s & ~3ull) != tag) continue;
        if ((s & 3ull) != 3ull) {       // claimed by a wavefront that is solving it right now
            if (no_wait) {              // the caller has something better to do than to sleep
                *slot = (int)idx;
                return MT_BUSY;
            }
            const long long t_wait = wall_clock64();
            do {
                if (wall_clock64() - t_start > watchdog_ticks) return MT_NONE;
                __builtin_amdgcn_s_sleep(8);
                s = __hip_atomic_load(&M.state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((s & 3ull) != 3ull);
            if (waited) *waited += wall_clock64() - t_wait;
        }
        *slot = (int)idx;
        return MT_HIT;
    }
    return MT_NONE;
}

// ANY lane, read-only: the slot of a PUBLISHED entry with this tag, or -1 (absent, or still being
// solved -- nobody waits here).  An empty slot ends the probe sequence: entries are claimed at the
// first empty slot of their sequence and never removed.
__device__ inline int mt_find(const MidTable& M, unsigned long long tag, unsigned int idx) {
    for (int probe = 0; probe < MT_PROBES; ++probe, idx = (idx + 1u) & M.mask) {
        const unsigned long long s = __hip_atomic_load(&M.state[idx], __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        if (s == 0ull) return -1;
        if ((s & ~3ull) == tag) return ((s & 3ull) == 3ull) ? (int)idx : -1;
    }
    return -1;
}

// ALL lanes of the owner: write the entry through, then flip the state word.
__device__ inline void mt_publish(const MidTable& M, int slot, unsigned long long tag, int lane,
                                  const double* mid, int p, double J, int status, int conv,
                                  int iters, const double* u0, int n_u, const double* grad) {
    double* e = M.data + (size_t)slot * MT_DOUBLES;
    if (lane < MT_DOUBLES) {
        double v = 0.0;
        if (lane < 8) v = lane < p ? mid[lane] : 0.0;
        else if (lane == 8) v = J;
        else if (lane == 9) v = (double)(status | (conv << 8) | (iters << 16));
        else if (lane < 18) v = lane - 10 < n_u ? u0[lane - 10] : 0.0;
        else if (lane < 26) v = (grad && lane - 18 < p) ? grad[lane - 18] : 0.0;
        __hip_atomic_store(e + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // Payload before state word, in the memory model's terms: the payload stores are agent-scope
    // (write-through: they reach the device coherence point without an L2 write-back), a
    // workgroup-scope release fence orders them before the state store for the compiler, and
    // s_waitcnt(0) holds the wavefront until the hardware has completed them.  A full agent-scope
    // RELEASE on the state store would say the same and cost a write-back of this XCD's whole L2
    // (measured in round 2: 47 GB per partition); readers pair it with an agent-scope ACQUIRE fence
    // (mt_read, the callers of mt_find).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0)
        __hip_atomic_store(&M.state[slot], tag | 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ALL lanes after MT_HIT: the entry's doubles (lane k < MT_DOUBLES gets entry[k]); *same = the
// stored key is the asker's midpoint, bit for bit (uniform).
__device__ inline double mt_read(const MidTable& M, int slot, int lane, const double* mid, int p,
                                 bool* same) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const double* e = M.data + (size_t)slot * MT_DOUBLES;
    const double v = lane < MT_DOUBLES ? e[lane] : 0.0;
    const bool differs = lane < p && __double_as_longlong(v) != __double_as_longlong(mid[lane]);
    *same = __builtin_amdgcn_ballot_w64(differs) == 0ull;
    return v;
}

}  // namespace ehm
