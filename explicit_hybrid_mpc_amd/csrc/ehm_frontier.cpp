// ehm_frontier.cpp -- the native partition driver of include/ehm_frontier.h.
//
// Plain C++ on the public entry points of this library: the searches' bookkeeping of
// ehm_search.h (point ids, memo of phase-one verdicts, lockstep descents, best-first queues) and
// the batched table solvers of ehmpc.h (ehm_point_idx_batch, ehm_simplex_idx_batch,
// ehm_problem_update_blocks, ehm_split_batch).  What used to be the interpreter's share of a
// configs[4] cell -- the round loop of bnb_frontier.grow_frontier, the pair lists of every launch,
// the slot maps of the two tables, the condensation of the relaxation blocks (numpy) -- is here.
//
// Per-cell semantics: lib/worker.py:241-417; the oracles' canonical answers: bnb.py / sequences.py
// (whose decisions this file reproduces one for one; tests/test_host_frontier_native.py runs both
// drivers on the CPU statement of the table and compares the trees).

#include "../../include/ehm_frontier.h"
#include "../../include/ehm_search.h"
#include "../../include/ehmpc.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <atomic>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

struct Fail {                                  // thrown inside the driver, turned into a code
    int code;
    std::string msg;
};
[[noreturn]] void raise(int code, const char* fmt, ...) {
    char buf[400];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Fail{code, buf};
}
void chk_search(int rc, const char* what) {
    if (rc) raise(rc, "%s: %s", what, ehm_search_last_error());
}
void chk_dev(int rc, const char* what) {
    if (rc) raise(rc, "%s: %s", what, ehm_last_error());
}

constexpr double FEAS_TOL = 1e-8;              // EHM_FEAS_TOL of ehm_capi.hip (sequences.py)
constexpr double SLIVER_TOL = 1e-7;            // EHM_SLIVER_TOL
constexpr double INHERIT_GUARD = 1e-6;         // bnb_frontier.INHERIT_GUARD
constexpr int BATCH = 16;                      // bnb.BATCH
constexpr int PID_BITS = 38;
constexpr size_t INHERIT_MAX = 2048;           // bounds handed down before the non-refuting ones are
                                               // dropped (bnb_frontier.INHERIT_MAX: 8192; the copies cost
                                               // more than the few problems the rest saves)
constexpr int64_t FEAS_MEMO_LIMIT = 3000000;   // sequences.PrefixSearch.FEAS_MEMO_LIMIT
constexpr size_t OPTIMA_MEMO_LIMIT = 2000000;
const double INF = std::numeric_limits<double>::infinity();

double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- the law and the condensation of prefix relaxations -----------------------------------------
struct Law {
    int n_x = 0, n_u = 0, n_modes = 0, N = 0, n_gx = 0, n_gu = 0, n_q = 0, n_r = 0, max_reg = 0;
    std::vector<double> A, B, w, Hx, hx, Gx, gx, Gu, gu, Q, R;
    std::vector<int> reg_rows, reg_off;
};

// (G, w, S) of the relaxation shared by every sequence that starts with a prefix, over the
// variables z = [u_0 .. u_{H-1} | ex_1 .. ex_H | eu_0 .. eu_{H-1}] of the law with horizon H:
// PWAMPC.condense_prefix (explicit_hybrid_mpc_amd/mpc_library.py) -- the rows of the undecided
// steps are 0 <= 1, their epigraph rows e >= 0.
struct Condenser {
    const Law* L = nullptr;
    int H = 0, nU = 0, n = 0, m = 0, off_u = 0, off_q = 0, off_r = 0, off_reg = 0;
    std::vector<double> tG, tw, tS;            // the block of the empty prefix
    std::vector<double> Phi, Gam, om, Phi2, Gam2, om2;

    void init(const Law* law, int horizon) {
        L = law; H = horizon;
        const int nx = L->n_x, nu = L->n_u;
        nU = H * nu;
        n = nU + 2 * H;
        off_u = H * L->n_gx;
        off_q = off_u + H * L->n_gu;
        off_r = off_q + 2 * H * L->n_q;
        off_reg = off_r + 2 * H * L->n_r;
        m = off_reg + H * L->max_reg;
        tG.assign((size_t)m * n, 0.0);
        tw.assign((size_t)m, 1.0);             // 0 <= 1 wherever nothing else is written
        tS.assign((size_t)m * nx, 0.0);
        for (int k = 0; k < H; ++k)            // input constraints Gu u_k <= gu
            for (int r = 0; r < L->n_gu; ++r) {
                const int row = off_u + k * L->n_gu + r;
                for (int c = 0; c < nu; ++c) tG[(size_t)row * n + k * nu + c] = L->Gu[(size_t)r * nu + c];
                tw[row] = L->gu[r];
            }
        for (int kk = 1; kk <= H; ++kk)        // +-Q x_kk <= ex_kk, relaxed: ex_kk >= 0
            for (int b = 0; b < 2; ++b)
                for (int r = 0; r < L->n_q; ++r) {
                    const int row = off_q + (2 * (kk - 1) + b) * L->n_q + r;
                    tG[(size_t)row * n + nU + (kk - 1)] = -1.0;
                    tw[row] = 0.0;
                }
        for (int k = 0; k < H; ++k)            // +-R u_k <= eu_k
            for (int b = 0; b < 2; ++b)
                for (int r = 0; r < L->n_r; ++r) {
                    const int row = off_r + (2 * k + b) * L->n_r + r;
                    const double sgn = b ? -1.0 : 1.0;
                    for (int c = 0; c < nu; ++c)
                        tG[(size_t)row * n + k * nu + c] = sgn * L->R[(size_t)r * nu + c];
                    tG[(size_t)row * n + nU + H + k] = -1.0;
                    tw[row] = 0.0;
                }
        Phi.resize((size_t)nx * nx); Gam.resize((size_t)nx * nU); om.resize(nx);
        Phi2 = Phi; Gam2 = Gam; om2 = om;
    }
    // rows [row, row + nr) <- M (nr x n_x) applied to the state (Phi, Gam, om):
    //   G = sgn M Gam,  w = rhs - sgn M om,  S = -sgn M Phi
    void rows(double* G, double* w, double* S, int row, const double* M, const double* rhs, int nr,
              double sgn) const {
        const int nx = L->n_x;
        for (int r = 0; r < nr; ++r) {
            double* g = G + (size_t)(row + r) * n;
            for (int c = 0; c < nU; ++c) {
                double acc = 0.0;
                for (int t = 0; t < nx; ++t) acc += sgn * M[(size_t)r * nx + t] * Gam[(size_t)t * nU + c];
                g[c] = acc;
            }
            double a = 0.0;
            for (int t = 0; t < nx; ++t) a += sgn * M[(size_t)r * nx + t] * om[t];
            w[row + r] = (rhs ? rhs[r] : 0.0) - a;
            for (int c = 0; c < nx; ++c) {
                double acc = 0.0;
                for (int t = 0; t < nx; ++t) acc += sgn * M[(size_t)r * nx + t] * Phi[(size_t)t * nx + c];
                S[(size_t)(row + r) * nx + c] = -acc;
            }
        }
    }
    void block(const int* prefix, int len, double* G, double* w, double* S) {
        const int nx = L->n_x, nu = L->n_u;
        std::memcpy(G, tG.data(), tG.size() * 8);
        std::memcpy(w, tw.data(), tw.size() * 8);
        std::memcpy(S, tS.data(), tS.size() * 8);
        std::fill(Phi.begin(), Phi.end(), 0.0);
        for (int i = 0; i < nx; ++i) Phi[(size_t)i * nx + i] = 1.0;
        std::fill(Gam.begin(), Gam.end(), 0.0);
        std::fill(om.begin(), om.end(), 0.0);
        int row_reg = off_reg;
        for (int j = 0; j < len; ++j) {
            const int mode = prefix[j];
            if (L->reg_rows[mode]) {           // the mode's region holds x_j
                rows(G, w, S, row_reg, &L->Hx[(size_t)L->reg_off[mode] * nx],
                     &L->hx[L->reg_off[mode]], L->reg_rows[mode], 1.0);
                row_reg += L->reg_rows[mode];
            }
            const double* A = &L->A[(size_t)mode * nx * nx];
            const double* B = &L->B[(size_t)mode * nx * nu];
            for (int r = 0; r < nx; ++r) {
                for (int c = 0; c < nx; ++c) {
                    double acc = 0.0;
                    for (int t = 0; t < nx; ++t) acc += A[(size_t)r * nx + t] * Phi[(size_t)t * nx + c];
                    Phi2[(size_t)r * nx + c] = acc;
                }
                for (int c = 0; c < nU; ++c) {
                    double acc = 0.0;
                    for (int t = 0; t < nx; ++t) acc += A[(size_t)r * nx + t] * Gam[(size_t)t * nU + c];
                    Gam2[(size_t)r * nU + c] = acc;
                }
                for (int c = 0; c < nu; ++c) Gam2[(size_t)r * nU + j * nu + c] += B[(size_t)r * nu + c];
                double acc = 0.0;
                for (int t = 0; t < nx; ++t) acc += A[(size_t)r * nx + t] * om[t];
                om2[r] = acc + L->w[(size_t)mode * nx + r];
            }
            Phi.swap(Phi2); Gam.swap(Gam2); om.swap(om2);
            const int kk = j + 1;              // x_kk: state constraints and the +-Q epigraphs
            rows(G, w, S, (kk - 1) * L->n_gx, L->Gx.data(), L->gx.data(), L->n_gx, 1.0);
            for (int b = 0; b < 2; ++b)
                rows(G, w, S, off_q + (2 * (kk - 1) + b) * L->n_q, L->Q.data(), nullptr, L->n_q,
                     b ? -1.0 : 1.0);
        }
    }
};

int copy_law(const ehm_pwa_law* in, Law& L) {
    if (!in || in->n_x < 1 || in->n_u < 1 || in->n_modes < 1 || in->N < 1 || !in->A || !in->B ||
        !in->w || !in->region_rows || in->n_gx < 0 || in->n_gu < 0 || in->n_q < 0 || in->n_r < 0)
        return fail(EHM_E_INVALID, "ehm_pwa_law: bad dimensions or NULL matrices");
    L.n_x = in->n_x; L.n_u = in->n_u; L.n_modes = in->n_modes; L.N = in->N;
    L.n_gx = in->n_gx; L.n_gu = in->n_gu; L.n_q = in->n_q; L.n_r = in->n_r;
    const size_t nx = L.n_x, nu = L.n_u, nm = L.n_modes;
    L.A.assign(in->A, in->A + nm * nx * nx);
    L.B.assign(in->B, in->B + nm * nx * nu);
    L.w.assign(in->w, in->w + nm * nx);
    L.reg_rows.assign(in->region_rows, in->region_rows + nm);
    L.reg_off.assign(nm, 0);
    int tot = 0;
    L.max_reg = 0;
    for (size_t i = 0; i < nm; ++i) {
        if (L.reg_rows[i] < 0) return fail(EHM_E_INVALID, "ehm_pwa_law: negative region_rows");
        L.reg_off[i] = tot;
        tot += L.reg_rows[i];
        L.max_reg = std::max(L.max_reg, L.reg_rows[i]);
    }
    if (tot && (!in->Hx || !in->hx)) return fail(EHM_E_INVALID, "ehm_pwa_law: Hx / hx is NULL");
    if (tot) { L.Hx.assign(in->Hx, in->Hx + (size_t)tot * nx); L.hx.assign(in->hx, in->hx + tot); }
    if ((L.n_gx && (!in->Gx || !in->gx)) || (L.n_gu && (!in->Gu || !in->gu)) || (L.n_q && !in->Q) ||
        (L.n_r && !in->R))
        return fail(EHM_E_INVALID, "ehm_pwa_law: a constraint or weight matrix is NULL");
    if (L.n_gx) { L.Gx.assign(in->Gx, in->Gx + (size_t)L.n_gx * nx); L.gx.assign(in->gx, in->gx + L.n_gx); }
    if (L.n_gu) { L.Gu.assign(in->Gu, in->Gu + (size_t)L.n_gu * nu); L.gu.assign(in->gu, in->gu + L.n_gu); }
    if (L.n_q) L.Q.assign(in->Q, in->Q + (size_t)L.n_q * nx);
    if (L.n_r) L.R.assign(in->R, in->R + (size_t)L.n_r * nu);
    return EHM_OK;
}

// ---- the device form of the pair solvers (sequences.PrefixTable / SplitPrefixTable) -------------
struct Table {
    ehm_problem* P = nullptr;
    int slots = 0, horizon = 0, n = 0, m = 0, full_length = -1;
    std::unordered_map<uint64_t, int> slot_of;
    std::vector<int> slot_len;
    std::vector<uint64_t> slot_code;           // what a slot holds (for eviction)
    std::vector<int64_t> slot_stamp;           // the launch that used it last
    int64_t stamp = 0, evicted = 0;
    Condenser cond;
    std::vector<int64_t> counts;               // [5][N + 1]
    std::vector<double> bG, bw, bS;            // staging of the blocks that go up
    int64_t lp = 0, loaded = 0;
};

struct DeviceSolver {
    Law law;
    int device = 0, short_len = 0, p = 0, nv = 0, n_u = 0, N = 0;
    uint64_t base = 1;
    // one table per horizon, ascending; tab_of_len[k] = the table of the prefixes of k steps (the
    // smallest horizon that holds them: the relaxation of a prefix constrains and prices its own
    // steps only, so as a block of the law with a shorter horizon it is the same problem in
    // fewer columns and rows -- PWAMPC.with_horizon)
    std::vector<Table> tab;
    std::vector<int> tab_of_len;
    int64_t total_lp() const { int64_t a = 0; for (const Table& T : tab) a += T.lp; return a; }
    int64_t total_loaded() const { int64_t a = 0; for (const Table& T : tab) a += T.loaded; return a; }
    // (the tables of one solver call run on threads of their own: shared tallies are atomic)
    std::atomic<int64_t> stalled{0}, slivers{0}, stalled_relax{0}, launches{0}, accepted_inaccurate{0};
    bool parallel_tables = true;
    // sequences.decisive_inaccurate: status word 1 | (decade << 8) of a stalled solve (ehm_dev.h)
    static bool decisive(int32_t status, double value) {
        const int dec = (status >> 8) & 0xff;
        if (!(status & 1) || dec > 6 || !std::isfinite(value)) return false;
        const double err = 100.0 * std::pow(10.0, dec - 10) * (1.0 + std::fabs(value));
        return std::fabs(value) > err;
    }

    int len_of(uint64_t code) const {
        int k = 0;
        while (code) { ++k; code /= base; }
        return k;
    }
    void digits(uint64_t code, int* out) const {
        int k = 0;
        while (code) { out[k++] = (int)(code % base) - 1; code /= base; }
    }
    void tally(Table& T, int kind, const std::vector<int32_t>& slot) {
        for (int32_t s : slot) ++T.counts[(size_t)kind * (N + 1) + T.slot_len[s]];
    }
    // slots of the distinct prefixes `uniq` (at most T.slots of them), loading what is missing; a
    // full table gives up the blocks that have gone unused the longest (never one of `uniq`)
    void ensure(Table& T, const std::vector<uint64_t>& uniq, std::vector<int32_t>& slot_u) {
        ++T.stamp;
        std::vector<uint64_t> missing;
        for (uint64_t c : uniq) {
            auto it = T.slot_of.find(c);
            if (it == T.slot_of.end()) missing.push_back(c);
            else T.slot_stamp[(size_t)it->second] = T.stamp;
        }
        if (!missing.empty()) {
            std::vector<int> place(missing.size());
            const int64_t n_free = (int64_t)T.slots - (int64_t)T.slot_of.size();
            size_t k = 0;
            int next_free = (int)T.slot_of.size();       // slots fill up from 0 before any eviction
            for (; k < missing.size() && (int64_t)k < n_free; ++k) place[k] = next_free++;
            if (k < missing.size()) {
                std::vector<int> victims;
                victims.reserve((size_t)T.slots);
                for (int sl = 0; sl < T.slots; ++sl)
                    if (T.slot_stamp[(size_t)sl] != T.stamp) victims.push_back(sl);
                const size_t need = missing.size() - k;
                if (victims.size() < need) raise(EHM_E_CAPACITY, "a launch names more prefixes than the table holds");
                std::partial_sort(victims.begin(), victims.begin() + (long)need, victims.end(),
                                  [&](int x, int y) { return T.slot_stamp[(size_t)x] < T.slot_stamp[(size_t)y]; });
                for (size_t v = 0; k < missing.size(); ++k, ++v) {
                    place[k] = victims[v];
                    T.slot_of.erase(T.slot_code[(size_t)victims[v]]);
                    ++T.evicted;
                }
            }
            // contiguous runs of slots go up together (the staging buffers stay small)
            std::vector<size_t> order(missing.size());
            for (size_t q = 0; q < order.size(); ++q) order[q] = q;
            std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return place[x] < place[y]; });
            const size_t nG = (size_t)T.m * T.n, nS = (size_t)T.m * p;
            const size_t group = 256;
            int pre[64];
            for (size_t g0 = 0; g0 < order.size();) {
                size_t cnt = 1;
                while (g0 + cnt < order.size() && cnt < group &&
                       place[order[g0 + cnt]] == place[order[g0]] + (int)cnt)
                    ++cnt;
                T.bG.resize(cnt * nG); T.bw.resize(cnt * T.m); T.bS.resize(cnt * nS);
                for (size_t q = 0; q < cnt; ++q) {
                    const uint64_t c = missing[order[g0 + q]];
                    const int len = len_of(c);
                    digits(c, pre);
                    T.cond.block(pre, len, &T.bG[q * nG], &T.bw[q * T.m], &T.bS[q * nS]);
                    const int sl = place[order[g0 + q]];
                    T.slot_of[c] = sl;
                    T.slot_len[(size_t)sl] = len;
                    T.slot_code[(size_t)sl] = c;
                    T.slot_stamp[(size_t)sl] = T.stamp;
                }
                chk_dev(ehm_problem_update_blocks(T.P, place[order[g0]], (int)cnt, T.bG.data(), T.bw.data(),
                                                  T.bS.data()), "ehm_problem_update_blocks");
                g0 += cnt;
            }
            T.loaded += (int64_t)missing.size();
        }
        slot_u.resize(uniq.size());
        for (size_t k = 0; k < uniq.size(); ++k) slot_u[k] = T.slot_of[uniq[k]];
    }
    // the pairs `sel` (indices into the caller's arrays) in parts that name at most T.slots
    // distinct prefixes each; f(part indices, their slots)
    template <class F>
    void chunks(Table& T, const uint64_t* code, const std::vector<int64_t>& sel, F f) {
        std::unordered_map<uint64_t, int64_t> order;
        std::vector<uint64_t> uniq;
        std::vector<int64_t> ord(sel.size());
        for (size_t k = 0; k < sel.size(); ++k) {
            auto it = order.find(code[sel[k]]);
            if (it == order.end()) {
                it = order.emplace(code[sel[k]], (int64_t)uniq.size()).first;
                uniq.push_back(code[sel[k]]);
            }
            ord[k] = it->second;
        }
        std::vector<int32_t> slot_u, slot;
        std::vector<int64_t> part;
        for (size_t c0 = 0; c0 < uniq.size(); c0 += (size_t)T.slots) {
            const size_t c1 = std::min(uniq.size(), c0 + (size_t)T.slots);
            std::vector<uint64_t> u(uniq.begin() + c0, uniq.begin() + c1);
            ensure(T, u, slot_u);
            part.clear(); slot.clear();
            for (size_t k = 0; k < sel.size(); ++k)
                if ((size_t)ord[k] >= c0 && (size_t)ord[k] < c1) {
                    part.push_back(sel[k]);
                    slot.push_back(slot_u[(size_t)ord[k] - c0]);
                }
            if (!part.empty()) f(part, slot);
        }
    }
    // body(t) does table t's share of one solver call.  Every table is a problem handle with a
    // stream of its own: the tables that have work run on threads of their own, so their kernels
    // share the device and their host sides (pair gathers, block condensation, copies) the cores.
    template <class Body>
    void per_table(const std::vector<std::vector<int64_t>>& sel, Body body) {
        std::vector<size_t> busy;
        for (size_t t = 0; t < sel.size(); ++t)
            if (!sel[t].empty()) busy.push_back(t);
        if (busy.size() <= 1 || !parallel_tables) {
            for (size_t t : busy) body(t);
            return;
        }
        std::vector<Fail> errs(busy.size(), Fail{0, ""});
        std::vector<std::thread> pool;
        for (size_t i = 1; i < busy.size(); ++i)
            pool.emplace_back([&, i] {
                try { body(busy[i]); }
                catch (const Fail& e) { errs[i] = e; }
                catch (const std::exception& e) { errs[i] = Fail{EHM_E_CAPACITY, e.what()}; }
            });
        try { body(busy[0]); }
        catch (const Fail& e) { errs[0] = e; }
        catch (const std::exception& e) { errs[0] = Fail{EHM_E_CAPACITY, e.what()}; }
        for (std::thread& th : pool) th.join();
        for (const Fail& e : errs)
            if (e.code) throw e;
    }
    void split_tables(int64_t n_pairs, const uint64_t* code, std::vector<std::vector<int64_t>>& sel) const {
        sel.assign(tab.size(), std::vector<int64_t>());
        for (int64_t k = 0; k < n_pairs; ++k) sel[(size_t)tab_of_len[(size_t)len_of(code[k])]].push_back(k);
    }
    // sequences.PrefixTable._phase_one_verdict
    void verdicts(const std::vector<double>& tau, const std::vector<int32_t>& st,
                  std::vector<uint8_t>& ok) {
        ok.resize(tau.size());
        int64_t unknown = 0;
        double worst = INF;
        for (size_t k = 0; k < tau.size(); ++k) {
            ok[k] = tau[k] <= FEAS_TOL;
            if (st[k] != 0) {
                ++stalled;
                if (!ok[k]) { ++unknown; worst = std::min(worst, tau[k]); }
            }
        }
        if (unknown)
            raise(EHM_E_NUMERIC, "%lld phase-one problem(s) stalled above the feasibility tolerance "
                  "(smallest tau %.3g): verdict unknown", (long long)unknown, worst);
    }

    void points(int64_t n_pairs, const uint64_t* code, const double* theta, int feas_only,
                int known_feasible, double* J, double* u0) {
        for (int64_t k = 0; k < n_pairs; ++k) J[k] = INF;
        if (u0) std::fill(u0, u0 + (size_t)n_pairs * n_u, 0.0);
        std::vector<std::vector<int64_t>> sel;
        split_tables(n_pairs, code, sel);
        per_table(sel, [&](size_t t) {
            Table& T = tab[t];
            std::vector<double> th, tau, uu, Jk;
            std::vector<int32_t> st, slot2;
            std::vector<uint8_t> ok;
            chunks(T, code, sel[t], [&](const std::vector<int64_t>& part, const std::vector<int32_t>& slot) {
                const size_t cnt = part.size();
                th.resize(cnt * p);
                for (size_t k = 0; k < cnt; ++k)
                    std::memcpy(&th[k * p], theta + (size_t)part[k] * p, 8 * (size_t)p);
                if (known_feasible) {
                    ok.assign(cnt, 1);
                } else {
                    tau.resize(cnt); st.resize(cnt); uu.resize(cnt * n_u);
                    chk_dev(ehm_point_idx_batch(T.P, (int64_t)cnt, th.data(), slot.data(), 1,
                                                tau.data(), uu.data(), st.data()), "ehm_point_idx_batch");
                    ++launches;
                    T.lp += (int64_t)cnt;
                    tally(T, 0, slot);
                    verdicts(tau, st, ok);
                }
                if (feas_only) {
                    for (size_t k = 0; k < cnt; ++k)
                        if (ok[k]) J[part[k]] = 0.0;
                    return;
                }
                std::vector<size_t> good;
                for (size_t k = 0; k < cnt; ++k)
                    if (ok[k]) good.push_back(k);
                if (good.empty()) return;
                std::vector<double> th2(good.size() * p);
                slot2.resize(good.size());
                for (size_t g = 0; g < good.size(); ++g) {
                    std::memcpy(&th2[g * p], &th[good[g] * p], 8 * (size_t)p);
                    slot2[g] = slot[good[g]];
                }
                Jk.resize(good.size()); st.resize(good.size()); uu.resize(good.size() * n_u);
                chk_dev(ehm_point_idx_batch(T.P, (int64_t)good.size(), th2.data(), slot2.data(), 0,
                                            Jk.data(), uu.data(), st.data()), "ehm_point_idx_batch");
                ++launches;
                T.lp += (int64_t)good.size();
                tally(T, 1, slot2);
                for (size_t g = 0; g < good.size(); ++g) {
                    const int64_t k = part[good[g]];
                    if (st[g] != 0) { ++stalled; J[k] = INF; }     // a failed solve, not an optimum
                    else J[k] = Jk[g];
                    if (u0) std::memcpy(u0 + (size_t)k * n_u, &uu[g * n_u], 8 * (size_t)n_u);
                }
            });
        });
    }

    // sequences.PrefixTable.slack_by_prefix_index (+ _settle_stalled)
    void slack(int64_t n_pairs, const uint64_t* code, const double* R, const double* Vbar,
               const uint8_t* known, double* t_out, double* alpha_out) {
        for (int64_t k = 0; k < n_pairs; ++k) t_out[k] = -INF;
        if (alpha_out) std::fill(alpha_out, alpha_out + (size_t)n_pairs * nv, 0.0);
        std::vector<std::vector<int64_t>> sel;
        split_tables(n_pairs, code, sel);
        const size_t sx = (size_t)nv * p;
        per_table(sel, [&](size_t tt) {
            Table& T = tab[tt];
            std::vector<double> Rb, Vb, tau, al, tk;
            std::vector<int32_t> st, sl;
            std::vector<uint8_t> ok, v;
            chunks(T, code, sel[tt], [&](const std::vector<int64_t>& part, const std::vector<int32_t>& slot) {
                const size_t cnt = part.size();
                ok.assign(cnt, 1);
                std::vector<size_t> todo;
                for (size_t k = 0; k < cnt; ++k)
                    if (!(known && known[part[k]])) todo.push_back(k);
                if (!todo.empty()) {
                    Rb.resize(todo.size() * sx); sl.resize(todo.size());
                    for (size_t q = 0; q < todo.size(); ++q) {
                        std::memcpy(&Rb[q * sx], R + (size_t)part[todo[q]] * sx, 8 * sx);
                        sl[q] = slot[todo[q]];
                    }
                    tau.resize(todo.size()); al.resize(todo.size() * nv); st.resize(todo.size());
                    chk_dev(ehm_simplex_idx_batch(T.P, (int64_t)todo.size(), Rb.data(), nullptr, sl.data(),
                                                  2, tau.data(), al.data(), st.data()),
                            "ehm_simplex_idx_batch");
                    ++launches;
                    T.lp += (int64_t)todo.size();
                    tally(T, 2, sl);
                    verdicts(tau, st, v);
                    for (size_t q = 0; q < todo.size(); ++q) ok[todo[q]] = v[q];
                }
                std::vector<size_t> good;
                for (size_t k = 0; k < cnt; ++k)
                    if (ok[k]) good.push_back(k);
                if (good.empty()) return;
                Rb.resize(good.size() * sx); Vb.resize(good.size() * nv); sl.resize(good.size());
                for (size_t g = 0; g < good.size(); ++g) {
                    std::memcpy(&Rb[g * sx], R + (size_t)part[good[g]] * sx, 8 * sx);
                    std::memcpy(&Vb[g * nv], Vbar + (size_t)part[good[g]] * nv, 8 * (size_t)nv);
                    sl[g] = slot[good[g]];
                }
                tk.resize(good.size()); al.resize(good.size() * nv); st.resize(good.size());
                chk_dev(ehm_simplex_idx_batch(T.P, (int64_t)good.size(), Rb.data(), Vb.data(), sl.data(), 1,
                                              tk.data(), al.data(), st.data()), "ehm_simplex_idx_batch");
                ++launches;
                T.lp += (int64_t)good.size();
                tally(T, 4, sl);
                // a slack problem that stalled: phase one decides whether the pair is an
                // interior-free sliver (infeasible) or a feasible relaxation without a value
                std::vector<size_t> bad;
                for (size_t g = 0; g < good.size(); ++g)
                    if (st[g] != 0) bad.push_back(g);
                if (!bad.empty()) {
                    stalled += (int64_t)bad.size();
                    std::vector<double> R2(bad.size() * sx), tau2(bad.size()), al2(bad.size() * nv);
                    std::vector<int32_t> sl2(bad.size()), st2(bad.size());
                    for (size_t q = 0; q < bad.size(); ++q) {
                        std::memcpy(&R2[q * sx], &Rb[bad[q] * sx], 8 * sx);
                        sl2[q] = sl[bad[q]];
                    }
                    chk_dev(ehm_simplex_idx_batch(T.P, (int64_t)bad.size(), R2.data(), nullptr, sl2.data(),
                                                  2, tau2.data(), al2.data(), st2.data()),
                            "ehm_simplex_idx_batch");
                    ++launches;
                    T.lp += (int64_t)bad.size();
                    int64_t full_err = 0;
                    for (size_t q = 0; q < bad.size(); ++q) {
                        const bool sliver = st2[q] == 0 && tau2[q] >= -SLIVER_TOL;
                        const bool feasible = !sliver && (tau2[q] <= FEAS_TOL || st2[q] != 0);
                        slivers += sliver;
                        if (feasible && T.full_length >= 0 && T.slot_len[sl2[q]] == T.full_length) {
                            // a full sequence's slack is an ANSWER.  A solve that stalled at merit
                            // 10^dec holds it to ~10^(dec - 10): where that is decisive -- the value
                            // is a hundred error bars away from zero -- it is taken, as the
                            // reference takes OPTIMAL_INACCURATE (lib/oracle.py:440-442)
                            if (decisive(st[bad[q]], tk[bad[q]])) {
                                ++accepted_inaccurate;
                                continue;
                            }
                            ++full_err;
                        }
                        tk[bad[q]] = feasible ? INF : -INF;
                        stalled_relax += feasible;
                    }
                    if (full_err)
                        raise(EHM_E_NUMERIC, "%lld suboptimality-test problem(s) of full mode "
                              "sequences did not converge on the device", (long long)full_err);
                }
                for (size_t g = 0; g < good.size(); ++g) {
                    t_out[part[good[g]]] = tk[g];
                    if (alpha_out)
                        std::memcpy(alpha_out + (size_t)part[good[g]] * nv, &al[g * nv], 8 * (size_t)nv);
                }
            });
        });
    }

    // sequences.PrefixTable.solve_min for pairs KNOWN to be feasible on their simplex (the node's
    // own commutation): +inf infeasible (an interior-free sliver), -inf a solve that stalled
    void minimum(int64_t n_pairs, const uint64_t* code, const double* R, const uint8_t* known,
                 double* J_out) {
        for (int64_t k = 0; k < n_pairs; ++k) J_out[k] = INF;
        std::vector<std::vector<int64_t>> sel;
        split_tables(n_pairs, code, sel);
        const size_t sx = (size_t)nv * p;
        per_table(sel, [&](size_t tt) {
            Table& T = tab[tt];
            std::vector<double> Rb, tau, al, Jk;
            std::vector<int32_t> st, sl;
            std::vector<uint8_t> ok, v;
            chunks(T, code, sel[tt], [&](const std::vector<int64_t>& part, const std::vector<int32_t>& slot) {
                const size_t cnt = part.size();
                ok.assign(cnt, 1);
                std::vector<size_t> todo;
                for (size_t k = 0; k < cnt; ++k)
                    if (!(known && known[part[k]])) todo.push_back(k);
                if (!todo.empty()) {
                    Rb.resize(todo.size() * sx); sl.resize(todo.size());
                    for (size_t q = 0; q < todo.size(); ++q) {
                        std::memcpy(&Rb[q * sx], R + (size_t)part[todo[q]] * sx, 8 * sx);
                        sl[q] = slot[todo[q]];
                    }
                    tau.resize(todo.size()); al.resize(todo.size() * nv); st.resize(todo.size());
                    chk_dev(ehm_simplex_idx_batch(T.P, (int64_t)todo.size(), Rb.data(), nullptr, sl.data(),
                                                  2, tau.data(), al.data(), st.data()),
                            "ehm_simplex_idx_batch");
                    ++launches;
                    T.lp += (int64_t)todo.size();
                    tally(T, 2, sl);
                    verdicts(tau, st, v);
                    for (size_t q = 0; q < todo.size(); ++q) ok[todo[q]] = v[q];
                }
                std::vector<size_t> good;
                for (size_t k = 0; k < cnt; ++k)
                    if (ok[k]) good.push_back(k);
                if (good.empty()) return;
                Rb.resize(good.size() * sx); sl.resize(good.size());
                for (size_t g = 0; g < good.size(); ++g) {
                    std::memcpy(&Rb[g * sx], R + (size_t)part[good[g]] * sx, 8 * sx);
                    sl[g] = slot[good[g]];
                }
                Jk.resize(good.size()); al.resize(good.size() * nv); st.resize(good.size());
                chk_dev(ehm_simplex_idx_batch(T.P, (int64_t)good.size(), Rb.data(), nullptr, sl.data(), 0,
                                              Jk.data(), al.data(), st.data()), "ehm_simplex_idx_batch");
                ++launches;
                T.lp += (int64_t)good.size();
                tally(T, 3, sl);
                std::vector<size_t> bad;
                for (size_t g = 0; g < good.size(); ++g)
                    if (st[g] != 0) bad.push_back(g);
                if (!bad.empty()) {
                    stalled += (int64_t)bad.size();
                    std::vector<double> R2(bad.size() * sx), tau2(bad.size()), al2(bad.size() * nv);
                    std::vector<int32_t> sl2(bad.size()), st2(bad.size());
                    for (size_t q = 0; q < bad.size(); ++q) {
                        std::memcpy(&R2[q * sx], &Rb[bad[q] * sx], 8 * sx);
                        sl2[q] = sl[bad[q]];
                    }
                    chk_dev(ehm_simplex_idx_batch(T.P, (int64_t)bad.size(), R2.data(), nullptr, sl2.data(),
                                                  2, tau2.data(), al2.data(), st2.data()),
                            "ehm_simplex_idx_batch");
                    ++launches;
                    T.lp += (int64_t)bad.size();
                    for (size_t q = 0; q < bad.size(); ++q) {
                        const bool sliver = st2[q] == 0 && tau2[q] >= -SLIVER_TOL;
                        const bool feasible = !sliver && (tau2[q] <= FEAS_TOL || st2[q] != 0);
                        slivers += sliver;
                        Jk[bad[q]] = feasible ? -INF : INF;     // (a minimum is a pruning bound)
                        stalled_relax += feasible;
                    }
                }
                for (size_t g = 0; g < good.size(); ++g) J_out[part[good[g]]] = Jk[g];
            });
        });
    }
};

int dev_points(void* user, int64_t n, const uint64_t* code, const double* theta, int32_t fo,
               int32_t kf, double* J, double* u0) {
    try {
        static_cast<DeviceSolver*>(user)->points(n, code, theta, fo, kf, J, u0);
    } catch (const Fail& e) {
        return fail(e.code, "%s", e.msg.c_str());
    }
    return EHM_OK;
}
int dev_slack(void* user, int64_t n, const uint64_t* code, const double* R, const double* V,
              const uint8_t* known, double* t, double* alpha) {
    try {
        static_cast<DeviceSolver*>(user)->slack(n, code, R, V, known, t, alpha);
    } catch (const Fail& e) {
        return fail(e.code, "%s", e.msg.c_str());
    }
    return EHM_OK;
}
int dev_min(void* user, int64_t n, const uint64_t* code, const double* R, const uint8_t* known,
            double* J) {
    try {
        static_cast<DeviceSolver*>(user)->minimum(n, code, R, known, J);
    } catch (const Fail& e) {
        return fail(e.code, "%s", e.msg.c_str());
    }
    return EHM_OK;
}
int dev_split(void* user, int64_t n, const double* R, double* S1, double* S2, int32_t* ij) {
    DeviceSolver* D = static_cast<DeviceSolver*>(user);
    int rc = ehm_split_batch(D->device, n, D->p, R, S1, S2, ij);
    if (rc) return fail(rc, "ehm_split_batch: %s", ehm_last_error());
    return EHM_OK;
}

}  // namespace

// ---- the driver -------------------------------------------------------------------------------
// (hidden: the header forward-declares the handle type inside its visibility pragma, which would
// otherwise export every member function)
struct __attribute__((visibility("hidden"))) ehm_frontier {
    int p = 0, nv = 0, n_u = 0, n_modes = 0, N = 0;
    uint64_t base = 1;
    std::vector<uint64_t> pw;
    double eps_a = 1.0, eps_r = 1.0;
    ehm_pair_solvers sol{};
    DeviceSolver* dev = nullptr;
    ehm_search* S = nullptr;
    int64_t feas_n = 0;
    // the tree (flat; node k: vertices, point ids, record)
    std::vector<double> verts, costs, inputs;
    std::vector<int64_t> pids;
    std::vector<int32_t> left, right, depth;
    std::vector<int64_t> seq, witness;          // code of the full sequence held / of the witness
    std::vector<int64_t> incumbent;             // lcss: the parent's best-slack sequence (-1: none)
    std::vector<uint8_t> flags;
    // a failure inside run() / p_theta() leaves cells flagged PENDING in no work list and memo
    // entries claimed by a launch that never answered: the handle then refuses everything but
    // ehm_frontier_reset (and destroy) instead of handing out a silently incomplete tree
    bool poisoned = false;
    int64_t n_roots = 0;
    std::vector<int32_t> ecc_work, lcss_work;
    // vertex optima by (sequence, point id)
    std::unordered_map<uint64_t, int64_t> opt_of;
    std::vector<double> opt_J, opt_u;
    ehm_frontier_stats st{};
    int32_t max_depth = 0;                      // of the run in progress (0 = none)
    // EHM_FR_TALLY=1: slack problems asked by caller, printed at the end of a run (diagnostics)
    int64_t tally_seed = 0, tally_bar_e = 0, tally_bar_d = 0, tally_cells_e = 0, tally_cells_d = 0,
            tally_closed = 0;
    int64_t base_ctr[5] = {0, 0, 0, 0, 0};      // the device solver's counters at the last reset

    int64_t n_nodes() const { return (int64_t)left.size(); }
    uint64_t seq_code(const int32_t* s) const {
        uint64_t c = 0;
        for (int i = 0; i < N; ++i) c += (uint64_t)(s[i] + 1) * pw[i];
        return c;
    }
    int32_t new_node(const double* R, int32_t d) {
        const int32_t k = (int32_t)n_nodes();
        verts.insert(verts.end(), R, R + (size_t)nv * p);
        costs.insert(costs.end(), (size_t)nv, NAN);
        inputs.insert(inputs.end(), (size_t)nv * n_u, NAN);
        pids.insert(pids.end(), (size_t)nv, -1);
        left.push_back(-1); right.push_back(-1); depth.push_back(d);
        seq.push_back(-1); witness.push_back(-1); incumbent.push_back(-1);
        node_bounds.emplace_back();
        flags.push_back(EHM_FR_PENDING);
        return k;
    }
    void solver_points(int64_t n, const uint64_t* code, const double* theta, int fo, int kf,
                       double* J, double* u0) {
        const double t0 = now();
        int rc = sol.points(sol.user, n, code, theta, fo, kf, J, u0);
        st.seconds_solvers += now() - t0;
        if (rc) raise(rc, "%s", dev ? g_err : "the caller's point solver failed");
    }
    void solver_slack(int64_t n, const uint64_t* code, const double* R, const double* V,
                      const uint8_t* known, double* t, double* alpha) {
        const double t0 = now();
        int rc = sol.slack(sol.user, n, code, R, V, known, t, alpha);
        st.seconds_solvers += now() - t0;
        if (rc) raise(rc, "%s", dev ? g_err : "the caller's slack solver failed");
    }
    void solver_min(int64_t n, const uint64_t* code, const double* R, const uint8_t* known,
                    double* J) {
        const double t0 = now();
        int rc = sol.min(sol.user, n, code, R, known, J);
        st.seconds_solvers += now() - t0;
        if (rc) raise(rc, "%s", dev ? g_err : "the caller's minimum solver failed");
    }
    void forget_if_full() {
        if (feas_n > FEAS_MEMO_LIMIT) {
            chk_search(ehm_search_forget(S), "ehm_search_forget");
            feas_n = 0;
        }
    }
    // one phase-one launch over the pairs the search state has pending
    void solve_pending(int64_t n_ask, int64_t n_prefix, std::vector<uint8_t>& ok) {
        std::vector<uint64_t> codes((size_t)n_prefix), pair((size_t)n_ask);
        std::vector<int64_t> idx((size_t)n_ask);
        std::vector<double> th((size_t)n_ask * p), J((size_t)n_ask);
        chk_search(ehm_search_asks(S, codes.data(), idx.data(), th.data()), "ehm_search_asks");
        for (int64_t a = 0; a < n_ask; ++a) pair[(size_t)a] = codes[(size_t)idx[(size_t)a]];
        try {
            solver_points(n_ask, pair.data(), th.data(), 1, 0, J.data(), nullptr);
        } catch (...) {
            ehm_search_abandon(S);
            throw;
        }
        feas_n += n_ask;
        ok.resize((size_t)n_ask);
        for (int64_t a = 0; a < n_ask; ++a) ok[(size_t)a] = std::isfinite(J[(size_t)a]) ? 1 : 0;
    }
    // sequences.PrefixSearch.feasible_sets with the points given by id
    void feasible_sets(const std::vector<uint64_t>& codes, const std::vector<int64_t>& begin,
                       const std::vector<int64_t>& pid, std::vector<uint8_t>& flags_out) {
        const int64_t n = (int64_t)codes.size();
        flags_out.assign((size_t)n, 1);
        if (!n) return;
        forget_if_full();
        int64_t n_ask = 0, n_prefix = 0;
        chk_search(ehm_search_query(S, n, codes.data(), begin.data(), pid.data(), flags_out.data(),
                                    &n_ask, &n_prefix), "ehm_search_query");
        if (n_ask) {
            std::vector<uint8_t> ok;
            solve_pending(n_ask, n_prefix, ok);
            chk_search(ehm_search_answer(S, ok.data(), flags_out.data()), "ehm_search_answer");
        }
    }
    // sequences.PrefixSearch.first_feasible_many: result[j] = code of the sequence or -1
    void first_feasible(const std::vector<int64_t>& begin, const std::vector<int64_t>& pid,
                        std::vector<int64_t>& result) {
        const int64_t n = (int64_t)begin.size() - 1;
        result.assign((size_t)std::max<int64_t>(n, 0), -1);
        if (n <= 0) return;
        forget_if_full();
        chk_search(ehm_search_descent_begin(S, n, begin.data(), pid.data(), nullptr, nullptr),
                   "ehm_search_descent_begin");
        int64_t n_ask = 0, n_prefix = 0;
        std::vector<uint8_t> ok;
        bool have = false;
        for (;;) {
            chk_search(ehm_search_descent_step(S, have ? ok.data() : nullptr, &n_ask, &n_prefix),
                       "ehm_search_descent_step");
            if (!n_ask) break;
            solve_pending(n_ask, n_prefix, ok);
            have = true;
        }
        std::vector<int32_t> sq((size_t)n * N);
        chk_search(ehm_search_descent_result(S, sq.data(), nullptr), "ehm_search_descent_result");
        for (int64_t j = 0; j < n; ++j)
            if (sq[(size_t)j * N] >= 0) result[(size_t)j] = (int64_t)seq_code(&sq[(size_t)j * N]);
    }

    // what one lcss visit has solved on its cell: prefix code -> (t, index of its maximiser or -1)
    struct Known { double t; int64_t alpha; };
    typedef std::unordered_map<uint64_t, Known> Learned;
    std::vector<double> alpha_pool;             // maximisers of the round, nv doubles each
    // upper bounds of the suboptimality-test optimum per prefix, proven on a cell's ancestors: a
    // child lies inside its parent and the optimal cost of a commutation is convex, so the
    // interpolant of its vertex costs lies below the parent's and t*(child) <= t*(parent) for
    // every prefix (+ the largest increase of a vertex cost where a commutation with larger costs
    // is adopted) -- bounds that refute, or stay below what a search still accepts, without a solve
    // (bnb_frontier.grow_frontier)
    // (flat and sorted by prefix code: handing bounds down is a merge and two memcpy, a look-up a
    // binary search -- a hash map per open cell cost the driver more than the solves it saved)
    struct BoundMap {
        std::vector<uint64_t> code;
        std::vector<double> t;
        size_t size() const { return code.size(); }
        bool empty() const { return code.empty(); }
        const double* find(uint64_t c) const {
            auto it = std::lower_bound(code.begin(), code.end(), c);
            return (it != code.end() && *it == c) ? &t[(size_t)(it - code.begin())] : nullptr;
        }
    };
    std::vector<std::shared_ptr<BoundMap>> node_bounds;     // per node, null = nothing inherited
    struct Split { int32_t node; int64_t code; std::vector<double> c, u; int64_t star;
                   std::shared_ptr<BoundMap> down; };

    void feasible_somewhere(const std::vector<uint64_t>& codes, const std::vector<int32_t>& node_of,
                            std::vector<uint8_t>& known);
    void slack_pairs(const std::vector<uint64_t>& codes, const std::vector<int32_t>& node_of,
                     std::vector<double>& t, std::vector<double>* alpha);
    void optima(const std::vector<uint64_t>& code, const std::vector<int64_t>& pid,
                const std::vector<double>& pts, std::vector<double>& J, std::vector<double>& u);
    void bisect(std::vector<Split>& todo);
    void bar_d(const std::vector<int32_t>& O, std::vector<Learned>& learned, int launch_target,
               std::vector<Split>& to_split);
    void p_theta(int64_t n, const double* theta, double* J, double* u0, int32_t* sequence);
    void ecc_round(const std::vector<int32_t>& E, std::vector<Split>& to_split);
    void lcss_round(const std::vector<int32_t>& Lc, int launch_target, std::vector<Split>& to_split);
    void run(const ehm_frontier_opts& o);
};

// sequences.PrefixSearch.feasible_somewhere_codes: a relaxation known to be feasible at a vertex of
// the cell is feasible on the cell; where nothing is held the first vertex without a verdict is
// asked (one point problem, remembered: the neighbours that ask about the same prefix get it free)
void ehm_frontier::feasible_somewhere(const std::vector<uint64_t>& codes,
                                      const std::vector<int32_t>& node_of,
                                      std::vector<uint8_t>& known) {
    const int64_t n = (int64_t)codes.size();
    known.assign((size_t)n, 0);
    if (!n) return;
    std::vector<int64_t> pp((size_t)n * nv), pb((size_t)n + 1);
    std::vector<int32_t> first((size_t)n);
    for (int64_t a = 0; a < n; ++a) {
        pb[(size_t)a] = a * nv;
        std::memcpy(&pp[(size_t)a * nv], &pids[(size_t)node_of[(size_t)a] * nv], 8 * (size_t)nv);
    }
    pb[(size_t)n] = n * nv;
    chk_search(ehm_search_peek_any(S, n, codes.data(), pb.data(), pp.data(), known.data(), first.data()),
               "ehm_search_peek_any");
    std::vector<uint64_t> qc;
    std::vector<int64_t> qb(1, 0), qp, qa;
    for (int64_t a = 0; a < n; ++a)
        if (!known[(size_t)a] && first[(size_t)a] >= 0) {
            qc.push_back(codes[(size_t)a]);
            qp.push_back(pp[(size_t)a * nv + first[(size_t)a]]);
            qb.push_back((int64_t)qp.size());
            qa.push_back(a);
        }
    if (!qc.empty()) {
        std::vector<uint8_t> fl;
        feasible_sets(qc, qb, qp, fl);
        for (size_t q = 0; q < qa.size(); ++q) known[(size_t)qa[q]] = fl[q];
    }
}

// suboptimality-test optima of (prefix, cell) pairs: t (and the maximisers, nv doubles per pair)
void ehm_frontier::slack_pairs(const std::vector<uint64_t>& codes, const std::vector<int32_t>& node_of,
                               std::vector<double>& t, std::vector<double>* alpha) {
    const size_t n = codes.size(), sx = (size_t)nv * p;
    t.assign(n, -INF);
    if (alpha) alpha->assign(n * nv, 0.0);
    if (!n) return;
    std::vector<uint8_t> known;
    feasible_somewhere(codes, node_of, known);
    std::vector<double> Rp(n * sx), Vp(n * nv);
    for (size_t a = 0; a < n; ++a) {
        std::memcpy(&Rp[a * sx], &verts[(size_t)node_of[a] * sx], 8 * sx);
        std::memcpy(&Vp[a * nv], &costs[(size_t)node_of[a] * nv], 8 * (size_t)nv);
    }
    solver_slack((int64_t)n, codes.data(), Rp.data(), Vp.data(), known.data(), t.data(),
                 alpha ? alpha->data() : nullptr);
}

// sequences.PrefixSearch.optima_at: optimal cost and first input of full sequence code[k] at point
// k (id pid[k]) for pairs the caller KNOWS to be feasible; computed once per (sequence, point)
void ehm_frontier::optima(const std::vector<uint64_t>& code, const std::vector<int64_t>& pid,
                          const std::vector<double>& pts, std::vector<double>& J,
                          std::vector<double>& u) {
    const size_t n = code.size();
    J.resize(n); u.resize(n * n_u);
    if (!n) return;
    if (opt_of.size() > OPTIMA_MEMO_LIMIT) { opt_of.clear(); opt_J.clear(); opt_u.clear(); }
    std::vector<uint64_t> mcode, mkey;
    std::vector<double> mth;
    for (size_t k = 0; k < n; ++k) {
        const uint64_t key = (code[k] << PID_BITS) | (uint64_t)pid[k];
        if (opt_of.count(key)) continue;
        opt_of[key] = -1 - (int64_t)mkey.size();            // claimed by this launch
        mkey.push_back(key);
        mcode.push_back(code[k]);
        mth.insert(mth.end(), &pts[k * p], &pts[k * p] + p);
    }
    st.optima_asked += (int64_t)n;
    st.optima_solved += (int64_t)mkey.size();
    if (!mkey.empty()) {
        std::vector<double> Jm(mkey.size()), um(mkey.size() * n_u);
        solver_points((int64_t)mkey.size(), mcode.data(), mth.data(), 0, 1, Jm.data(), um.data());
        for (size_t a = 0; a < mkey.size(); ++a) {
            opt_of[mkey[a]] = (int64_t)opt_J.size();
            opt_J.push_back(Jm[a]);
            opt_u.insert(opt_u.end(), &um[a * n_u], &um[a * n_u] + n_u);
        }
    }
    for (size_t k = 0; k < n; ++k) {
        const int64_t at = opt_of[(code[k] << PID_BITS) | (uint64_t)pid[k]];
        J[k] = opt_J[(size_t)at];
        std::memcpy(&u[k * n_u], &opt_u[(size_t)at * n_u], 8 * (size_t)n_u);
    }
}

// Longest-edge bisection of the cells in `todo` (lib/worker.py:277-291, 403-417): children of a cell
// without a commutation look for one (ecc); children of a cell that holds one inherit its vertex
// costs and inputs with the new vertex's slot replaced by the optimum at the midpoint
// (lib/worker.py:356-365) and go on with lcss.
void ehm_frontier::bisect(std::vector<Split>& todo_all) {
    // a cell at the depth limit is not bisected: it stays an open leaf (EHM_FR_DEPTH)
    std::vector<Split> todo;
    for (Split& sp : todo_all) {
        if (max_depth > 0 && depth[sp.node] >= max_depth) {
            flags[sp.node] = (uint8_t)((flags[sp.node] & EHM_FR_HAS_RECORD) | EHM_FR_DEPTH);
            node_bounds[(size_t)sp.node].reset();
            ++st.depth_limited;
            continue;
        }
        todo.push_back(std::move(sp));
    }
    const size_t n = todo.size(), sx = (size_t)nv * p;
    if (!n) return;
    std::vector<double> Rs(n * sx), S1(n * sx), S2(n * sx);
    std::vector<int32_t> ij(n * 2);
    for (size_t i = 0; i < n; ++i) std::memcpy(&Rs[i * sx], &verts[(size_t)todo[i].node * sx], 8 * sx);
    {
        const double t0 = now();
        int rc = sol.split(sol.user, (int64_t)n, Rs.data(), S1.data(), S2.data(), ij.data());
        st.seconds_solvers += now() - t0;
        if (rc) raise(rc, "%s", dev ? g_err : "the caller's bisection failed");
    }
    std::vector<double> mids(n * p);
    for (size_t i = 0; i < n; ++i) std::memcpy(&mids[i * p], &S1[i * sx + (size_t)ij[2 * i] * p], 8 * (size_t)p);
    std::vector<int64_t> mid_id(n), a_id(n), b_id(n);
    chk_search(ehm_search_point_ids(S, (int64_t)n, mids.data(), mid_id.data()), "ehm_search_point_ids");
    for (size_t i = 0; i < n; ++i) {
        a_id[i] = pids[(size_t)todo[i].node * nv + ij[2 * i]];
        b_id[i] = pids[(size_t)todo[i].node * nv + ij[2 * i + 1]];
    }
    chk_search(ehm_search_register_midpoints(S, (int64_t)n, mid_id.data(), a_id.data(), b_id.data()),
               "ehm_search_register_midpoints");
    // the adopted commutation is feasible at both ends of the edge, so at its midpoint: no phase
    // one; the cells around an edge that hold the same commutation share the optimum
    std::vector<size_t> with;
    std::vector<uint64_t> mc;
    std::vector<int64_t> mp;
    std::vector<double> mpt, Jm, um;
    for (size_t i = 0; i < n; ++i)
        if (todo[i].code >= 0) {
            with.push_back(i);
            mc.push_back((uint64_t)todo[i].code);
            mp.push_back(mid_id[i]);
            mpt.insert(mpt.end(), &mids[i * p], &mids[i * p] + p);
        }
    optima(mc, mp, mpt, Jm, um);
    for (size_t w = 0; w < with.size(); ++w)
        if (!std::isfinite(Jm[w]))
            raise(EHM_E_NUMERIC, "midpoint solve of the adopted commutation failed");
    size_t w = 0;
    for (size_t i = 0; i < n; ++i) {
        const int32_t nd = todo[i].node;
        const int32_t l = new_node(&S1[i * sx], depth[nd] + 1);
        const int32_t r = new_node(&S2[i * sx], depth[nd] + 1);
        left[nd] = l; right[nd] = r;
        flags[nd] &= (uint8_t)EHM_FR_HAS_RECORD;
        node_bounds[(size_t)nd].reset();
        for (int v = 0; v < nv; ++v) pids[(size_t)l * nv + v] = pids[(size_t)r * nv + v] = pids[(size_t)nd * nv + v];
        const int vi = ij[2 * i], vj = ij[2 * i + 1];
        pids[(size_t)l * nv + vi] = mid_id[i];
        pids[(size_t)r * nv + vj] = mid_id[i];
        st.depth = std::max(st.depth, depth[nd] + 1);
        if (todo[i].code < 0) {
            witness[l] = witness[r] = witness[nd];
            ecc_work.push_back(l);
            ecc_work.push_back(r);
            continue;
        }
        for (int32_t ch : {l, r}) {
            std::memcpy(&costs[(size_t)ch * nv], todo[i].c.data(), 8 * (size_t)nv);
            std::memcpy(&inputs[(size_t)ch * nv * n_u], todo[i].u.data(), 8 * (size_t)nv * n_u);
            seq[ch] = todo[i].code;
            incumbent[ch] = todo[i].star;
            node_bounds[(size_t)ch] = todo[i].down;
            flags[ch] = EHM_FR_HAS_RECORD | EHM_FR_PENDING;
            lcss_work.push_back(ch);
        }
        costs[(size_t)l * nv + vi] = Jm[w];
        costs[(size_t)r * nv + vj] = Jm[w];
        std::memcpy(&inputs[((size_t)l * nv + vi) * n_u], &um[w * n_u], 8 * (size_t)n_u);
        std::memcpy(&inputs[((size_t)r * nv + vj) * n_u], &um[w * n_u], 8 * (size_t)n_u);
        ++w;
    }
}

void ehm_frontier::ecc_round(const std::vector<int32_t>& E, std::vector<Split>& to_split) {
    const size_t ne = E.size();
    if (!ne) return;
    st.calls_v_r += (int64_t)ne;
    st.ecc_visits += (int64_t)ne;
    std::vector<int64_t> begin(ne + 1, 0), pid(ne * nv), found;
    for (size_t k = 0; k < ne; ++k) {
        begin[k + 1] = (int64_t)(k + 1) * nv;
        std::memcpy(&pid[k * nv], &pids[(size_t)E[k] * nv], 8 * (size_t)nv);
    }
    first_feasible(begin, pid, found);
    // lib/worker.py:264-266 checks the barycentre first; a sequence feasible at every vertex is
    // feasible there too, so only the cells V_R finds nothing for need the check: the sequence that
    // was feasible at the parent's barycentre first (one problem), the descent where that fails
    std::vector<size_t> none;
    for (size_t k = 0; k < ne; ++k)
        if (found[k] < 0) none.push_back(k);
    st.calls_p_theta += (int64_t)none.size();
    if (!none.empty()) {
        std::vector<double> centre(none.size() * p);
        for (size_t i = 0; i < none.size(); ++i) {
            const double* R = &verts[(size_t)E[none[i]] * nv * p];
            for (int c = 0; c < p; ++c) {
                double sum = R[c];
                for (int v = 1; v < nv; ++v) sum += R[(size_t)v * p + c];
                centre[i * p + c] = sum / (double)nv;
            }
        }
        std::vector<int64_t> cid(none.size());
        chk_search(ehm_search_point_ids(S, (int64_t)none.size(), centre.data(), cid.data()),
                   "ehm_search_point_ids");
        std::vector<uint8_t> held(none.size(), 0);
        std::vector<size_t> tried;
        for (size_t i = 0; i < none.size(); ++i)
            if (witness[E[none[i]]] >= 0) tried.push_back(i);
        if (!tried.empty()) {
            std::vector<uint64_t> wc(tried.size());
            std::vector<int64_t> wb(tried.size() + 1, 0), wp(tried.size());
            for (size_t q = 0; q < tried.size(); ++q) {
                wc[q] = (uint64_t)witness[E[none[tried[q]]]];
                wb[q + 1] = (int64_t)q + 1;
                wp[q] = cid[tried[q]];
            }
            std::vector<uint8_t> fl;
            feasible_sets(wc, wb, wp, fl);
            for (size_t q = 0; q < tried.size(); ++q) {
                held[tried[q]] = fl[q];
                st.witness_hits += fl[q];
            }
        }
        std::vector<size_t> miss;
        for (size_t i = 0; i < none.size(); ++i)
            if (!held[i]) miss.push_back(i);
        if (!miss.empty()) {
            std::vector<int64_t> mb(miss.size() + 1, 0), mp(miss.size()), fresh;
            for (size_t q = 0; q < miss.size(); ++q) { mb[q + 1] = (int64_t)q + 1; mp[q] = cid[miss[q]]; }
            first_feasible(mb, mp, fresh);
            for (size_t q = 0; q < miss.size(); ++q) {
                if (fresh[q] < 0) raise(EHM_E_INFEASIBLE, "STOP, Theta contains infeasible regions");
                witness[E[none[miss[q]]]] = fresh[q];
            }
        }
    }
    // the cells that hold a commutation now: its optimum at every vertex (feasible there: no phase
    // one; an optimum a neighbouring cell has computed already is not computed again)
    std::vector<size_t> have;
    for (size_t k = 0; k < ne; ++k)
        if (found[k] >= 0) have.push_back(k);
    if (!have.empty()) {
        std::vector<uint64_t> oc;
        std::vector<int64_t> op;
        std::vector<double> opt_pts, J, u;
        for (size_t h = 0; h < have.size(); ++h) {
            const int32_t nd = E[have[h]];
            for (int v = 0; v < nv; ++v) {
                oc.push_back((uint64_t)found[have[h]]);
                op.push_back(pids[(size_t)nd * nv + v]);
                opt_pts.insert(opt_pts.end(), &verts[((size_t)nd * nv + v) * p],
                               &verts[((size_t)nd * nv + v) * p] + p);
            }
        }
        optima(oc, op, opt_pts, J, u);
        for (size_t h = 0; h < have.size(); ++h) {
            const int32_t nd = E[have[h]];
            bool finite = true;
            for (int v = 0; v < nv; ++v) {
                costs[(size_t)nd * nv + v] = J[h * nv + v];
                std::memcpy(&inputs[((size_t)nd * nv + v) * n_u], &u[(h * nv + v) * n_u], 8 * (size_t)n_u);
                finite = finite && std::isfinite(J[h * nv + v]);
            }
            if (!finite) {
                // lib/oracle.py:214-218: a failed vertex solve blacklists the commutation and V_R is
                // asked again -- the caller's one-cell oracle does that
                flags[nd] = EHM_FR_OPEN | EHM_FR_NEEDS_ECC;
                ++st.open_cells;
                continue;
            }
            seq[nd] = found[have[h]];
            flags[nd] = EHM_FR_HAS_RECORD | EHM_FR_PENDING;
            lcss_work.push_back(nd);
        }
    }
    for (size_t i : none) to_split.push_back(Split{E[i], -1, {}, {}, -1, nullptr});
}

namespace {
// heapq on (-t, prefix tuple): larger t first; ties by the tuples' lexicographic order (digit by
// digit from step 0; a prefix sorts before its extensions) -- ehm_search.cpp, BareItem
struct HeapItem { double t; uint64_t code; int32_t len; };
struct PrefixHeap {
    uint64_t base = 1;
    std::vector<HeapItem> h;
    bool before(const HeapItem& a, const HeapItem& b) const {
        if (a.t != b.t) return a.t > b.t;
        uint64_t ca = a.code, cb = b.code;
        const int32_t m = a.len < b.len ? a.len : b.len;
        for (int32_t i = 0; i < m; ++i) {
            const uint64_t da = ca % base, db = cb % base;
            if (da != db) return da < db;
            ca /= base; cb /= base;
        }
        return a.len < b.len;
    }
    void push(const HeapItem& it) {
        h.push_back(it);
        size_t c = h.size() - 1;
        while (c > 0) {
            const size_t q = (c - 1) / 2;
            if (!before(h[c], h[q])) break;
            std::swap(h[c], h[q]);
            c = q;
        }
    }
    HeapItem pop() {
        HeapItem top = h.front();
        h.front() = h.back();
        h.pop_back();
        size_t q = 0;
        const size_t n_ = h.size();
        for (;;) {
            size_t l = 2 * q + 1, r = l + 1, b = q;
            if (l < n_ && before(h[l], h[b])) b = l;
            if (r < n_ && before(h[r], h[b])) b = r;
            if (b == q) break;
            std::swap(h[q], h[b]);
            q = b;
        }
        return top;
    }
};
constexpr double TIE_TOL = 1e-6;               // sequences.TIE_TOL
constexpr double PLATEAU = 1e-7;               // bnb.PLATEAU
inline double rel(double x) { return 1.0 + std::fabs(x); }
}  // namespace

// bnb_frontier.p_theta_many (lib/oracle.py:104-139 as a search): P_theta at n parameters in
// lockstep.  Phase one: best-first on the relaxations' optimal costs (a prefix's cost is a lower
// bound of each completion's) for the optimal VALUE; phase two: the first sequence, in enumeration
// order, within the tie tolerance of it -- a walk over prefixes whose children phase one has
// mostly solved already (their values are kept per parameter: without ties the walk needs no
// problem at all).
void ehm_frontier::p_theta(int64_t n, const double* theta, double* J_out, double* u_out,
                           int32_t* seq_out) {
    struct Search {
        int phase = 1;
        PrefixHeap heap;                        // keyed by -J: the cheapest relaxation first
        double best = INF, limit = INF;
        std::vector<HeapItem> stack;
        std::unordered_map<uint64_t, std::pair<double, int64_t>> seen;   // code -> (J, index of u0)
        bool done = false;
    };
    std::vector<Search> sr((size_t)n);
    std::vector<double> upool;                  // first inputs of every solved (prefix, parameter)
    for (int64_t j = 0; j < n; ++j) {
        sr[(size_t)j].heap.base = base;
        sr[(size_t)j].heap.push(HeapItem{-0.0, 0, 0});
        J_out[j] = INF;
        if (u_out) std::fill(u_out + (size_t)j * n_u, u_out + (size_t)(j + 1) * n_u, NAN);
        if (seq_out) std::fill(seq_out + (size_t)j * N, seq_out + (size_t)(j + 1) * N, -1);
    }
    auto cut_of = [](const Search& q) {
        return std::isfinite(q.best) ? q.best - PLATEAU * rel(q.best) : INF;
    };
    std::vector<int64_t> active((size_t)n);
    for (int64_t j = 0; j < n; ++j) active[(size_t)j] = j;
    std::vector<uint64_t> ask_code;
    std::vector<double> ask_th, Ja, ua;
    std::vector<std::pair<int64_t, uint64_t>> ask_at;
    while (!active.empty()) {
        ask_code.clear(); ask_th.clear(); ask_at.clear();
        std::vector<int64_t> still;
        std::vector<std::vector<HeapItem>> kids_of(active.size());
        for (size_t a = 0; a < active.size(); ++a) {
            const int64_t j = active[a];
            Search& q = sr[(size_t)j];
            std::vector<HeapItem>& kids = kids_of[a];
            if (q.phase == 1) {
                const double cut = cut_of(q);
                int taken = 0;
                while (!q.heap.h.empty() && taken < BATCH && -q.heap.h.front().t < cut) {
                    const HeapItem b = q.heap.pop();
                    ++taken;
                    ++st.prefixes_expanded;
                    for (int i = 0; i < n_modes; ++i)
                        kids.push_back(HeapItem{0.0, b.code + (uint64_t)(i + 1) * pw[(size_t)b.len], b.len + 1});
                }
            } else {
                // walk as far as the known values carry it
                for (;;) {
                    if (q.stack.empty())
                        raise(EHM_E_NUMERIC, "P_theta: the optimum found in phase one was not reproduced");
                    const HeapItem top = q.stack.back();
                    std::vector<HeapItem> ch;
                    bool missing = false;
                    for (int i = 0; i < n_modes; ++i) {
                        const uint64_t c = top.code + (uint64_t)(i + 1) * pw[(size_t)top.len];
                        ch.push_back(HeapItem{0.0, c, top.len + 1});
                        if (!q.seen.count(c)) { missing = true; kids.push_back(ch.back()); }
                    }
                    if (missing) break;
                    q.stack.pop_back();
                    ++st.prefixes_expanded;
                    std::vector<HeapItem> good;
                    for (const HeapItem& c : ch)
                        if (q.seen[c.code].first <= q.limit) good.push_back(c);
                    if (!good.empty() && good[0].len == N) {
                        const auto& hit = q.seen[good[0].code];
                        J_out[j] = hit.first;
                        if (u_out) std::memcpy(u_out + (size_t)j * n_u, &upool[(size_t)hit.second * n_u], 8 * (size_t)n_u);
                        if (seq_out) {
                            uint64_t c = good[0].code;
                            for (int i = 0; i < N; ++i) { seq_out[(size_t)j * N + i] = (int32_t)(c % base) - 1; c /= base; }
                        }
                        q.done = true;
                        kids.clear();
                        break;
                    }
                    for (size_t g = good.size(); g-- > 0;) q.stack.push_back(good[g]);
                }
                if (q.done) continue;
            }
            still.push_back(j);
            for (const HeapItem& k : kids) {
                ask_at.emplace_back(j, k.code);
                ask_code.push_back(k.code);
                ask_th.insert(ask_th.end(), theta + (size_t)j * p, theta + (size_t)(j + 1) * p);
            }
        }
        // (kids_of is indexed like `active`; keep the pairing while the answers come in)
        std::vector<int64_t> act_before = active;
        active.swap(still);
        if (active.empty()) break;
        Ja.resize(ask_code.size()); ua.resize(ask_code.size() * n_u);
        if (!ask_code.empty())
            solver_points((int64_t)ask_code.size(), ask_code.data(), ask_th.data(), 0, 0, Ja.data(), ua.data());
        for (size_t a = 0; a < ask_code.size(); ++a) {
            Search& q = sr[(size_t)ask_at[a].first];
            const int64_t ui = (int64_t)(upool.size() / n_u);
            upool.insert(upool.end(), &ua[a * n_u], &ua[a * n_u] + n_u);
            q.seen[ask_at[a].second] = std::make_pair(Ja[a], ui);
        }
        std::vector<int64_t> next;
        for (size_t a = 0; a < act_before.size(); ++a) {
            const int64_t j = act_before[a];
            Search& q = sr[(size_t)j];
            if (q.done) continue;
            if (q.phase == 1) {
                for (const HeapItem& k : kids_of[a]) {
                    const double jq = q.seen[k.code].first;
                    if (!std::isfinite(jq)) continue;
                    if (k.len == N) q.best = std::min(q.best, jq);
                    else q.heap.push(HeapItem{-jq, k.code, k.len});
                }
                if (!(!q.heap.h.empty() && -q.heap.h.front().t < cut_of(q))) {
                    if (!std::isfinite(q.best)) { q.done = true; continue; }     // infeasible parameter
                    q.phase = 2;
                    q.limit = q.best + TIE_TOL * rel(q.best);
                    q.stack.assign(1, HeapItem{0.0, 0, 0});
                }
            }
            next.push_back(j);
        }
        active.swap(next);
    }
}

// bnb_frontier.bar_d_many (lib/oracle.py:311-414 as a search): for every open cell the commutation
// with the LARGEST slack among those feasible at every vertex with t* >= 0 (ties: first in
// enumeration order) -- best-first for the value, a lexicographic walk for the sequence --, then
// the winners' vertex optima and variability test (lib/oracle.py:220-283), and the cell's fate
// (lib/worker.py:377-417): adopt in place and look again, or bisect.
void ehm_frontier::bar_d(const std::vector<int32_t>& O, std::vector<Learned>& learned,
                         int launch_target, std::vector<Split>& to_split) {
    const size_t n = O.size();
    if (!n) return;
    st.calls_bar_d += (int64_t)n;
    tally_cells_d += (int64_t)n;
    struct Search {
        int phase = 1;
        PrefixHeap heap;
        double best = -INF, limit = 0.0;
        std::vector<HeapItem> stack;
        int star = 0;                           // 0 searching, 1 found, -1 there is none
        uint64_t star_code = 0;
        int64_t star_alpha = -1;
        // the step in flight
        std::vector<uint64_t> kid;
        std::vector<int32_t> klen;
        std::vector<double> kt;
        std::vector<int64_t> kalpha;
        std::vector<uint8_t> kdead;
    };
    std::vector<Search> sr(n);
    std::vector<double> guard(n);
    for (size_t j = 0; j < n; ++j) {
        double mx = 0.0;
        for (int v = 0; v < nv; ++v) mx = std::max(mx, std::fabs(costs[(size_t)O[j] * nv + v]));
        guard[j] = INHERIT_GUARD * (1.0 + mx);
    }
    auto floor_of = [](const Search& q) {
        return std::isfinite(q.best) ? std::max(0.0, q.best + PLATEAU * rel(q.best)) : 0.0;
    };
    for (size_t j = 0; j < n; ++j) {
        sr[j].heap.base = base;
        sr[j].heap.push(HeapItem{INF, 0, 0});
    }
    // warm start: the parent's best-slack sequence, evaluated on the cell by bar_E's seeding
    {
        std::vector<uint64_t> wc;
        std::vector<int64_t> wb(1, 0), wp;
        std::vector<size_t> wj;
        for (size_t j = 0; j < n; ++j) {
            const int64_t inc = incumbent[O[j]];
            if (inc < 0) continue;
            auto it = learned[j].find((uint64_t)inc);
            if (it == learned[j].end() || !(it->second.t >= 0.0)) continue;
            wc.push_back((uint64_t)inc);
            for (int v = 0; v < nv; ++v) wp.push_back(pids[(size_t)O[j] * nv + v]);
            wb.push_back((int64_t)wp.size());
            wj.push_back(j);
        }
        if (!wc.empty()) {
            std::vector<uint8_t> fl;
            feasible_sets(wc, wb, wp, fl);
            for (size_t q = 0; q < wj.size(); ++q)
                if (fl[q]) sr[wj[q]].best = learned[wj[q]][wc[q]].t;
        }
    }
    std::vector<size_t> active(n);
    for (size_t j = 0; j < n; ++j) active[j] = j;
    while (!active.empty()) {
        const int width = (int)std::max<int64_t>(1, std::min<int64_t>(
            BATCH, launch_target / std::max<int64_t>(1, (int64_t)active.size() * n_modes)));
        std::vector<uint64_t> ask_code;
        std::vector<int32_t> ask_node;
        std::vector<std::pair<size_t, size_t>> ask_at;          // (search, kid index)
        for (size_t j : active) {
            Search& q = sr[j];
            q.kid.clear(); q.klen.clear(); q.kt.clear(); q.kalpha.clear(); q.kdead.clear();
            std::vector<HeapItem> batch;
            double need = q.limit;
            if (q.phase == 1) {
                const double fl = floor_of(q);
                need = fl;
                while (!q.heap.h.empty() && (int)batch.size() < width && q.heap.h.front().t >= fl)
                    batch.push_back(q.heap.pop());
            } else {
                batch.push_back(q.stack.back());
                q.stack.pop_back();
            }
            st.prefixes_expanded += (int64_t)batch.size();
            for (const HeapItem& b : batch)
                for (int i = 0; i < n_modes; ++i) {
                    const uint64_t c = b.code + (uint64_t)(i + 1) * pw[(size_t)b.len];
                    const int32_t len = b.len + 1;
                    double t = 0.0;
                    int64_t al = -1;
                    bool have = false;
                    auto it = learned[j].find(c);
                    if (it != learned[j].end() &&
                        (it->second.t < 0.0 || len < N || it->second.alpha >= 0)) {
                        t = it->second.t; al = it->second.alpha; have = true;
                        ++st.answered_without_a_problem;
                    } else if (const BoundMap* bm = node_bounds[(size_t)O[j]].get()) {
                        // an inherited upper bound that refutes, or that lies below what this
                        // phase still accepts: never solved
                        if (const double* bt = bm->find(c)) {
                            if (*bt < -guard[j]) { t = *bt; have = true; }
                            else if (*bt < need - guard[j]) { t = -INF; have = true; }
                            if (have) ++st.answered_without_a_problem;
                        }
                    }
                    if (!have) {
                        ask_at.emplace_back(j, q.kid.size());
                        ask_code.push_back(c);
                        ask_node.push_back(O[j]);
                    }
                    q.kid.push_back(c); q.klen.push_back(len); q.kt.push_back(t);
                    q.kalpha.push_back(al); q.kdead.push_back(0);
                }
        }
        if (!ask_code.empty()) {
            std::vector<double> t, al;
            tally_bar_d += (int64_t)ask_code.size();
            slack_pairs(ask_code, ask_node, t, &al);
            for (size_t a = 0; a < ask_code.size(); ++a) {
                Search& q = sr[ask_at[a].first];
                const size_t k = ask_at[a].second;
                int64_t at = -1;
                if (q.klen[k] == N) {
                    at = (int64_t)(alpha_pool.size() / nv);
                    alpha_pool.insert(alpha_pool.end(), &al[a * nv], &al[a * nv] + nv);
                }
                q.kt[k] = t[a];
                q.kalpha[k] = at;
                learned[ask_at[a].first][ask_code[a]] = Known{t[a], at};
            }
        }
        // feasibility at every vertex, for the prefixes whose slack bound is not negative
        {
            std::vector<uint64_t> lc;
            std::vector<int64_t> lb(1, 0), lp;
            std::vector<std::pair<size_t, size_t>> lat;
            for (size_t j : active) {
                Search& q = sr[j];
                for (size_t k = 0; k < q.kid.size(); ++k)
                    if (q.kt[k] >= 0.0) {
                        lc.push_back(q.kid[k]);
                        for (int v = 0; v < nv; ++v) lp.push_back(pids[(size_t)O[j] * nv + v]);
                        lb.push_back((int64_t)lp.size());
                        lat.emplace_back(j, k);
                    }
            }
            if (!lc.empty()) {
                std::vector<uint8_t> fl;
                feasible_sets(lc, lb, lp, fl);
                for (size_t a = 0; a < lat.size(); ++a)
                    if (!fl[a]) sr[lat[a].first].kdead[lat[a].second] = 1;
            }
        }
        std::vector<size_t> still;
        for (size_t j : active) {
            Search& q = sr[j];
            const size_t nk = q.kid.size();
            if (q.phase == 1) {
                for (size_t k = 0; k < nk; ++k) {
                    const double tq = q.kdead[k] ? -INF : q.kt[k];
                    if (!(tq >= 0.0)) continue;
                    if (q.klen[k] == N) q.best = std::max(q.best, tq);
                    else q.heap.push(HeapItem{tq, q.kid[k], q.klen[k]});
                }
                if (!(!q.heap.h.empty() && q.heap.h.front().t >= floor_of(q))) {
                    if (!std::isfinite(q.best)) {
                        q.star = -1;
                        continue;
                    }
                    q.phase = 2;
                    q.limit = std::max(0.0, q.best - TIE_TOL * rel(q.best));
                    q.stack.assign(1, HeapItem{0.0, 0, 0});
                }
                still.push_back(j);
            } else {
                std::vector<size_t> good;
                for (size_t k = 0; k < nk; ++k) {
                    const double tk = q.kdead[k] ? -INF : q.kt[k];
                    if (tk >= q.limit) good.push_back(k);
                }
                if (!good.empty() && q.klen[good[0]] == N) {
                    q.star = 1;
                    q.star_code = q.kid[good[0]];
                    q.star_alpha = q.kalpha[good[0]];
                    continue;
                }
                for (size_t g = good.size(); g-- > 0;)
                    q.stack.push_back(HeapItem{0.0, q.kid[good[g]], q.klen[good[g]]});
                if (q.stack.empty())
                    raise(EHM_E_NUMERIC, "bar_D: the slack found in phase one was not reproduced");
                still.push_back(j);
            }
        }
        active.swap(still);
    }
    // the winners' vertex solves and variability checks, one call each
    std::vector<size_t> win;
    for (size_t j = 0; j < n; ++j)
        if (sr[j].star == 1 && (int64_t)sr[j].star_code != seq[O[j]]) win.push_back(j);
    std::vector<double> Jv, uv, thetas, Jmin, Jth;
    if (!win.empty()) {
        const size_t nw = win.size(), sx = (size_t)nv * p;
        std::vector<uint64_t> vc(nw * nv), rc(nw), sc(nw);
        std::vector<double> vpts(nw * sx), Rw(nw * sx), uth(nw * n_u);
        std::vector<uint8_t> ones(nw, 1);
        thetas.assign(nw * p, 0.0);
        for (size_t w = 0; w < nw; ++w) {
            const int32_t nd = O[win[w]];
            for (int v = 0; v < nv; ++v) vc[w * nv + v] = sr[win[w]].star_code;
            std::memcpy(&vpts[w * sx], &verts[(size_t)nd * sx], 8 * sx);
            std::memcpy(&Rw[w * sx], &verts[(size_t)nd * sx], 8 * sx);
            rc[w] = (uint64_t)seq[nd];
            sc[w] = sr[win[w]].star_code;
            if (sr[win[w]].star_alpha < 0) raise(EHM_E_NUMERIC, "bar_D: a winner without its maximiser");
            const double* al = &alpha_pool[(size_t)sr[win[w]].star_alpha * nv];
            for (int c = 0; c < p; ++c) {
                double acc = 0.0;
                for (int v = 0; v < nv; ++v) acc += al[v] * verts[((size_t)nd * nv + v) * p + c];
                thetas[w * p + c] = acc;
            }
        }
        Jv.resize(nw * nv); uv.resize(nw * nv * n_u); Jmin.resize(nw); Jth.resize(nw);
        solver_points((int64_t)(nw * nv), vc.data(), vpts.data(), 0, 0, Jv.data(), uv.data());
        // (the cell's own commutation is feasible at every vertex: no phase one over the simplex)
        solver_min((int64_t)nw, rc.data(), Rw.data(), ones.data(), Jmin.data());
        solver_points((int64_t)nw, sc.data(), thetas.data(), 0, 0, Jth.data(), uth.data());
    }
    size_t w = 0;
    for (size_t j = 0; j < n; ++j) {
        const int32_t nd = O[j];
        const int64_t star = sr[j].star == 1 ? (int64_t)sr[j].star_code : -1;
        const bool winner = w < win.size() && win[w] == j;
        // what the children inherit: the optima solved on this cell bound theirs from above, on
        // top of what the cell inherited itself
        std::shared_ptr<BoundMap> down = std::make_shared<BoundMap>();
        {
            std::vector<std::pair<uint64_t, double>> own;
            own.reserve(learned[j].size());
            for (const auto& kv : learned[j]) own.emplace_back(kv.first, kv.second.t);
            std::sort(own.begin(), own.end());
            const BoundMap* inh = node_bounds[(size_t)nd].get();
            const size_t ni = inh ? inh->size() : 0;
            down->code.reserve(ni + own.size());
            down->t.reserve(ni + own.size());
            size_t a = 0, b = 0;                // merge; what was solved HERE overrides what was inherited
            while (a < ni || b < own.size()) {
                if (b == own.size() || (a < ni && inh->code[a] < own[b].first)) {
                    down->code.push_back(inh->code[a]); down->t.push_back(inh->t[a]); ++a;
                } else {
                    if (a < ni && inh->code[a] == own[b].first) ++a;
                    down->code.push_back(own[b].first); down->t.push_back(own[b].second); ++b;
                }
            }
            if (down->size() > INHERIT_MAX) {   // keep what refutes; the rest only orders bar_D
                size_t w2 = 0;
                for (size_t q = 0; q < down->code.size(); ++q)
                    if (down->t[q] < 0.0) { down->code[w2] = down->code[q]; down->t[w2] = down->t[q]; ++w2; }
                down->code.resize(w2); down->t.resize(w2);
            }
        }
        if (!winner) {                          // no better commutation: bisect with the cell's own
            to_split.push_back(Split{nd, seq[nd],
                                     std::vector<double>(&costs[(size_t)nd * nv], &costs[(size_t)nd * nv] + nv),
                                     std::vector<double>(&inputs[(size_t)nd * nv * n_u],
                                                         &inputs[(size_t)nd * nv * n_u] + (size_t)nv * n_u),
                                     star, down});
            continue;
        }
        bool finite = std::isfinite(Jmin[w]) && std::isfinite(Jth[w]);
        for (int v = 0; v < nv; ++v) finite = finite && std::isfinite(Jv[w * nv + v]);
        if (!finite) {
            // a failed solve: the blacklist-and-retry path of the one-cell oracle
            // (lib/oracle.py:406-414) -- the caller's driver has it
            flags[nd] = EHM_FR_HAS_RECORD | EHM_FR_OPEN;
            ++st.open_cells;
            ++w;
            continue;
        }
        std::vector<double> nc(&Jv[w * nv], &Jv[w * nv] + nv);
        std::vector<double> nu(&uv[w * nv * n_u], &uv[w * nv * n_u] + (size_t)nv * n_u);
        double vmax = -INF, rise = -INF;
        for (int v = 0; v < nv; ++v) {
            vmax = std::max(vmax, costs[(size_t)nd * nv + v]);
            rise = std::max(rise, nc[(size_t)v] - costs[(size_t)nd * nv + v]);
        }
        // the bounds were proven against the interpolation of the OLD vertex costs; where the
        // adopted commutation's are larger the interpolant rises by at most the largest increase
        if (rise > 0.0)
            for (double& tb : down->t) tb += rise;
        const double rhs = std::max(eps_a, eps_r * Jth[w]);
        const bool small = vmax - Jmin[w] < rhs;
        if (small) {                            // lib/worker.py:396-401: adopt in place, look again
            seq[nd] = (int64_t)sr[j].star_code;
            std::memcpy(&costs[(size_t)nd * nv], nc.data(), 8 * (size_t)nv);
            std::memcpy(&inputs[(size_t)nd * nv * n_u], nu.data(), 8 * (size_t)nv * n_u);
            incumbent[nd] = star;
            node_bounds[(size_t)nd] = down;
            flags[nd] = EHM_FR_HAS_RECORD | EHM_FR_PENDING;
            lcss_work.push_back(nd);
            ++st.swaps;
        } else {
            to_split.push_back(Split{nd, (int64_t)sr[j].star_code, nc, nu, star, down});
        }
        ++w;
    }
}

// lib/worker.py:340-417 for the cells that hold a commutation: bar_E by best-first search over
// prefixes (bnb_frontier.bar_e_many: native queues, the parent's best-slack sequence tried first),
// closed cells are leaves, open ones go through bar_D.
void ehm_frontier::lcss_round(const std::vector<int32_t>& Lc, int launch_target,
                              std::vector<Split>& to_split) {
    const size_t n = Lc.size();
    if (!n) return;
    st.calls_bar_e += (int64_t)n;
    st.lcss_visits += (int64_t)n;
    tally_cells_e += (int64_t)n;
    alpha_pool.clear();
    std::vector<double> guard(n);
    for (size_t j = 0; j < n; ++j) {
        double mx = 0.0;
        for (int v = 0; v < nv; ++v) mx = std::max(mx, std::fabs(costs[(size_t)Lc[j] * nv + v]));
        guard[j] = INHERIT_GUARD * (1.0 + mx);
    }
    ehm_search_bare* Bq = nullptr;
    chk_search(ehm_search_bare_create((int32_t)n, n_modes, N, guard.data(), &Bq), "ehm_search_bare_create");
    struct Guard { ehm_search_bare* b; ~Guard() { ehm_search_bare_destroy(b); } } g{Bq};
    std::vector<Learned> learned(n);
    int64_t n_ask = 0, left_n = (int64_t)n;
    {   // inherited bounds (the queues keep the refuting ones)
        std::vector<uint64_t> bc;
        std::vector<double> bt;
        for (size_t j = 0; j < n; ++j) {
            const BoundMap* bm = node_bounds[(size_t)Lc[j]].get();
            if (!bm || bm->empty()) continue;
            bc.clear(); bt.clear();
            for (size_t q = 0; q < bm->code.size(); ++q)    // (the queues keep the refuting ones only)
                if (bm->t[q] < -guard[j]) { bc.push_back(bm->code[q]); bt.push_back(bm->t[q]); }
            if (bc.empty()) continue;
            chk_search(ehm_search_bare_bounds(Bq, (int32_t)j, (int64_t)bc.size(), bc.data(), bt.data()),
                       "ehm_search_bare_bounds");
        }
    }
    // the parent's best-slack sequence first: where its slack is not negative the cell is open and
    // the search is not run
    {
        std::vector<uint64_t> ic;
        std::vector<int32_t> in;
        std::vector<size_t> ij;
        for (size_t j = 0; j < n; ++j)
            if (incumbent[Lc[j]] >= 0) {
                ic.push_back((uint64_t)incumbent[Lc[j]]);
                in.push_back(Lc[j]);
                ij.push_back(j);
            }
        if (!ic.empty()) {
            std::vector<double> t, al;
            tally_seed += (int64_t)ic.size();
            slack_pairs(ic, in, t, &al);
            for (size_t q = 0; q < ic.size(); ++q) {
                const int64_t at = (int64_t)(alpha_pool.size() / nv);
                alpha_pool.insert(alpha_pool.end(), &al[q * nv], &al[q * nv] + nv);
                learned[ij[q]][ic[q]] = Known{t[q], at};
                const int open = t[q] >= 0.0;
                left_n -= open;
                chk_search(ehm_search_bare_seed(Bq, (int32_t)ij[q], ic[q], t[q], open),
                           "ehm_search_bare_seed");
            }
        }
    }
    std::vector<uint64_t> codes;
    std::vector<int32_t> owner, node_of;
    std::vector<double> t;
    while (left_n > 0) {
        const int width = (int)std::max<int64_t>(1, std::min<int64_t>(BATCH, launch_target /
                                                                     std::max<int64_t>(1, left_n * n_modes)));
        chk_search(ehm_search_bare_step(Bq, width, &n_ask, &left_n), "ehm_search_bare_step");
        if (!n_ask) break;
        codes.resize((size_t)n_ask); owner.resize((size_t)n_ask); node_of.resize((size_t)n_ask);
        chk_search(ehm_search_bare_asks(Bq, codes.data(), owner.data()), "ehm_search_bare_asks");
        for (int64_t a = 0; a < n_ask; ++a) node_of[(size_t)a] = Lc[owner[(size_t)a]];
        tally_bar_e += (int64_t)codes.size();
        slack_pairs(codes, node_of, t, nullptr);
        chk_search(ehm_search_bare_answer(Bq, t.data(), &left_n), "ehm_search_bare_answer");
    }
    std::vector<int8_t> closed(n);
    std::vector<double> margin(n);
    int64_t counts[2] = {0, 0};
    chk_search(ehm_search_bare_result(Bq, closed.data(), margin.data(), counts), "ehm_search_bare_result");
    st.prefixes_expanded += counts[0];
    st.answered_without_a_problem += counts[1];
    std::vector<int32_t> O;
    std::vector<Learned> lo;
    std::vector<uint64_t> lc;
    std::vector<double> lt;
    for (size_t j = 0; j < n; ++j) {
        const int32_t nd = Lc[j];
        if (closed[j]) {
            flags[nd] = EHM_FR_HAS_RECORD | EHM_FR_CLOSED;
            ++st.regions;
            continue;
        }
        // what bar_E solved on the cell: the same problems bar_D asks about
        int64_t cnt = 0;
        chk_search(ehm_search_bare_learned(Bq, (int32_t)j, &cnt, nullptr, nullptr), "ehm_search_bare_learned");
        if (cnt) {
            lc.resize((size_t)cnt); lt.resize((size_t)cnt);
            chk_search(ehm_search_bare_learned(Bq, (int32_t)j, &cnt, lc.data(), lt.data()),
                       "ehm_search_bare_learned");
            for (int64_t q = 0; q < cnt; ++q) learned[j][lc[(size_t)q]] = Known{lt[(size_t)q], -1};
        }
        O.push_back(nd);
        lo.push_back(std::move(learned[j]));
    }
    bar_d(O, lo, launch_target, to_split);
    // (a cell that adopted a commutation in place holds its new bounds; everything else has
    // handed them on or is a leaf)
    for (size_t j = 0; j < n; ++j)
        if (!(flags[Lc[j]] & EHM_FR_PENDING)) node_bounds[(size_t)Lc[j]].reset();
}

void ehm_frontier::run(const ehm_frontier_opts& o) {
    const int cap = o.round_cap > 0 ? o.round_cap : 4096;
    max_depth = o.max_depth;
    const int target = o.launch_target > 0 ? o.launch_target : 65536;
    const double t0 = now();
    st.truncated = 0;
    while (!ecc_work.empty() || !lcss_work.empty()) {
        if ((o.max_visits > 0 && st.visits >= o.max_visits) ||
            (o.min_regions > 0 && st.regions >= o.min_regions)) {
            st.truncated = 1;
            break;
        }
        int64_t room = cap;
        if (o.max_visits > 0) room = std::min<int64_t>(room, o.max_visits - st.visits);
        // cells that hold a commutation before cells that still look for one, deepest first
        // (bnb_frontier.grow_frontier, order 'lcss-first'; stable)
        auto deepest_first = [&](std::vector<int32_t>& w) {
            std::stable_sort(w.begin(), w.end(), [&](int32_t a, int32_t b) { return depth[a] > depth[b]; });
        };
        deepest_first(lcss_work);
        deepest_first(ecc_work);
        std::vector<int32_t> Lc, E;
        const size_t nl = (size_t)std::min<int64_t>(room, (int64_t)lcss_work.size());
        Lc.assign(lcss_work.begin(), lcss_work.begin() + nl);
        lcss_work.erase(lcss_work.begin(), lcss_work.begin() + nl);
        room -= (int64_t)nl;
        const size_t ne = (size_t)std::min<int64_t>(room, (int64_t)ecc_work.size());
        E.assign(ecc_work.begin(), ecc_work.begin() + ne);
        ecc_work.erase(ecc_work.begin(), ecc_work.begin() + ne);
        ++st.rounds;
        st.visits += (int64_t)(nl + ne);
        std::vector<Split> to_split;
        ecc_round(E, to_split);
        lcss_round(Lc, target, to_split);
        bisect(to_split);
    }
    st.n_nodes = n_nodes();
    st.seconds_total += now() - t0;
    if (const char* e = std::getenv("EHM_FR_TALLY"))
        if (e[0] == '1')
            std::fprintf(stderr, "ehm_frontier tally: lcss cells %lld (slack problems: incumbent seeds %lld, "
                         "bar_E search %lld), bar_D cells %lld (slack problems %lld), regions %lld\n",
                         (long long)tally_cells_e, (long long)tally_seed, (long long)tally_bar_e,
                         (long long)tally_cells_d, (long long)tally_bar_d, (long long)st.regions);
    if (dev) {                                  // since the last reset
        st.lp_solves = dev->total_lp() - base_ctr[0];
        st.launches = dev->launches - base_ctr[1];
        st.blocks_loaded = dev->total_loaded() - base_ctr[2];
        st.stalled = dev->stalled - base_ctr[3];
        st.slivers = dev->slivers - base_ctr[4];
    }
}

// ---- C-ABI --------------------------------------------------------------------------------------
extern "C" {

const char* ehm_frontier_last_error(void) { return g_err; }

static int init_common(ehm_frontier* f, int n_x, int n_u, int n_modes, int N, double eps_a,
                       double eps_r) {
    f->p = n_x; f->nv = n_x + 1; f->n_u = n_u; f->n_modes = n_modes; f->N = N;
    f->base = (uint64_t)n_modes + 1;
    f->pw.assign((size_t)N + 1, 1);
    for (int i = 1; i <= N; ++i) f->pw[(size_t)i] = f->pw[(size_t)i - 1] * f->base;
    f->eps_a = eps_a; f->eps_r = eps_r;
    int rc = ehm_search_create(n_x, n_modes, N, &f->S);
    if (rc) return fail(rc, "ehm_search_create: %s", ehm_search_last_error());
    return EHM_OK;
}

int ehm_frontier_create_custom(int32_t n_x, int32_t n_u, int32_t n_modes, int32_t N,
                               const ehm_pair_solvers* solvers, double eps_a, double eps_r,
                               ehm_frontier** out) {
    if (!out || !solvers || !solvers->points || !solvers->slack || !solvers->min ||
        !solvers->split || n_x < 1 ||
        n_u < 1 || n_modes < 1 || N < 1)
        return fail(EHM_E_INVALID, "ehm_frontier_create_custom: bad argument");
    ehm_frontier* f = new (std::nothrow) ehm_frontier();
    if (!f) return fail(EHM_E_CAPACITY, "ehm_frontier_create_custom: out of memory");
    f->sol = *solvers;
    int rc = init_common(f, n_x, n_u, n_modes, N, eps_a, eps_r);
    if (rc) { delete f; return rc; }
    *out = f;
    return EHM_OK;
}

static int make_table(DeviceSolver* D, Table& T, int horizon, int slots, int full_length,
                      double eps_a, double eps_r) {
    T.horizon = horizon; T.slots = slots; T.full_length = full_length;
    T.cond.init(&D->law, horizon);
    T.n = T.cond.n; T.m = T.cond.m;
    T.slot_len.assign((size_t)slots, 0);
    T.slot_code.assign((size_t)slots, 0);
    T.slot_stamp.assign((size_t)slots, 0);
    T.counts.assign((size_t)5 * (D->N + 1), 0);
    // every slot starts as the block of the sequence 0 .. 0 (as sequences.PrefixTable creates its
    // handle: ehm_problem_create derives the eliminated columns from FULL blocks); problem data is
    // written on demand
    const size_t nG = (size_t)T.m * T.n, nS = (size_t)T.m * D->p;
    std::vector<double> G0(nG), w0((size_t)T.m), S0(nS);
    {
        std::vector<int> zero((size_t)horizon, 0);
        T.cond.block(zero.data(), horizon, G0.data(), w0.data(), S0.data());
    }
    std::vector<double> G, w, S, c((size_t)T.n, 0.0);
    std::vector<uint8_t> deltas;
    try {
        G.resize((size_t)slots * nG); w.resize((size_t)slots * T.m); S.resize((size_t)slots * nS);
        deltas.assign((size_t)slots * D->law.n_modes * horizon, 0);
    } catch (const std::bad_alloc&) {
        return fail(EHM_E_CAPACITY, "ehm_frontier_create: out of host memory for %d slots", slots);
    }
    for (int k = 0; k < slots; ++k) {
        std::memcpy(&G[(size_t)k * nG], G0.data(), nG * 8);
        std::memcpy(&w[(size_t)k * T.m], w0.data(), (size_t)T.m * 8);
        std::memcpy(&S[(size_t)k * nS], S0.data(), nS * 8);
    }
    for (int j = horizon * D->law.n_u; j < T.n; ++j) c[(size_t)j] = 1.0;
    ehm_problem_desc d{};
    d.n = T.n; d.m = T.m; d.p = D->p; d.n_u = D->n_u; d.n_delta = slots;
    d.delta_len = D->law.n_modes * horizon;
    d.G = G.data(); d.w = w.data(); d.S = S.data(); d.c = c.data(); d.deltas = deltas.data();
    d.eps_a = eps_a; d.eps_r = eps_r;
    int rc = ehm_problem_create(&d, D->device, &T.P);
    if (rc) return fail(rc, "ehm_problem_create (%d slots, n=%d m=%d): %s", slots, T.n, T.m,
                        ehm_last_error());
    return EHM_OK;
}

int ehm_frontier_create(const ehm_pwa_law* law, int32_t n_tables, const int32_t* horizons,
                        const int32_t* slots, int device, double eps_a, double eps_r,
                        ehm_frontier** out) {
    if (!out) return fail(EHM_E_INVALID, "ehm_frontier_create: out is NULL");
    DeviceSolver* D = new (std::nothrow) DeviceSolver();
    if (!D) return fail(EHM_E_CAPACITY, "ehm_frontier_create: out of memory");
    int rc = copy_law(law, D->law);
    if (rc) { delete D; return rc; }
    const int N = D->law.N;
    bool ok = n_tables >= 1 && n_tables <= 16 && horizons && slots && N <= 60 &&
              horizons[n_tables - 1] == N;
    for (int t = 0; ok && t < n_tables; ++t)
        ok = horizons[t] >= 1 && (t == 0 || horizons[t] > horizons[t - 1]) && slots[t] >= 0;
    if (!ok) {
        delete D;
        return fail(EHM_E_INVALID, "ehm_frontier_create: horizons must ascend to N, slots >= 0");
    }
    D->device = device; D->short_len = n_tables > 1 ? horizons[0] : 0;
    if (const char* e = getenv("EHM_FRONTIER_THREADS")) D->parallel_tables = atoi(e) != 0;
    D->p = D->law.n_x; D->nv = D->p + 1; D->n_u = D->law.n_u; D->N = N;
    D->base = (uint64_t)D->law.n_modes + 1;
    D->tab.resize((size_t)n_tables);
    D->tab_of_len.assign((size_t)N + 1, n_tables - 1);
    int lo = 0;
    for (int t = 0; !rc && t < n_tables; ++t) {
        // the prefixes this table holds: lo .. horizons[t] steps (the first one from the empty prefix)
        double count = 0.0, pwr = 1.0;
        for (int k = 0; k <= horizons[t]; ++k) {
            if (k >= lo) count += pwr;
            pwr *= D->law.n_modes;
        }
        for (int k = lo; k <= horizons[t]; ++k) D->tab_of_len[(size_t)k] = t;
        int want = slots[t];
        if (want == 0 || (double)want > count) want = (int)std::min(count, 1e6);   // one slot each
        want = std::max(want, 16);
        rc = make_table(D, D->tab[(size_t)t], horizons[t], want, horizons[t] == N ? N : -1, eps_a, eps_r);
        lo = horizons[t] + 1;
    }
    ehm_frontier* f = rc ? nullptr : new (std::nothrow) ehm_frontier();
    if (!rc && !f) rc = fail(EHM_E_CAPACITY, "ehm_frontier_create: out of memory");
    if (!rc) {
        f->dev = D;
        f->sol.user = D;
        f->sol.points = dev_points; f->sol.slack = dev_slack; f->sol.min = dev_min;
        f->sol.split = dev_split;
        rc = init_common(f, D->law.n_x, D->law.n_u, D->law.n_modes, D->law.N, eps_a, eps_r);
    }
    if (rc) {
        for (Table& T : D->tab)
            if (T.P) ehm_problem_destroy(T.P);
        delete D;
        delete f;
        return rc;
    }
    *out = f;
    return EHM_OK;
}

int ehm_frontier_destroy(ehm_frontier* f) {
    if (!f) return EHM_OK;
    if (f->S) ehm_search_destroy(f->S);
    if (f->dev) {
        for (Table& T : f->dev->tab)
            if (T.P) ehm_problem_destroy(T.P);
        delete f->dev;
    }
    delete f;
    return EHM_OK;
}

int ehm_frontier_set_eps(ehm_frontier* f, double eps_a, double eps_r) {
    if (!f) return fail(EHM_E_INVALID, "ehm_frontier_set_eps: NULL handle");
    f->eps_a = eps_a; f->eps_r = eps_r;
    if (f->dev)
        for (Table& T : f->dev->tab)
            if (T.P) {
                int rc = ehm_problem_set_eps(T.P, eps_a, eps_r);
                if (rc) return fail(rc, "ehm_problem_set_eps: %s", ehm_last_error());
            }
    return EHM_OK;
}

int ehm_frontier_table(ehm_frontier* f, int32_t index, ehm_problem** table, int32_t* horizon,
                       int32_t* slots, int64_t* evicted) {
    if (!f) return fail(EHM_E_INVALID, "ehm_frontier_table: NULL handle");
    const int32_t n = f->dev ? (int32_t)f->dev->tab.size() : 0;
    if (index < 0 || index >= n) {
        if (table) *table = nullptr;
        return index == n ? EHM_OK : fail(EHM_E_INVALID, "ehm_frontier_table: index %d of %d", index, n);
    }
    const Table& T = f->dev->tab[(size_t)index];
    if (table) *table = T.P;
    if (horizon) *horizon = T.horizon;
    if (slots) *slots = T.slots;
    if (evicted) *evicted = T.evicted;
    return EHM_OK;
}

int ehm_frontier_reset(ehm_frontier* f) {
    if (!f) return fail(EHM_E_INVALID, "ehm_frontier_reset: NULL handle");
    if (f->S) ehm_search_destroy(f->S);
    f->S = nullptr;
    int rc = ehm_search_create(f->p, f->n_modes, f->N, &f->S);
    if (rc) return fail(rc, "ehm_search_create: %s", ehm_search_last_error());
    f->feas_n = 0;
    f->verts.clear(); f->costs.clear(); f->inputs.clear(); f->pids.clear();
    f->left.clear(); f->right.clear(); f->depth.clear(); f->seq.clear(); f->witness.clear();
    f->incumbent.clear(); f->node_bounds.clear();
    f->flags.clear(); f->n_roots = 0;
    f->ecc_work.clear(); f->lcss_work.clear();
    f->opt_of.clear(); f->opt_J.clear(); f->opt_u.clear();
    f->poisoned = false;
    f->st = ehm_frontier_stats{};
    if (f->dev) {
        f->base_ctr[0] = f->dev->total_lp();
        f->base_ctr[1] = f->dev->launches;
        f->base_ctr[2] = f->dev->total_loaded();
        f->base_ctr[3] = f->dev->stalled;
        f->base_ctr[4] = f->dev->slivers;
    }
    return EHM_OK;
}

int ehm_frontier_add_root(ehm_frontier* f, const double* vertices) {
    if (!f || !vertices) return fail(EHM_E_INVALID, "ehm_frontier_add_root: bad argument");
    if (f->poisoned)
        return fail(EHM_E_INVALID, "ehm_frontier_add_root: an earlier call failed inside a round; "
                                   "ehm_frontier_reset first");
    if (f->n_roots != f->n_nodes())
        return fail(EHM_E_INVALID, "ehm_frontier_add_root: the tree has been grown already (reset first)");
    try {
        const int32_t k = f->new_node(vertices, 0);
        int rc = ehm_search_point_ids(f->S, f->nv, vertices, &f->pids[(size_t)k * f->nv]);
        if (rc) return fail(rc, "ehm_search_point_ids: %s", ehm_search_last_error());
        f->ecc_work.push_back(k);
        ++f->n_roots;
    } catch (const std::bad_alloc&) {
        return fail(EHM_E_CAPACITY, "ehm_frontier_add_root: out of memory");
    }
    return EHM_OK;
}

int ehm_frontier_pending(const ehm_frontier* f, int64_t* n_pending) {
    if (!f || !n_pending) return fail(EHM_E_INVALID, "ehm_frontier_pending: bad argument");
    *n_pending = (int64_t)(f->ecc_work.size() + f->lcss_work.size());
    return EHM_OK;
}

// Work leaves this handle: the shallowest pending cells (the largest sub-trees still to grow) come
// out with their records; here they stay leaves flagged EHM_FR_REMOTE.
int ehm_frontier_take(ehm_frontier* f, int64_t max_cells, int64_t* n_taken, int32_t* node,
                      double* vertices, int32_t* sequence, double* vertex_costs,
                      double* vertex_inputs, int32_t* depth) {
    if (!f || !n_taken || max_cells < 0 || (max_cells && (!node || !vertices || !sequence ||
                                                          !vertex_costs || !vertex_inputs || !depth)))
        return fail(EHM_E_INVALID, "ehm_frontier_take: bad argument");
    if (f->poisoned)
        return fail(EHM_E_INVALID, "ehm_frontier_take: an earlier call failed inside a round; "
                                   "ehm_frontier_reset first");
    *n_taken = 0;
    try {
        // (work list, position) of every pending cell, shallowest first; ties: cells that hold a
        // commutation first, then the order of the lists -- a fixed rule, so a run that is repeated
        // hands over the same cells
        struct Pick { int32_t depth; int list; size_t at; };
        std::vector<Pick> all;
        for (size_t a = 0; a < f->lcss_work.size(); ++a) all.push_back({f->depth[f->lcss_work[a]], 0, a});
        for (size_t a = 0; a < f->ecc_work.size(); ++a) all.push_back({f->depth[f->ecc_work[a]], 1, a});
        std::stable_sort(all.begin(), all.end(), [](const Pick& x, const Pick& y) { return x.depth < y.depth; });
        const size_t n = (size_t)std::min<int64_t>(max_cells, (int64_t)all.size());
        const size_t sx = (size_t)f->nv * f->p;
        std::vector<uint8_t> gone_l(f->lcss_work.size(), 0), gone_e(f->ecc_work.size(), 0);
        for (size_t i = 0; i < n; ++i) {
            const int32_t k = all[i].list ? f->ecc_work[all[i].at] : f->lcss_work[all[i].at];
            (all[i].list ? gone_e : gone_l)[all[i].at] = 1;
            node[i] = k;
            depth[i] = f->depth[k];
            std::memcpy(vertices + i * sx, &f->verts[(size_t)k * sx], 8 * sx);
            std::memcpy(vertex_costs + i * f->nv, &f->costs[(size_t)k * f->nv], 8 * (size_t)f->nv);
            std::memcpy(vertex_inputs + i * f->nv * f->n_u, &f->inputs[(size_t)k * f->nv * f->n_u],
                        8 * (size_t)f->nv * f->n_u);
            int32_t* sq = sequence + i * f->N;
            if (f->seq[k] < 0) {
                for (int j = 0; j < f->N; ++j) sq[j] = -1;
            } else {
                uint64_t c = (uint64_t)f->seq[k];
                for (int j = 0; j < f->N; ++j) { sq[j] = (int32_t)(c % f->base) - 1; c /= f->base; }
            }
            f->flags[k] = (uint8_t)((f->flags[k] & EHM_FR_HAS_RECORD) | EHM_FR_REMOTE);
            f->node_bounds[(size_t)k].reset();
        }
        auto compact = [](std::vector<int32_t>& w, const std::vector<uint8_t>& gone) {
            size_t o = 0;
            for (size_t a = 0; a < w.size(); ++a)
                if (!gone[a]) w[o++] = w[a];
            w.resize(o);
        };
        compact(f->lcss_work, gone_l);
        compact(f->ecc_work, gone_e);
        *n_taken = (int64_t)n;
    } catch (const std::bad_alloc&) {
        return fail(EHM_E_CAPACITY, "ehm_frontier_take: out of memory");
    }
    return EHM_OK;
}

// Work arrives: cells another handle gave up become roots of this handle's forest (a cell that
// holds a commutation goes on with lcss from its record, lib/scheduler.py:633-639; one without
// looks for one with ecc).  Depths are kept, so a depth limit means the same in both handles.
int ehm_frontier_give(ehm_frontier* f, int64_t n, const double* vertices, const int32_t* sequence,
                      const double* vertex_costs, const double* vertex_inputs, const int32_t* depth) {
    if (!f || n < 0 || (n && (!vertices || !sequence || !vertex_costs || !vertex_inputs)))
        return fail(EHM_E_INVALID, "ehm_frontier_give: bad argument");
    if (f->poisoned)
        return fail(EHM_E_INVALID, "ehm_frontier_give: an earlier call failed inside a round; "
                                   "ehm_frontier_reset first");
    if (f->n_roots != f->n_nodes())
        return fail(EHM_E_INVALID, "ehm_frontier_give: the tree has been grown already (reset first)");
    const size_t sx = (size_t)f->nv * f->p;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t* sq = sequence + (size_t)i * f->N;
        const bool has = sq[0] >= 0;
        for (int j = 0; j < f->N; ++j)
            if ((sq[j] >= 0) != has || sq[j] >= f->n_modes)
                return fail(EHM_E_INVALID, "ehm_frontier_give: cell %lld: bad mode sequence", (long long)i);
        if (has)
            for (int v = 0; v < f->nv; ++v)
                if (!std::isfinite(vertex_costs[(size_t)i * f->nv + v]))
                    return fail(EHM_E_INVALID, "ehm_frontier_give: cell %lld holds a commutation but no "
                                               "vertex costs", (long long)i);
    }
    try {
        for (int64_t i = 0; i < n; ++i) {
            const int32_t k = f->new_node(vertices + (size_t)i * sx, depth ? depth[i] : 0);
            int rc = ehm_search_point_ids(f->S, f->nv, vertices + (size_t)i * sx, &f->pids[(size_t)k * f->nv]);
            if (rc) return fail(rc, "ehm_search_point_ids: %s", ehm_search_last_error());
            ++f->n_roots;
            const int32_t* sq = sequence + (size_t)i * f->N;
            if (sq[0] < 0) {
                f->ecc_work.push_back(k);
                continue;
            }
            f->seq[k] = (int64_t)f->seq_code(sq);
            std::memcpy(&f->costs[(size_t)k * f->nv], vertex_costs + (size_t)i * f->nv, 8 * (size_t)f->nv);
            std::memcpy(&f->inputs[(size_t)k * f->nv * f->n_u], vertex_inputs + (size_t)i * f->nv * f->n_u,
                        8 * (size_t)f->nv * f->n_u);
            f->flags[k] = EHM_FR_HAS_RECORD | EHM_FR_PENDING;
            f->lcss_work.push_back(k);
            f->st.depth = std::max(f->st.depth, f->depth[k]);
        }
    } catch (const std::bad_alloc&) {
        return fail(EHM_E_CAPACITY, "ehm_frontier_give: out of memory");
    }
    return EHM_OK;
}

int ehm_frontier_run(ehm_frontier* f, const ehm_frontier_opts* opts, ehm_frontier_stats* stats) {
    if (!f) return fail(EHM_E_INVALID, "ehm_frontier_run: NULL handle");
    if (f->poisoned)
        return fail(EHM_E_INVALID, "ehm_frontier_run: an earlier call failed inside a round -- the "
                                   "tree is incomplete; ehm_frontier_reset first");
    ehm_frontier_opts o{};
    if (opts) o = *opts;
    try {
        f->run(o);
    } catch (const Fail& e) {
        f->poisoned = true;
        if (f->S) (void)ehm_search_abandon(f->S);
        return fail(e.code, "%s", e.msg.c_str());
    } catch (const std::bad_alloc&) {
        f->poisoned = true;
        if (f->S) (void)ehm_search_abandon(f->S);
        return fail(EHM_E_CAPACITY, "ehm_frontier_run: out of memory");
    }
    if (stats) *stats = f->st;
    return EHM_OK;
}

int ehm_frontier_p_theta(ehm_frontier* f, int64_t n, const double* theta, double* J, double* u0,
                         int32_t* sequence) {
    if (!f || n < 0 || (n && (!theta || !J))) return fail(EHM_E_INVALID, "ehm_frontier_p_theta: bad argument");
    if (f->poisoned)
        return fail(EHM_E_INVALID, "ehm_frontier_p_theta: an earlier call failed inside a round; "
                                   "ehm_frontier_reset first");
    try {
        f->p_theta(n, theta, J, u0, sequence);
    } catch (const Fail& e) {
        f->poisoned = true;
        return fail(e.code, "%s", e.msg.c_str());
    } catch (const std::bad_alloc&) {
        f->poisoned = true;
        return fail(EHM_E_CAPACITY, "ehm_frontier_p_theta: out of memory");
    }
    return EHM_OK;
}

int ehm_frontier_sizes(const ehm_frontier* f, int64_t* n_nodes, int64_t* n_roots) {
    if (!f) return fail(EHM_E_INVALID, "ehm_frontier_sizes: NULL handle");
    if (n_nodes) *n_nodes = f->n_nodes();
    if (n_roots) *n_roots = f->n_roots;
    return EHM_OK;
}

int ehm_frontier_export(const ehm_frontier* f, double* vertices, int32_t* left, int32_t* right,
                        int32_t* sequence, double* vertex_costs, double* vertex_inputs,
                        uint8_t* flags) {
    if (!f) return fail(EHM_E_INVALID, "ehm_frontier_export: NULL handle");
    if (f->poisoned)
        return fail(EHM_E_INVALID, "ehm_frontier_export: an earlier call failed inside a round -- "
                                   "the tree is incomplete; ehm_frontier_reset first");
    const size_t n = (size_t)f->n_nodes();
    if (vertices) std::memcpy(vertices, f->verts.data(), f->verts.size() * 8);
    if (left) std::memcpy(left, f->left.data(), n * 4);
    if (right) std::memcpy(right, f->right.data(), n * 4);
    if (vertex_costs) std::memcpy(vertex_costs, f->costs.data(), f->costs.size() * 8);
    if (vertex_inputs) std::memcpy(vertex_inputs, f->inputs.data(), f->inputs.size() * 8);
    if (flags) std::memcpy(flags, f->flags.data(), n);
    if (sequence)
        for (size_t k = 0; k < n; ++k) {
            int32_t* s = sequence + k * f->N;
            if (f->seq[k] < 0) {
                for (int i = 0; i < f->N; ++i) s[i] = -1;
                continue;
            }
            uint64_t c = (uint64_t)f->seq[k];
            for (int i = 0; i < f->N; ++i) { s[i] = (int32_t)(c % f->base) - 1; c /= f->base; }
        }
    return EHM_OK;
}

int ehm_frontier_lp_counts(const ehm_frontier* f, int64_t* out) {
    if (!f || !out) return fail(EHM_E_INVALID, "ehm_frontier_lp_counts: bad argument");
    const size_t per = (size_t)5 * (f->N + 1);
    if (f->dev)
        for (size_t t = 0; t < f->dev->tab.size(); ++t)
            std::memcpy(out + t * per, f->dev->tab[t].counts.data(), per * 8);
    return EHM_OK;
}

int ehm_frontier_condense(const ehm_pwa_law* law, int32_t horizon, int32_t len,
                          const int32_t* prefix, int32_t dims[2], double* G, double* w, double* S) {
    Law L;
    int rc = copy_law(law, L);
    if (rc) return rc;
    if (horizon < 1 || horizon > L.N || len < 0 || len > horizon || (len && !prefix) || !dims)
        return fail(EHM_E_INVALID, "ehm_frontier_condense: bad argument");
    for (int i = 0; i < len; ++i)
        if (prefix[i] < 0 || prefix[i] >= L.n_modes)
            return fail(EHM_E_INVALID, "ehm_frontier_condense: mode out of range");
    Condenser C;
    C.init(&L, horizon);
    dims[0] = C.n; dims[1] = C.m;
    if (!G) return EHM_OK;
    if (!w || !S) return fail(EHM_E_INVALID, "ehm_frontier_condense: w / S is NULL");
    std::vector<int> pre(prefix, prefix + len);
    C.block(pre.data(), len, G, w, S);
    return EHM_OK;
}

}  // extern "C"
