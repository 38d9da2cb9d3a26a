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
#include <cstring>
#include <limits>
#include <new>
#include <stdexcept>
#include <string>
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
    Condenser cond;
    std::vector<int64_t> counts;               // [5][N + 1]
    int64_t lp = 0, loaded = 0;
};

struct DeviceSolver {
    Law law;
    int device = 0, short_len = 0, p = 0, nv = 0, n_u = 0, N = 0;
    uint64_t base = 1;
    Table tab[2];                              // 0 short, 1 long
    int64_t stalled = 0, slivers = 0, stalled_relax = 0, launches = 0, accepted_inaccurate = 0;
    // sequences.decisive_inaccurate: status word 1 | (decade << 8) of a stalled solve (ehm_dev.h)
    static bool decisive(int32_t status, double value) {
        const int dec = (status >> 8) & 0xff;
        if (!(status & 1) || dec > 6 || !std::isfinite(value)) return false;
        const double err = 100.0 * std::pow(10.0, dec - 10) * (1.0 + std::fabs(value));
        return std::fabs(value) > err;
    }
    // scratch
    std::vector<double> bG, bw, bS;

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
    // slots of the distinct prefixes `uniq` (at most T.slots of them), loading what is missing
    void ensure(Table& T, const std::vector<uint64_t>& uniq, std::vector<int32_t>& slot_u) {
        std::vector<uint64_t> missing;
        for (uint64_t c : uniq)
            if (!T.slot_of.count(c)) missing.push_back(c);
        if (!missing.empty() && (int64_t)T.slot_of.size() + (int64_t)missing.size() > T.slots) {
            T.slot_of.clear();
            missing = uniq;
        }
        if (!missing.empty()) {
            const int first = (int)T.slot_of.size();
            const size_t nG = (size_t)T.m * T.n, nS = (size_t)T.m * p;
            // blocks go up in groups: the staging buffers stay small whatever a launch names
            const size_t group = 256;
            int pre[64];
            for (size_t g0 = 0; g0 < missing.size(); g0 += group) {
                const size_t cnt = std::min(group, missing.size() - g0);
                bG.resize(cnt * nG); bw.resize(cnt * T.m); bS.resize(cnt * nS);
                for (size_t k = 0; k < cnt; ++k) {
                    const uint64_t c = missing[g0 + k];
                    const int len = len_of(c);
                    digits(c, pre);
                    T.cond.block(pre, len, &bG[k * nG], &bw[k * T.m], &bS[k * nS]);
                    T.slot_of[c] = first + (int)(g0 + k);
                    T.slot_len[first + g0 + k] = len;
                }
                chk_dev(ehm_problem_update_blocks(T.P, first + (int)g0, (int)cnt, bG.data(), bw.data(),
                                                  bS.data()), "ehm_problem_update_blocks");
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
    void split_tables(int64_t n_pairs, const uint64_t* code, std::vector<int64_t> sel[2]) const {
        sel[0].clear(); sel[1].clear();
        for (int64_t k = 0; k < n_pairs; ++k)
            sel[(short_len > 0 && len_of(code[k]) <= short_len) ? 0 : 1].push_back(k);
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
        std::vector<int64_t> sel[2];
        split_tables(n_pairs, code, sel);
        std::vector<double> th, tau, uu, Jk;
        std::vector<int32_t> st, slot2;
        std::vector<uint8_t> ok;
        for (int t = 0; t < 2; ++t) {
            if (sel[t].empty()) continue;
            Table& T = tab[t];
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
        }
    }

    // sequences.PrefixTable.slack_by_prefix_index (+ _settle_stalled)
    void slack(int64_t n_pairs, const uint64_t* code, const double* R, const double* Vbar,
               const uint8_t* known, double* t_out) {
        for (int64_t k = 0; k < n_pairs; ++k) t_out[k] = -INF;
        std::vector<int64_t> sel[2];
        split_tables(n_pairs, code, sel);
        const size_t sx = (size_t)nv * p;
        std::vector<double> Rb, Vb, tau, al, tk;
        std::vector<int32_t> st, sl;
        std::vector<uint8_t> ok, v;
        for (int tt = 0; tt < 2; ++tt) {
            if (sel[tt].empty()) continue;
            Table& T = tab[tt];
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
                for (size_t g = 0; g < good.size(); ++g) t_out[part[good[g]]] = tk[g];
            });
        }
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
              const uint8_t* known, double* t) {
    try {
        static_cast<DeviceSolver*>(user)->slack(n, code, R, V, known, t);
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
struct ehm_frontier {
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
    std::vector<uint8_t> flags;
    int64_t n_roots = 0;
    std::vector<int32_t> ecc_work, lcss_work;
    // vertex optima by (sequence, point id)
    std::unordered_map<uint64_t, int64_t> opt_of;
    std::vector<double> opt_J, opt_u;
    ehm_frontier_stats st{};
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
        seq.push_back(-1); witness.push_back(-1);
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
                      const uint8_t* known, double* t) {
        const double t0 = now();
        int rc = sol.slack(sol.user, n, code, R, V, known, t);
        st.seconds_solvers += now() - t0;
        if (rc) raise(rc, "%s", dev ? g_err : "the caller's slack solver failed");
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

    void ecc_round(const std::vector<int32_t>& E);
    void lcss_round(const std::vector<int32_t>& Lc, int launch_target);
    void run(const ehm_frontier_opts& o);
};

void ehm_frontier::ecc_round(const std::vector<int32_t>& E) {
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
                double s = R[c];
                for (int v = 1; v < nv; ++v) s += R[(size_t)v * p + c];
                centre[i * p + c] = s / (double)nv;
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
        if (opt_of.size() > OPTIMA_MEMO_LIMIT) { opt_of.clear(); opt_J.clear(); opt_u.clear(); }
        std::vector<uint64_t> mcode;
        std::vector<double> mth;
        std::vector<uint64_t> mkey;
        for (size_t h = 0; h < have.size(); ++h) {
            const int32_t nd = E[have[h]];
            for (int v = 0; v < nv; ++v) {
                const uint64_t key = ((uint64_t)found[have[h]] << PID_BITS) | (uint64_t)pids[(size_t)nd * nv + v];
                if (opt_of.count(key)) continue;
                opt_of[key] = -1 - (int64_t)mkey.size();        // claimed by this launch
                mkey.push_back(key);
                mcode.push_back((uint64_t)found[have[h]]);
                mth.insert(mth.end(), &verts[((size_t)nd * nv + v) * p], &verts[((size_t)nd * nv + v) * p] + p);
            }
        }
        st.optima_asked += (int64_t)have.size() * nv;
        st.optima_solved += (int64_t)mkey.size();
        if (!mkey.empty()) {
            std::vector<double> J(mkey.size()), u(mkey.size() * n_u);
            solver_points((int64_t)mkey.size(), mcode.data(), mth.data(), 0, 1, J.data(), u.data());
            for (size_t a = 0; a < mkey.size(); ++a) {
                opt_of[mkey[a]] = (int64_t)opt_J.size();
                opt_J.push_back(J[a]);
                opt_u.insert(opt_u.end(), &u[a * n_u], &u[a * n_u] + n_u);
            }
        }
        for (size_t h = 0; h < have.size(); ++h) {
            const int32_t nd = E[have[h]];
            bool finite = true;
            for (int v = 0; v < nv; ++v) {
                const uint64_t key = ((uint64_t)found[have[h]] << PID_BITS) | (uint64_t)pids[(size_t)nd * nv + v];
                const int64_t at = opt_of[key];
                costs[(size_t)nd * nv + v] = opt_J[(size_t)at];
                std::memcpy(&inputs[((size_t)nd * nv + v) * n_u], &opt_u[(size_t)at * n_u], 8 * (size_t)n_u);
                finite = finite && std::isfinite(opt_J[(size_t)at]);
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
    // the cells without a commutation: bisect, the children look again
    if (!none.empty()) {
        const size_t sx = (size_t)nv * p;
        std::vector<double> Rs(none.size() * sx), S1(none.size() * sx), S2(none.size() * sx);
        std::vector<int32_t> ij(none.size() * 2);
        for (size_t i = 0; i < none.size(); ++i)
            std::memcpy(&Rs[i * sx], &verts[(size_t)E[none[i]] * sx], 8 * sx);
        {
            const double t0 = now();
            int rc = sol.split(sol.user, (int64_t)none.size(), Rs.data(), S1.data(), S2.data(), ij.data());
            st.seconds_solvers += now() - t0;
            if (rc) raise(rc, "%s", dev ? g_err : "the caller's bisection failed");
        }
        std::vector<double> mids(none.size() * p);
        for (size_t i = 0; i < none.size(); ++i)
            std::memcpy(&mids[i * p], &S1[i * sx + (size_t)ij[2 * i] * p], 8 * (size_t)p);
        std::vector<int64_t> mid_id(none.size()), a_id(none.size()), b_id(none.size());
        chk_search(ehm_search_point_ids(S, (int64_t)none.size(), mids.data(), mid_id.data()),
                   "ehm_search_point_ids");
        for (size_t i = 0; i < none.size(); ++i) {
            a_id[i] = pids[(size_t)E[none[i]] * nv + ij[2 * i]];
            b_id[i] = pids[(size_t)E[none[i]] * nv + ij[2 * i + 1]];
        }
        chk_search(ehm_search_register_midpoints(S, (int64_t)none.size(), mid_id.data(), a_id.data(),
                                                 b_id.data()), "ehm_search_register_midpoints");
        for (size_t i = 0; i < none.size(); ++i) {
            const int32_t nd = E[none[i]];
            const int32_t l = new_node(&S1[i * sx], depth[nd] + 1);
            const int32_t r = new_node(&S2[i * sx], depth[nd] + 1);
            left[nd] = l; right[nd] = r;
            flags[nd] = 0;
            for (int v = 0; v < nv; ++v) pids[(size_t)l * nv + v] = pids[(size_t)r * nv + v] = pids[(size_t)nd * nv + v];
            pids[(size_t)l * nv + ij[2 * i]] = mid_id[i];
            pids[(size_t)r * nv + ij[2 * i + 1]] = mid_id[i];
            witness[l] = witness[r] = witness[nd];
            ecc_work.push_back(l);
            ecc_work.push_back(r);
            st.depth = std::max(st.depth, depth[nd] + 1);
        }
    }
}

// bnb_frontier.bar_e_many for cells on their first lcss visit (nothing inherited, no incumbent)
void ehm_frontier::lcss_round(const std::vector<int32_t>& Lc, int launch_target) {
    const size_t n = Lc.size();
    if (!n) return;
    st.calls_bar_e += (int64_t)n;
    st.lcss_visits += (int64_t)n;
    std::vector<double> guard(n);
    for (size_t j = 0; j < n; ++j) {
        double mx = 0.0;
        for (int v = 0; v < nv; ++v) mx = std::max(mx, std::fabs(costs[(size_t)Lc[j] * nv + v]));
        guard[j] = INHERIT_GUARD * (1.0 + mx);
    }
    ehm_search_bare* Bq = nullptr;
    chk_search(ehm_search_bare_create((int32_t)n, n_modes, N, guard.data(), &Bq), "ehm_search_bare_create");
    struct Guard { ehm_search_bare* b; ~Guard() { ehm_search_bare_destroy(b); } } g{Bq};
    const size_t sx = (size_t)nv * p;
    int64_t n_ask = 0, left_n = (int64_t)n;
    std::vector<uint64_t> codes, rep;
    std::vector<int32_t> owner;
    std::vector<int64_t> pp;
    std::vector<int8_t> ver;
    std::vector<uint8_t> known;
    std::vector<double> Rp, Vp, t;
    while (left_n > 0) {
        const int width = (int)std::max<int64_t>(1, std::min<int64_t>(BATCH, launch_target /
                                                                     std::max<int64_t>(1, left_n * n_modes)));
        chk_search(ehm_search_bare_step(Bq, width, &n_ask, &left_n), "ehm_search_bare_step");
        if (!n_ask) break;
        codes.resize((size_t)n_ask); owner.resize((size_t)n_ask);
        chk_search(ehm_search_bare_asks(Bq, codes.data(), owner.data()), "ehm_search_bare_asks");
        // feasible_somewhere_codes: a relaxation known to be feasible at a vertex is feasible on the
        // simplex; where nothing is held the first vertex without a verdict is asked (remembered)
        rep.resize((size_t)n_ask * nv); pp.resize((size_t)n_ask * nv); ver.resize((size_t)n_ask * nv);
        for (int64_t a = 0; a < n_ask; ++a)
            for (int v = 0; v < nv; ++v) {
                rep[(size_t)a * nv + v] = codes[(size_t)a];
                pp[(size_t)a * nv + v] = pids[(size_t)Lc[owner[(size_t)a]] * nv + v];
            }
        chk_search(ehm_search_peek(S, n_ask * nv, rep.data(), pp.data(), ver.data()), "ehm_search_peek");
        known.assign((size_t)n_ask, 0);
        std::vector<uint64_t> qc;
        std::vector<int64_t> qb(1, 0), qp, qa;
        for (int64_t a = 0; a < n_ask; ++a) {
            int first_unknown = -1;
            for (int v = 0; v < nv; ++v) {
                const int8_t r = ver[(size_t)a * nv + v];
                if (r == 1) known[(size_t)a] = 1;
                if (r == -1 && first_unknown < 0) first_unknown = v;
            }
            if (!known[(size_t)a] && first_unknown >= 0) {
                qc.push_back(codes[(size_t)a]);
                qp.push_back(pp[(size_t)a * nv + first_unknown]);
                qb.push_back((int64_t)qp.size());
                qa.push_back(a);
            }
        }
        if (!qc.empty()) {
            std::vector<uint8_t> fl;
            feasible_sets(qc, qb, qp, fl);
            for (size_t q = 0; q < qa.size(); ++q) known[(size_t)qa[q]] = fl[q];
        }
        Rp.resize((size_t)n_ask * sx); Vp.resize((size_t)n_ask * nv); t.resize((size_t)n_ask);
        for (int64_t a = 0; a < n_ask; ++a) {
            std::memcpy(&Rp[(size_t)a * sx], &verts[(size_t)Lc[owner[(size_t)a]] * sx], 8 * sx);
            std::memcpy(&Vp[(size_t)a * nv], &costs[(size_t)Lc[owner[(size_t)a]] * nv], 8 * (size_t)nv);
        }
        solver_slack(n_ask, codes.data(), Rp.data(), Vp.data(), known.data(), t.data());
        chk_search(ehm_search_bare_answer(Bq, t.data(), &left_n), "ehm_search_bare_answer");
    }
    std::vector<int8_t> closed(n);
    std::vector<double> margin(n);
    int64_t counts[2] = {0, 0};
    chk_search(ehm_search_bare_result(Bq, closed.data(), margin.data(), counts), "ehm_search_bare_result");
    st.prefixes_expanded += counts[0];
    st.answered_without_a_problem += counts[1];
    for (size_t j = 0; j < n; ++j) {
        const int32_t nd = Lc[j];
        if (closed[j]) {
            flags[nd] = EHM_FR_HAS_RECORD | EHM_FR_CLOSED;
            ++st.regions;
        } else {
            flags[nd] = EHM_FR_HAS_RECORD | EHM_FR_OPEN;
            ++st.open_cells;
        }
    }
}

void ehm_frontier::run(const ehm_frontier_opts& o) {
    const int cap = o.round_cap > 0 ? o.round_cap : 4096;
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
        ecc_round(E);
        lcss_round(Lc, target);
    }
    st.n_nodes = n_nodes();
    st.seconds_total += now() - t0;
    if (dev) {                                  // since the last reset
        st.lp_solves = dev->tab[0].lp + dev->tab[1].lp - base_ctr[0];
        st.launches = dev->launches - base_ctr[1];
        st.blocks_loaded = dev->tab[0].loaded + dev->tab[1].loaded - base_ctr[2];
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
    if (!out || !solvers || !solvers->points || !solvers->slack || !solvers->split || n_x < 1 ||
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

int ehm_frontier_create(const ehm_pwa_law* law, int32_t short_len, int32_t long_slots, int device,
                        double eps_a, double eps_r, ehm_frontier** out) {
    if (!out) return fail(EHM_E_INVALID, "ehm_frontier_create: out is NULL");
    DeviceSolver* D = new (std::nothrow) DeviceSolver();
    if (!D) return fail(EHM_E_CAPACITY, "ehm_frontier_create: out of memory");
    int rc = copy_law(law, D->law);
    if (rc) { delete D; return rc; }
    if (short_len < 0 || short_len >= D->law.N || long_slots < 16 || D->law.N > 60) {
        delete D;
        return fail(EHM_E_INVALID, "ehm_frontier_create: short_len %d, long_slots %d", short_len, long_slots);
    }
    D->device = device; D->short_len = short_len;
    D->p = D->law.n_x; D->nv = D->p + 1; D->n_u = D->law.n_u; D->N = D->law.N;
    D->base = (uint64_t)D->law.n_modes + 1;
    if (short_len > 0) {
        int n_short = 0, pwr = 1;
        for (int k = 0; k <= short_len; ++k) { n_short += pwr; pwr *= D->law.n_modes; }
        rc = make_table(D, D->tab[0], short_len, std::max(16, n_short), -1, eps_a, eps_r);
    }
    if (!rc) rc = make_table(D, D->tab[1], D->law.N, long_slots, D->law.N, eps_a, eps_r);
    ehm_frontier* f = rc ? nullptr : new (std::nothrow) ehm_frontier();
    if (!rc && !f) rc = fail(EHM_E_CAPACITY, "ehm_frontier_create: out of memory");
    if (!rc) {
        f->dev = D;
        f->sol.user = D;
        f->sol.points = dev_points; f->sol.slack = dev_slack; f->sol.split = dev_split;
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

int ehm_frontier_tables(ehm_frontier* f, ehm_problem** short_table, ehm_problem** long_table) {
    if (!f) return fail(EHM_E_INVALID, "ehm_frontier_tables: NULL handle");
    if (short_table) *short_table = f->dev ? f->dev->tab[0].P : nullptr;
    if (long_table) *long_table = f->dev ? f->dev->tab[1].P : nullptr;
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
    f->flags.clear(); f->n_roots = 0;
    f->ecc_work.clear(); f->lcss_work.clear();
    f->opt_of.clear(); f->opt_J.clear(); f->opt_u.clear();
    f->st = ehm_frontier_stats{};
    if (f->dev) {
        f->base_ctr[0] = f->dev->tab[0].lp + f->dev->tab[1].lp;
        f->base_ctr[1] = f->dev->launches;
        f->base_ctr[2] = f->dev->tab[0].loaded + f->dev->tab[1].loaded;
        f->base_ctr[3] = f->dev->stalled;
        f->base_ctr[4] = f->dev->slivers;
    }
    return EHM_OK;
}

int ehm_frontier_add_root(ehm_frontier* f, const double* vertices) {
    if (!f || !vertices) return fail(EHM_E_INVALID, "ehm_frontier_add_root: bad argument");
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

int ehm_frontier_run(ehm_frontier* f, const ehm_frontier_opts* opts, ehm_frontier_stats* stats) {
    if (!f) return fail(EHM_E_INVALID, "ehm_frontier_run: NULL handle");
    ehm_frontier_opts o{};
    if (opts) o = *opts;
    try {
        f->run(o);
    } catch (const Fail& e) {
        return fail(e.code, "%s", e.msg.c_str());
    } catch (const std::bad_alloc&) {
        return fail(EHM_E_CAPACITY, "ehm_frontier_run: out of memory");
    }
    if (stats) *stats = f->st;
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
    std::fill(out, out + 2 * per, 0);
    if (f->dev)
        for (int t = 0; t < 2; ++t)
            if (!f->dev->tab[t].counts.empty())
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
