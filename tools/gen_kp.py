"""Derives csrc/ehm_kp.hip (persistent frontier kernel at two solver widths) from the body of
k2_persist in csrc/ehm_k2.hip, so that the two stay one piece of code.  Run after editing
k2_persist:  python tools/gen_kp.py"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'explicit_hybrid_mpc_amd', 'csrc')
src = open(os.path.join(CSRC, 'ehm_k2.hip')).read()
a = src.index("__global__ __launch_bounds__(EHM_K2_THREADS) void k2_persist(")
b = src.index("// ---- vertex solves that seed a node's costs / inputs")
k = src[a:b]


def sub(old, new, count=1):
    global k
    assert k.count(old) >= 1, old
    k = k.replace(old, new) if count == 0 else k.replace(old, new, count)


sub("void k2_persist(", "void kp_persist(")
sub('''    K2_PROLOGUE();
''', '''    double* sm = reinterpret_cast<double*>(k2_smem);
    const int tid = threadIdx.x;
    const int lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    kd::Shared S;           // the two namespaces describe the same LDS image
    kd::carve_shared(S, sm, P);
    ke::Shared Se;
    ke::carve_shared(Se, sm, P);
#if EHM2_PROF
    S.gprof = cnt ? cnt->phase : nullptr;
    Se.gprof = cnt ? cnt->phase : nullptr;
#endif
    kd::NodeBuf nb;
    kd::carve_node(nb, sm + kd::shared_doubles(P) + (size_t)wave * wave_doubles, P.p, P.n_u);
    ke::NodeBuf nbe;
    ke::carve_node(nbe, sm + kd::shared_doubles(P) + (size_t)wave * wave_doubles, P.p, P.n_u);
''')
sub("    load_shared(P, 0, sm, tid, blockDim.x);", "    kd::load_shared(P, 0, sm, tid, blockDim.x);")
sub("wsync();", "kd::wsync();", 0)
sub("pin(", "kd::pin(", 0)
sub('''        Wave W;
        IpmResult r;
        int its = 0;
        for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
            double b[SLOTS];
            const int ln = kd::pin(lane);
            assemble_simplex(S, W, nb, node, node + rec_off_vcost(p), SX_SLACK, P.eps_a, P.eps_r,
                             b, ln, P, 0);
            r = ipm_solve(S, W, b, ln, sign_only != 0, step_fraction(attempt));
            its += r.iters;
            if (r.status == 0) break;
        }
        r.iters = its;''', '''        kd::IpmResult r;
        int its = 0;
        {
            kd::Wave W;
            for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
                double b[kd::SLOTS];
                const int ln = kd::pin(lane);
                kd::assemble_simplex(S, W, nb, node, node + rec_off_vcost(p), SX_SLACK, P.eps_a,
                                     P.eps_r, b, ln, P, 0);
                r = kd::ipm_solve(S, W, b, ln, sign_only != 0, kd::step_fraction(attempt));
                its += r.iters;
                if (r.status == 0) break;
            }
        }
        r.iters = its;''')
sub('''        its = 0;
        for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
            double b[SLOTS];
            const int ln = kd::pin(lane);
            assemble_point(S, W, nb.lp, mid, false, b, ln, P, 0);
            r = ipm_solve(S, W, b, ln, false, step_fraction(attempt), T.grad ? nb.F : nullptr);
            its += r.iters;
            if (r.status == 0) break;
        }
        r.iters = its;''', '''        // the midpoint LP has p + 1 columns less: it runs in the narrower instance
        ke::Wave W;
        ke::IpmResult re;
        its = 0;
        for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
            double b[ke::SLOTS];
            const int ln = kd::pin(lane);
            ke::assemble_point(Se, W, nbe.lp, mid, false, b, ln, P, 0);
            re = ke::ipm_solve(Se, W, b, ln, false, ke::step_fraction(attempt),
                               T.grad ? nb.F : nullptr);
            its += re.iters;
            if (re.status == 0) break;
        }
        r.iters = its;
        r.status = re.status;
        r.obj = re.obj;''')
sub("        const double* xmid = W.xb;\n#endif", "        const double* xmid = W.xb;\n#endif")

# --- the midpoint-first flow (EHM_PERSIST_MIDFIRST objects) -------------------------------------
sub('''            Wave Wm;
            IpmResult rm;
            for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
                double b[SLOTS];
                const int ln = kd::pin(lane);
                assemble_point(S, Wm, nb.lp, mid, false, b, ln, P, 0);
                rm = ipm_solve(S, Wm, b, ln, false, step_fraction(attempt), T.grad ? nb.F : nullptr);''',
    '''            ke::Wave Wm;
            ke::IpmResult rm;
            for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
                double b[ke::SLOTS];
                const int ln = kd::pin(lane);
                ke::assemble_point(Se, Wm, nbe.lp, mid, false, b, ln, P, 0);
                rm = ke::ipm_solve(Se, Wm, b, ln, false, ke::step_fraction(attempt),
                                   T.grad ? nb.F : nullptr);''')
sub('''            Wave W;
            IpmResult r;
            its = 0;
            for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
                double b[SLOTS];
                const int ln = kd::pin(lane);
                assemble_simplex(S, W, nb, node, node + rec_off_vcost(p), SX_SLACK, P.eps_a, P.eps_r,
                                 b, ln, P, 0);
                r = ipm_solve(S, W, b, ln, sign_only != 0, step_fraction(attempt));''',
    '''            kd::Wave W;
            kd::IpmResult r;
            its = 0;
            for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
                double b[kd::SLOTS];
                const int ln = kd::pin(lane);
                kd::assemble_simplex(S, W, nb, node, node + rec_off_vcost(p), SX_SLACK, P.eps_a,
                                     P.eps_r, b, ln, P, 0);
                r = kd::ipm_solve(S, W, b, ln, sign_only != 0, kd::step_fraction(attempt));''')

HEAD = '''// GENERATED by tools/gen_kp.py from k2_persist in ehm_k2.hip -- edit there, then regenerate.
//
// Persistent frontier kernel at two solver widths (gfx950).  Same kernel as k2_persist in
// ehm_k2.hip (see there and DESIGN.md section 4 for the queue protocol), but the suboptimality-
// test LPs (n + p + 1 columns) and the midpoint LPs (n columns) each run in the instance of
// ehm_ipm2.h that fits them: a row of the normal matrix lives in registers, so the column
// capacity is a compile-time size and every elimination step costs that many FMAs per lane.
// Compiled per (EHM_NPD, EHM_NPE, EHM_SLOTS); ehm_capi.hip uses it when a matching pair exists.
#include <hip/hip_runtime.h>

#ifndef EHM_NPD
#error "EHM_NPD / EHM_NPE (column capacities of the two LP kinds) must be defined"
#endif

#define EHM_NP EHM_NPD
#include "ehm_k2_asm.h"
namespace kd = EHM2_NS;
#undef EHM_NP
#define EHM_NP EHM_NPE
#include "ehm_k2_asm.h"
namespace ke = EHM2_NS;
#undef EHM_NP

using namespace ehm;

// one named namespace per instance (kernels of different objects must not share a symbol)
#define KP_CAT2(a, b, c, d) a##b##_##c##_##d
#define KP_CAT(a, b, c, d) KP_CAT2(a, b, c, d)
#ifndef EHM_PERSIST_MIDFIRST
#define EHM_PERSIST_MIDFIRST 0
#endif
#if EHM2_QUAD
#define KP_NS KP_CAT(ehm_kpq_, EHM_NPD, EHM_NPE, EHM_SLOTS)
#elif EHM_PERSIST_MIDFIRST
#define KP_NS KP_CAT(ehm_kpm_, EHM_NPD, EHM_NPE, EHM_SLOTS)
#else
#define KP_NS KP_CAT(ehm_kp_, EHM_NPD, EHM_NPE, EHM_SLOTS)
#endif

namespace KP_NS {

#define EHM_PERSIST_WATCHDOG_TICKS (60LL * 100000000LL)    // 60 s of the 100 MHz wall clock
constexpr int SLOTS = EHM_SLOTS;

'''
TAIL = '''
hipError_t set_lds(int bytes) {
    return hipFuncSetAttribute((const void*)kp_persist,
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
size_t wave_doubles_for(const DevProblem& P, int n_lp_d, int ne_d, int n_lp_e) {
    const int nE = P.n - P.nd0;
    const size_t a = kd::wave_lp_doubles(n_lp_d, ne_d, nE, P.p), b = ke::wave_lp_doubles(n_lp_e, 0, nE, P.p);
    // (+ k2_stash_doubles: the midpoint-first flow parks the midpoint solve's input and gradient and the
    // node's witness there)
    return k2_node_doubles(P.p, P.n_u) + (a > b ? a : b) + (EHM_PERSIST_MIDFIRST ? k2_stash_doubles(P.p, P.n_u) : 0);
}
size_t shared_doubles_for(const DevProblem& P) { return kd::shared_doubles(P); }
void l_persist(const K2Launch& L, DevProblem P, DevTree T, int32_t* slots, int n_slots,
               PersistCtl* ctl, int node_cap, DevCounters* cnt, int sign_only, int max_depth,
               PersistDeal deal) {
    P.wc_lds = L.wc_lds;
    hipLaunchKernelGGL(kp_persist, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream, P, T,
                       slots, n_slots, ctl, node_cap, cnt, L.wave_doubles, sign_only, max_depth,
                       deal);
}

const KpApi g_api = {EHM_NPD, EHM_NPE, EHM_SLOTS, EHM_K2_THREADS, set_lds, wave_doubles_for,
                     shared_doubles_for, l_persist};

}  // namespace KP_NS

#if EHM2_QUAD
extern "C" const ehm::KpApi* KP_CAT(ehm_kpq_api_, EHM_NPD, EHM_NPE, EHM_SLOTS)() {
    return &KP_NS::g_api;
}
#elif EHM_PERSIST_MIDFIRST
extern "C" const ehm::KpApi* KP_CAT(ehm_kpm_api_, EHM_NPD, EHM_NPE, EHM_SLOTS)() {
    return &KP_NS::g_api;
}
#else
extern "C" const ehm::KpApi* KP_CAT(ehm_kp_api_, EHM_NPD, EHM_NPE, EHM_SLOTS)() {
    return &KP_NS::g_api;
}
#endif
'''
open(os.path.join(CSRC, 'ehm_kp.hip'), 'w').write(HEAD + k + TAIL)
print('ehm_kp.hip regenerated (%d lines)' % (HEAD + k + TAIL).count('\n'))
