// HIP kernels and the extern-"C" shim of the batched iLQG solver (gfx950).
//
// One translation unit per problem: the generated problem file iLQG_func.c is
// #included below, UNMODIFIED, inside a region that marks every function as a
// device function, so the kernels call the very callbacks the reference's
// solver calls on the host (ddpf, ddpL, ddpF, clampU, limitsU, bp_derivsL,
// bp_derivsF, calc*Aux*, init_running/init_final — reference
// iLQG_func.tem:40-347).  A Maxima-generated problem file drops in the same way.
//
// Mapping ("lane mapping"): one lane = one trajectory, 64 trajectories per
// wavefront.  Device arrays are [time step][field][trajectory] so that a
// wavefront reads/writes 512 contiguous bytes per field.  All small matrices
// of a trajectory live in that lane's VGPRs (ilqg_device.hpp).
//
// Kernels                             replaces (reference)
//   k_derivs     lane = (traj, step)   calc_derivs            iLQG_func.tem:187-221
//   k_backward   lane = traj           back_pass + retry loop back_pass.c:38-257, iLQG.c:261-303
//   k_rollout    lane = (traj, alpha)  forward_pass           iLQG_func.tem:121-185
//   k_select     lane = traj           line_search selection  line_search.c:37-75
//   k_update     lane = traj           accept / reject        iLQG.c:311-361
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "mex.h"

// NaN/Inf guards in generated code still `return 0`; their printing is dropped on the device
#define PRNT(...) ((void)0)

// The generated callbacks call sin(x) and cos(x) of the same few arguments many times, spread
// over several functions (calcXUVariableAux, ddpf, bp_derivsL, ...).  The device math
// library's sin/cos contain branches (huge-argument reduction), so once the callbacks are
// inlined into a kernel every call is a separate ~130-instruction body the optimiser cannot
// merge; and a non-inlined helper would stall on the function-call ABI's `s_waitcnt vmcnt(0)`.
// ilqg_sincos below is STRAIGHT-LINE code for |x| < 8e5 (anything larger, NaN and Inf go to
// the library through a rarely taken branch), so value numbering merges all evaluations of
// the same argument: one argument reduction + one sine and one cosine polynomial per distinct
// argument and loop iteration.
//
// Algorithm: Cody-Waite reduction with pi/2 split into three 33-bit pieces, always carried to
// the third piece (the medium-size path of fdlibm's e_rem_pio2.c), then the minimax kernels of
// fdlibm / FreeBSD msun k_sin.c and k_cos.c on [-pi/4, pi/4] with the reduction tail.  Error
// below 1 ulp, the same class as the host libm and the device library (tests/test_gpu_parity.py
// checks it against numpy).
#ifndef ILQG_NO_SHARED_SINCOS
struct ilqg_sc { double s, c; };

__device__ __attribute__((noinline)) static ilqg_sc ilqg_sincos_slow(double x) {
    ilqg_sc r;
    sincos(x, &r.s, &r.c);
    return r;
}

__device__ __forceinline__ static ilqg_sc ilqg_sincos(double x) {
    const double fn = rint(x * 6.36619772367581382433e-01);
    // x - fn*(P1 + P2 + P3 + P3t) as y0 + y1; P1, P2, P3 have 33 significant bits each, so the
    // products fn*Pi are exact for |fn| < 2^20; e1, e2 are the rounding errors of the two subtractions
    const double a = x - fn * 1.57079632673412561417e+00;
    const double b = fn * 6.07710050630396597660e-11;
    const double r1 = a - b;
    const double e1 = (a - r1) - b;
    const double c3 = fn * 2.02226624871116645580e-21;
    const double r2 = r1 - c3;
    const double e2 = (r1 - r2) - c3;
    const double w = (fn * 8.47842766036889956997e-32 - e2) - e1;
    const double y0 = r2 - w;
    const double y1 = (r2 - y0) - w;

    const double z = y0 * y0;
    const double zz = z * z;
    // sine kernel with tail
    const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * 2.75573137070700676789e-06) +
                      z * zz * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10);
    const double v = z * y0;
    const double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * -1.66666666666666324348e-01);
    // cosine kernel with tail
    const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * 2.48015872894767294178e-05)) +
                      (zz * zz) * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11));
    const double hz = 0.5 * z;
    const double wc = 1.0 - hz;
    const double kc = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));

    const int q = ((int)fn) & 3;
    ilqg_sc out;
    out.s = (q & 1) ? kc : ks;
    out.c = (q & 1) ? ks : kc;
    if(q == 1 || q == 2) out.c = -out.c;
    if(q >= 2) out.s = -out.s;
    if(!(fabs(x) < 8.0e5)) out = ilqg_sincos_slow(x);  // huge arguments, NaN, Inf: device library
    return out;
}
#if defined(ILQG_SINCOS_CALL)
// Large generated files (thousands of sin/cos call sites, e.g. the tensors of an n = 16 problem): keep
// every evaluation a CALL to a side-effect-free function.  Calls with equal arguments are merged before
// anything is inlined, which also keeps the compile time bounded.
__device__ __attribute__((noinline, const)) static ilqg_sc ilqg_sincos_call(double x) { return ilqg_sincos(x); }
#define sin(x) (ilqg_sincos_call(x).s)
#define cos(x) (ilqg_sincos_call(x).c)
#else
#define sin(x) (ilqg_sincos(x).s)
#define cos(x) (ilqg_sincos(x).c)
#endif
#endif

extern "C" {
#pragma clang attribute push(__attribute__((device)), apply_to = function)
#pragma clang attribute push(__attribute__((internal_linkage)), apply_to = variable(is_global))
#include "iLQG.h"
#include "matMult.h"
#include "iLQG_func.c"
#pragma clang attribute pop
#pragma clang attribute pop
}
#undef sin
#undef cos

// Mapping: lane mapping (one lane per trajectory, everything in registers) for small problems,
// wave mapping (one wavefront per trajectory, matrices in LDS) when a lane's registers cannot
// hold the matrices.  -DILQG_WAVE_MAP=1 forces the wave mapping for a small problem.
#ifndef ILQG_WAVE_MAP
#define ILQG_WAVE_MAP (N_X > 8)
#endif

#include "ilqg_device.hpp"
#include "ilqg_wave.hpp"
#include "ilqg_param_layout.h"  // generated at build time from the problem's paramdesc[]
#include "ilqg_shim.h"

namespace {

using namespace ilqg;

constexpr int NX = N_X, NU = N_U;
constexpr bool FULL = FULL_DDP != 0;
#ifdef ILQG_STATE_DEPENDENT_LIMITS
constexpr bool HX = ILQG_STATE_DEPENDENT_LIMITS != 0;
#else
constexpr bool HX = true;  // Maxima-generated header: assume the general case
#endif
using RL = RecLayout<NX, NU, FULL, HX>;
constexpr int SXX = RL::SXX, SUU = RL::SUU, NXU = RL::NXU, REC = RL::SIZE, REC_HOST = RL::HOST_SIZE;
constexpr int FIN = NX + SXX;
constexpr int WAVE = 64;
constexpr bool WAVE_MAP = ILQG_WAVE_MAP;

thread_local std::string g_err;

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if(e_ != hipSuccess) {                                                              \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                      \
            return 1;                                                                       \
        }                                                                                   \
    } while(0)

struct DevPtrs {
    double *f[ILQG_F_COUNT];
    int *i[ILQG_I_COUNT];
    int *derivs_failed;
    int *pending;        // trajectories that go to the second line-search stage
    int *n_pending;      // their count (read by the second stage)
    int *n_pending_next; // counter the first-stage selection appends with (same word as n_pending)
    trajEl_t *work;      // wave mapping: derivative records of one chunk of trajectories, [chunk][N]
    double **p;
    int B, Bp, N;
};

// element (step k, component i of W) of trajectory b in a per-step field of `steps` steps:
// lane mapping [k][i][b] (batch-innermost), wave mapping [b][k][i] (trajectory-major)
__device__ __forceinline__ size_t ix(const DevPtrs &P, int W, int steps, int k, int i, int b) {
    return WAVE_MAP ? ((size_t)b * steps + k) * W + i : ((size_t)k * W + i) * (size_t)P.Bp + b;
}

// Per-lane snapshot of the problem parameters.  The generated callbacks read parameters as
// p[i][j] through a `double **`; read from global memory, every such value would have to be
// re-loaded after each store of the kernel (the compiler cannot prove that the parameter
// arrays do not alias the output arrays), which costs two dependent memory round trips per
// use.  Fixed-size parameters are therefore copied once into a private array that the
// optimiser keeps in registers; per-time-step parameters (size -1) stay in global memory.
struct ParamValues {
    double v[ILQG_PTOTAL];
};
struct ParamTable {
    double *ptr[ILQG_NP > 0 ? ILQG_NP : 1];
};
__device__ __forceinline__ void load_params(ParamValues &V, ParamTable &T, double **p) {
    constexpr int sizes[ILQG_NP > 0 ? ILQG_NP : 1] = ILQG_PSIZES;
    constexpr int offs[ILQG_NP > 0 ? ILQG_NP : 1] = ILQG_POFFSETS;
#pragma unroll
    for(int i = 0; i < ILQG_NP; i++) {
        double *src = p[i];
        if(sizes[i] > 0) {
#pragma unroll
            for(int j = 0; j < sizes[i]; j++) V.v[offs[i] + j] = src[j];
            T.ptr[i] = &V.v[offs[i]];
        } else {
            T.ptr[i] = src;
        }
    }
}

__device__ __forceinline__ void make_optset(tOptSet &o, const DevPtrs &P, const ilqg_dev_opts_t &O,
                                            ParamValues &V, ParamTable &T) {
    load_params(V, T, P.p);
    o.p = T.ptr;
    o.n_hor = P.N;
    o.w_pen_l = O.w_pen_init_l;
    o.w_pen_f = O.w_pen_init_f;
    o.tolConstraint = O.tolConstraint;
    o.w_pen_fact1 = O.w_pen_fact1;
    o.w_pen_fact2 = O.w_pen_fact2;
    o.w_pen_max_l = O.w_pen_max_l;
    o.w_pen_max_f = O.w_pen_max_f;
}

// ---------------------------------------------------------------------------
// host layout [b][k][f]  <->  device layout [k][f][b]
// ---------------------------------------------------------------------------
__global__ void k_to_soa(const double *__restrict__ aos, double *__restrict__ soa, int B, int Bp, int steps, int wh,
                         int wd) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * steps * wd;
    if(i >= total) return;
    const int fcol = (int)(i % wd);
    const int k = (int)((i / wd) % steps);
    const int b = (int)(i / ((size_t)wd * steps));
    soa[((size_t)k * wd + fcol) * Bp + b] = aos[((size_t)b * steps + k) * wh + fcol];
}

__global__ void k_to_aos(const double *__restrict__ soa, double *__restrict__ aos, int B, int Bp, int steps, int wh,
                         int wd) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * steps * wh;
    if(i >= total) return;
    const int fcol = (int)(i % wh);
    const int k = (int)((i / wh) % steps);
    const int b = (int)(i / ((size_t)wh * steps));
    aos[i] = (fcol < wd) ? soa[((size_t)k * wd + fcol) * Bp + b] : 0.0;
}

#if !ILQG_WAVE_MAP
// ---------------------------------------------------------------------------
// calc_derivs: one lane per (trajectory, time step); step N is the final record
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_derivs(DevPtrs P, ilqg_dev_opts_t O) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = (int)(tid % P.Bp);
    const int k = (int)(tid / P.Bp);
    if(k > P.N || b >= P.B) return;
    if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE || !P.i[ILQG_I_NEED_DERIVS][b]) return;
    const size_t Bp = P.Bp;

    tOptSet o;
    ParamValues pval;
    ParamTable ptab;
    make_optset(o, P, O, pval, ptab);
    int ok = 1;
    if(k < P.N) {
        tOptSet o1 = o;
        o1.n_hor = 1;  // init_running loops over n_hor elements: write this element's constants only
        trajEl_t t;
        init_running(&t, &o1);
#pragma unroll
        for(int i = 0; i < NX; i++) t.x[i] = P.f[ILQG_F_X][((size_t)k * NX + i) * Bp + b];
#pragma unroll
        for(int i = 0; i < NU; i++) t.u[i] = P.f[ILQG_F_U][((size_t)k * NU + i) * Bp + b];
        // auxiliaries are not kept in HBM: recompute them from the stored (x,u) exactly as
        // forward_pass did (iLQG_func.tem:160-164), then the reference's calc_derivs body
        ok &= calcXVariableAux(&t, nullptr, k, &o);
        ok &= calcXUVariableAux(&t, nullptr, k, &o);
        ok &= calcLAuxDeriv(&t, nullptr, k, &o);
        ok &= bp_derivsL(&t, k, o.p);
        limitsU(&t, k, o.p, P.N);

        double *out = P.f[ILQG_F_DER] + (size_t)k * REC * Bp + b;
#define PUT(off, arr, cnt) \
    _Pragma("unroll") for(int i = 0; i < (cnt); i++) out[(size_t)((off) + i) * Bp] = (arr)[i];
        PUT(RL::CX, t.cx, NX)
        PUT(RL::CXX, t.cxx, SXX)
        PUT(RL::CU, t.cu, NU)
        PUT(RL::CUU, t.cuu, SUU)
        PUT(RL::CXU, t.cxu, NXU)
        PUT(RL::FX, t.fx, NX * NX)
        PUT(RL::FU, t.fu, NXU)
        PUT(RL::LOWER, t.lower, NU)
        PUT(RL::UPPER, t.upper, NU)
#if FULL_DDP
        PUT(RL::FXX, t.fxx, NX * SXX)
        PUT(RL::FUU, t.fuu, NX * SUU)
        PUT(RL::FXU, t.fxu, NX * NXU)
#endif
        if(HX) {
            PUT(RL::LSIGN, t.lower_sign, NU)
            PUT(RL::USIGN, t.upper_sign, NU)
            PUT(RL::LHX, t.lower_hx, NXU)
            PUT(RL::UHX, t.upper_hx, NXU)
        }
    } else {
        trajFin_t fin;
        init_final(&fin, &o);
#pragma unroll
        for(int i = 0; i < NX; i++) fin.x[i] = P.f[ILQG_F_X][((size_t)P.N * NX + i) * Bp + b];
        ok &= calcFVariableAux(&fin, nullptr, &o);
        ok &= calcFAuxDeriv(&fin, nullptr, &o);
        ok &= bp_derivsF(&fin, P.N, o.p);
        double *out = P.f[ILQG_F_FIN] + b;
        PUT(0, fin.cx, NX)
        PUT(NX, fin.cxx, SXX)
#undef PUT
    }
    if(!ok) P.derivs_failed[b] = 1;
}

// ---------------------------------------------------------------------------
// back_pass: one lane per trajectory, sequential in time, next record prefetched
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load_record(double *dst, double *udst, const DevPtrs &P, int k, int b) {
    const size_t Bp = P.Bp;
    const double *src = P.f[ILQG_F_DER] + (size_t)k * REC * Bp + b;
#pragma unroll
    for(int i = 0; i < REC; i++) dst[i] = src[(size_t)i * Bp];
    const double *us = P.f[ILQG_F_U] + (size_t)k * NU * Bp + b;
#pragma unroll
    for(int i = 0; i < NU; i++) udst[i] = us[(size_t)i * Bp];
}

// Gains of one step, kept in registers for one more loop iteration: stores are issued at the
// TOP of the next iteration, ahead of that iteration's prefetch loads.  The memory counter
// (vmcnt) retires loads and stores in issue order, and the compiler waits with vmcnt(0) at
// the loop head; with this order everything outstanding at that wait is a full step old.
struct PendingGains {
    double l[NU], K[NXU];
    int k;
    bool valid;
};

__device__ __forceinline__ void flush_gains(const DevPtrs &P, int b, const PendingGains &g) {
    if(!g.valid) return;
    const size_t Bp = P.Bp;
    double *lo = P.f[ILQG_F_LG] + (size_t)g.k * NU * Bp + b;
#pragma unroll
    for(int i = 0; i < NU; i++) lo[(size_t)i * Bp] = g.l[i];
    double *ko = P.f[ILQG_F_KG] + (size_t)g.k * NXU * Bp + b;
#pragma unroll
    for(int i = 0; i < NXU; i++) ko[(size_t)i * Bp] = g.K[i];
}

__device__ __forceinline__ void hold_gains(PendingGains &g, const double *l, const double *K, int k) {
#pragma unroll
    for(int i = 0; i < NU; i++) g.l[i] = l[i];
#pragma unroll
    for(int i = 0; i < NXU; i++) g.K[i] = K[i];
    g.k = k;
    g.valid = true;
}

// one sweep k = N-1..0; returns 0 ok, 1 box-QP failed (back_pass.c:168-171)
__device__ __forceinline__ int backward_sweep(const DevPtrs &P, int b, double lambda, int regType, double &dV0,
                                              double &dV1, double &g_norm) {
    const size_t Bp = P.Bp;
    const int N = P.N;
    double Vx[NX], Vxx[SXX], l[NU], K[NXU];
#pragma unroll
    for(int i = 0; i < NX; i++) Vx[i] = P.f[ILQG_F_FIN][(size_t)i * Bp + b];
#pragma unroll
    for(int i = 0; i < SXX; i++) Vxx[i] = P.f[ILQG_F_FIN][(size_t)(NX + i) * Bp + b];
#pragma unroll
    for(int i = 0; i < NU; i++) l[i] = 0.0;  // warm start of the last step (back_pass.c:163-164)
    dV0 = 0.0;
    dV1 = 0.0;
    double gsum = 0.0;

    double cur[REC], ucur[NU];
    load_record(cur, ucur, P, N - 1, b);
    int failed = 0;
    PendingGains pend;
    pend.valid = false;
    for(int k = N - 1; k >= 0; k--) {
        flush_gains(P, b, pend);
        pend.valid = false;
        double nxt[REC], unxt[NU];
        if(k > 0) load_record(nxt, unxt, P, k - 1, b);  // in flight while this step computes
        // l still holds the solution of step k+1: the warm start (back_pass.c:165-166)
        const int rc = back_step<NX, NU, FULL, HX>(cur, ucur, Vx, Vxx, l, K, lambda, regType, dV0, dV1, gsum);
        if(rc < 1) {
            failed = 1;
            break;
        }
        hold_gains(pend, l, K, k);
#pragma unroll
        for(int i = 0; i < REC; i++) cur[i] = nxt[i];
#pragma unroll
        for(int i = 0; i < NU; i++) ucur[i] = unxt[i];
    }
    flush_gains(P, b, pend);
    if(!failed) g_norm = gsum / ((double)(N - 1));  // N summands over N-1 (back_pass.c:254)
    return failed;
}

// The same sweep with the derivative record of each step evaluated on the fly from the stored
// (x_k, u_k) by the generated callbacks instead of being read from HBM: per step 6 doubles are
// read and 10 written, instead of 57 + 10 (and k_derivs' 61 are not moved at all).  The values
// are the ones k_derivs would have stored (same callbacks, same inputs).
// Returns 0 ok, 1 box-QP failed, 2 NaN/Inf in the derivatives (iLQG.c:247-249).
__device__ __forceinline__ int backward_sweep_fused(const DevPtrs &P, const ilqg_dev_opts_t &O, int b, double lambda,
                                                    double &dV0, double &dV1, double &g_norm) {
    const size_t Bp = P.Bp;
    const int N = P.N;
    tOptSet o;
    ParamValues pval;
    ParamTable ptab;
    make_optset(o, P, O, pval, ptab);
    tOptSet o1 = o;
    o1.n_hor = 1;

    double Vx[NX], Vxx[SXX], l[NU], K[NXU];
    {
        trajFin_t fin;
        init_final(&fin, &o);
#pragma unroll
        for(int i = 0; i < NX; i++) fin.x[i] = P.f[ILQG_F_X][((size_t)N * NX + i) * Bp + b];
        int ok = calcFVariableAux(&fin, nullptr, &o);
        ok &= calcFAuxDeriv(&fin, nullptr, &o);
        ok &= bp_derivsF(&fin, N, o.p);
        if(!ok) return 2;
#pragma unroll
        for(int i = 0; i < NX; i++) Vx[i] = fin.cx[i];
#pragma unroll
        for(int i = 0; i < SXX; i++) Vxx[i] = fin.cxx[i];
    }
#pragma unroll
    for(int i = 0; i < NU; i++) l[i] = 0.0;
    dV0 = 0.0;
    dV1 = 0.0;
    double gsum = 0.0;

    trajEl_t t;
    init_running(&t, &o1);  // constant entries of the record (iLQG_func.tem:312-347)
    double xk[NX], uk[NU];
#pragma unroll
    for(int i = 0; i < NX; i++) xk[i] = P.f[ILQG_F_X][((size_t)(N - 1) * NX + i) * Bp + b];
#pragma unroll
    for(int i = 0; i < NU; i++) uk[i] = P.f[ILQG_F_U][((size_t)(N - 1) * NU + i) * Bp + b];
    int result = 0;
    PendingGains pend;
    pend.valid = false;
    for(int k = N - 1; k >= 0; k--) {
        flush_gains(P, b, pend);
        pend.valid = false;
        double xn[NX], un[NU];
        if(k > 0) {
#pragma unroll
            for(int i = 0; i < NX; i++) xn[i] = P.f[ILQG_F_X][((size_t)(k - 1) * NX + i) * Bp + b];
#pragma unroll
            for(int i = 0; i < NU; i++) un[i] = P.f[ILQG_F_U][((size_t)(k - 1) * NU + i) * Bp + b];
        }
#pragma unroll
        for(int i = 0; i < NX; i++) t.x[i] = xk[i];
#pragma unroll
        for(int i = 0; i < NU; i++) t.u[i] = uk[i];
        int ok = calcXVariableAux(&t, nullptr, k, &o);
        ok &= calcXUVariableAux(&t, nullptr, k, &o);
        ok &= calcLAuxDeriv(&t, nullptr, k, &o);
        ok &= bp_derivsL(&t, k, o.p);
        limitsU(&t, k, o.p, N);
        if(!ok) {
            result = 2;
            break;
        }
        double cur[REC];
#define GETF(off, arr, cnt) _Pragma("unroll") for(int i = 0; i < (cnt); i++) cur[(off) + i] = (arr)[i];
        GETF(RL::CX, t.cx, NX)
        GETF(RL::CXX, t.cxx, SXX)
        GETF(RL::CU, t.cu, NU)
        GETF(RL::CUU, t.cuu, SUU)
        GETF(RL::CXU, t.cxu, NXU)
        GETF(RL::FX, t.fx, NX * NX)
        GETF(RL::FU, t.fu, NXU)
        GETF(RL::LOWER, t.lower, NU)
        GETF(RL::UPPER, t.upper, NU)
#if FULL_DDP
        GETF(RL::FXX, t.fxx, NX * SXX)
        GETF(RL::FUU, t.fuu, NX * SUU)
        GETF(RL::FXU, t.fxu, NX * NXU)
#endif
        if(HX) {
            GETF(RL::LSIGN, t.lower_sign, NU)
            GETF(RL::USIGN, t.upper_sign, NU)
            GETF(RL::LHX, t.lower_hx, NXU)
            GETF(RL::UHX, t.upper_hx, NXU)
        }
#undef GETF
        const int rc = back_step<NX, NU, FULL, HX>(cur, uk, Vx, Vxx, l, K, lambda, O.regType, dV0, dV1, gsum);
        if(rc < 1) {
            result = 1;
            break;
        }
        hold_gains(pend, l, K, k);
#pragma unroll
        for(int i = 0; i < NX; i++) xk[i] = xn[i];
#pragma unroll
        for(int i = 0; i < NU; i++) uk[i] = un[i];
    }
    flush_gains(P, b, pend);
    if(!result) g_norm = gsum / ((double)(N - 1));
    return result;
}

// mode: 0 = records from HBM, lambda retry loop and gradient test (iLQG.c:261-303)
//       1 = records from HBM, ONE sweep (the drop-in back_pass(): the caller owns the retry loop)
//       2 = as 0 with the derivatives evaluated on the fly (k_derivs is not needed)
template <int mode>
__global__ __launch_bounds__(WAVE, 1) void k_backward(DevPtrs P, ilqg_dev_opts_t O) {
    const int b = blockIdx.x * WAVE + threadIdx.x;
    if(b >= P.B) return;
    if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE) return;
    const int single_sweep = (mode == 1);
    P.i[ILQG_I_NEED_DERIVS][b] = 0;
    if(P.derivs_failed[b]) {  // iLQG.c:247-249
        P.i[ILQG_I_STATUS][b] = ILQG_ST_DERIVS_FAILED;
        return;
    }
    double lambda = P.f[ILQG_F_LAMBDA][b], dlambda = P.f[ILQG_F_DLAMBDA][b];
    double dV0 = 0.0, dV1 = 0.0, g_norm = P.f[ILQG_F_GNORM][b];
    int calls = 0, rc;
    for(;;) {
        if(mode == 2)
            rc = backward_sweep_fused(P, O, b, lambda, dV0, dV1, g_norm);
        else
            rc = backward_sweep(P, b, lambda, O.regType, dV0, dV1, g_norm);
        calls++;
        if(single_sweep || rc != 1) break;
        // raise the regularisation and retry (iLQG.c:271-274)
        const double t1 = dlambda * O.lambdaFactor;
        dlambda = (t1 > O.lambdaFactor) ? t1 : O.lambdaFactor;
        const double t2 = lambda * dlambda;
        lambda = (t2 > O.lambdaMin) ? t2 : O.lambdaMin;
        if(lambda > O.lambdaMax) break;
    }
    if(!single_sweep) {
        if(rc == 2) {
            P.i[ILQG_I_STATUS][b] = ILQG_ST_DERIVS_FAILED;
        } else if(rc) {
            P.i[ILQG_I_STATUS][b] = ILQG_ST_NO_DESCENT;
        } else if(g_norm < O.tolGrad && lambda < 1e-5) {  // iLQG.c:297-303
            const double t1 = dlambda / O.lambdaFactor, t2 = 1.0 / O.lambdaFactor;
            dlambda = (t1 < t2) ? t1 : t2;
            lambda = lambda * dlambda * (lambda > O.lambdaMin);
            P.i[ILQG_I_STATUS][b] = ILQG_ST_CONVERGED_GRAD;
        }
    }
    P.f[ILQG_F_LAMBDA][b] = lambda;
    P.f[ILQG_F_DLAMBDA][b] = dlambda;
    P.f[ILQG_F_DV0][b] = dV0;
    P.f[ILQG_F_DV1][b] = dV1;
    P.f[ILQG_F_GNORM][b] = g_norm;
    P.i[ILQG_I_BP_CALLS][b] = calls;
    P.i[ILQG_I_BP_RC][b] = rc;
}

#else  // ILQG_WAVE_MAP
// ---------------------------------------------------------------------------
// wave mapping: calc_derivs straight into the device trajEl_t records, one lane per
// (trajectory of the chunk, time step); step N is the final record
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_derivs_wave(DevPtrs P, ilqg_dev_opts_t O, int chunk_first, int chunk_count,
                                                    int init_consts) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int bw = (int)(tid / (P.N + 1));
    const int k = (int)(tid % (P.N + 1));
    const int b = chunk_first + bw;
    if(bw >= chunk_count || b >= P.B) return;
    if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE) return;
    tOptSet o;
    ParamValues pval;
    ParamTable ptab;
    make_optset(o, P, O, pval, ptab);
    int ok = 1;
    if(k < P.N) {
        trajEl_t *t = P.work + (size_t)bw * P.N + k;
        if(init_consts) {  // constant entries, once per buffer (init_opt, iLQG_func.tem:402-415)
            tOptSet o1 = o;
            o1.n_hor = 1;
            init_running(t, &o1);
        }
        for(int i = 0; i < NX; i++) t->x[i] = P.f[ILQG_F_X][ix(P, NX, P.N + 1, k, i, b)];
        for(int i = 0; i < NU; i++) t->u[i] = P.f[ILQG_F_U][ix(P, NU, P.N, k, i, b)];
        ok &= calcXVariableAux(t, nullptr, k, &o);
        ok &= calcXUVariableAux(t, nullptr, k, &o);
        ok &= calcLAuxDeriv(t, nullptr, k, &o);
        ok &= bp_derivsL(t, k, o.p);
        limitsU(t, k, o.p, P.N);
    } else {
        trajFin_t fin;
        init_final(&fin, &o);
        for(int i = 0; i < NX; i++) fin.x[i] = P.f[ILQG_F_X][ix(P, NX, P.N + 1, P.N, i, b)];
        ok &= calcFVariableAux(&fin, nullptr, &o);
        ok &= calcFAuxDeriv(&fin, nullptr, &o);
        ok &= bp_derivsF(&fin, P.N, o.p);
        double *out = P.f[ILQG_F_FIN] + (size_t)b * FIN;
        for(int i = 0; i < NX; i++) out[i] = fin.cx[i];
        for(int i = 0; i < SXX; i++) out[NX + i] = fin.cxx[i];
    }
    if(!ok) P.derivs_failed[b] = 1;
}

// one sweep of one trajectory on one wave; returns 0 ok, 1 box-QP failed (wave-uniform)
__device__ __forceinline__ int backward_sweep_wave(WaveLds<NX, NU> &S, const DevPtrs &P, int b, int bw, double lambda,
                                                   int regType, double &dV0, double &dV1, double &g_norm) {
    const int lane = threadIdx.x & 63;
    const int N = P.N;
    const double *fin = P.f[ILQG_F_FIN] + (size_t)b * FIN;
    for(int i = lane; i < NX; i += 64) S.Vx[i] = fin[i];
    for(int i = lane; i < SXX; i += 64) S.Vxx[i] = fin[NX + i];
    for(int i = lane; i < NU; i += 64) S.l[i] = 0.0;
    __syncthreads();
    dV0 = 0.0;
    dV1 = 0.0;
    double gsum = 0.0;
    for(int k = N - 1; k >= 0; k--) {
        const trajEl_t *t = P.work + (size_t)bw * N + k;
        StepFields<NX, NU> F;
        F.cx = t->cx; F.cxx = t->cxx; F.cu = t->cu; F.cuu = t->cuu; F.cxu = t->cxu;
        F.fx = t->fx; F.fu = t->fu; F.lower = t->lower; F.upper = t->upper;
#if FULL_DDP
        F.fxx = t->fxx; F.fuu = t->fuu; F.fxu = t->fxu;
#else
        F.fxx = F.fuu = F.fxu = nullptr;
#endif
        F.lower_sign = t->lower_sign; F.upper_sign = t->upper_sign;
        F.lower_hx = t->lower_hx; F.upper_hx = t->upper_hx;
        F.u = P.f[ILQG_F_U] + ix(P, NU, N, k, 0, b);
        const int rc = back_step_wave<NX, NU, FULL, HX>(S, F, P.f[ILQG_F_LG] + ix(P, NU, N, k, 0, b),
                                                        P.f[ILQG_F_KG] + ix(P, NXU, N, k, 0, b), lambda, regType, dV0,
                                                        dV1, gsum);
        if(rc < 1) return 1;
    }
    g_norm = gsum / ((double)(N - 1));
    return 0;
}

// back_pass + retry loop, one wavefront (= one block) per trajectory of the chunk.  single_sweep: 1 = the
// drop-in back_pass() (caller owns the retry loop)
__global__ __launch_bounds__(64, 1) void k_backward_wave(DevPtrs P, ilqg_dev_opts_t O, int single_sweep,
                                                         int chunk_first, int chunk_count) {
    __shared__ WaveLds<NX, NU> S;
    const int bw = blockIdx.x;
    const int b = chunk_first + bw;
    const int lane = threadIdx.x & 63;
    if(bw >= chunk_count || b >= P.B) return;
    if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE) return;
    if(P.derivs_failed[b]) {
        if(lane == 0) {
            P.i[ILQG_I_NEED_DERIVS][b] = 0;
            P.i[ILQG_I_STATUS][b] = ILQG_ST_DERIVS_FAILED;
        }
        return;
    }
    double lambda = P.f[ILQG_F_LAMBDA][b], dlambda = P.f[ILQG_F_DLAMBDA][b];
    double dV0 = 0.0, dV1 = 0.0, g_norm = P.f[ILQG_F_GNORM][b];
    int calls = 0, rc, status = ILQG_ST_ACTIVE;
    for(;;) {
        rc = backward_sweep_wave(S, P, b, bw, lambda, O.regType, dV0, dV1, g_norm);
        calls++;
        if(single_sweep || rc != 1) break;
        const double t1 = dlambda * O.lambdaFactor;
        dlambda = (t1 > O.lambdaFactor) ? t1 : O.lambdaFactor;
        const double t2 = lambda * dlambda;
        lambda = (t2 > O.lambdaMin) ? t2 : O.lambdaMin;
        if(lambda > O.lambdaMax) break;
        __syncthreads();
    }
    if(!single_sweep) {
        if(rc) {
            status = ILQG_ST_NO_DESCENT;
        } else if(g_norm < O.tolGrad && lambda < 1e-5) {
            const double t1 = dlambda / O.lambdaFactor, t2 = 1.0 / O.lambdaFactor;
            dlambda = (t1 < t2) ? t1 : t2;
            lambda = lambda * dlambda * (lambda > O.lambdaMin);
            status = ILQG_ST_CONVERGED_GRAD;
        }
    }
    if(lane == 0) {
        P.i[ILQG_I_NEED_DERIVS][b] = 0;
        P.i[ILQG_I_STATUS][b] = status;
        P.f[ILQG_F_LAMBDA][b] = lambda;
        P.f[ILQG_F_DLAMBDA][b] = dlambda;
        P.f[ILQG_F_DV0][b] = dV0;
        P.f[ILQG_F_DV1][b] = dV1;
        P.f[ILQG_F_GNORM][b] = g_norm;
        P.i[ILQG_I_BP_CALLS][b] = calls;
        P.i[ILQG_I_BP_RC][b] = rc;
    }
}
#endif  // ILQG_WAVE_MAP

// ---------------------------------------------------------------------------
// forward_pass: one lane per (trajectory, step size)
// ---------------------------------------------------------------------------
enum { ROLL_INIT = 0, ROLL_SEARCH = 1, ROLL_WINNER = 2, ROLL_COST = 3, ROLL_SEARCH_LIST = 4 };

// nominal data of one step (what forward_pass reads of the nominal trajectory, iLQG_func.tem:145-155)
struct NomStep {
    double x[NX], u[NU], l[NU];
    double K[WAVE_MAP ? 1 : NXU];  // wave mapping: L is too large to prefetch, it is streamed (below)
};

__device__ __forceinline__ void load_nominal(NomStep &s, const DevPtrs &P, int k, int b, bool gains) {
#pragma unroll
    for(int i = 0; i < NX; i++) s.x[i] = P.f[ILQG_F_X][ix(P, NX, P.N + 1, k, i, b)];
#pragma unroll
    for(int i = 0; i < NU; i++) s.u[i] = P.f[ILQG_F_U][ix(P, NU, P.N, k, i, b)];
    if(gains) {
#pragma unroll
        for(int i = 0; i < NU; i++) s.l[i] = P.f[ILQG_F_LG][ix(P, NU, P.N, k, i, b)];
        if(!WAVE_MAP) {
#pragma unroll
            for(int i = 0; i < NXU; i++) s.K[i] = P.f[ILQG_F_KG][ix(P, NXU, P.N, k, i, b)];
        }
    }
}

// `mode` is a run-time argument on purpose: the search passes and the winner pass
// must execute the same machine code so that the re-rolled winner reproduces
// the cost the selection was based on, bit for bit.
//   ROLL_SEARCH       lane = (trajectory blockIdx.x*64+lane, step size a0 + blockIdx.y)
//   ROLL_SEARCH_LIST  as ROLL_SEARCH for the trajectories listed in P.pending (second stage)
//   ROLL_WINNER       lane = trajectory, accepted step size, result stored in place
//   ROLL_INIT         lane = trajectory, alpha = 0 (initial roll-out, iLQG_mex.c:116), stored
//   ROLL_COST         lane = trajectory, cost of the stored trajectory (forward_pass cost_only = 1)
__global__ __launch_bounds__(WAVE) void k_rollout(DevPtrs P, ilqg_dev_opts_t O, int mode, int a0) {
    int b = blockIdx.x * WAVE + threadIdx.x;
    const int ai = a0 + blockIdx.y;
    if(mode == ROLL_SEARCH_LIST) {
        if(b >= *P.n_pending) return;
        b = P.pending[b];
    }
    if(b >= P.B) return;
    const int N = P.N;
    double alpha = 0.0;
    if(mode == ROLL_SEARCH || mode == ROLL_SEARCH_LIST) {
        if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE) return;
        alpha = O.alpha[ai];
    } else if(mode == ROLL_WINNER) {
        if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE || !P.i[ILQG_I_ACCEPTED][b]) return;
        alpha = O.alpha[P.i[ILQG_I_ALPHA_IDX][b] - 1];
    } else if(mode == ROLL_COST) {
        if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE || !P.i[ILQG_I_ACCEPTED][b]) return;
    }
    const bool cost_only = (mode == ROLL_COST);
    const bool store = (mode == ROLL_INIT || mode == ROLL_WINNER);
    const bool gains = !cost_only && alpha != 0.0;

    tOptSet o;
    ParamValues pval;
    ParamTable ptab;
    make_optset(o, P, O, pval, ptab);
    tOptSet o1 = o;
    o1.n_hor = 1;
    trajEl_t ct;
    init_running(&ct, &o1);  // constant auxiliaries of this problem (iLQG_func.tem:312-347)

    double xc[NX];
#pragma unroll
    for(int i = 0; i < NX; i++) xc[i] = P.f[ILQG_F_X][ix(P, NX, N + 1, 0, i, b)];  // x0 (iLQG_func.tem:141-142)
    double csum = 0.0;
    int ok = 1;
    NomStep cur;
    load_nominal(cur, P, 0, b, gains);
    // stores of a step are issued at the top of the next iteration, ahead of its prefetch (see PendingGains)
    double px[NX], pu[NU];
    int pk = -1;
    for(int k = 0; k < N; k++) {
        if(store && pk >= 0) {
#pragma unroll
            for(int i = 0; i < NX; i++) P.f[ILQG_F_X][ix(P, NX, N + 1, pk, i, b)] = px[i];
#pragma unroll
            for(int i = 0; i < NU; i++) P.f[ILQG_F_U][ix(P, NU, N, pk, i, b)] = pu[i];
        }
        NomStep nxt;
        if(k + 1 < N) load_nominal(nxt, P, k + 1, b, gains);  // in flight while this step computes
        if(cost_only) {
#pragma unroll
            for(int i = 0; i < NX; i++) ct.x[i] = cur.x[i];
#pragma unroll
            for(int i = 0; i < NU; i++) ct.u[i] = cur.u[i];
        } else {
#pragma unroll
            for(int i = 0; i < NX; i++) ct.x[i] = xc[i];
            if(alpha) {
                // u = u_nom + alpha*l + L (x - x_nom), state by state (iLQG_func.tem:146-155)
#pragma unroll
                for(int j = 0; j < NU; j++) ct.u[j] = cur.u[j] + cur.l[j] * alpha;
#pragma unroll
                for(int i = 0; i < NX; i++) {
                    const double dx = ct.x[i] - cur.x[i];
                    if(WAVE_MAP) {
                        const double *Kk = P.f[ILQG_F_KG] + ix(P, NXU, N, k, i * NU, b);
#pragma unroll
                        for(int j = 0; j < NU; j++) ct.u[j] += Kk[j] * dx;
                    } else {
#pragma unroll
                        for(int j = 0; j < NU; j++) ct.u[j] += cur.K[j + i * NU] * dx;
                    }
                }
            } else {
#pragma unroll
                for(int j = 0; j < NU; j++) ct.u[j] = cur.u[j];
            }
        }
        if(!calcXVariableAux(&ct, nullptr, k, &o)) { ok = 0; break; }
        if(!cost_only) clampU(ct.u, &ct, k, o.p, N);
        if(!calcXUVariableAux(&ct, nullptr, k, &o)) { ok = 0; break; }
        double xnext[NX];
        if(!cost_only) {
            if(!ddpf(xnext, &ct, k, o.p, N)) { ok = 0; break; }
        }
        if(!ddpL(&ct, k, &o)) { ok = 0; break; }
        csum += ct.c;
        if(store) {
#pragma unroll
            for(int i = 0; i < NX; i++) px[i] = ct.x[i];
#pragma unroll
            for(int i = 0; i < NU; i++) pu[i] = ct.u[i];
            pk = k;
        }
        if(!cost_only) {
#pragma unroll
            for(int i = 0; i < NX; i++) xc[i] = xnext[i];
        }
        cur = nxt;
    }
    if(store && pk >= 0) {
#pragma unroll
        for(int i = 0; i < NX; i++) P.f[ILQG_F_X][ix(P, NX, N + 1, pk, i, b)] = px[i];
#pragma unroll
        for(int i = 0; i < NU; i++) P.f[ILQG_F_U][ix(P, NU, N, pk, i, b)] = pu[i];
    }
    if(ok) {
        trajFin_t cf;
        init_final(&cf, &o);
        if(cost_only) {
#pragma unroll
            for(int i = 0; i < NX; i++) cf.x[i] = P.f[ILQG_F_X][ix(P, NX, N + 1, N, i, b)];
        } else {
#pragma unroll
            for(int i = 0; i < NX; i++) cf.x[i] = xc[i];
        }
        if(!calcFVariableAux(&cf, nullptr, &o)) ok = 0;
        if(ok && !ddpF(&cf, &o)) ok = 0;
        if(ok) {
            csum += cf.c;
            if(store) {
#pragma unroll
                for(int i = 0; i < NX; i++) P.f[ILQG_F_X][ix(P, NX, N + 1, N, i, b)] = cf.x[i];
            }
        }
    }

    if(mode == ROLL_SEARCH || mode == ROLL_SEARCH_LIST) {
        P.f[ILQG_F_ALPHA_COST][(size_t)ai * P.Bp + b] = csum;
        P.i[ILQG_I_ALPHA_OK][(size_t)ai * P.Bp + b] = ok;
    } else if(mode == ROLL_WINNER) {
        P.f[ILQG_F_NEW_COST][b] = csum;
    } else if(mode == ROLL_COST) {
        P.f[ILQG_F_COST][b] = csum;
    } else {
        P.f[ILQG_F_COST][b] = csum;
        if(!ok) P.i[ILQG_I_STATUS][b] = ILQG_ST_INIT_FAILED;
    }
}

// line_search.c:37-75: the FIRST step size (lowest index) whose forward pass was finite and
// whose z = dcost/expected exceeds zMin wins.  The scan over the step sizes can be cut in two
// stages [0,a1) and [a1,n_alpha): a trajectory that finds no acceptable step size in the first
// stage is appended to P.pending and only those are rolled out for the remaining step sizes.
// The scan state (last cnew / dcost / expected) is carried between the stages, so the result
// is exactly that of one scan over all step sizes.
//   from_list = 0: lane = trajectory, scans [a0,a1); from_list = 1: lane = entry of P.pending
__global__ void k_select(DevPtrs P, ilqg_dev_opts_t O, int a0, int a1, int from_list) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if(from_list) {
        if(b >= *P.n_pending) return;
        b = P.pending[b];
    }
    if(b >= P.B || P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE) return;
    const size_t Bp = P.Bp;
    const double cost = P.f[ILQG_F_COST][b], dV0 = P.f[ILQG_F_DV0][b], dV1 = P.f[ILQG_F_DV1][b];
    double cnew = (a0 > 0) ? P.f[ILQG_F_NEW_COST][b] : 0.0;
    double dcost = P.f[ILQG_F_DCOST][b], expected = P.f[ILQG_F_EXPECTED][b];
    int i, ok = 0;
    for(i = a0; i < a1; i++) {
        const double a = O.alpha[i];
        ok = P.i[ILQG_I_ALPHA_OK][(size_t)i * Bp + b];
        cnew = P.f[ILQG_F_ALPHA_COST][(size_t)i * Bp + b];
        if(!ok) continue;
        dcost = cost - cnew;
        expected = -a * (dV0 + a * dV1);
        const double z = (expected > 0) ? dcost / expected : 0.0;
        if(z > O.zMin) break;
        ok = 0;
    }
    if(!ok && a1 < O.n_alpha) P.pending[atomicAdd(P.n_pending_next, 1)] = b;  // to the second stage
    P.i[ILQG_I_ALPHA_IDX][b] = i + 1;
    P.i[ILQG_I_ACCEPTED][b] = ok;
    P.f[ILQG_F_NEW_COST][b] = cnew;
    P.f[ILQG_F_DCOST][b] = dcost;
    P.f[ILQG_F_EXPECTED][b] = expected;
}

// iLQG.c:311-361 and the loop bookkeeping of iLQG.c:239,365-378
__global__ void k_update(DevPtrs P, ilqg_dev_opts_t O) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if(b >= P.B || P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE) return;
    double lambda = P.f[ILQG_F_LAMBDA][b], dlambda = P.f[ILQG_F_DLAMBDA][b];
    int iter = P.i[ILQG_I_ITER][b];
    int status = ILQG_ST_ACTIVE;
    if(P.i[ILQG_I_ACCEPTED][b]) {
        const double t1 = dlambda / O.lambdaFactor, t2 = 1.0 / O.lambdaFactor;
        dlambda = (t1 < t2) ? t1 : t2;
        lambda = lambda * dlambda * (lambda > O.lambdaMin);
        P.f[ILQG_F_COST][b] = P.f[ILQG_F_NEW_COST][b];
        P.i[ILQG_I_NEED_DERIVS][b] = 1;
        if(P.f[ILQG_F_DCOST][b] < O.tolFun) status = ILQG_ST_CONVERGED_FUN;
    } else {
        const double t1 = dlambda * O.lambdaFactor;
        dlambda = (t1 > O.lambdaFactor) ? t1 : O.lambdaFactor;
        const double t2 = lambda * dlambda;
        lambda = (t2 > O.lambdaMin) ? t2 : O.lambdaMin;
        if(lambda > O.lambdaMax) status = ILQG_ST_LAMBDA_MAX;
    }
    if(status == ILQG_ST_ACTIVE) {
        iter++;
        if(iter >= O.max_iter) status = ILQG_ST_MAX_ITER;
    }
    P.f[ILQG_F_LAMBDA][b] = lambda;
    P.f[ILQG_F_DLAMBDA][b] = dlambda;
    P.i[ILQG_I_ITER][b] = iter;
    P.i[ILQG_I_STATUS][b] = status;
}

// solver entry state (iLQG.c:226-237)
__global__ void k_reset(DevPtrs P, ilqg_dev_opts_t O) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if(b >= P.Bp) return;
    const bool live = b < P.B && P.i[ILQG_I_STATUS][b] != ILQG_ST_INIT_FAILED;
    P.f[ILQG_F_LAMBDA][b] = O.lambdaInit;
    P.f[ILQG_F_DLAMBDA][b] = O.dlambdaInit;
    P.i[ILQG_I_ITER][b] = 0;
    P.i[ILQG_I_NEED_DERIVS][b] = 1;
    P.i[ILQG_I_ACCEPTED][b] = 0;
    P.i[ILQG_I_BP_CALLS][b] = 0;
    P.derivs_failed[b] = 0;
    if(live) P.i[ILQG_I_STATUS][b] = (O.max_iter > 0) ? ILQG_ST_ACTIVE : ILQG_ST_MAX_ITER;
}

__global__ void k_count_active(const int *status, int B, int *out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int active = (b < B && status[b] == ILQG_ST_ACTIVE) ? 1 : 0;
    const unsigned long long m = __ballot(active);
    if((threadIdx.x & 63) == 0 && m) atomicAdd(out, __popcll(m));
}

// unit-test kernel for the shared sincos the generated callbacks are routed through
__global__ void k_sincos_test(int n, const double *x, double *s, double *c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
#ifndef ILQG_NO_SHARED_SINCOS
    const ilqg_sc r = ilqg_sincos(x[i]);
    s[i] = r.s;
    c[i] = r.c;
#else
    sincos(x[i], &s[i], &c[i]);
#endif
}

// The reference's small dense helpers as callable entry points (matMult.h:11-14, cholesky.h:4-6), executed by
// the same device templates the lane-mapped kernels use.  One problem on one lane; sizes are those of this
// build's problem (matMult) resp. 1, 2, 3, 8 and N_U (Cholesky).
enum { DENSE_MULVEC = 0, DENSE_SQUARETRI = 1, DENSE_MUL2TRI = 2, DENSE_CHOL = 3, DENSE_CHOLINV = 4 };

template <int M>
__device__ void dense_chol(int op, const double *in, double *out, int *flag) {
    constexpr int T = tri(M);
    double A[T], U[T];
#pragma unroll
    for(int i = 0; i < T; i++) A[i] = in[i];
    if(op == DENSE_CHOL) {
        *flag = chol_factor<M>(A, U) ? 1 : 0;  // on failure the content of L is unspecified (as in the reference)
#pragma unroll
        for(int i = 0; i < T; i++) out[i] = U[i];
    } else {
        chol_inverse<M>(A, U);
#pragma unroll
        for(int i = 0; i < T; i++) out[i] = U[i];
    }
}

__global__ void k_dense_test(int op, int v0, int v1, int v2, const double *in0, const double *in1, const double *in2,
                             double *out, int *flag) {
    if(blockIdx.x || threadIdx.x) return;
    *flag = 1;
    if(op == DENSE_CHOL || op == DENSE_CHOLINV) {
        switch(v0) {
            case 1: dense_chol<1>(op, in0, out, flag); break;
            case 2: dense_chol<2>(op, in0, out, flag); break;
            case 3: dense_chol<3>(op, in0, out, flag); break;
            case 8: dense_chol<8>(op, in0, out, flag); break;
            default:
                if(v0 == NU) dense_chol<NU>(op, in0, out, flag); else *flag = -1;
        }
        return;
    }
#if !ILQG_WAVE_MAP
    // v0 selects the shape: 0 = (N_X, N_U), 1 = (N_X, N_X), 2 = (N_U, N_X) / (N_U, N_X, 1)
    if(op == DENSE_MULVEC) {
        if(v0 == 0) add_mul_vec<NX, NU>(out, in0, in1); else add_mul_vec<NX, NX>(out, in0, in1);
    } else if(op == DENSE_SQUARETRI) {
        if(v0 == 0) add_square_tri<NX, NU>(out, in0, in1);
        else if(v0 == 1) add_square_tri<NX, NX>(out, in0, in1);
        else add_square_tri<NU, NX>(out, in0, in1);
    } else if(op == DENSE_MUL2TRI) {
        if(v0 == 0) add_mul2_tri<NX, NX, NU>(out, in0, in1, in2); else add_mul2_tri<NU, NX, 1>(out, in0, in1, in2);
    }
#else
    *flag = -1;
#endif
}

// unit-test kernel for box_qp<M>, one problem per lane
template <int M>
__global__ __launch_bounds__(64, 1) void k_boxqp_test(int count, const double *H, const double *g, const double *lower, const double *upper,
                             double *x, int *clamp, int *n_free, double *invH, int *rc) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= count) return;
    constexpr int T = tri(M);
    double h[T], gg[M], lo[M], up[M], xx[M], inv[T];
    int cl[M], nf;
#pragma unroll
    for(int i = 0; i < T; i++) h[i] = H[(size_t)t * T + i];
#pragma unroll
    for(int i = 0; i < M; i++) {
        gg[i] = g[(size_t)t * M + i];
        lo[i] = lower[(size_t)t * M + i];
        up[i] = upper[(size_t)t * M + i];
        xx[i] = x[(size_t)t * M + i];
    }
    rc[t] = box_qp<M>(h, gg, lo, up, xx, cl, nf, inv);
    n_free[t] = nf;
#pragma unroll
    for(int i = 0; i < M; i++) {
        x[(size_t)t * M + i] = xx[i];
        clamp[t * M + i] = cl[i];
    }
#pragma unroll
    for(int i = 0; i < T; i++) invH[(size_t)t * T + i] = inv[i];
}

}  // namespace

// ===========================================================================
// shim
// ===========================================================================
struct ilqg_dev {
    int device, B, Bp, N;
    hipStream_t stream;
    DevPtrs P;
    ilqg_dev_opts_t O;
    std::vector<double *> param_bufs;
    double *staging;
    size_t staging_bytes;
    int *counter;
    int chunk;            // wave mapping: trajectories whose derivative records fit the work buffer
    bool work_consts;     // wave mapping: constant entries of the records written (init_running)
    bool timing;
    struct Span { int kernel; hipEvent_t a, b; };
    std::vector<Span> spans;
    double t_ms[ILQG_K_COUNT];
    int t_n[ILQG_K_COUNT];
};

namespace {

struct FieldInfo { int steps_plus; int wd, wh; };  // steps = steps_plus<0 ? 1 : N + steps_plus

FieldInfo field_info(int f) {
    switch(f) {
        case ILQG_F_X: return {1, NX, NX};
        case ILQG_F_U: return {0, NU, NU};
        case ILQG_F_LG: return {0, NU, NU};
        case ILQG_F_KG: return {0, NXU, NXU};
        case ILQG_F_DER: return {0, WAVE_MAP ? REC_HOST : REC, REC_HOST};
        case ILQG_F_FIN: return {-1, FIN, FIN};
        case ILQG_F_ALPHA_COST: return {-1, ILQG_MAX_ALPHA, ILQG_MAX_ALPHA};
        default: return {-1, 1, 1};
    }
}

int field_steps(const ilqg_dev *d, int f) {
    const FieldInfo fi = field_info(f);
    return fi.steps_plus < 0 ? 1 : d->N + fi.steps_plus;
}

int int_field_width(int f) { return f == ILQG_I_ALPHA_OK ? ILQG_MAX_ALPHA : 1; }

int ensure_staging(ilqg_dev *d, size_t bytes) {
    if(bytes <= d->staging_bytes) return 0;
    if(d->staging) HIP_TRY(hipFree(d->staging));
    d->staging = nullptr;
    d->staging_bytes = 0;
    HIP_TRY(hipMalloc((void **)&d->staging, bytes));
    d->staging_bytes = bytes;
    return 0;
}

struct Timed {
    ilqg_dev *d;
    int kernel;
    hipEvent_t a, b;
    Timed(ilqg_dev *d_, int k) : d(d_), kernel(k), a(nullptr), b(nullptr) {
        if(d->timing) {
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a, d->stream);
        }
    }
    ~Timed() {
        if(d->timing) {
            hipEventRecord(b, d->stream);
            d->spans.push_back({kernel, a, b});
        }
    }
};

int drain_spans(ilqg_dev *d) {
    if(d->spans.empty()) return 0;
    HIP_TRY(hipStreamSynchronize(d->stream));
    for(auto &s : d->spans) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, s.a, s.b);
        d->t_ms[s.kernel] += ms;
        d->t_n[s.kernel]++;
        hipEventDestroy(s.a);
        hipEventDestroy(s.b);
    }
    d->spans.clear();
    return 0;
}

inline dim3 grid1(size_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace

extern "C" {

const char *ilqg_dev_error(void) { return g_err.c_str(); }

int ilqg_dev_count(void) {
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void ilqg_dev_dims(int *out) {
    out[0] = NX;
    out[1] = NU;
    out[2] = FULL ? 1 : 0;
    out[3] = REC_HOST;
    out[4] = REC;
    out[5] = HX ? 1 : 0;
    out[6] = 0;
    out[7] = WAVE_MAP ? 1 : 0;
}

const char *ilqg_dev_kernel_name(int k) {
    static const char *names[ILQG_K_COUNT] = {"k_derivs", "k_backward", "k_rollout[search]", "k_select",
                                              "k_rollout[winner]", "k_update", "k_rollout[cost]", "k_rollout[init]",
                                              "k_to_soa/k_to_aos", "k_backward[fused derivs]", "k_rollout[search stage 2]"};
    return (k >= 0 && k < ILQG_K_COUNT) ? names[k] : "?";
}

int ilqg_dev_create(ilqg_dev_t **out, int device, int batch, int n_hor) {
    *out = nullptr;
    if(batch < 1 || n_hor < 2) {
        g_err = "ilqg_dev_create: need batch >= 1 and n_hor >= 2";
        return 1;
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if(device < 0 || device >= ndev) {
        g_err = "ilqg_dev_create: no such HIP device";
        return 1;
    }
    HIP_TRY(hipSetDevice(device));
    ilqg_dev *d = new ilqg_dev();
    d->device = device;
    d->B = batch;
    d->Bp = (batch + WAVE - 1) / WAVE * WAVE;
    d->N = n_hor;
    d->staging = nullptr;
    d->staging_bytes = 0;
    d->timing = false;
    memset(d->t_ms, 0, sizeof(d->t_ms));
    memset(d->t_n, 0, sizeof(d->t_n));
    memset(&d->P, 0, sizeof(d->P));
    memset(&d->O, 0, sizeof(d->O));
    d->P.B = d->B;
    d->P.Bp = d->Bp;
    d->P.N = d->N;
    HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    d->chunk = 0;
    d->work_consts = false;
    for(int f = 0; f < ILQG_F_COUNT; f++) {
        if(WAVE_MAP && f == ILQG_F_DER) {
            // derivative records = device trajEl_t structs for as many trajectories as fit the budget
            // budget: ILQG_WORK_GB if set, else half of the free device memory
            const char *e = getenv("ILQG_WORK_GB");
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            const double budget = e ? atof(e) * 1e9 : 0.5 * (double)free_b;
            const size_t per_traj = (size_t)d->N * sizeof(trajEl_t);
            size_t c = (size_t)(budget / (double)per_traj);
            if(c < 1) c = 1;
            if(c > (size_t)d->B) c = d->B;
            d->chunk = (int)c;
            HIP_TRY(hipMalloc((void **)&d->P.work, c * per_traj));
            HIP_TRY(hipMemsetAsync(d->P.work, 0, c * per_traj, d->stream));
            d->P.f[f] = nullptr;
            continue;
        }
        const FieldInfo fi = field_info(f);
        const size_t bytes = (size_t)field_steps(d, f) * fi.wd * d->Bp * sizeof(double);
        HIP_TRY(hipMalloc((void **)&d->P.f[f], bytes));
        HIP_TRY(hipMemsetAsync(d->P.f[f], 0, bytes, d->stream));
    }
    for(int f = 0; f < ILQG_I_COUNT; f++) {
        const size_t bytes = (size_t)int_field_width(f) * d->Bp * sizeof(int);
        HIP_TRY(hipMalloc((void **)&d->P.i[f], bytes));
        HIP_TRY(hipMemsetAsync(d->P.i[f], 0, bytes, d->stream));
    }
    HIP_TRY(hipMalloc((void **)&d->P.derivs_failed, d->Bp * sizeof(int)));
    HIP_TRY(hipMemsetAsync(d->P.derivs_failed, 0, d->Bp * sizeof(int), d->stream));
    HIP_TRY(hipMalloc((void **)&d->counter, sizeof(int)));
    HIP_TRY(hipMalloc((void **)&d->P.pending, d->Bp * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&d->P.n_pending, sizeof(int)));
    HIP_TRY(hipMemsetAsync(d->P.n_pending, 0, sizeof(int), d->stream));
    d->P.n_pending_next = d->P.n_pending;
    d->P.p = nullptr;
    HIP_TRY(hipStreamSynchronize(d->stream));
    *out = d;
    return 0;
}

void ilqg_dev_destroy(ilqg_dev_t *d) {
    if(!d) return;
    hipSetDevice(d->device);
    hipStreamSynchronize(d->stream);
    for(auto &s : d->spans) {
        hipEventDestroy(s.a);
        hipEventDestroy(s.b);
    }
    for(int f = 0; f < ILQG_F_COUNT; f++)
        if(d->P.f[f]) hipFree(d->P.f[f]);
    if(d->P.work) hipFree(d->P.work);
    for(int f = 0; f < ILQG_I_COUNT; f++) hipFree(d->P.i[f]);
    hipFree(d->P.derivs_failed);
    hipFree(d->counter);
    hipFree(d->P.pending);
    hipFree(d->P.n_pending);
    for(double *p : d->param_bufs) hipFree(p);
    if(d->P.p) hipFree(d->P.p);
    if(d->staging) hipFree(d->staging);
    hipStreamDestroy(d->stream);
    delete d;
}

int ilqg_dev_set_params(ilqg_dev_t *d, int n_params, const int *sizes, const double *const *values) {
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipStreamSynchronize(d->stream));
    for(double *p : d->param_bufs) hipFree(p);
    d->param_bufs.clear();
    if(d->P.p) hipFree(d->P.p);
    d->P.p = nullptr;
    std::vector<double *> ptrs(n_params > 0 ? n_params : 1, nullptr);
    for(int i = 0; i < n_params; i++) {
        const int sz = sizes[i] == -1 ? d->N + 1 : sizes[i];
        double *buf = nullptr;
        HIP_TRY(hipMalloc((void **)&buf, sz * sizeof(double)));
        HIP_TRY(hipMemcpy(buf, values[i], sz * sizeof(double), hipMemcpyHostToDevice));
        d->param_bufs.push_back(buf);
        ptrs[i] = buf;
    }
    HIP_TRY(hipMalloc((void **)&d->P.p, ptrs.size() * sizeof(double *)));
    HIP_TRY(hipMemcpy(d->P.p, ptrs.data(), ptrs.size() * sizeof(double *), hipMemcpyHostToDevice));
    d->work_consts = false;  // constant record entries depend on the parameters
    return 0;
}

int ilqg_dev_set_opts(ilqg_dev_t *d, const ilqg_dev_opts_t *o) {
    if(o->n_alpha < 1 || o->n_alpha > ILQG_MAX_ALPHA) {
        g_err = "ilqg_dev_set_opts: n_alpha must be in 1..16";
        return 1;
    }
    d->O = *o;
    return 0;
}

int ilqg_dev_field_width(int field) { return field_info(field).wh; }
int ilqg_dev_field_steps(ilqg_dev_t *d, int field) { return field_steps(d, field); }
void *ilqg_dev_field_ptr(ilqg_dev_t *d, int field) { return d->P.f[field]; }
void *ilqg_dev_stream(ilqg_dev_t *d) { return (void *)d->stream; }

int ilqg_dev_write(ilqg_dev_t *d, int field, const double *host) {
    return ilqg_dev_write_steps(d, field, host, field_steps(d, field));
}

// wave mapping: these fields are trajectory-major on the device, i.e. already in host layout
static bool is_traj_major(int field) {
    return WAVE_MAP && (field == ILQG_F_X || field == ILQG_F_U || field == ILQG_F_LG || field == ILQG_F_KG ||
                        field == ILQG_F_FIN);
}

#if ILQG_WAVE_MAP
// derivative records live in device trajEl_t structs: convert to/from the packed host record
#define REC_FIELDS(OP)                                                                            \
    OP(cx, NX) OP(cxx, SXX) OP(cu, NU) OP(cuu, SUU) OP(cxu, NXU) OP(fx, NX * NX) OP(fu, NXU)     \
    OP(lower, NU) OP(upper, NU) REC_FIELDS_FULL(OP)                                               \
    OP(lower_sign, NU) OP(upper_sign, NU) OP(lower_hx, NXU) OP(upper_hx, NXU)
#if FULL_DDP
#define REC_FIELDS_FULL(OP) OP(fxx, NX * SXX) OP(fuu, NX * SUU) OP(fxu, NX * NXU)
#else
#define REC_FIELDS_FULL(OP)
#endif

static int der_io(ilqg_dev *d, double *host_rw, const double *host_ro) {
    if(d->B > d->chunk) {
        g_err = "derivative records of the whole batch do not fit the work buffer (wave mapping): read/write them "
                "with a batch <= the chunk size";
        return 1;
    }
    const size_t n = (size_t)d->B * d->N;
    std::vector<trajEl_t> tmp(n);
    HIP_TRY(hipMemcpy(tmp.data(), d->P.work, n * sizeof(trajEl_t), hipMemcpyDeviceToHost));
    for(size_t e = 0; e < n; e++) {
        trajEl_t &t = tmp[e];
        if(host_ro) {
            const double *r = host_ro + e * REC_HOST;
#define OP(field, cnt) memcpy(t.field, r, sizeof(double) * (cnt)); r += (cnt);
            REC_FIELDS(OP)
#undef OP
        } else {
            double *r = host_rw + e * REC_HOST;
#define OP(field, cnt) memcpy(r, t.field, sizeof(double) * (cnt)); r += (cnt);
            REC_FIELDS(OP)
#undef OP
        }
    }
    if(host_ro) HIP_TRY(hipMemcpy(d->P.work, tmp.data(), n * sizeof(trajEl_t), hipMemcpyHostToDevice));
    return 0;
}
#endif

int ilqg_dev_write_steps(ilqg_dev_t *d, int field, const double *host, int steps) {
    HIP_TRY(hipSetDevice(d->device));
    const FieldInfo fi = field_info(field);
    if(steps < 1 || steps > field_steps(d, field)) {
        g_err = "ilqg_dev_write_steps: bad step count";
        return 1;
    }
#if ILQG_WAVE_MAP
    if(field == ILQG_F_DER) {
        HIP_TRY(hipStreamSynchronize(d->stream));
        return der_io(d, nullptr, host);
    }
#endif
    if(is_traj_major(field)) {
        const size_t row = (size_t)steps * fi.wd * sizeof(double);
        const size_t dpitch = (size_t)field_steps(d, field) * fi.wd * sizeof(double);
        HIP_TRY(hipMemcpy2DAsync(d->P.f[field], dpitch, host, row, row, d->B, hipMemcpyHostToDevice, d->stream));
        HIP_TRY(hipStreamSynchronize(d->stream));
        return 0;
    }
    const size_t n = (size_t)d->B * steps * fi.wh;
    if(ensure_staging(d, n * sizeof(double))) return 1;
    HIP_TRY(hipMemcpyAsync(d->staging, host, n * sizeof(double), hipMemcpyHostToDevice, d->stream));
    {
        Timed t(d, ILQG_K_TRANSPOSE);
        const size_t total = (size_t)d->B * steps * fi.wd;
        hipLaunchKernelGGL(k_to_soa, grid1(total, 256), dim3(256), 0, d->stream, d->staging, d->P.f[field], d->B, d->Bp,
                           steps, fi.wh, fi.wd);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(d->stream));
    return 0;
}

int ilqg_dev_read(ilqg_dev_t *d, int field, double *host) {
    HIP_TRY(hipSetDevice(d->device));
    const FieldInfo fi = field_info(field);
    const int steps = field_steps(d, field);
    const size_t n = (size_t)d->B * steps * fi.wh;
#if ILQG_WAVE_MAP
    if(field == ILQG_F_DER) {
        HIP_TRY(hipStreamSynchronize(d->stream));
        return der_io(d, host, nullptr);
    }
#endif
    if(is_traj_major(field)) {
        HIP_TRY(hipMemcpyAsync(host, d->P.f[field], n * sizeof(double), hipMemcpyDeviceToHost, d->stream));
        HIP_TRY(hipStreamSynchronize(d->stream));
        return 0;
    }
    if(ensure_staging(d, n * sizeof(double))) return 1;
    {
        Timed t(d, ILQG_K_TRANSPOSE);
        hipLaunchKernelGGL(k_to_aos, grid1(n, 256), dim3(256), 0, d->stream, d->P.f[field], d->staging, d->B, d->Bp,
                           steps, fi.wh, fi.wd);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host, d->staging, n * sizeof(double), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return 0;
}

// int fields are [width][Bp] on the device, [B][width] on the host
int ilqg_dev_write_int(ilqg_dev_t *d, int field, const int *host) {
    HIP_TRY(hipSetDevice(d->device));
    const int w = int_field_width(field);
    std::vector<int> tmp((size_t)w * d->Bp, 0);
    for(int b = 0; b < d->B; b++)
        for(int j = 0; j < w; j++) tmp[(size_t)j * d->Bp + b] = host[(size_t)b * w + j];
    HIP_TRY(hipMemcpyAsync(d->P.i[field], tmp.data(), tmp.size() * sizeof(int), hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return 0;
}

int ilqg_dev_read_int(ilqg_dev_t *d, int field, int *host) {
    HIP_TRY(hipSetDevice(d->device));
    const int w = int_field_width(field);
    std::vector<int> tmp((size_t)w * d->Bp, 0);
    HIP_TRY(hipMemcpyAsync(tmp.data(), d->P.i[field], tmp.size() * sizeof(int), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    for(int b = 0; b < d->B; b++)
        for(int j = 0; j < w; j++) host[(size_t)b * w + j] = tmp[(size_t)j * d->Bp + b];
    return 0;
}

#define NEED_PARAMS(d)                                                   \
    if(!(d)->P.p) {                                                      \
        g_err = "problem parameters not set (ilqg_dev_set_params)";      \
        return 1;                                                        \
    }

int ilqg_dev_reset(ilqg_dev_t *d) {
    HIP_TRY(hipSetDevice(d->device));
    hipLaunchKernelGGL(k_reset, grid1(d->Bp, 256), dim3(256), 0, d->stream, d->P, d->O);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int launch_rollout(ilqg_dev_t *d, int mode, int kernel_id, int a0, int n_alpha) {
    Timed t(d, kernel_id);
    hipLaunchKernelGGL(k_rollout, dim3(d->Bp / WAVE, n_alpha), dim3(WAVE), 0, d->stream, d->P, d->O, mode, a0);
    return 0;
}

int ilqg_dev_rollout_init(ilqg_dev_t *d) {
    NEED_PARAMS(d);
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipMemsetAsync(d->P.i[ILQG_I_STATUS], 0, d->Bp * sizeof(int), d->stream));
    launch_rollout(d, ROLL_INIT, ILQG_K_ROLLOUT_INIT, 0, 1);
    HIP_TRY(hipGetLastError());
    return 0;
}

#if ILQG_WAVE_MAP
// wave mapping: derivative records are evaluated chunk by chunk into the work buffer and consumed by
// the backward kernel of the same chunk.  do_derivs = 0 uses the records already in the buffer.
static int wave_backward(ilqg_dev_t *d, int single_sweep, int do_derivs, int do_backward) {
    for(int c0 = 0; c0 < d->B; c0 += d->chunk) {
        const int cnt = (d->B - c0 < d->chunk) ? d->B - c0 : d->chunk;
        if(do_derivs) {
            Timed t(d, ILQG_K_DERIVS);
            const size_t total = (size_t)cnt * (d->N + 1);
            hipLaunchKernelGGL(k_derivs_wave, grid1(total, 64), dim3(64), 0, d->stream, d->P, d->O, c0, cnt,
                               d->work_consts ? 0 : 1);
        }
        if(do_backward) {
            Timed t(d, ILQG_K_BACKWARD);
            hipLaunchKernelGGL(k_backward_wave, dim3(cnt), dim3(64), 0, d->stream, d->P, d->O, single_sweep, c0, cnt);
        }
        if(do_derivs && cnt == d->chunk) d->work_consts = true;  // every element of the buffer has its constants now
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
#endif

int ilqg_dev_derivs(ilqg_dev_t *d) {
    NEED_PARAMS(d);
    HIP_TRY(hipSetDevice(d->device));
#if ILQG_WAVE_MAP
    if(d->B > d->chunk) {
        g_err = "ilqg_dev_derivs: the batch does not fit the work buffer; use ilqg_dev_backward(mode 2) / iterate";
        return 1;
    }
    return wave_backward(d, 0, 1, 0);
#else
    {
        Timed t(d, ILQG_K_DERIVS);
        const size_t total = (size_t)d->Bp * (d->N + 1);
        hipLaunchKernelGGL(k_derivs, grid1(total, 256), dim3(256), 0, d->stream, d->P, d->O);
    }
    HIP_TRY(hipGetLastError());
    return 0;
#endif
}

int ilqg_dev_backward(ilqg_dev_t *d, int mode) {
    HIP_TRY(hipSetDevice(d->device));
    if(mode < 0 || mode > 2) {
        g_err = "ilqg_dev_backward: mode must be 0, 1 or 2";
        return 1;
    }
    if(mode == 2) NEED_PARAMS(d);
#if ILQG_WAVE_MAP
    if(mode != 2 && d->B > d->chunk) {
        g_err = "ilqg_dev_backward: stored records need the whole batch in the work buffer; use mode 2";
        return 1;
    }
    return wave_backward(d, mode == 1, mode == 2, 1);
#else
    {
        Timed t(d, mode == 2 ? ILQG_K_BACKWARD_FUSED : ILQG_K_BACKWARD);
        const dim3 grid(d->Bp / WAVE), block(WAVE);
        if(mode == 0)
            hipLaunchKernelGGL(k_backward<0>, grid, block, 0, d->stream, d->P, d->O);
        else if(mode == 1)
            hipLaunchKernelGGL(k_backward<1>, grid, block, 0, d->stream, d->P, d->O);
        else
            hipLaunchKernelGGL(k_backward<2>, grid, block, 0, d->stream, d->P, d->O);
    }
    HIP_TRY(hipGetLastError());
    return 0;
#endif
}

int ilqg_dev_search(ilqg_dev_t *d) {
    NEED_PARAMS(d);
    HIP_TRY(hipSetDevice(d->device));
    const int A = d->O.n_alpha;
    const int s1 = (d->O.ls_split > 0 && d->O.ls_split < A) ? d->O.ls_split : A;  // step sizes in stage 1
    HIP_TRY(hipMemsetAsync(d->P.n_pending, 0, sizeof(int), d->stream));
    launch_rollout(d, ROLL_SEARCH, ILQG_K_ROLLOUT_SEARCH, 0, s1);
    {
        Timed t(d, ILQG_K_SELECT);
        hipLaunchKernelGGL(k_select, grid1(d->Bp, 256), dim3(256), 0, d->stream, d->P, d->O, 0, s1, 0);
    }
    if(s1 < A) {
        // second stage: only the trajectories without an acceptable step size so far.  The grid covers the
        // worst case; blocks beyond the pending count return at once.
        launch_rollout(d, ROLL_SEARCH_LIST, ILQG_K_ROLLOUT_SEARCH2, s1, A - s1);
        Timed t(d, ILQG_K_SELECT);
        hipLaunchKernelGGL(k_select, grid1(d->Bp, 256), dim3(256), 0, d->stream, d->P, d->O, s1, A, 1);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int ilqg_dev_winner(ilqg_dev_t *d) {
    NEED_PARAMS(d);
    HIP_TRY(hipSetDevice(d->device));
    launch_rollout(d, ROLL_WINNER, ILQG_K_ROLLOUT_WINNER, 0, 1);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ilqg_dev_update(ilqg_dev_t *d) {
    NEED_PARAMS(d);
    HIP_TRY(hipSetDevice(d->device));
    {
        Timed t(d, ILQG_K_UPDATE);
        hipLaunchKernelGGL(k_update, grid1(d->Bp, 256), dim3(256), 0, d->stream, d->P, d->O);
    }
    if(d->O.resweep) launch_rollout(d, ROLL_COST, ILQG_K_ROLLOUT_COST, 0, 1);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ilqg_dev_iterate(ilqg_dev_t *d, int n) {
    for(int it = 0; it < n; it++) {
        if(d->O.fuse_derivs || WAVE_MAP) {  // wave mapping: derivatives + sweep chunk by chunk
            if(ilqg_dev_backward(d, 2)) return 1;
        } else {
            if(ilqg_dev_derivs(d)) return 1;
            if(ilqg_dev_backward(d, 0)) return 1;
        }
        if(ilqg_dev_search(d)) return 1;
        if(ilqg_dev_winner(d)) return 1;
        if(ilqg_dev_update(d)) return 1;
    }
    return 0;
}

int ilqg_dev_sync(ilqg_dev_t *d) {
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return 0;
}

int ilqg_dev_count_active(ilqg_dev_t *d, int *n_active) {
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipMemsetAsync(d->counter, 0, sizeof(int), d->stream));
    hipLaunchKernelGGL(k_count_active, grid1(d->Bp, 256), dim3(256), 0, d->stream, d->P.i[ILQG_I_STATUS], d->B,
                       d->counter);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(n_active, d->counter, sizeof(int), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return 0;
}

int ilqg_dev_timing(ilqg_dev_t *d, int enable) {
    if(drain_spans(d)) return 1;
    d->timing = enable != 0;
    memset(d->t_ms, 0, sizeof(d->t_ms));
    memset(d->t_n, 0, sizeof(d->t_n));
    return 0;
}

int ilqg_dev_get_timing(ilqg_dev_t *d, int kernel, int *launches, double *total_ms) {
    if(kernel < 0 || kernel >= ILQG_K_COUNT) {
        g_err = "ilqg_dev_get_timing: bad kernel id";
        return 1;
    }
    if(drain_spans(d)) return 1;
    *launches = d->t_n[kernel];
    *total_ms = d->t_ms[kernel];
    return 0;
}

// op / shape as in k_dense_test; in*/out are host arrays of n_in0/n_in1/n_in2/n_out doubles (out is in/out)
int ilqg_dev_dense(int device, int op, int shape, const double *in0, int n_in0, const double *in1, int n_in1,
                   const double *in2, int n_in2, double *out, int n_out, int *flag) {
    HIP_TRY(hipSetDevice(device));
    double *d0 = nullptr, *d1 = nullptr, *d2 = nullptr, *dout = nullptr;
    int *dflag = nullptr;
    HIP_TRY(hipMalloc((void **)&d0, (n_in0 > 0 ? n_in0 : 1) * 8));
    HIP_TRY(hipMalloc((void **)&d1, (n_in1 > 0 ? n_in1 : 1) * 8));
    HIP_TRY(hipMalloc((void **)&d2, (n_in2 > 0 ? n_in2 : 1) * 8));
    HIP_TRY(hipMalloc((void **)&dout, (n_out > 0 ? n_out : 1) * 8));
    HIP_TRY(hipMalloc((void **)&dflag, 4));
    if(n_in0 > 0) HIP_TRY(hipMemcpy(d0, in0, n_in0 * 8, hipMemcpyHostToDevice));
    if(n_in1 > 0) HIP_TRY(hipMemcpy(d1, in1, n_in1 * 8, hipMemcpyHostToDevice));
    if(n_in2 > 0) HIP_TRY(hipMemcpy(d2, in2, n_in2 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dout, out, n_out * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dense_test, dim3(1), dim3(64), 0, 0, op, shape, 0, 0, d0, d1, d2, dout, dflag);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout, n_out * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(flag, dflag, 4, hipMemcpyDeviceToHost));
    hipFree(d0); hipFree(d1); hipFree(d2); hipFree(dout); hipFree(dflag);
    return 0;
}

int ilqg_dev_sincos_batch(int device, int n, const double *x, double *s, double *c) {
    HIP_TRY(hipSetDevice(device));
    double *dx, *ds, *dc;
    HIP_TRY(hipMalloc((void **)&dx, n * 8));
    HIP_TRY(hipMalloc((void **)&ds, n * 8));
    HIP_TRY(hipMalloc((void **)&dc, n * 8));
    HIP_TRY(hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sincos_test, grid1(n, 256), dim3(256), 0, 0, n, dx, ds, dc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(s, ds, n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(c, dc, n * 8, hipMemcpyDeviceToHost));
    hipFree(dx); hipFree(ds); hipFree(dc);
    return 0;
}

int ilqg_dev_boxqp_batch(int device, int n, int count, const double *H, const double *g, const double *lower,
                         const double *upper, double *x, int *clamp, int *n_free, double *invH, int *rc) {
    if(n != 2 && n != 8 && n != NU) {
        g_err = "ilqg_dev_boxqp_batch: n must be 2, 8 or N_U";
        return 1;
    }
    HIP_TRY(hipSetDevice(device));
    const size_t T = n * (n + 1) / 2;
    double *dH, *dg, *dlo, *dup, *dx, *dinv;
    int *dcl, *dnf, *drc;
    HIP_TRY(hipMalloc((void **)&dH, count * T * 8));
    HIP_TRY(hipMalloc((void **)&dinv, count * T * 8));
    HIP_TRY(hipMalloc((void **)&dg, count * n * 8));
    HIP_TRY(hipMalloc((void **)&dlo, count * n * 8));
    HIP_TRY(hipMalloc((void **)&dup, count * n * 8));
    HIP_TRY(hipMalloc((void **)&dx, count * n * 8));
    HIP_TRY(hipMalloc((void **)&dcl, count * n * 4));
    HIP_TRY(hipMalloc((void **)&dnf, count * 4));
    HIP_TRY(hipMalloc((void **)&drc, count * 4));
    HIP_TRY(hipMemcpy(dH, H, count * T * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dg, g, count * n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dlo, lower, count * n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dup, upper, count * n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dx, x, count * n * 8, hipMemcpyHostToDevice));
    const dim3 grid = grid1(count, 64), block(64);
    if(n == 2)
        hipLaunchKernelGGL(k_boxqp_test<2>, grid, block, 0, 0, count, dH, dg, dlo, dup, dx, dcl, dnf, dinv, drc);
    else if(n == 8)
        hipLaunchKernelGGL(k_boxqp_test<8>, grid, block, 0, 0, count, dH, dg, dlo, dup, dx, dcl, dnf, dinv, drc);
    else
        hipLaunchKernelGGL(k_boxqp_test<NU>, grid, block, 0, 0, count, dH, dg, dlo, dup, dx, dcl, dnf, dinv, drc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(x, dx, count * n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(invH, dinv, count * T * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(clamp, dcl, count * n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(n_free, dnf, count * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rc, drc, count * 4, hipMemcpyDeviceToHost));
    hipFree(dH); hipFree(dinv); hipFree(dg); hipFree(dlo); hipFree(dup); hipFree(dx);
    hipFree(dcl); hipFree(dnf); hipFree(drc);
    return 0;
}

}  // extern "C"
