// "Wave mapping": ONE WAVEFRONT PER TRAJECTORY, for problems whose matrices do not fit one
// lane's registers (n = 16, m = 8: the derivative record of one step is 5 524 doubles with
// FULL_DDP).  The 64 lanes of a wave share the step's matrices through LDS; every OUTPUT
// element of a product is owned by one lane, which accumulates its sum sequentially in
// ascending index order — the reference's order (matMult.c:14-72, back_pass.c:80-241) — so
// the results equal the lane mapping's and the CPU's up to FMA contraction.
//
// Data: the derivative records are the reference's own `trajEl_t` structs in HBM
// (iLQG_problem.tem:23-51), written in place by the generated callbacks (k_derivs_wave);
// a wave reads each array field with consecutive lanes on consecutive doubles (coalesced).
// x, u, l, L are trajectory-major [trajectory][step][field].
//
// The box QP of a step (boxQP.c:39-238, size NU) is evaluated redundantly by all lanes in
// registers with the same template the lane mapping uses (box_qp_uniform<NU>); its results go through
// LDS because the gain formula indexes them per lane.
#pragma once
#include <hip/hip_runtime.h>
#include "ilqg_device.hpp"

namespace ilqg {

// packed upper-triangle index e -> (r, c), r <= c
__device__ __forceinline__ void tri_rc(int e, int &r, int &c) {
    c = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while((c + 1) * (c + 2) / 2 <= e) c++;
    while(c * (c + 1) / 2 > e) c--;
    r = e - c * (c + 1) / 2;
}

// Column-major matrices that are read one COLUMN per lane (a'b-type products) are stored with the
// leading dimension padded by one double: a stride of 16 doubles (128 B) would put every lane of such
// a read on the same two LDS banks (8-way conflict, measured 6 700 conflict cycles per step); with 17
// (resp. 9) doubles the 16 column starts fall on distinct bank pairs.
template <int NX, int NU>
struct WaveLds {
    static constexpr int SXX = tri(NX), SUU = tri(NU), NXU = NX * NU;
    static constexpr int LDX = NX + 1, LDU = NU + 1;
    double Vx[NX], Vxx[SXX];
    double fx[LDX * NX], fu[LDX * NU];   // NX x NX, NX x NU, leading dimension LDX
    double T1[LDX * NX], T2[LDX * NU];   // Vxx*fx, Vxx*fu, leading dimension LDX
    double Qx[NX], Qu[NU], Qxx[SXX], Qxu[NXU], Quu[SUU];
    double QuuF[SUU], Qxur[NXU];
    double K[LDU * NX], l[NU], invH[SUU]; // NU x NX, leading dimension LDU
    double ba[LDU * NX], bc[NU];          // Quu*K (leading dimension LDU), Quu*l
    int clamp[NU];
};

// ---------------------------------------------------------------------------
// boxQP.c:39-238 for the wave mapping, cooperative: lane i < M owns variable i — its x, g, limits, clamp flag,
// row i of H and of the inverse, column i of the Cholesky factor — and lanes >= M mirror lane (lane mod M), so
// every lane takes part in every broadcast.  (The first version solved the whole problem redundantly on every
// lane: for M = 8 that is four packed M x M matrices per lane, 512 registers and one wavefront per SIMD.)
//
// Every scalar of the reference is computed by ONE expression tree in the reference's own operand order —
// a row sum runs j = 0..M-1 on the row's lane with x[j] broadcast (v_readlane), a sum over the variables
// (value, gradient norm, search'grad) runs i = 0..M-1 on broadcast operands, uniformly on all lanes — so
// the results are those of box_qp_uniform bit for bit (asserted by a unit test on the reference's goldens).
// Sums over a subset (free or clamped variables) skip the others through wave-uniform masks, as the
// reference's loops do; where the reference's loop bounds depend on the row (Cholesky, inverse) the extra
// terms are exact zeros.
// Results: x -> S_l[i], flags -> S_clamp[i], inverse of the free Hessian (full-index form, packed) -> S_invH,
// all in LDS, where the gain formula reads them.  Returns the reference's code, wave-uniform.
// ---------------------------------------------------------------------------
// Hand-over of LDS data between the lanes of ONE wavefront.  A wavefront's LDS operations execute in issue order, so
// a lane's read behind another lane's write of the same wavefront sees it; all that is needed is that the compiler
// keeps that order.  (No s_barrier: a workgroup may hold several wavefronts, each working on a trajectory of its own at
// its own pace.)
ILQG_DEV void wave_sync() {
#ifdef ILQG_WAVE_SYNC_BARRIER
    __syncthreads();
    return;
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

ILQG_DEV double lane_bcast(double v, int src) {  // src: wave-uniform lane number
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

template <int M>
ILQG_DEV int box_qp_rows(const double *Hpacked /* LDS */, const double g, const double lower, const double upper,
                         double *S_l, int *S_clamp, double *S_invH, int &n_free_out) {
    static_assert(M <= 32, "one lane per variable, masks in 32 bits");
    constexpr int T = tri(M);
    const int lane = threadIdx.x & 63, me = lane % M;
    const unsigned all = (M == 32) ? 0xffffffffu : ((1u << M) - 1u);
    const int max_iter = 100;
    const double min_grad = 1e-8, min_rel_improve = 1e-8, step_dec = 0.6, min_step = 1e-22, armijo = 0.1;

    double Hrow[M], invrow[M], Ucol[M];
#pragma unroll
    for(int j = 0; j < M; j++) {
        Hrow[j] = Hpacked[sy(me, j)];
        invrow[j] = 0.0;
        Ucol[j] = 0.0;
    }
    double x = S_l[me];  // warm start
    if(x > upper) x = upper;
    if(x < lower) x = lower;
    int clamp = 0;
    for(int e = lane; e < T; e += 64) S_invH[e] = 0.0;
    n_free_out = 0;

    // value(y) = sum_i y_i (g_i + 0.5 (H y)_i), boxQP.c:17-37
    auto qp_value_rows = [&](double y) {
        double hx = 0.0;
#pragma unroll
        for(int j = 0; j < M; j++) hx += Hrow[j] * lane_bcast(y, j);
        const double w = g + 0.5 * hx;
        double v = 0.0;
#pragma unroll
        for(int i = 0; i < M; i++) v += lane_bcast(y, i) * lane_bcast(w, i);
        return v;
    };

    double value = qp_value_rows(x), oldvalue = 0.0;
    int rc = 1;  // max_iter iterations (boxQP.c:237)
    for(int iter = 0; iter < max_iter; iter++) {
        if(iter > 0 && (oldvalue - value) < min_rel_improve * fabs(oldvalue)) { rc = 4; break; }
        oldvalue = value;

        // gradient and clamped set (boxQP.c:101-124)
        double hx = 0.0;
#pragma unroll
        for(int j = 0; j < M; j++) hx += Hrow[j] * lane_bcast(x, j);
        const double grad = g + hx;
        const int was = clamp;
        if(x <= lower && grad > 0)
            clamp = 1;
        else if(x >= upper && grad < 0)
            clamp = 2;
        else
            clamp = 0;
        const unsigned cm = (unsigned)__ballot(clamp != 0) & all;               // clamped variables
        const bool changed = ((unsigned)__ballot((!was) != (!clamp)) & all) != 0;
        const int n_free = M - __popc(cm);
        n_free_out = n_free;
        if(cm == all) { rc = 6; break; }
        double gnorm = 0.0;
#pragma unroll
        for(int i = 0; i < M; i++) {
            const double gi = lane_bcast(grad, i);
            if(!((cm >> i) & 1u)) gnorm += gi * gi;
        }

        if(iter == 0 || changed) {
            // Cholesky of the Hessian with clamped rows and columns replaced by identity (boxQP.c:131-160,
            // cholesky.c:6-27): lane i computes column i of U, row j in step j
            bool pd = true;
#pragma unroll
            for(int j = 0; j < M; j++) {
                double dot = 0.0;
#pragma unroll
                for(int k = 0; k < j; k++) dot += Ucol[k] * lane_bcast(Ucol[k], j);
                const bool masked = clamp != 0 || ((cm >> j) & 1u);
                const double a = masked ? ((me == j) ? 1.0 : 0.0) : Hrow[j];
                const double sv = a - dot;
                const double piv = lane_bcast(sv, j);
                if(piv <= 0.0) pd = false;
                const double d = sqrt(piv);
                Ucol[j] = (me == j) ? d : ((me > j) ? 1.0 / d * sv : 0.0);
            }
            if(!pd) { rc = -1; break; }
            // explicit inverse (cholesky.c:51-74): lane l solves U'U y = e_l; y[k] for k >= l is row l of the inverse
            double y[M];
#pragma unroll
            for(int k = 0; k < M; k++) {
                double v = (k == me) ? 1.0 : 0.0;
#pragma unroll
                for(int i = 0; i < k; i++) v -= y[i] * lane_bcast(Ucol[i], k);  // y[i] = 0 for i < l: exact zeros
                y[k] = v / lane_bcast(Ucol[k], k);
            }
#pragma unroll
            for(int k = M - 1; k >= 0; k--) {
                double v = y[k];
#pragma unroll
                for(int i = k + 1; i < M; i++) v -= y[i] * lane_bcast(Ucol[k], i);
                y[k] = v / lane_bcast(Ucol[k], k);
            }
            wave_sync();
            if(lane < M) {
#pragma unroll
                for(int k = 0; k < M; k++)
                    if(k >= me) S_invH[ut(me, k)] = y[k];
            }
            wave_sync();
#pragma unroll
            for(int j = 0; j < M; j++) invrow[j] = S_invH[sy(me, j)];
        }

        if(gnorm < min_grad * min_grad) { rc = 5; break; }

        // search(free) = -invH(free,free) (g + H x_clamped)(free) - x(free); search(clamped) = 0 (boxQP.c:170-196)
        double hc = 0.0;
#pragma unroll
        for(int j = 0; j < M; j++) {
            const double xj = lane_bcast(x, j);
            if((cm >> j) & 1u) hc += Hrow[j] * xj;
        }
        const double gc = g + hc;
        double sr = -x;
#pragma unroll
        for(int j = 0; j < M; j++) {
            const double gj = lane_bcast(gc, j);
            if(!((cm >> j) & 1u)) sr -= invrow[j] * gj;
        }
        const double search = clamp ? 0.0 : sr;

        double sdotg = 0.0;
#pragma unroll
        for(int i = 0; i < M; i++) sdotg += lane_bcast(search, i) * lane_bcast(grad, i);
        if(sdotg >= 0.0) { rc = -2; break; }

        // Armijo line search (boxQP.c:203-228)
        double step = 1.0, vc, xc;
        bool tiny = false;
        for(;;) {
            xc = x + step * search;
            if(xc > upper) xc = upper;
            if(xc < lower) xc = lower;
            vc = qp_value_rows(xc);
            if(((vc - oldvalue) / (step * sdotg)) >= armijo) break;
            step = step * step_dec;
            if(step < min_step) { tiny = true; break; }
        }
        if(tiny) { rc = 2; break; }
        x = xc;
        value = vc;
    }
    wave_sync();
    if(lane < M) {
        S_l[me] = x;
        S_clamp[me] = clamp;
    }
    wave_sync();
    return rc;
}

// The operands of a lane's dot product, LDS -> registers, ALL reads issued before the first multiply-add.  Left to
// itself the scheduler (minimising registers in a kernel this large) issues one read, waits for it, does two
// multiply-adds, issues the next: 8 exposed LDS latencies per 16-term sum, with nothing else on the SIMD to hide
// them (measured: 191 full LDS waits in the step).  The summation order is untouched.
template <int N>
ILQG_DEV void lds_fetch(double (&dst)[N], const double *src, int stride = 1) {
#pragma unroll
    for(int i = 0; i < N; i++) dst[i] = src[i * stride];
}
ILQG_DEV void reads_before_math() { __builtin_amdgcn_sched_barrier(0); }
template <int N>
ILQG_DEV double dot_acc(double acc, const double (&a)[N], const double (&b)[N]) {
#pragma unroll
    for(int i = 0; i < N; i++) acc += a[i] * b[i];
    return acc;
}

// rec: this step's trajEl_t in global memory (fields read through the pointers below)
template <int NX, int NU>
struct StepFields {
    const double *cx, *cxx, *cu, *cuu, *cxu, *fx, *fu, *lower, *upper, *fxx, *fuu, *fxu;
    const double *lower_sign, *upper_sign, *lower_hx, *upper_hx, *u;
};

// Everything a lane reads of a step's record, in registers.  With one wavefront per SIMD nothing else hides the
// latency of a global load, and a step used to expose it about eight times (one load-then-use per section).  The
// whole set is loaded in one go at the start of the step instead.  (Loading the NEXT step's set as soon as the Q
// assembly has consumed the registers, in flight during the box QP, was measured as well: the 100 doubles then live
// across the whole step push the kernel over 512 registers and the spills cost what the overlap gains.)
template <int NX, int NU, bool FULL>
struct StepRegs {
    static constexpr int SXX = tri(NX), SUU = tri(NU), NXU = NX * NU;
    static constexpr int NFX = (NX * NX + 63) / 64, NFU = (NXU + 63) / 64, NC = (NU + NX + 63) / 64;
    static constexpr int NJ = (NXU + 63) / 64, NQ = (SUU + SXX + 63) / 64;
    double fx[NFX], fu[NFU], c1[NC];      // this lane's elements of fx, fu, (cu | cx)
    double cxu[NJ], txu[NJ][FULL ? NX : 1];  // of cxu and the NX slices of fxu
    double cq[NQ], tq[NQ][FULL ? NX : 1];    // of (cuu | cxx) and the NX slices of (fuu | fxx)
    double lo, up;                        // limits of input (lane mod NU)
    double u[NU];                         // nominal inputs (gradient norm)
};

template <int NX, int NU, bool FULL>
__device__ __forceinline__ void load_step(StepRegs<NX, NU, FULL> &R, const StepFields<NX, NU> &F) {
    using SR = StepRegs<NX, NU, FULL>;
    constexpr int SXX = SR::SXX, SUU = SR::SUU, NXU = SR::NXU;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for(int a = 0; a < SR::NFX; a++) {
        const int i = lane + 64 * a;
        R.fx[a] = (i < NX * NX) ? F.fx[i] : 0.0;
    }
#pragma unroll
    for(int a = 0; a < SR::NFU; a++) {
        const int i = lane + 64 * a;
        R.fu[a] = (i < NXU) ? F.fu[i] : 0.0;
    }
#pragma unroll
    for(int a = 0; a < SR::NC; a++) {
        const int c = lane + 64 * a;
        R.c1[a] = (c < NU) ? F.cu[c] : ((c < NU + NX) ? F.cx[c - NU] : 0.0);
    }
#pragma unroll
    for(int a = 0; a < SR::NJ; a++) {
        const int j = lane + 64 * a;
        R.cxu[a] = (j < NXU) ? F.cxu[j] : 0.0;
        if(FULL) {
#pragma unroll
            for(int i = 0; i < NX; i++) R.txu[a][i] = (j < NXU) ? F.fxu[j + i * NXU] : 0.0;
        }
    }
#pragma unroll
    for(int a = 0; a < SR::NQ; a++) {
        const int o = lane + 64 * a;
        const bool in = o < SUU + SXX, isxx = o >= SUU;
        const int e = isxx ? o - SUU : o;
        R.cq[a] = in ? (isxx ? F.cxx[e] : F.cuu[e]) : 0.0;
        if(FULL) {
            const double *ten = isxx ? F.fxx : F.fuu;
            const int stride = isxx ? SXX : SUU;
#pragma unroll
            for(int i = 0; i < NX; i++) R.tq[a][i] = in ? ten[e + i * stride] : 0.0;
        }
    }
    R.lo = F.lower[lane % NU];
    R.up = F.upper[lane % NU];
#pragma unroll
    for(int i = 0; i < NU; i++) R.u[i] = F.u[i];
}

// One backward step on a wave.  S: LDS block of this wave; F: global fields of the step;
// lout/Kout: where the step's gains go in global memory (trajectory-major).
// Returns the box-QP code (wave-uniform); < 1 aborts the sweep.
template <int NX, int NU, bool FULL, bool HX>
__device__ __forceinline__ int back_step_wave(WaveLds<NX, NU> &S, const StepFields<NX, NU> &F, double *lout,
                                              double *Kout, const double lambda, const int regType, double &dV0,
                                              double &dV1, double &gsum, Prof *pf = nullptr) {
    using SR = StepRegs<NX, NU, FULL>;
    constexpr int SXX = tri(NX), SUU = tri(NU), NXU = NX * NU;
    constexpr int LDX = NX + 1, LDU = NU + 1;  // padded leading dimensions of the LDS copies
    const int lane = threadIdx.x & 63;
    StepRegs<NX, NU, FULL> R;
    load_step<NX, NU, FULL>(R, F);

    // stage fx, fu (read NX resp. NU times each) in LDS
#pragma unroll
    for(int a = 0; a < SR::NFX; a++) {
        const int i = lane + 64 * a;
        if(i < NX * NX) S.fx[(i % NX) + (i / NX) * LDX] = R.fx[a];
    }
#pragma unroll
    for(int a = 0; a < SR::NFU; a++) {
        const int i = lane + 64 * a;
        if(i < NXU) S.fu[(i % NX) + (i / NX) * LDX] = R.fu[a];
    }
    // what the later sections need of this step's record
    const double lo_k = R.lo, up_k = R.up;
    double u_k[NU];
#pragma unroll
    for(int i = 0; i < NU; i++) u_k[i] = R.u[i];
    wave_sync();

    // Qu = cu + fu'Vx ; Qx = cx + fx'Vx   (addMulVec, matMult.c:3-12)
#pragma unroll
    for(int a = 0; a < SR::NC; a++) {
        const int c = lane + 64 * a;
        if(c >= NU + NX) continue;
        double vx[NX], col[NX];
        lds_fetch(vx, S.Vx);
        lds_fetch(col, (c < NU) ? S.fu + c * LDX : S.fx + (c - NU) * LDX);
        reads_before_math();
        const double acc = dot_acc(R.c1[a], vx, col);
        if(c < NU)
            S.Qu[c] = acc;
        else
            S.Qx[c - NU] = acc;
    }
    // T2 = Vxx fu (the `bc` of addMul2Tri and the `ba` of addSquareTri for Quu), T1 = Vxx fx
    for(int o = lane; o < NXU + NX * NX; o += 64) {
        const bool second = o >= NXU;
        const int oo = second ? o - NXU : o;
        const int r = oo % NX, q = oo / NX;
        const double *A = second ? S.fx : S.fu;
        double row[NX], col[NX];
#pragma unroll
        for(int s = 0; s < NX; s++) row[s] = S.Vxx[sy(r, s)];
        lds_fetch(col, A + q * LDX);
        reads_before_math();
        (second ? S.T1 : S.T2)[r + q * LDX] = dot_acc(0.0, row, col);
    }
    wave_sync();

    if(pf) pf->probe(0);
    // Qxu = cxu + fx' T2 (+ sum_i Vx_i fxu_i)        (back_pass.c:90-102)
#pragma unroll
    for(int a = 0; a < SR::NJ; a++) {
        const int j = lane + 64 * a;
        if(j >= NXU) continue;
        const int r = j % NX, q = j / NX;
        double ca[NX], cb[NX];
        lds_fetch(ca, S.fx + r * LDX);
        lds_fetch(cb, S.T2 + q * LDX);
        reads_before_math();
        const double d = dot_acc(0.0, ca, cb);
        double v = R.cxu[a] + d;
        if(FULL) {
            double d1 = 0.0;
            #pragma unroll
            for(int i = 0; i < NX; i++) d1 += S.Vx[i] * R.txu[a][i];
            v += d1;
        }
        S.Qxu[j] = v;
    }
    // Quu = cuu + fu' T2 symmetrised (+ sum_i Vx_i fuu_i) ; Qxx likewise with fx, T1   (back_pass.c:104-131)
#pragma unroll
    for(int a = 0; a < SR::NQ; a++) {
        const int o = lane + 64 * a;
        if(o >= SUU + SXX) continue;
        const bool isxx = o >= SUU;
        const int e = isxx ? o - SUU : o;
        int r, c;
        tri_rc(e, r, c);
        const double *A = isxx ? S.fx : S.fu;
        const double *T = isxx ? S.T1 : S.T2;
        double ar[NX], tc[NX], ac[NX], tr[NX];
        lds_fetch(ar, A + r * LDX);
        lds_fetch(tc, T + c * LDX);
        lds_fetch(ac, A + c * LDX);  // (diagonal entries do not use these two)
        lds_fetch(tr, T + r * LDX);
        reads_before_math();
        double acc = dot_acc(0.0, ar, tc);
        if(r != c) {
            acc = dot_acc(acc, ac, tr);
            acc *= 0.5;
        }
        double v = R.cq[a] + acc;
        if(FULL) {
            double d1 = 0.0;
            #pragma unroll
            for(int i = 0; i < NX; i++) d1 += S.Vx[i] * R.tq[a][i];
            v += d1;
        }
        (isxx ? S.Qxx : S.Quu)[e] = v;
    }
    wave_sync();

    if(pf) pf->probe(1);
    // regularisation (back_pass.c:134-159); regType 2 literally as in the reference
    for(int e = lane; e < SUU; e += 64) {
        int r, c;
        tri_rc(e, r, c);
        double v = S.Quu[e];
        if(regType == 2) {
            double acc = 0.0;
            #pragma unroll
            for(int q = 0; q < NU; q++) acc += S.fu[(sy(q, r) % NX) + (sy(q, r) / NX) * LDX] * S.fu[(sy(q, c) % NX) + (sy(q, c) / NX) * LDX];
            v += acc * lambda;
        }
        if(regType == 1 && r == c) v += lambda;
        S.QuuF[e] = v;
    }
    for(int j = lane; j < NXU; j += 64) {
        double v = S.Qxu[j];
        if(regType == 2) {
            const int i = j % NX, q = j / NX;
            double acc = 0.0;
            #pragma unroll
            for(int s = 0; s < NX; s++) acc += S.fx[s + i * LDX] * S.fu[((s + q * NU) % NX) + ((s + q * NU) / NX) * LDX];
            v += acc * lambda;
        }
        S.Qxur[j] = v;
    }
    wave_sync();

    if(pf) pf->probe(2);
    // box QP, one lane per input (box_qp_rows); warm start: the later step's solution in S.l (back_pass.c:163-166)
    int rc, nf;
    rc = box_qp_rows<NU>(S.QuuF, S.Qu[lane % NU], lo_k, up_k, S.l, S.clamp, S.invH, nf);
    if(pf) pf->probe(3);
    if(rc < 1) return rc;

    // feedback gains (back_pass.c:175-201); invH is in full-index form (see box_qp)
    for(int o = lane; o < NXU; o += 64) {
        const int i = o % NU, q = o / NU;  // K[i + q*NU]
        double v = 0.0;
        int cl[NU];
        double ih[NU], qx[NU];
#pragma unroll
        for(int j = 0; j < NU; j++) {
            cl[j] = S.clamp[j];
            ih[j] = S.invH[sy(i, j)];
        }
        lds_fetch(qx, S.Qxur + q, NX);
        reads_before_math();
        if(S.clamp[i]) {
            if(HX) {
                const double sg = (S.clamp[i] == 1) ? F.lower_sign[i] : F.upper_sign[i];
                const double hx = (S.clamp[i] == 1) ? F.lower_hx[q + i * NX] : F.upper_hx[q + i * NX];
                v -= sg * hx;
            }
        } else {
            #pragma unroll
            for(int j = 0; j < NU; j++) {
                if(!cl[j]) {
                    v -= ih[j] * qx[j];
                } else if(HX) {
                    double w = 0.0;
                    #pragma unroll
                    for(int s = 0; s < NU; s++)
                        if(!S.clamp[s]) w -= S.invH[sy(i, s)] * S.QuuF[sy(s, j)];
                    const double sg = (S.clamp[j] == 1) ? F.lower_sign[j] : F.upper_sign[j];
                    const double hx = (S.clamp[j] == 1) ? F.lower_hx[q + j * NX] : F.upper_hx[q + j * NX];
                    v -= w * (sg * hx);
                }
            }
        }
        S.K[i + q * LDU] = v;
        Kout[o] = v;
    }
    for(int i = lane; i < NU; i += 64) lout[i] = S.l[i];
    wave_sync();

    if(pf) pf->probe(4);
    // expected cost change, redundantly on every lane (back_pass.c:205-214)
    {
        double quv[NU], lv[NU], quu[SUU];
        lds_fetch(quv, S.Qu);
        lds_fetch(lv, S.l);
        lds_fetch(quu, S.Quu);
        reads_before_math();
        dV0 = dot_acc(dV0, quv, lv);
        #pragma unroll
        for(int i = 0; i < NU; i++) {
            double acc = 0.0;
            #pragma unroll
            for(int j = 0; j < NU; j++) acc += lv[j] * quu[sy(j, i)];
            dV1 += 0.5 * lv[i] * acc;
        }
    }

    // Quu*l and Quu*K (the `bc` / `ba` temporaries of addMul2Tri / addSquareTri)
    for(int o = lane; o < NU + NXU; o += 64) {
        if(o < NU) {
            double acc = 0.0;
            #pragma unroll
            for(int s = 0; s < NU; s++) acc += S.Quu[sy(o, s)] * S.l[s];
            S.bc[o] = acc;
        } else {
            const int oo = o - NU, r = oo % NU, c = oo / NU;
            double qrow[NU], kcol[NU];
#pragma unroll
            for(int s = 0; s < NU; s++) qrow[s] = S.Quu[sy(r, s)];
            lds_fetch(kcol, S.K + c * LDU);
            reads_before_math();
            S.ba[r + c * LDU] = dot_acc(0.0, qrow, kcol);
        }
    }
    wave_sync();

    if(pf) pf->probe(5);
    // Vx, Vxx with the unregularised Quu / Qxu (back_pass.c:219-241)
    for(int o = lane; o < NX + SXX; o += 64) {
        if(o < NX) {
            const int i = o;
            double ki[NU], bcv[NU], quv[NU], qi[NU], lv[NU];
            lds_fetch(ki, S.K + i * LDU);
            lds_fetch(bcv, S.bc);
            lds_fetch(quv, S.Qu);
            lds_fetch(qi, S.Qxu + i, NX);
            lds_fetch(lv, S.l);
            const double qx = S.Qx[i];
            reads_before_math();
            const double d = dot_acc(0.0, ki, bcv);
            double v = qx + d;
            v = dot_acc(v, ki, quv);
            v = dot_acc(v, qi, lv);
            S.Vx[i] = v;
        } else {
            const int e = o - NX;
            int r, c;
            tri_rc(e, r, c);
            double kr[NU], kc[NU], bar[NU], bac[NU], qr[NU], qc[NU];
            lds_fetch(kr, S.K + r * LDU);
            lds_fetch(kc, S.K + c * LDU);
            lds_fetch(bac, S.ba + c * LDU);
            lds_fetch(bar, S.ba + r * LDU);
            lds_fetch(qr, S.Qxu + r, NX);
            lds_fetch(qc, S.Qxu + c, NX);
            const double qxx = S.Qxx[e];
            reads_before_math();
            double acc = dot_acc(0.0, kr, bac);
            if(r != c) {
                acc = dot_acc(acc, kc, bar);
                acc *= 0.5;
            }
            double v = qxx + acc;
            // the reference's i-major loop touches packed entry (r,c) first as (i=r,j=c), then as (i=c,j=r)
            if(r == c) {
                #pragma unroll
                for(int q = 0; q < NU; q++) v += (kr[q] * qr[q]) * 2.0;
            } else {
                v = dot_acc(v, kr, qc);
                v = dot_acc(v, kc, qr);
            }
            S.Vxx[e] = v;
        }
    }

    // gradient-norm summand (back_pass.c:246-251)
    double gmax = 0.0;
    #pragma unroll
    for(int i = 0; i < NU; i++) {
        const double gi = fabs(S.l[i]) / (fabs(u_k[i]) + 1.0);
        if(gi > gmax) gmax = gi;
    }
    gsum += gmax;
    wave_sync();
    if(pf) pf->probe(6);
    return rc;
}

}  // namespace ilqg
