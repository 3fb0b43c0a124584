// "Row mapping" of the backward step for the one-wavefront-per-trajectory kernels (N_X <= 16, N_U <= 16).
//
// The 64 lanes of the wavefront are 4 DPP rows of 16 lanes: lane (g, c), g = lane / 16, c = lane % 16, owns the
// entries (4g + j, c), j = 0..3, of every matrix product of the step.  A product C = A B is evaluated as
//
//     C[4g+j, c] = sum_s A[4g+j, s] * B[s, c]            s ascending, as in the reference (matMult.c:14-72)
//
// with B[s, c] in the lane's own registers (column c of B) and A[4g+j, s] read from lane (g, s), which holds it as
// ITS entry (4g+j, s): one v_fmac_f64_dpp with row_newbcast:s per multiply-add, no LDS operand at all (measured,
// tools/ubench/dpp_row_fma.hip: 2.6 ns per instruction and wavefront, the rate of a plain v_fma_f64; the
// two-instruction form v_mov_b64_dpp + v_fma_f64 takes 6.3 ns).  LDS is only the place where a product's result
// changes layout (4 rows per lane group -> whole column per lane) for the next product: ~100 LDS accesses per
// lane and step where the first version of the wave mapping (ilqg_wave.hpp: one output element per lane, both
// operands of every multiply-add from LDS) needed ~900.
//
// Summation order and temporaries are the reference's (back_pass.c:80-241), so the results equal the lane mapping's
// and the CPU's up to FMA contraction; the -ffp-contract=off build (ILQG_STRICT_FP) multiplies and adds separately.
// Symmetric results (Quu, Qxx, Vxx) are produced by the lanes with 4g+j <= c in the reference's order of the two
// half sums; the other lanes run the same instructions and their results are discarded.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "ilqg_device.hpp"
#include "ilqg_wave.hpp"

namespace ilqg {

template <int I, int N, class F>
ILQG_DEV void static_for(F &&f) {
    if constexpr(I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// acc += (a of lane S of this lane's 16-lane row) * b
// The statements are `volatile` for one reason: a row broadcast reads OTHER lanes' registers, so it must execute with
// every lane of the row active.  A plain asm is a pure function of its operands to the optimiser, which sinks it into a
// conditional block if its result is only used there (the guarded stores of the triangular results) — where the lanes
// that hold the operand may be masked off (seen: the -ffp-contract=off build of the stored-tensor kernel, 1e-4 off).
template <int S>
ILQG_DEV void row_fma(double &acc, const double a, const double b) {
#ifdef ILQG_STRICT_FP
    double t;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a), "n"(S));
    acc = acc + t * b;
#else
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(S));
#endif
}
// A value written by a VALU instruction must not be read through DPP in the next two issue slots (the hazard
// recogniser does not look into inline assembly): every array that is about to be broadcast passes through here.
ILQG_DEV void dpp_source(double &a) { asm volatile("s_nop 1" : "+v"(a)); }
ILQG_DEV void dpp_source(double (&a)[4]) { asm volatile("s_nop 1" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); }

// acc[j] += sum_{s < K} (a[j] of lane s) * b[s]        the product of the header comment, 4 rows per lane
template <int K>
ILQG_DEV void row_product(double (&acc)[4], const double (&a)[4], const double *b) {
    static_for<0, K>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for(int j = 0; j < 4; j++) row_fma<s>(acc[j], a[j], b[s]);
    });
}
// acc[j] += sum_{s < K} a[s] * (b[j] of lane s)        (own operand first: same products, same order)
template <int K>
ILQG_DEV void row_product_t(double (&acc)[4], const double *a, const double (&b)[4]) {
    static_for<0, K>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for(int j = 0; j < 4; j++) row_fma<s>(acc[j], b[j], a[s]);
    });
}
// acc += sum_{s < K} (a of lane s) * b[s]
template <int K>
ILQG_DEV void row_dot(double &acc, const double a, const double *b) {
    static_for<0, K>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        row_fma<s>(acc, a, b[s]);
    });
}

// acc += (a of lane S of the row) * (b of lane S of the row)
template <int S>
ILQG_DEV void row_fma2(double &acc, const double a, const double b) {
    double t;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(b), "n"(S));
    row_fma<S>(acc, a, t);
}
// the value of lane S of the row
template <int S>
ILQG_DEV double row_get(const double a) {
    double t;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a), "n"(S));
    return t;
}

// ---------------------------------------------------------------------------
// boxQP.c:39-238 for the wave mapping, cooperative as box_qp_rows (ilqg_wave.hpp: lane i owns variable i — its x, g,
// limits, clamp flag, row i of H and of the inverse, column i of the Cholesky factor), with every exchange between the
// lanes a row broadcast inside the consuming multiply-add instead of two v_readlane and a scalar operand: lane j of
// EVERY 16-lane row holds variable j (the other lanes mirror lane mod M), so row_newbcast:j reaches it from anywhere.
// Expression trees, operand order and exits are those of box_qp_rows, i.e. of the reference, bit for bit.
// ---------------------------------------------------------------------------
template <int M>
ILQG_DEV int box_qp_row(const double *Hpacked /* LDS */, const double g, const double lower, const double upper,
                        double *S_l, int *S_clamp, double *S_invH, int &n_free_out) {
    static_assert(M <= 16, "one 16-lane row holds all variables");
    constexpr int T = tri(M);
    const int lane = threadIdx.x & 63, me = (lane & 15) % M;
    const unsigned all = (1u << M) - 1u;
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
    auto qp_value = [&](double y) {
        dpp_source(y);
        double hx = 0.0;
        row_dot<M>(hx, y, Hrow);
        double w = g + 0.5 * hx;
        dpp_source(w);
        double v = 0.0;
        static_for<0, M>([&](auto ic) { row_fma2<decltype(ic)::value>(v, y, w); });
        return v;
    };

    double value = qp_value(x), oldvalue = 0.0;
    int rc = 1;  // max_iter iterations (boxQP.c:237)
    for(int iter = 0; iter < max_iter; iter++) {
        if(iter > 0 && (oldvalue - value) < min_rel_improve * fabs(oldvalue)) { rc = 4; break; }
        oldvalue = value;

        // gradient and clamped set (boxQP.c:101-124)
        dpp_source(x);
        double hx = 0.0;
        row_dot<M>(hx, x, Hrow);
        double grad = g + hx;
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
        dpp_source(grad);
        double gnorm = 0.0;
        static_for<0, M>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if(!((cm >> i) & 1u)) row_fma2<i>(gnorm, grad, grad);
        });

        if(iter == 0 || changed) {
            // Cholesky of the Hessian with clamped rows and columns replaced by identity (boxQP.c:131-160,
            // cholesky.c:6-27): lane i computes column i of U, row j in step j
            bool pd = true;
            static_for<0, M>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                double dot = 0.0;
                static_for<0, j>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    row_fma<j>(dot, Ucol[k], Ucol[k]);  // U[k, j] * U[k, me]
                });
                const bool masked = clamp != 0 || ((cm >> j) & 1u);
                const double a = masked ? ((me == j) ? 1.0 : 0.0) : Hrow[j];
                double sv = a - dot;
                dpp_source(sv);
                const double piv = row_get<j>(sv);
                if(piv <= 0.0) pd = false;
                const double d = sqrt(piv);
                Ucol[j] = (me == j) ? d : ((me > j) ? 1.0 / d * sv : 0.0);
                dpp_source(Ucol[j]);
            });
            if(!pd) { rc = -1; break; }
            // explicit inverse (cholesky.c:51-74): lane l solves U'U y = e_l; y[k] for k >= l is row l of the inverse
            double y[M], ny[M];
            static_for<0, M>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                double v = (k == me) ? 1.0 : 0.0;
                static_for<0, k>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    row_fma<k>(v, Ucol[i], ny[i]);  // v -= y[i] * U[i, k]    (y[i] = 0 for i < l: exact zeros)
                });
                y[k] = v / row_get<k>(Ucol[k]);
                ny[k] = -y[k];
            });
            static_for<0, M>([&](auto kr) {
                constexpr int k = M - 1 - decltype(kr)::value;
                double v = y[k];
                static_for<k + 1, M>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    row_fma<i>(v, Ucol[k], ny[i]);  // v -= y[i] * U[k, i]
                });
                y[k] = v / row_get<k>(Ucol[k]);
                ny[k] = -y[k];
            });
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
        static_for<0, M>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if((cm >> j) & 1u) row_fma<j>(hc, x, Hrow[j]);
        });
        double gc = g + hc;
        dpp_source(gc);
        double sr = -x;
        static_for<0, M>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if(!((cm >> j) & 1u)) row_fma<j>(sr, gc, -invrow[j]);
        });
        double search = clamp ? 0.0 : sr;
        dpp_source(search);

        double sdotg = 0.0;
        static_for<0, M>([&](auto ic) { row_fma2<decltype(ic)::value>(sdotg, search, grad); });
        if(sdotg >= 0.0) { rc = -2; break; }

        // Armijo line search (boxQP.c:203-228)
        double step = 1.0, vc, xc;
        bool tiny = false;
        for(;;) {
            xc = x + step * search;
            if(xc > upper) xc = upper;
            if(xc < lower) xc = lower;
            vc = qp_value(xc);
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

template <int NX, int NU>
struct RowLds {
    static constexpr int SXX = tri(NX), SUU = tri(NU), NXU = NX * NU;
    static constexpr int LDX = NX + 1, LDU = NU + 1;  // padded leading dimensions: column reads hit distinct banks
    double Vx[NX], Vxx[SXX];
    double fx[LDX * NX], fu[LDX * NU];   // NX x NX, NX x NU
    double T1[LDX * NX], T2[LDX * NU];   // Vxx fx, Vxx fu
    double Qxu[LDX * NU];                // NX x NU
    double Qu[NU], Quu[SUU], QuuF[SUU];
    double l[NU];
    union {
        struct {                         // from the box QP to the end of the step
            double K[LDU * NX], BA[LDU * NX];  // gains (NU x NX), Quu K
            double invH[SUU];
        };
        struct {                         // from the start of the step to the assembly of the Q blocks
            double dxx[SXX], duu[SUU], dxu[NXU];  // sum_i Vx[i] * (fxx_i, fuu_i, fxu_i), in the arrays' own order
        };
    };
    int clamp[NU];
};

// Where a step's derivative entries come from.  RecordSource: the trajEl_t the generated calc_derivs code has
// written to HBM (k_derivs_wave).  Every accessor returns the entry THIS lane needs; out-of-range lanes get an
// in-range address (their results are never used).
template <int NX, int NU, bool FULL>
struct RecordSource {
    static constexpr int SXX = tri(NX), SUU = tri(NU), NXU = NX * NU;
    StepFields<NX, NU> F;
    ILQG_DEV double fx(int i) const { return F.fx[i]; }
    ILQG_DEV double fu(int i) const { return F.fu[i]; }
    ILQG_DEV double cx(int i) const { return F.cx[i]; }
    ILQG_DEV double cu(int i) const { return F.cu[i]; }
    ILQG_DEV double cxu(int i) const { return F.cxu[i]; }
    ILQG_DEV double cxx(int e) const { return F.cxx[e]; }
    ILQG_DEV double cuu(int e) const { return F.cuu[e]; }
    // the tensors slice by slice: slice(i) is whatever identifies slice i, f??(slice, e) entry e of it
    ILQG_DEV int slice(int i) const { return i; }
    ILQG_DEV double fxu(int i, int e) const { return F.fxu[e + i * NXU]; }
    ILQG_DEV double fuu(int i, int e) const { return F.fuu[e + i * SUU]; }
    ILQG_DEV double fxx(int i, int e) const { return F.fxx[e + i * SXX]; }
};

// One backward step.  S: LDS block of the wavefront (Vx, Vxx, l carry over between steps); lout / Kout: the step's
// gains in global memory.  Returns the box-QP code (wave-uniform); < 1 abandons the sweep (back_pass.c:168-171).
template <int NX, int NU, bool FULL, bool HX, class Source>
__device__ __forceinline__ int back_step_row(RowLds<NX, NU> &S, const Source &D, const StepFields<NX, NU> &F, double *lout,
                                             double *Kout, const double lambda, const int regType, double &dV0,
                                             double &dV1, double &gsum, Prof *pf = nullptr) {
    static_assert(NX <= 16 && NU <= 16, "one 16-lane row per matrix row block");
    constexpr int SXX = tri(NX), SUU = tri(NU), NXU = NX * NU;
    constexpr int LDX = NX + 1, LDU = NU + 1;
    // Everything below that depends on the lane alone (rows, columns, the LDS and record addresses made of them) is
    // loop invariant in the sweep, and the optimiser would move all of it — some hundred values — in front of the
    // loop and keep it in registers, i.e. spill it.  The lane number is made opaque here, once per step: the
    // addresses are recomputed by a few integer instructions where they are used.
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4, c = lane & 15;
    const int cx_ = (c < NX) ? c : 0, cu_ = (c < NU) ? c : 0;  // this lane's column, clamped into range
    int a[4], ax[4], au[4];                                    // this lane's rows 4g+j, clamped into range
#pragma unroll
    for(int j = 0; j < 4; j++) {
        a[j] = 4 * g + j;
        ax[j] = (a[j] < NX) ? a[j] : 0;
        au[j] = (a[j] < NU) ? a[j] : 0;
    }

    // ---- value function of step k+1
    double vxl = S.Vx[cx_];  // Vx[c]
    double vxx_r[4];         // Vxx[4g+j, c]
#pragma unroll
    for(int j = 0; j < 4; j++) vxx_r[j] = S.Vxx[sy(ax[j], cx_)];
    dpp_source(vxl);
    dpp_source(vxx_r);

    // ---- the step's record.  Everything but the tensors is requested here and consumed behind the tensor
    // contraction, which hides its latency: fx, fu (on their way to LDS), this lane's entries of the cost derivatives
    constexpr int NFX = (NX * NX + 63) / 64, NFU = (NXU + 63) / 64;
    double fx_in[NFX], fu_in[NFU];
#pragma unroll
    for(int q = 0; q < NFX; q++) fx_in[q] = D.fx((lane + 64 * q < NX * NX) ? lane + 64 * q : 0);
#pragma unroll
    for(int q = 0; q < NFU; q++) fu_in[q] = D.fu((lane + 64 * q < NXU) ? lane + 64 * q : 0);
    const double cxl = D.cx(cx_), cul = D.cu(cu_);
    // this lane's 4 entries of a packed triangle are consecutive: rows 4g..4g+3 of column c where 4g <= c (rows
    // beyond the diagonal, and all four where 4g > c, are never used: any address inside the array will do)
    const int rxx = (4 * g <= cx_) ? 4 * g : 0, ruu = (4 * g <= cu_) ? 4 * g : 0;
    int exu[4], euu[4], exx[4];
#pragma unroll
    for(int j = 0; j < 4; j++) {
        exu[j] = ax[j] + cu_ * NX;
        euu[j] = (ut(ruu, cu_) + j < SUU) ? ut(ruu, cu_) + j : SUU - 1;
        exx[j] = (ut(rxx, cx_) + j < SXX) ? ut(rxx, cx_) + j : SXX - 1;
    }
    double cxu_e[4], cuu_e[4], cxx_e[4];
#pragma unroll
    for(int j = 0; j < 4; j++) {
        cxu_e[j] = D.cxu(exu[j]);
        cuu_e[j] = D.cuu(euu[j]);
        cxx_e[j] = D.cxx(exx[j]);
    }
    const double lo_k = F.lower[(lane & 15) % NU], up_k = F.upper[(lane & 15) % NU];
    const double u_l = F.u[cu_];

    // ---- second-order terms of the dynamics (back_pass.c:95-131): d[e] = sum_i Vx[i] * tensor_i[e], i ascending.
    // Here the lanes take the entries e of a tensor slice in the array's own order (e = lane, lane + 64, ...: consecutive
    // lanes on consecutive doubles, every lane busy) and hand the sums to the lanes that own them through LDS.
    if(FULL) {
        constexpr int NTX = (SXX + 63) / 64, NTU = (SUU + 63) / 64, NTC = (NXU + 63) / 64;
        double dxx[NTX], duu[NTU], dxu[NTC];
#pragma unroll
        for(int q = 0; q < NTX; q++) dxx[q] = 0.0;
#pragma unroll
        for(int q = 0; q < NTU; q++) duu[q] = 0.0;
#pragma unroll
        for(int q = 0; q < NTC; q++) dxu[q] = 0.0;
#ifndef ILQG_ROW_UNROLL
#define ILQG_ROW_UNROLL 8
#endif
#pragma unroll ILQG_ROW_UNROLL
        for(int i = 0; i < NX; i++) {
            const double vxi = lane_bcast(vxl, i);  // Vx[i] (lane i holds it)
            const auto slice = D.slice(i);
#pragma unroll
            for(int q = 0; q < NTC; q++) dxu[q] += vxi * D.fxu(slice, (lane + 64 * q < NXU) ? lane + 64 * q : 0);
#pragma unroll
            for(int q = 0; q < NTU; q++) duu[q] += vxi * D.fuu(slice, (lane + 64 * q < SUU) ? lane + 64 * q : 0);
#pragma unroll
            for(int q = 0; q < NTX; q++) dxx[q] += vxi * D.fxx(slice, (lane + 64 * q < SXX) ? lane + 64 * q : 0);
        }
#pragma unroll
        for(int q = 0; q < NTC; q++)
            if(lane + 64 * q < NXU) S.dxu[lane + 64 * q] = dxu[q];
#pragma unroll
        for(int q = 0; q < NTU; q++)
            if(lane + 64 * q < SUU) S.duu[lane + 64 * q] = duu[q];
#pragma unroll
        for(int q = 0; q < NTX; q++)
            if(lane + 64 * q < SXX) S.dxx[lane + 64 * q] = dxx[q];
    }
    // fx, fu into LDS
#pragma unroll
    for(int q = 0; q < NFX; q++) {
        const int i = lane + 64 * q;
        if(i < NX * NX) S.fx[(i % NX) + (i / NX) * LDX] = fx_in[q];
    }
#pragma unroll
    for(int q = 0; q < NFU; q++) {
        const int i = lane + 64 * q;
        if(i < NXU) S.fu[(i % NX) + (i / NX) * LDX] = fu_in[q];
    }
    wave_sync();

    // ---- T1 = Vxx fx, T2 = Vxx fu (the `ba` / `bc` temporaries of matMult.c); Qu = cu + fu'Vx, Qx = cx + fx'Vx
    double qxl = cxl, qul = cul;  // Qx[c], Qu[c]
    {
        double fxc[NX];  // column c of fx
#pragma unroll
        for(int s = 0; s < NX; s++) fxc[s] = S.fx[s + cx_ * LDX];
        row_dot<NX>(qxl, vxl, fxc);
        double t1[4] = {0.0, 0.0, 0.0, 0.0};
        row_product<NX>(t1, vxx_r, fxc);
#pragma unroll
        for(int j = 0; j < 4; j++)
            if(a[j] < NX && c < NX) S.T1[a[j] + c * LDX] = t1[j];
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        double fuc[NX];  // column c of fu
#pragma unroll
        for(int s = 0; s < NX; s++) fuc[s] = S.fu[s + cu_ * LDX];
        row_dot<NX>(qul, vxl, fuc);
        double t2[4] = {0.0, 0.0, 0.0, 0.0};
        row_product<NX>(t2, vxx_r, fuc);
#pragma unroll
        for(int j = 0; j < 4; j++)
            if(a[j] < NX && c < NU) S.T2[a[j] + c * LDX] = t2[j];
    }
    if(g == 0 && c < NU) S.Qu[c] = qul;
    wave_sync();
    if(pf) pf->probe(0);

    // ---- Qxu = cxu + fx'T2, Quu = cuu + fu'T2 (symmetrised), Qxx = cxx + fx'T1 (symmetrised)   back_pass.c:90-131
    // Three blocks, each with its operands read from LDS right in front of it and nothing but its results alive behind
    // it (the scheduling barriers keep the reads of a later block from being issued — and held in registers — early).
    double qxu_r[4], qxx_r[4];  // Qxu[4g+j, c], Qxx[4g+j, c]
    {
        double t2c[NX];  // column c of T2
#pragma unroll
        for(int s = 0; s < NX; s++) t2c[s] = S.T2[s + cu_ * LDX];
        {
            double fxt[4];  // fx[c, 4g+j]: what the row mates read
#pragma unroll
            for(int j = 0; j < 4; j++) fxt[j] = S.fx[cx_ + ax[j] * LDX];
            dpp_source(fxt);
            double dxu[4] = {0.0, 0.0, 0.0, 0.0};
            row_product<NX>(dxu, fxt, t2c);  // sum_si fx[si, r] * T2[si, c]
#pragma unroll
            for(int j = 0; j < 4; j++) {
                double v = cxu_e[j] + dxu[j];
                if(FULL) v += S.dxu[exu[j]];
                qxu_r[j] = v;
                if(a[j] < NX && c < NU) S.Qxu[a[j] + c * LDX] = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            double fuc2[NX], fut[4], t2t[4];  // column c of fu; fu[c, 4g+j], T2[c, 4g+j]
#pragma unroll
            for(int s = 0; s < NX; s++) fuc2[s] = S.fu[s + cu_ * LDX];
#pragma unroll
            for(int j = 0; j < 4; j++) {
                fut[j] = S.fu[cx_ + au[j] * LDX];
                t2t[j] = S.T2[cx_ + au[j] * LDX];
            }
            dpp_source(fut);
            dpp_source(t2t);
            double suu[4] = {0.0, 0.0, 0.0, 0.0}, suu_d[4];
            row_product<NX>(suu, fut, t2c);     // sum_si fu[si, r] * T2[si, c] ...
#pragma unroll
            for(int j = 0; j < 4; j++) suu_d[j] = suu[j];  // a diagonal entry stops here
            row_product_t<NX>(suu, fuc2, t2t);  // ... + sum_si fu[si, c] * T2[si, r]      (r < c)
#pragma unroll
            for(int j = 0; j < 4; j++) {
                double v = cuu_e[j] + ((a[j] == c) ? suu_d[j] : suu[j] * 0.5);
                if(FULL) v += S.duu[euu[j]];
                if(a[j] <= c && c < NU) {
                    S.Quu[ut(au[j], cu_)] = v;
                    S.QuuF[ut(au[j], cu_)] = (regType == 1 && a[j] == c) ? v + lambda : v;
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        double t1c[NX], fxc2[NX], fxt[4], t1t[4];  // columns c of T1 and fx; fx[c, 4g+j], T1[c, 4g+j]
#pragma unroll
        for(int s = 0; s < NX; s++) {
            t1c[s] = S.T1[s + cx_ * LDX];
            fxc2[s] = S.fx[s + cx_ * LDX];
        }
#pragma unroll
        for(int j = 0; j < 4; j++) {
            fxt[j] = S.fx[cx_ + ax[j] * LDX];
            t1t[j] = S.T1[cx_ + ax[j] * LDX];
        }
        dpp_source(fxt);
        dpp_source(t1t);
        double sxx[4] = {0.0, 0.0, 0.0, 0.0}, sxx_d[4];
        row_product<NX>(sxx, fxt, t1c);
#pragma unroll
        for(int j = 0; j < 4; j++) sxx_d[j] = sxx[j];
        row_product_t<NX>(sxx, fxc2, t1t);
#pragma unroll
        for(int j = 0; j < 4; j++) {
            double v = cxx_e[j] + ((a[j] == c) ? sxx_d[j] : sxx[j] * 0.5);
            if(FULL) v += S.dxx[exx[j]];
            qxx_r[j] = v;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    wave_sync();
    if(pf) pf->probe(1);

    // regType 2, literally as in the reference (back_pass.c:136-155)
    double qxur_r[4];  // regularised Qxu[4g+j, c]
#pragma unroll
    for(int j = 0; j < 4; j++) qxur_r[j] = qxu_r[j];
    if(regType == 2) {
        for(int e = lane; e < SUU; e += 64) {
            int r, cc;
            tri_rc(e, r, cc);
            double acc = 0.0;
#pragma unroll
            for(int q = 0; q < NU; q++)
                acc += S.fu[(sy(q, r) % NX) + (sy(q, r) / NX) * LDX] * S.fu[(sy(q, cc) % NX) + (sy(q, cc) / NX) * LDX];
            S.QuuF[e] = S.Quu[e] + acc * lambda;
        }
#pragma unroll
        for(int j = 0; j < 4; j++) {
            const int i = ax[j], q = cu_;
            double acc = 0.0;
#pragma unroll
            for(int s = 0; s < NX; s++) acc += S.fx[s + i * LDX] * S.fu[((s + q * NU) % NX) + ((s + q * NU) / NX) * LDX];
            qxur_r[j] = qxu_r[j] + acc * lambda;
        }
        wave_sync();
    }
    if(pf) pf->probe(2);

    // ---- box QP, one lane per input (box_qp_rows); warm start: the later step's solution in S.l (back_pass.c:163-166)
    int nf;
    const int rc = box_qp_row<NU>(S.QuuF, S.Qu[(lane & 15) % NU], lo_k, up_k, S.l, S.clamp, S.invH, nf);
    if(pf) pf->probe(3);
    if(rc < 1) return rc;

    // ---- feedback gains (back_pass.c:175-201): lane (g, i) computes K[i, 4g+j], input i = c, state 4g+j
    double kt[4] = {0.0, 0.0, 0.0, 0.0};
    {
        int cl[NU];
        double nih[NU];  // -invH[i, jj]
#pragma unroll
        for(int jj = 0; jj < NU; jj++) {
            cl[jj] = __builtin_amdgcn_readfirstlane(S.clamp[jj]);  // wave-uniform: a scalar branch below
            nih[jj] = -S.invH[sy(cu_, jj)];
        }
        dpp_source(qxur_r);
        static_for<0, NU>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            if(!cl[jj]) {  // wave-uniform
#pragma unroll
                for(int j = 0; j < 4; j++) row_fma<jj>(kt[j], qxur_r[j], nih[jj]);
            } else if(HX) {
                double w = 0.0;
#pragma unroll
                for(int s = 0; s < NU; s++)
                    if(!cl[s]) w -= S.invH[sy(cu_, s)] * S.QuuF[sy(s, jj)];
                const double sg = (cl[jj] == 1) ? F.lower_sign[jj] : F.upper_sign[jj];
#pragma unroll
                for(int j = 0; j < 4; j++) {
                    const double hx = (cl[jj] == 1) ? F.lower_hx[ax[j] + jj * NX] : F.upper_hx[ax[j] + jj * NX];
                    kt[j] -= w * (sg * hx);
                }
            }
        });
        const int mine = S.clamp[cu_];
        if(mine) {  // a clamped input follows its limit (back_pass.c:186-190)
#pragma unroll
            for(int j = 0; j < 4; j++) {
                double v = 0.0;
                if(HX) {
                    const double sg = (mine == 1) ? F.lower_sign[cu_] : F.upper_sign[cu_];
                    const double hx = (mine == 1) ? F.lower_hx[ax[j] + cu_ * NX] : F.upper_hx[ax[j] + cu_ * NX];
                    v -= sg * hx;
                }
                kt[j] = v;
            }
        }
#pragma unroll
        for(int j = 0; j < 4; j++)
            if(a[j] < NX && c < NU) {
                S.K[c + a[j] * LDU] = kt[j];
                Kout[c + a[j] * NU] = kt[j];
            }
        if(g == 0 && c < NU) lout[c] = S.l[c];
    }
    wave_sync();
    if(pf) pf->probe(4);

    // ---- Quu l, Quu K; expected cost change (back_pass.c:205-214)
    double kc[NU];           // column c of K: K[s, c], state c
    double ql[NU];           // Quu[c, s]: row c of Quu
    double quu_r[4];         // Quu[4g+j, c]
    double ll = S.l[cu_];    // l[c]
#pragma unroll
    for(int s = 0; s < NU; s++) {
        kc[s] = S.K[s + cx_ * LDU];
        ql[s] = S.Quu[sy(cu_, s)];
    }
#pragma unroll
    for(int j = 0; j < 4; j++) quu_r[j] = S.Quu[sy(au[j], cu_)];
    dpp_source(quu_r);
    dpp_source(ll);
    double bcl = 0.0;        // (Quu l)[c]
    row_dot<NU>(bcl, ll, ql);
    {
        double ba[4] = {0.0, 0.0, 0.0, 0.0};
        row_product<NU>(ba, quu_r, kc);  // (Quu K)[4g+j, c]
#pragma unroll
        for(int j = 0; j < 4; j++)
            if(a[j] < NU && c < NX) S.BA[a[j] + c * LDU] = ba[j];
    }
    // dV += [l'Qu, 0.5 l'Quu l], term by term over the inputs
#pragma unroll
    for(int i = 0; i < NU; i++) {
        const double qi = lane_bcast(qul, i), li = lane_bcast(ll, i), bi = lane_bcast(bcl, i);
        dV0 += qi * li;
        dV1 += 0.5 * li * bi;
    }
    wave_sync();
    if(pf) pf->probe(5);

    // ---- Vx, Vxx with the unregularised Quu / Qxu (back_pass.c:219-241)
    {
        double bac[NU], qxc[NU];  // column c of Quu K; row c of Qxu
        double bat[4];            // (Quu K)[c, 4g+j]
#pragma unroll
        for(int s = 0; s < NU; s++) {
            bac[s] = S.BA[s + cx_ * LDU];
            qxc[s] = S.Qxu[cx_ + s * LDX];
        }
#pragma unroll
        for(int j = 0; j < 4; j++) bat[j] = S.BA[cu_ + ax[j] * LDU];
        dpp_source(bat);
        dpp_source(kt);
        dpp_source(qxu_r);
        dpp_source(bcl);
        dpp_source(qul);

        // Vx[c] = Qx[c] + K[:, c]'(Quu l) + K[:, c]'Qu + Qxu[c, :] l
        double d = 0.0;
        row_dot<NU>(d, bcl, kc);
        double vx = qxl + d;
        row_dot<NU>(vx, qul, kc);
        row_dot<NU>(vx, ll, qxc);

        // Vxx[r, c], r = 4g+j <= c
        double sq[4] = {0.0, 0.0, 0.0, 0.0}, sq_d[4];
        row_product<NU>(sq, kt, bac);     // sum_si K[si, r] (Quu K)[si, c]
#pragma unroll
        for(int j = 0; j < 4; j++) sq_d[j] = sq[j];
        row_product_t<NU>(sq, kc, bat);   // + sum_si K[si, c] (Quu K)[si, r]
        double vv[4], qx2[NU];
#pragma unroll
        for(int s = 0; s < NU; s++) qx2[s] = qxc[s] * 2.0;
#pragma unroll
        for(int j = 0; j < 4; j++) vv[j] = qxx_r[j] + ((a[j] == c) ? sq_d[j] : sq[j] * 0.5);
        // the reference's loop nest touches packed entry (r, c) first as (i = r, j = c), then as (i = c, j = r);
        // a diagonal entry once, with the term doubled
        double vd[4];
#pragma unroll
        for(int j = 0; j < 4; j++) vd[j] = vv[j];
        row_product<NU>(vd, kt, qx2);
        row_product<NU>(vv, kt, qxc);
        row_product_t<NU>(vv, kc, qxu_r);
#pragma unroll
        for(int j = 0; j < 4; j++)
            if(a[j] <= c && c < NX) S.Vxx[ut(ax[j], cx_)] = (a[j] == c) ? vd[j] : vv[j];
        if(g == 0 && c < NX) S.Vx[c] = vx;
    }

    // gradient-norm summand (back_pass.c:246-251)
    {
        const double gl = fabs(ll) / (fabs(u_l) + 1.0);
        double gmax = 0.0;
#pragma unroll
        for(int i = 0; i < NU; i++) {
            const double gi = lane_bcast(gl, i);
            if(gi > gmax) gmax = gi;
        }
        gsum += gmax;
    }
    wave_sync();
    if(pf) pf->probe(6);
    return rc;
}

}  // namespace ilqg
