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

// rec: this step's trajEl_t in global memory (fields read through the pointers below)
template <int NX, int NU>
struct StepFields {
    const double *cx, *cxx, *cu, *cuu, *cxu, *fx, *fu, *lower, *upper, *fxx, *fuu, *fxu;
    const double *lower_sign, *upper_sign, *lower_hx, *upper_hx, *u;
};

// One backward step on a wave.  S: LDS block of this wave; F: global fields of the step;
// lout/Kout: where the step's gains go in global memory (trajectory-major).
// Returns the box-QP code (wave-uniform); < 1 aborts the sweep.
template <int NX, int NU, bool FULL, bool HX>
__device__ __forceinline__ int back_step_wave(WaveLds<NX, NU> &S, const StepFields<NX, NU> &F, double *lout,
                                              double *Kout, const double lambda, const int regType, double &dV0,
                                              double &dV1, double &gsum, Prof *pf = nullptr) {
    constexpr int SXX = tri(NX), SUU = tri(NU), NXU = NX * NU;
    constexpr int LDX = NX + 1, LDU = NU + 1;  // padded leading dimensions of the LDS copies
    const int lane = threadIdx.x & 63;

    // stage fx, fu (read NX resp. NU times each) in LDS
    for(int i = lane; i < NX * NX; i += 64) S.fx[(i % NX) + (i / NX) * LDX] = F.fx[i];
    for(int i = lane; i < NXU; i += 64) S.fu[(i % NX) + (i / NX) * LDX] = F.fu[i];
    __syncthreads();

    // Qu = cu + fu'Vx ; Qx = cx + fx'Vx   (addMulVec, matMult.c:3-12)
    for(int c = lane; c < NU + NX; c += 64) {
        if(c < NU) {
            double acc = F.cu[c];
            #pragma unroll
            for(int r = 0; r < NX; r++) acc += S.Vx[r] * S.fu[r + c * LDX];
            S.Qu[c] = acc;
        } else {
            const int cc = c - NU;
            double acc = F.cx[cc];
            #pragma unroll
            for(int r = 0; r < NX; r++) acc += S.Vx[r] * S.fx[r + cc * LDX];
            S.Qx[cc] = acc;
        }
    }
    // T2 = Vxx fu (the `bc` of addMul2Tri and the `ba` of addSquareTri for Quu), T1 = Vxx fx
    for(int o = lane; o < NXU + NX * NX; o += 64) {
        const bool second = o >= NXU;
        const int oo = second ? o - NXU : o;
        const int r = oo % NX, q = oo / NX;
        const double *A = second ? S.fx : S.fu;
        double acc = 0.0;
        #pragma unroll
        for(int s = 0; s < NX; s++) acc += S.Vxx[sy(r, s)] * A[s + q * LDX];
        (second ? S.T1 : S.T2)[r + q * LDX] = acc;
    }
    __syncthreads();

    if(pf) pf->probe(0);
    // Qxu = cxu + fx' T2 (+ sum_i Vx_i fxu_i)        (back_pass.c:90-102)
    for(int j = lane; j < NXU; j += 64) {
        const int r = j % NX, q = j / NX;
        double d = 0.0;
        #pragma unroll
        for(int s = 0; s < NX; s++) d += S.fx[s + r * LDX] * S.T2[s + q * LDX];
        double v = F.cxu[j] + d;
        if(FULL) {
            double d1 = 0.0;
            #pragma unroll
            for(int i = 0; i < NX; i++) d1 += S.Vx[i] * F.fxu[j + i * NXU];
            v += d1;
        }
        S.Qxu[j] = v;
    }
    // Quu = cuu + fu' T2 symmetrised (+ sum_i Vx_i fuu_i) ; Qxx likewise with fx, T1   (back_pass.c:104-131)
    for(int o = lane; o < SUU + SXX; o += 64) {
        const bool isxx = o >= SUU;
        const int e = isxx ? o - SUU : o;
        int r, c;
        tri_rc(e, r, c);
        const double *A = isxx ? S.fx : S.fu;
        const double *T = isxx ? S.T1 : S.T2;
        double acc = 0.0;
        #pragma unroll
        for(int s = 0; s < NX; s++) acc += A[s + r * LDX] * T[s + c * LDX];
        if(r != c) {
            #pragma unroll
            for(int s = 0; s < NX; s++) acc += A[s + c * LDX] * T[s + r * LDX];
            acc *= 0.5;
        }
        double v = (isxx ? F.cxx[e] : F.cuu[e]) + acc;
        if(FULL) {
            const double *ten = isxx ? F.fxx : F.fuu;
            const int stride = isxx ? SXX : SUU;
            double d1 = 0.0;
            #pragma unroll
            for(int i = 0; i < NX; i++) d1 += S.Vx[i] * ten[e + i * stride];
            v += d1;
        }
        (isxx ? S.Qxx : S.Quu)[e] = v;
    }
    __syncthreads();

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
    __syncthreads();

    if(pf) pf->probe(2);
    // box QP, redundantly on every lane (wave-uniform data)
    int rc;
    {
        double H[SUU], g[NU], lo[NU], up[NU], x[NU], inv[SUU];
        int cl[NU], nf;
#pragma unroll
        for(int i = 0; i < SUU; i++) H[i] = S.QuuF[i];
        #pragma unroll
        for(int i = 0; i < NU; i++) {
            g[i] = S.Qu[i];
            lo[i] = F.lower[i];
            up[i] = F.upper[i];
            x[i] = S.l[i];  // warm start: the later step's solution (back_pass.c:163-166)
        }
        rc = box_qp_uniform<NU>(H, g, lo, up, x, cl, nf, inv);
        __syncthreads();
        if(lane == 0) {
            #pragma unroll
            for(int i = 0; i < NU; i++) {
                S.l[i] = x[i];
                S.clamp[i] = cl[i];
            }
#pragma unroll
            for(int i = 0; i < SUU; i++) S.invH[i] = inv[i];
        }
        __syncthreads();
    }
    if(pf) pf->probe(3);
    if(rc < 1) return rc;

    // feedback gains (back_pass.c:175-201); invH is in full-index form (see box_qp)
    for(int o = lane; o < NXU; o += 64) {
        const int i = o % NU, q = o / NU;  // K[i + q*NU]
        double v = 0.0;
        if(S.clamp[i]) {
            if(HX) {
                const double sg = (S.clamp[i] == 1) ? F.lower_sign[i] : F.upper_sign[i];
                const double hx = (S.clamp[i] == 1) ? F.lower_hx[q + i * NX] : F.upper_hx[q + i * NX];
                v -= sg * hx;
            }
        } else {
            #pragma unroll
            for(int j = 0; j < NU; j++) {
                if(!S.clamp[j]) {
                    v -= S.invH[sy(i, j)] * S.Qxur[q + j * NX];
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
    __syncthreads();

    if(pf) pf->probe(4);
    // expected cost change, redundantly on every lane (back_pass.c:205-214)
    #pragma unroll
    for(int i = 0; i < NU; i++) dV0 += S.Qu[i] * S.l[i];
    #pragma unroll
    for(int i = 0; i < NU; i++) {
        double acc = 0.0;
        #pragma unroll
        for(int j = 0; j < NU; j++) acc += S.l[j] * S.Quu[sy(j, i)];
        dV1 += 0.5 * S.l[i] * acc;
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
            double acc = 0.0;
            #pragma unroll
            for(int s = 0; s < NU; s++) acc += S.Quu[sy(r, s)] * S.K[s + c * LDU];
            S.ba[r + c * LDU] = acc;
        }
    }
    __syncthreads();

    if(pf) pf->probe(5);
    // Vx, Vxx with the unregularised Quu / Qxu (back_pass.c:219-241)
    for(int o = lane; o < NX + SXX; o += 64) {
        if(o < NX) {
            const int i = o;
            double d = 0.0;
            #pragma unroll
            for(int s = 0; s < NU; s++) d += S.K[s + i * LDU] * S.bc[s];
            double v = S.Qx[i] + d;
            #pragma unroll
            for(int j = 0; j < NU; j++) v += S.K[j + i * LDU] * S.Qu[j];
            #pragma unroll
            for(int j = 0; j < NU; j++) v += S.Qxu[i + j * NX] * S.l[j];
            S.Vx[i] = v;
        } else {
            const int e = o - NX;
            int r, c;
            tri_rc(e, r, c);
            double acc = 0.0;
            #pragma unroll
            for(int s = 0; s < NU; s++) acc += S.K[s + r * LDU] * S.ba[s + c * LDU];
            if(r != c) {
                #pragma unroll
                for(int s = 0; s < NU; s++) acc += S.K[s + c * LDU] * S.ba[s + r * LDU];
                acc *= 0.5;
            }
            double v = S.Qxx[e] + acc;
            // the reference's i-major loop touches packed entry (r,c) first as (i=r,j=c), then as (i=c,j=r)
            if(r == c) {
                #pragma unroll
                for(int q = 0; q < NU; q++) v += (S.K[q + r * LDU] * S.Qxu[r + q * NX]) * 2.0;
            } else {
                #pragma unroll
                for(int q = 0; q < NU; q++) v += S.K[q + r * LDU] * S.Qxu[c + q * NX];
                #pragma unroll
                for(int q = 0; q < NU; q++) v += S.K[q + c * LDU] * S.Qxu[r + q * NX];
            }
            S.Vxx[e] = v;
        }
    }

    // gradient-norm summand (back_pass.c:246-251)
    double gmax = 0.0;
    #pragma unroll
    for(int i = 0; i < NU; i++) {
        const double gi = fabs(S.l[i]) / (fabs(F.u[i]) + 1.0);
        if(gi > gmax) gmax = gi;
    }
    gsum += gmax;
    __syncthreads();
    if(pf) pf->probe(6);
    return rc;
}

}  // namespace ilqg
