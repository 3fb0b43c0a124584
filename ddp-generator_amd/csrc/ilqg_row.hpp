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
//
// The kernel is bound by the NUMBER of vector instructions it issues (two wavefronts per SIMD keep it ~80 % busy), and
// an integer instruction costs what a 64-bit multiply-add costs.  So the step spends as few as it can on addresses:
//   * symmetric matrices live in LDS as full squares (both triangles stored): every access is base + constant, where a
//     packed triangle needs a multiply, a compare and a select per index;
//   * every group of LDS accesses goes through ONE address register made of a wave-uniform part (block + member) and
//     a lane part, given to the optimiser as opaque (lds_base): the accesses then carry their distance as immediate
//     offsets — left to itself the optimiser re-derives a base per pair of accesses;
//   * the record of the step is read as (uniform base) + (32-bit lane offset), the form global loads take without
//     64-bit address arithmetic;
//   * values every lane needs from one lane of its row (Vx[i], l[i], ...) are broadcast inside the consuming
//     multiply-add (row_newbcast) instead of through two v_readlane.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
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
// acc -= (a of lane S of the row) * b          (the sign rides on the operand: no instruction for the negation)
template <int S>
ILQG_DEV void row_fnma(double &acc, const double a, const double b) {
#ifdef ILQG_STRICT_FP
    double t;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a), "n"(S));
    acc = acc + t * -b;
#else
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(S));
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
// LDS addresses.  An LDS pointer is a 32-bit number; LdsBase is one the optimiser has to take as it is (the `asm`),
// so that p[i] with a constant i becomes the instruction's immediate offset and a group of accesses shares ONE
// address register.
// ---------------------------------------------------------------------------
using lds_double = __attribute__((address_space(3))) double;
ILQG_DEV unsigned lds_addr(const void *p) {  // p: a generic pointer INTO LDS
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}
struct LdsBase {
    unsigned a;
    ILQG_DEV lds_double &operator[](int i) const { return ((lds_double *)(uintptr_t)a)[i]; }
    // a read the optimiser neither merges with a neighbour (a merged pair takes small offsets only and gets a new
    // base register computed for it) nor moves: offset i (doubles) is the instruction's immediate
    ILQG_DEV double fetch(int i) const { return ((volatile lds_double *)(uintptr_t)a)[i]; }
};
ILQG_DEV LdsBase lds_base(unsigned a) {
    asm("" : "+v"(a));
    return {a};
}
constexpr int pad64(int n) { return (n + 63) / 64 * 64; }

// Selections that depend on the lane NUMBER alone: the set of lanes is a 64-bit constant, handed to v_cndmask as a
// literal — no compare, and no scalar register pair that has to survive (or be spilled) between its uses.
// lanes_of<M>(j, kind): the lanes whose variable me = (lane mod 16) mod M is == j (kind 0) resp. > j (kind 1)
template <int M>
constexpr unsigned long long lanes_of(int j, int kind) {
    unsigned long long m = 0;
    for(int l = 0; l < 64; l++) {
        const int me = (l & 15) % M;
        if(kind == 0 ? me == j : me > j) m |= 1ull << l;
    }
    return m;
}
// (lane in MASK) ? a : b
template <unsigned long long MASK>
ILQG_DEV double lane_pick(const double a, const double b) {
    int lo, hi;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(b)), "v"(__double2loint(a)), "s"(MASK));
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(b)), "v"(__double2hiint(a)), "s"(MASK));
    return __hiloint2double(hi, lo);
}
// (lane in MASK) ? 1.0 : 0.0        (the low words agree: one selection)
template <unsigned long long MASK>
ILQG_DEV double lane_unit() {
    int hi;
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(hi) : "v"(0x3ff00000), "s"(MASK));
    return __hiloint2double(hi, 0);
}

// (sqrt_plain / rcp_plain / div_plain: ilqg_device.hpp)
// x in [2^-200, 2^200]?  x wave-uniform (the same in every lane); evaluated on the scalar unit
ILQG_DEV bool plain_range(const double x) {
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(x));
    return hi - 0x33700000u < 0x4c700000u - 0x33700000u;
}

// ---------------------------------------------------------------------------
// boxQP.c:39-238 for the wave mapping, cooperative as box_qp_rows (ilqg_wave.hpp: lane i owns variable i — its x, g,
// limits, clamp flag, row i of H and of the inverse, column i of the Cholesky factor), with every exchange between the
// lanes a row broadcast inside the consuming multiply-add instead of two v_readlane and a scalar operand: lane j of
// EVERY 16-lane row holds variable j (the other lanes mirror lane mod M), so row_newbcast:j reaches it from anywhere.
// Expression trees, operand order and exits are those of box_qp_rows, i.e. of the reference, bit for bit.
//
// LD = 0: H and the inverse are packed upper triangles (the reference's storage; the inverse zeroed on entry).
// LD > 0: H is a full square of leading dimension LD with both triangles stored (the backward step's), S_invH (LD x M)
// is scratch, and the inverse is handed back in registers: inv_row_out[j] = entry (me, j), me = (lane mod 16) mod M
// (zeros without a factorisation — every variable clamped at once — where nothing uses it).  Also handed back: this
// lane's clamp flag (of variable me) and the flags of all variables as two wave-uniform masks (at the lower / at the
// upper limit).
// ---------------------------------------------------------------------------
template <int M, int LD = 0>
ILQG_DEV int box_qp_row(const double *H /* LDS */, const double g, const double lower, const double upper, double *S_l,
                        int *S_clamp, double *S_invH, int &n_free_out, int *clamp_out = nullptr, unsigned *lo_mask = nullptr,
                        unsigned *hi_mask = nullptr, double *inv_row_out = nullptr) {
    static_assert(M <= 16, "one 16-lane row holds all variables");
    constexpr int T = tri(M);
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));  // (as in back_step_row: what depends on the lane alone is not kept across steps)
    lane &= 63;
    const int me = (lane & 15) % M;
    const unsigned all = (1u << M) - 1u;
    const int max_iter = 100;
    const double min_grad = 1e-8, min_rel_improve = 1e-8, step_dec = 0.6, min_step = 1e-22, armijo = 0.1;
    unsigned inv_at = 0;  // LDS address of the inverse (full-square form)
    if constexpr(LD > 0) inv_at = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(S_invH));

    double Hrow[M], invrow[M], Ucol[M];
    if constexpr(LD > 0) {
        const LdsBase ph = lds_base((unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(H)) + me * (LD * 8));
#pragma unroll
        for(int j = 0; j < M; j++) Hrow[j] = ph[j];
    } else {
#pragma unroll
        for(int j = 0; j < M; j++) Hrow[j] = H[sy(me, j)];
    }
#pragma unroll
    for(int j = 0; j < M; j++) {
        invrow[j] = 0.0;
        Ucol[j] = 0.0;
    }
    double x = S_l[me];  // warm start
    if(x > upper) x = upper;
    if(x < lower) x = lower;
    int clamp = 0;
    if constexpr(LD == 0)
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
            bool pd = true, plain = true;  // plain: every pivot in the range of the short forms above
            double dg[M], rdg[M];          // the factor's diagonal and its reciprocals (the same in every lane)
            static_for<0, M>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                double dot = 0.0;
                static_for<0, j>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    row_fma<j>(dot, Ucol[k], Ucol[k]);  // U[k, j] * U[k, me]
                });
                const bool masked = clamp != 0 || ((cm >> j) & 1u);
                const double unit = lane_unit<lanes_of<M>(j, 0)>();  // (me == j) ? 1 : 0
                const double a = masked ? unit : Hrow[j];
                double sv = a - dot;
                dpp_source(sv);
                const double piv = row_get<j>(sv);
                if(plain_range(piv)) {  // wave-uniform
                    dg[j] = sqrt_plain(piv);
                    rdg[j] = rcp_plain(dg[j]);
                } else {
                    plain = false;
                    if(piv <= 0.0) pd = false;
                    dg[j] = sqrt(piv);
                    rdg[j] = 1.0 / dg[j];
                }
                // (me == j) ? d : ((me > j) ? 1.0 / d * sv : 0.0)
                Ucol[j] = lane_pick<lanes_of<M>(j, 0)>(dg[j], lane_pick<lanes_of<M>(j, 1)>(rdg[j] * sv, 0.0));
                dpp_source(Ucol[j]);
            });
            if(!pd) { rc = -1; break; }
            // explicit inverse (cholesky.c:51-74): lane l solves U'U y = e_l; y[k] for k >= l is row l of the inverse
            double y[M];
            auto solve = [&](auto plain_c) {
                constexpr bool PLAIN = decltype(plain_c)::value;
                static_for<0, M>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    double v = lane_unit<lanes_of<M>(k, 0)>();  // (k == me) ? 1 : 0
                    static_for<0, k>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        row_fnma<k>(v, Ucol[i], y[i]);  // v -= y[i] * U[i, k]    (y[i] = 0 for i < l: exact zeros)
                    });
                    y[k] = PLAIN ? div_plain(v, dg[k], rdg[k]) : v / dg[k];
                });
                static_for<0, M>([&](auto kr) {
                    constexpr int k = M - 1 - decltype(kr)::value;
                    double v = y[k];
                    static_for<k + 1, M>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        row_fnma<i>(v, Ucol[k], y[i]);  // v -= y[i] * U[k, i]
                    });
                    y[k] = PLAIN ? div_plain(v, dg[k], rdg[k]) : v / dg[k];
                });
            };
            if(plain)
                solve(std::true_type{});
            else
                solve(std::false_type{});
            wave_sync();
            if constexpr(LD > 0) {
                // Row `me` of the (symmetric) inverse for every lane: entries j >= me are the lane's own y[j]; entry
                // j < me is y[me] of lane j, fetched through LDS — every lane < M lays down its y (a whole row; what
                // lies left of the diagonal is never read) and every lane reads column `me`.  No store predicates.
                if(lane < M) {
                    const LdsBase pr = lds_base(inv_at + me * (LD * 8));
#pragma unroll
                    for(int k = 0; k < M; k++) pr[k] = y[k];
                }
                wave_sync();
                const LdsBase pc = lds_base(inv_at + me * 8);
                static_for<0, M>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const double other = pc[j * LD];
                    invrow[j] = lane_pick<lanes_of<M>(j, 1)>(other, y[j]);  // (me > j) ? lane j's : own
                });
            } else {
                if(lane < M) {
#pragma unroll
                    for(int k = 0; k < M; k++)
                        if(k >= me) S_invH[ut(me, k)] = y[k];
                }
                wave_sync();
#pragma unroll
                for(int j = 0; j < M; j++) invrow[j] = S_invH[sy(me, j)];
            }
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
            if(!((cm >> j) & 1u)) row_fnma<j>(sr, gc, invrow[j]);
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
    if(clamp_out) *clamp_out = clamp;
    if(inv_row_out) {
#pragma unroll
        for(int j = 0; j < M; j++) inv_row_out[j] = invrow[j];
    }
    if(lo_mask) *lo_mask = (unsigned)__ballot(clamp == 1) & all;
    if(hi_mask) *hi_mask = (unsigned)__ballot(clamp == 2) & all;
    wave_sync();
    return rc;
}

template <int NX, int NU>
struct RowLds {
    static constexpr int SXX = tri(NX), SUU = tri(NU), NXU = NX * NU;
    static constexpr int LDX = NX + 1, LDU = NU + 1;  // padded leading dimensions: column reads hit distinct banks
    double Vx[NX], Vxx[LDX * NX];        // Vxx: full square
    double Qxu[LDX * NU];                // NX x NU
    double Qu[NU], Quu[LDU * NU], QuuF[LDU * NU];  // full squares
    double l[NU];
    union {
        struct {                         // from the start of the step to the assembly of the Q blocks
            double fx[LDX * NX], fu[LDX * NU];   // NX x NX, NX x NU
            double T1[LDX * NX], T2[LDX * NU];   // Vxx fx, Vxx fu
            // sum_i Vx[i] * (fxx_i, fuu_i, fxu_i), packed as the tensors' slices are; padded: every lane stores its sums
            double dxx[pad64(SXX)], duu[pad64(SUU)], dxu[pad64(NXU)];
            double basis[64];                    // factored tensors: the step's products (see FactoredSource)
        };
        struct {                         // from the box QP to the end of the step
            double K[LDU * NX], BA[LDU * NX];    // gains (NU x NX), Quu K
            double invH[LDU * NU];               // (scratch of the box QP)
        };
    };
    int clamp[NU + NU % 2];
    double spare[4];                     // (rows 4g..4g+3 are read as a group also where NX is not a multiple of 4)
    // the value function behind the last step (the final cost's: packed triangle in global memory)
    ILQG_DEV void set_value(const double *vx, const double *vxx_packed, int lane) {
        for(int i = lane; i < NX; i += 64) Vx[i] = vx[i];
        for(int e = lane; e < NX * NX; e += 64) Vxx[(e % NX) + (e / NX) * LDX] = vxx_packed[sy(e % NX, e / NX)];
    }
};

// Where a step's derivative entries come from.  RecordSource: the trajEl_t the generated calc_derivs code has
// written to HBM (k_derivs_wave), addressed as uniform base + 32-bit lane offset.  R: byte offsets of the members
// (a struct of constexpr unsigned: cx, cxx, cu, cuu, cxu, fx, fu, lower, upper, lower_sign, upper_sign, lower_hx,
// upper_hx and — FULL — fxx, fuu, fxu).
// (Round 6: the tensors of step k-1 requested one step ahead into 192 registers, behind the first-order entries of step k,
// were measured — one wavefront per SIMD instead of two, 590 against 530 ms per iteration of config 5 with stored tensors,
// profiles/r6_stored_path.txt — and taken out again.)
template <int NX, int NU, bool FULL, class R_>
struct RecordSource {
    using R = R_;
    static constexpr int SXX = tri(NX), SUU = tri(NU), NXU = NX * NU;
    static constexpr int NTX = (SXX + 63) / 64, NTU = (SUU + 63) / 64, NTC = (NXU + 63) / 64;
    const char *rec;  // wave-uniform
    template <unsigned OFF>
    ILQG_DEV double ld(unsigned byte_off) const {
        // the member's address in a scalar register pair of its own: the lane part stays a 32-bit offset
        using gchar = const __attribute__((address_space(1))) char;
        gchar *member = (gchar *)(rec + OFF);
        asm("" : "+s"(member));
        return *(const __attribute__((address_space(1))) double *)(member + byte_off);
    }
    // d??[q] += sum_i Vx[i] * f??_i[lane + 64 q], i ascending (vxl: Vx[c] in lane c of every row)
    ILQG_DEV void contract(const double vxl, double (&dxx)[NTX], double (&duu)[NTU], double (&dxu)[NTC], const int lane) const {
        if constexpr(FULL) {
#pragma unroll 8
            for(int i = 0; i < NX; i++) {
                const double vxi = lane_bcast(vxl, i);  // Vx[i] (lane i holds it)
#pragma unroll
                for(int q = 0; q < NTC; q++)
                    dxu[q] += vxi * ld<R::fxu>((unsigned)(((lane + 64 * q < NXU) ? lane + 64 * q : 0) + i * NXU) * 8u);
#pragma unroll
                for(int q = 0; q < NTU; q++)
                    duu[q] += vxi * ld<R::fuu>((unsigned)(((lane + 64 * q < SUU) ? lane + 64 * q : 0) + i * SUU) * 8u);
#pragma unroll
                for(int q = 0; q < NTX; q++)
                    dxx[q] += vxi * ld<R::fxx>((unsigned)(((lane + 64 * q < SXX) ? lane + 64 * q : 0) + i * SXX) * 8u);
            }
        }
    }
};

// the lanes (g, c) whose row 4g + j is == c (kind 0), <= c (kind 1)
constexpr unsigned long long row_lanes(int j, int kind) {
    unsigned long long m = 0;
    for(int l = 0; l < 64; l++) {
        const int r = 4 * (l >> 4) + j, c = l & 15;
        if(kind == 0 ? r == c : r <= c) m |= 1ull << l;
    }
    return m;
}

// One backward step.  S: LDS block of the wavefront (Vx, Vxx, l carry over between steps); nom_u / lout / Kout: the
// step's nominal inputs and gains in global memory.  Returns the box-QP code (wave-uniform); < 1 abandons the sweep
// (back_pass.c:168-171).
template <int NX, int NU, bool FULL, bool HX, class Source>
__device__ __forceinline__ int back_step_row(RowLds<NX, NU> &S, const Source &D, const double *nom_u, double *lout, double *Kout,
                                             const double lambda, const int regType, double &dV0, double &dV1, double &gsum,
                                             Prof *pf = nullptr) {
    static_assert(NX <= 16 && NU <= 16, "one 16-lane row per matrix row block");
    using L = RowLds<NX, NU>;
    using R = typename Source::R;
    constexpr int SXX = tri(NX), SUU = tri(NU), NXU = NX * NU;
    constexpr int LDX = NX + 1, LDU = NU + 1;
    constexpr bool P2U = (NU & (NU - 1)) == 0, P2GU = NU % 4 == 0 && ((NU / 4) & (NU / 4 - 1)) == 0;
#define ROW_OFF(m) ((unsigned)offsetof(L, m))
    // Everything that depends on the lane alone (rows, columns, the LDS and record addresses made of them, the
    // conditions on them) is loop invariant in the sweep, and the optimiser would move all of it — some hundred values
    // — in front of the loop and keep it in registers, i.e. spill it; conditions held in scalar register pairs are
    // spilled through v_writelane / v_readlane, which cost vector instructions.  So the lane number is made opaque at the
    // start of the step and again behind the box QP (ROW_LANE): what is needed is recomputed there by a few integer
    // instructions and nothing of it lives across the box QP or the step boundary.  `lane &= 63` gives the optimiser the range back: clamps that can never
    // act fold away.
    //   g, c      this lane's block of four rows (4g .. 4g+3) and its column
    //   cx_, cu_, gx, gu   the same moved into range where the matrix is smaller than 16 (such lanes compute along on
    //             entries that exist; their results are never stored)
    //   me        the input whose limits, clamp flag and row of the inverse this lane holds in the box QP
#define ROW_LANE                                                                              \
    lane = threadIdx.x & 63;                                                                  \
    asm volatile("" : "+v"(lane));                                                            \
    lane &= 63;                                                                               \
    g = lane >> 4, c = lane & 15;                                                             \
    cx_ = (c < NX) ? c : 0, cu_ = P2U ? (c & (NU - 1)) : ((c < NU) ? c : 0);                  \
    gx = (4 * g < NX) ? g : 0, gu = P2GU ? (g & (NU / 4 - 1)) : ((4 * g < NU) ? g : 0);       \
    me = (lane & 15) % NU;
    int lane, g, c, cx_, cu_, gx, gu, me;
    const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(&S));  // the block's LDS address

    ROW_LANE
    double vxl, vxx_r[4];                    // Vx[c], Vxx[4g+j, c] of step k+1
    double cxl, cul;                         // cx[c], cu[c]
    double cxu_e[4], cuu_e[4], cxx_e[4];     // cxu[4g+j, c], this lane's entries of the packed cuu and cxx
    double lo_k, up_k, u_l;                  // limits of input me, u[c]
    {
        // ---- value function of step k+1
        vxl = lds_base(sb + ROW_OFF(Vx) + cx_ * 8)[0];
        {
            const LdsBase p = lds_base(sb + ROW_OFF(Vxx) + (cx_ * LDX + 4 * gx) * 8);
#pragma unroll
            for(int j = 0; j < 4; j++) vxx_r[j] = p[j];
        }
        dpp_source(vxl);
        dpp_source(vxx_r);

        // ---- the step's record.  Everything but the tensors is requested here and consumed behind the tensor
        // contraction, which hides its latency: fx, fu (on their way to LDS), this lane's entries of the cost derivatives
        constexpr int NFX = (NX * NX + 63) / 64, NFU = (NXU + 63) / 64;
        double fx_in[NFX], fu_in[NFU];
#pragma unroll
        for(int q = 0; q < NFX; q++) fx_in[q] = D.template ld<R::fx>((unsigned)((lane + 64 * q < NX * NX) ? lane + 64 * q : 0) * 8u);
#pragma unroll
        for(int q = 0; q < NFU; q++) fu_in[q] = D.template ld<R::fu>((unsigned)((lane + 64 * q < NXU) ? lane + 64 * q : 0) * 8u);
        cxl = D.template ld<R::cx>((unsigned)cx_ * 8u);
        cul = D.template ld<R::cu>((unsigned)cu_ * 8u);
        // this lane's 4 entries of a packed triangle are consecutive: rows 4g..4g+3 of column c where 4g <= c (rows
        // beyond the diagonal, and all four where 4g > c, are never used: any address inside the array will do)
        const int rxx = (4 * g <= cx_) ? 4 * g : 0, ruu = (4 * g <= cu_) ? 4 * g : 0;
        const int txx0 = ut(rxx, cx_), tuu0 = ut(ruu, cu_);  // the first of the four
        const int xu0 = 4 * gx + cu_ * NX;                    // entry (4g, c) of cxu; the other three follow
#pragma unroll
        for(int j = 0; j < 4; j++) {
            // (clamped to the array's last entry unless the largest index a lane can form is inside anyway)
            constexpr bool in_xu = NX % 4 == 0, in_uu = NU >= 4 && ut(NU - 1 - (NU - 1) % 4, NU - 1) + 3 < SUU,
                           in_xx = NX >= 4 && ut(NX - 1 - (NX - 1) % 4, NX - 1) + 3 < SXX;
            cxu_e[j] = D.template ld<R::cxu>((unsigned)(in_xu ? xu0 + j : ((xu0 + j < NXU) ? xu0 + j : NXU - 1)) * 8u);
            cuu_e[j] = D.template ld<R::cuu>((unsigned)(in_uu ? tuu0 + j : ((tuu0 + j < SUU) ? tuu0 + j : SUU - 1)) * 8u);
            cxx_e[j] = D.template ld<R::cxx>((unsigned)(in_xx ? txx0 + j : ((txx0 + j < SXX) ? txx0 + j : SXX - 1)) * 8u);
        }
        lo_k = D.template ld<R::lower>((unsigned)me * 8u);
        up_k = D.template ld<R::upper>((unsigned)me * 8u);
        u_l = nom_u[(unsigned)cu_];

        // ---- second-order terms of the dynamics (back_pass.c:95-131): d[e] = sum_i Vx[i] * tensor_i[e], i ascending.
        // Here the lanes take the entries e of a tensor slice in the array's own order (e = lane, lane + 64, ...:
        // consecutive lanes on consecutive doubles, every lane busy) and hand the sums to the lanes that own them
        // through LDS.
        if constexpr(FULL) {
            constexpr int NTX = (SXX + 63) / 64, NTU = (SUU + 63) / 64, NTC = (NXU + 63) / 64;
            double dxx[NTX], duu[NTU], dxu[NTC];
#pragma unroll
            for(int q = 0; q < NTX; q++) dxx[q] = 0.0;
#pragma unroll
            for(int q = 0; q < NTU; q++) duu[q] = 0.0;
#pragma unroll
            for(int q = 0; q < NTC; q++) dxu[q] = 0.0;
            D.contract(vxl, dxx, duu, dxu, lane);
            const LdsBase pd = lds_base(sb + lane * 8);  // (the arrays are padded to whole wavefronts: no lane is left out)
#pragma unroll
            for(int q = 0; q < NTC; q++) pd[ROW_OFF(dxu) / 8 + 64 * q] = dxu[q];
#pragma unroll
            for(int q = 0; q < NTU; q++) pd[ROW_OFF(duu) / 8 + 64 * q] = duu[q];
#pragma unroll
            for(int q = 0; q < NTX; q++) pd[ROW_OFF(dxx) / 8 + 64 * q] = dxx[q];
        }
        // fx, fu into LDS
        if constexpr(NX == 16) {
            // entry lane + 64 q is (row c, column g + 4 q)
            const LdsBase pf_ = lds_base(sb + ROW_OFF(fx) + (c + g * LDX) * 8), pu_ = lds_base(sb + ROW_OFF(fu) + (c + g * LDX) * 8);
#pragma unroll
            for(int q = 0; q < NFX; q++) pf_[4 * q * LDX] = fx_in[q];
#pragma unroll
            for(int q = 0; q < NFU; q++)
                if(64 * (q + 1) <= NXU || lane + 64 * q < NXU) pu_[4 * q * LDX] = fu_in[q];
        } else {
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
        }
    }
    wave_sync();

    // ---- T1 = Vxx fx, T2 = Vxx fu (the `ba` / `bc` temporaries of matMult.c); Qu = cu + fu'Vx, Qx = cx + fx'Vx
    double qxl = cxl, qul = cul;  // Qx[c], Qu[c]
    {
        double fxc[NX];  // column c of fx
        const LdsBase p = lds_base(sb + ROW_OFF(fx) + cx_ * (LDX * 8));
#pragma unroll
        for(int s = 0; s < NX; s++) fxc[s] = p[s];
        row_dot<NX>(qxl, vxl, fxc);
        double t1[4] = {0.0, 0.0, 0.0, 0.0};
        row_product<NX>(t1, vxx_r, fxc);
        const LdsBase w = lds_base(sb + ROW_OFF(T1) + (cx_ * LDX + 4 * gx) * 8);
#pragma unroll
        for(int j = 0; j < 4; j++)
            if(4 * g + j < NX && c < NX) w[j] = t1[j];
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        double fuc[NX];  // column c of fu
        const LdsBase p = lds_base(sb + ROW_OFF(fu) + cu_ * (LDX * 8));
#pragma unroll
        for(int s = 0; s < NX; s++) fuc[s] = p[s];
        row_dot<NX>(qul, vxl, fuc);
        double t2[4] = {0.0, 0.0, 0.0, 0.0};
        row_product<NX>(t2, vxx_r, fuc);
        const LdsBase w = lds_base(sb + ROW_OFF(T2) + (cu_ * LDX + 4 * gx) * 8);
#pragma unroll
        for(int j = 0; j < 4; j++)
            if(4 * g + j < NX && c < NU) w[j] = t2[j];
        if(lane < NU) S.Qu[lane] = qul;
    }
    wave_sync();
    if(pf) pf->probe(0);

    // ---- Qxu = cxu + fx'T2, Quu = cuu + fu'T2 (symmetrised), Qxx = cxx + fx'T1 (symmetrised)   back_pass.c:90-131
    // Three blocks, each with its operands read from LDS right in front of it and nothing but its results alive behind
    // it (the scheduling barriers keep the reads of a later block from being issued — and held in registers — early).
    double qxu_r[4], qxx_r[4];  // Qxu[4g+j, c], Qxx[4g+j, c]
    {
        double t2c[NX];  // column c of T2
        {
                {
                const LdsBase p = lds_base(sb + ROW_OFF(T2) + cu_ * (LDX * 8));
#pragma unroll
                for(int s = 0; s < NX; s++) t2c[s] = p[s];
            }
            double fxt[4];  // fx[c, 4g+j]: what the row mates read
            const LdsBase p = lds_base(sb + ROW_OFF(fx) + (cx_ + 4 * gx * LDX) * 8);
#pragma unroll
            for(int j = 0; j < 4; j++) fxt[j] = p[j * LDX];
            dpp_source(fxt);
            double dxu[4] = {0.0, 0.0, 0.0, 0.0};
            row_product<NX>(dxu, fxt, t2c);  // sum_si fx[si, r] * T2[si, c]
            const LdsBase pd = lds_base(sb + ROW_OFF(dxu) + (4 * gx + cu_ * NX) * 8),
                          w = lds_base(sb + ROW_OFF(Qxu) + (cu_ * LDX + 4 * gx) * 8);
#pragma unroll
            for(int j = 0; j < 4; j++) {
                double v = cxu_e[j] + dxu[j];
                if(FULL) v += pd[j];
                qxu_r[j] = v;
                if(4 * g + j < NX && c < NU) w[j] = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
                double fuc2[NX], fut[4], t2t[4];  // column c of fu; fu[c, 4g+j], T2[c, 4g+j]
            {
                const LdsBase p = lds_base(sb + ROW_OFF(fu) + cu_ * (LDX * 8));
#pragma unroll
                for(int s = 0; s < NX; s++) fuc2[s] = p[s];
                const unsigned lt = (cx_ + 4 * gu * LDX) * 8;
                const LdsBase pa = lds_base(sb + ROW_OFF(fu) + lt), pb = lds_base(sb + ROW_OFF(T2) + lt);
#pragma unroll
                for(int j = 0; j < 4; j++) {
                    fut[j] = pa[j * LDX];
                    t2t[j] = pb[j * LDX];
                }
            }
            dpp_source(fut);
            dpp_source(t2t);
            double suu[4] = {0.0, 0.0, 0.0, 0.0}, suu_d[4];
            row_product<NX>(suu, fut, t2c);     // sum_si fu[si, r] * T2[si, c] ...
#pragma unroll
            for(int j = 0; j < 4; j++) suu_d[j] = suu[j];  // a diagonal entry stops here
            row_product_t<NX>(suu, fuc2, t2t);  // ... + sum_si fu[si, c] * T2[si, r]      (r < c)
            const int ruu = (4 * g <= cu_) ? 4 * g : 0;
            const LdsBase pd = lds_base(sb + ROW_OFF(duu) + ut(ruu, cu_) * 8);
            const unsigned la = (cu_ * LDU + 4 * gu) * 8, lb = (cu_ + 4 * gu * LDU) * 8;  // (4g+j, c) and its mirror image
            const LdsBase wa = lds_base(sb + ROW_OFF(Quu) + la), wb = lds_base(sb + ROW_OFF(Quu) + lb);
            const LdsBase fa = lds_base(sb + ROW_OFF(QuuF) + la), fb = lds_base(sb + ROW_OFF(QuuF) + lb);
            static_for<0, 4>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                double v = cuu_e[j] + lane_pick<row_lanes(j, 0)>(suu_d[j], suu[j] * 0.5);  // (4g+j == c) ? .. : ..
                if(FULL) v += pd[j];
                const double vf = lane_pick<row_lanes(j, 0)>(v + ((regType == 1) ? lambda : 0.0), v);
                if(4 * g + j <= c && c < NU) {
                    wa[j] = v;
                    wb[j * LDU] = v;
                    fa[j] = (regType == 1) ? vf : v;
                    fb[j * LDU] = (regType == 1) ? vf : v;
                }
            });
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        double t1c[NX], fxc2[NX], fxt[4], t1t[4];  // columns c of T1 and fx; fx[c, 4g+j], T1[c, 4g+j]
        {
            const LdsBase p1 = lds_base(sb + ROW_OFF(T1) + cx_ * (LDX * 8)), p2 = lds_base(sb + ROW_OFF(fx) + cx_ * (LDX * 8));
#pragma unroll
            for(int s = 0; s < NX; s++) {
                t1c[s] = p1[s];
                fxc2[s] = p2[s];
            }
            const unsigned lt = (cx_ + 4 * gx * LDX) * 8;
            const LdsBase pa = lds_base(sb + ROW_OFF(fx) + lt), pb = lds_base(sb + ROW_OFF(T1) + lt);
#pragma unroll
            for(int j = 0; j < 4; j++) {
                fxt[j] = pa[j * LDX];
                t1t[j] = pb[j * LDX];
            }
        }
        dpp_source(fxt);
        dpp_source(t1t);
        double sxx[4] = {0.0, 0.0, 0.0, 0.0}, sxx_d[4];
        row_product<NX>(sxx, fxt, t1c);
#pragma unroll
        for(int j = 0; j < 4; j++) sxx_d[j] = sxx[j];
        row_product_t<NX>(sxx, fxc2, t1t);
        const int rxx = (4 * g <= cx_) ? 4 * g : 0;
        const LdsBase pd = lds_base(sb + ROW_OFF(dxx) + ut(rxx, cx_) * 8);
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            double v = cxx_e[j] + lane_pick<row_lanes(j, 0)>(sxx_d[j], sxx[j] * 0.5);
            if(FULL) v += pd[j];
            qxx_r[j] = v;
        });
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
            const double v = S.Quu[r + cc * LDU] + acc * lambda;
            S.QuuF[r + cc * LDU] = v;
            S.QuuF[cc + r * LDU] = v;
        }
#pragma unroll
        for(int j = 0; j < 4; j++) {
            const int i = (4 * g + j < NX) ? 4 * g + j : 0, q = cu_;
            double acc = 0.0;
#pragma unroll
            for(int s = 0; s < NX; s++) acc += S.fx[s + i * LDX] * S.fu[((s + q * NU) % NX) + ((s + q * NU) / NX) * LDX];
            qxur_r[j] = qxu_r[j] + acc * lambda;
        }
        wave_sync();
    }
    if(pf) pf->probe(2);

    // ---- box QP, one lane per input (box_qp_rows); warm start: the later step's solution in S.l (back_pass.c:163-166)
    // (from here on fx .. dxu are dead: K, Quu K and the box QP's scratch take their place)
    int nf, mine;
    unsigned cm_lo, cm_hi;
    double ih[NU];  // invH[me, :]
    // (the gradient of input me: with NU a power of two that is this lane's own Qu[c])
    const int rc = box_qp_row<NU, LDU>(S.QuuF, P2U ? qul : S.Qu[me], lo_k, up_k, S.l, S.clamp, S.invH, nf, &mine, &cm_lo, &cm_hi, ih);
    if(pf) pf->probe(3);
    if(rc < 1) return rc;

    // ---- feedback gains (back_pass.c:175-201): lane (g, i) computes K[i, 4g+j], input i = c, state 4g+j
    double kt[4] = {0.0, 0.0, 0.0, 0.0};
    ROW_LANE
    {
        dpp_source(qxur_r);
        static_for<0, NU>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            const int cl = ((cm_lo >> jj) & 1u) ? 1 : (((cm_hi >> jj) & 1u) ? 2 : 0);  // wave-uniform: scalar branches
            if(!cl) {
#pragma unroll
                for(int j = 0; j < 4; j++) row_fnma<jj>(kt[j], qxur_r[j], ih[jj]);  // K -= Qxu(reg) invH
            } else if(HX) {
                double w = 0.0;
#pragma unroll
                for(int s = 0; s < NU; s++)
                    if(!(((cm_lo | cm_hi) >> s) & 1u)) w -= ih[s] * S.QuuF[s + jj * LDU];
                const double sg = (cl == 1) ? D.template ld<R::lower_sign>(jj * 8u) : D.template ld<R::upper_sign>(jj * 8u);
#pragma unroll
                for(int j = 0; j < 4; j++) {
                    const unsigned o = (unsigned)(((4 * g + j < NX) ? 4 * g + j : 0) + jj * NX) * 8u;
                    const double hx = (cl == 1) ? D.template ld<R::lower_hx>(o) : D.template ld<R::upper_hx>(o);
                    kt[j] -= w * (sg * hx);
                }
            }
        });
        if(mine) {  // a clamped input follows its limit (back_pass.c:186-190)
#pragma unroll
            for(int j = 0; j < 4; j++) {
                double v = 0.0;
                if(HX) {
                    const unsigned o = (unsigned)(((4 * g + j < NX) ? 4 * g + j : 0) + me * NX) * 8u;
                    const double sg = (mine == 1) ? D.template ld<R::lower_sign>((unsigned)me * 8u)
                                                  : D.template ld<R::upper_sign>((unsigned)me * 8u);
                    const double hx = (mine == 1) ? D.template ld<R::lower_hx>(o) : D.template ld<R::upper_hx>(o);
                    v -= sg * hx;
                }
                kt[j] = v;
            }
        }
        const LdsBase w = lds_base(sb + ROW_OFF(K) + (cu_ + 4 * gx * LDU) * 8);
        double *const ko = Kout + (unsigned)(cu_ + 4 * gx * NU);
#pragma unroll
        for(int j = 0; j < 4; j++)
            if(4 * g + j < NX && c < NU) {
                w[j * LDU] = kt[j];
                ko[j * NU] = kt[j];
            }
        if(lane < NU) lout[lane] = S.l[lane];
    }
    wave_sync();
    if(pf) pf->probe(4);

    // ---- Quu l, Quu K; expected cost change (back_pass.c:205-214)
    double kc[NU];           // column c of K: K[s, c], state c
    double ll, bcl = 0.0;    // l[c], (Quu l)[c]
    {
        double ql[NU];       // Quu[c, s]: row c of Quu
        double quu_r[4];     // Quu[4g+j, c]
        ll = lds_base(sb + ROW_OFF(l) + cu_ * 8)[0];
        {
            const LdsBase pk = lds_base(sb + ROW_OFF(K) + cx_ * (LDU * 8)), pq = lds_base(sb + ROW_OFF(Quu) + cu_ * (LDU * 8));
#pragma unroll
            for(int s = 0; s < NU; s++) {
                kc[s] = pk[s];
                ql[s] = pq[s];
            }
            const LdsBase pr = lds_base(sb + ROW_OFF(Quu) + (cu_ * LDU + 4 * gu) * 8);
#pragma unroll
            for(int j = 0; j < 4; j++) quu_r[j] = pr[j];
        }
        dpp_source(quu_r);
        dpp_source(ll);
        row_dot<NU>(bcl, ll, ql);
        double ba[4] = {0.0, 0.0, 0.0, 0.0};
        row_product<NU>(ba, quu_r, kc);  // (Quu K)[4g+j, c]
        const LdsBase w = lds_base(sb + ROW_OFF(BA) + (cx_ * LDU + 4 * gu) * 8);
#pragma unroll
        for(int j = 0; j < 4; j++)
            if(4 * g + j < NU && c < NX) w[j] = ba[j];
        // dV += [l'Qu, 0.5 l'Quu l], term by term over the inputs (the same sums in every lane)
        double hl = 0.5 * ll;
        dpp_source(hl);
        dpp_source(bcl);
        dpp_source(qul);
        static_for<0, NU>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            row_fma2<i>(dV0, qul, ll);
            row_fma2<i>(dV1, hl, bcl);
        });
    }
    wave_sync();
    if(pf) pf->probe(5);

    // ---- Vx, Vxx with the unregularised Quu / Qxu (back_pass.c:219-241)
    {
        double bac[NU], qxc[NU];  // column c of Quu K; row c of Qxu
        double bat[4];            // (Quu K)[c, 4g+j]
        {
            const LdsBase pb = lds_base(sb + ROW_OFF(BA) + cx_ * (LDU * 8)), pq = lds_base(sb + ROW_OFF(Qxu) + cx_ * 8);
#pragma unroll
            for(int s = 0; s < NU; s++) {
                bac[s] = pb[s];
                qxc[s] = pq[s * LDX];
            }
            const LdsBase pt = lds_base(sb + ROW_OFF(BA) + (cu_ + 4 * gx * LDU) * 8);
#pragma unroll
            for(int j = 0; j < 4; j++) bat[j] = pt[j * LDU];
        }
        dpp_source(bat);
        dpp_source(kt);
        dpp_source(qxu_r);

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
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            vv[j] = qxx_r[j] + lane_pick<row_lanes(j, 0)>(sq_d[j], sq[j] * 0.5);
        });
        // the reference's loop nest touches packed entry (r, c) first as (i = r, j = c), then as (i = c, j = r);
        // a diagonal entry once, with the term doubled
        double vd[4];
#pragma unroll
        for(int j = 0; j < 4; j++) vd[j] = vv[j];
        row_product<NU>(vd, kt, qx2);
        row_product<NU>(vv, kt, qxc);
        row_product_t<NU>(vv, kc, qxu_r);
        const LdsBase wa = lds_base(sb + ROW_OFF(Vxx) + (cx_ * LDX + 4 * gx) * 8), wb = lds_base(sb + ROW_OFF(Vxx) + (cx_ + 4 * gx * LDX) * 8);
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const double v = lane_pick<row_lanes(j, 0)>(vd[j], vv[j]);
            if(4 * g + j <= c && c < NX) {
                wa[j] = v;
                wb[j * LDX] = v;
            }
        });
        if(lane < NX) S.Vx[lane] = vx;
    }

    // gradient-norm summand (back_pass.c:246-251)
    {
        double gl = fabs(ll) / (fabs(u_l) + 1.0);
        dpp_source(gl);
        double gmax = 0.0;
        static_for<0, NU>([&](auto ic) {
            gmax = __builtin_fmax(gmax, row_get<decltype(ic)::value>(gl));  // (none of them negative; a NaN is skipped either way)
        });
        gsum += gmax;
    }
    wave_sync();
    if(pf) pf->probe(6);
#undef ROW_OFF
#undef ROW_LANE
    return rc;
}

}  // namespace ilqg
