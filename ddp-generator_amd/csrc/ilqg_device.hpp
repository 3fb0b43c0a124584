// Device-side dense kernels of the iLQG backward pass for small, compile-time
// problem sizes.  Every loop has constant bounds and every array index is a
// compile-time constant after unrolling, so all operands live in VGPRs (a
// runtime-indexed private array would be demoted to scratch memory).
//
// "Lane mapping": one lane owns one trajectory and runs this scalar code on
// its own registers; 64 trajectories per wavefront.  The operation order of
// the reference is kept (ascending sums, same temporaries), so results differ
// from the CPU only through fused multiply-add contraction.
//
// What each routine replaces in the reference:
//   add_mul_vec / add_square_tri / add_mul2_tri   matMult.c:3-72
//   chol_factor / chol_inverse                    cholesky.c:6-27, 51-74
//   box_qp                                        boxQP.c:39-238
//   back_step                                     back_pass.c:80-251 (one time step)
#pragma once
#include <hip/hip_runtime.h>

namespace ilqg {

#define ILQG_DEV __device__ __forceinline__

// Optional cycle accounting of the sections of a backward step (builds with -DILQG_PROFILE_SECTIONS only;
// tools/section_profile.py).  probe(i) charges the cycles since the previous probe to section i.
#ifdef ILQG_PROFILE_SECTIONS
__device__ unsigned long long ilqg_prof_cycles[8];  // summed over wavefronts: see tools/section_profile.py
#endif
// Where and when every wavefront of the backward and search kernels ran (builds with -DILQG_WAVE_PLACES only;
// tools/experiments/wave_places.py): HW_ID (SIMD, CU, shader array, engine), XCC_ID and the constant 100 MHz clock at entry and exit.
#ifdef ILQG_WAVE_PLACES
constexpr unsigned PLACES_MAX = 1u << 19;
__device__ unsigned long long ilqg_places[PLACES_MAX][3];
__device__ unsigned ilqg_places_n;
#endif
struct Place {
    unsigned slot;
    ILQG_DEV explicit Place(int kind) {
        slot = ~0u;
#ifdef ILQG_WAVE_PLACES
        if(threadIdx.x % 64 == 0) {
            slot = atomicAdd(&ilqg_places_n, 1u);
            if(slot < PLACES_MAX) {
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                ilqg_places[slot][0] = ((unsigned long long)kind << 40) | ((unsigned long long)(xcc & 15u) << 32) | hw;
                ilqg_places[slot][1] = wall_clock64();
            }
        }
#endif
    }
    ILQG_DEV ~Place() {
#ifdef ILQG_WAVE_PLACES
        if(slot < PLACES_MAX) ilqg_places[slot][2] = wall_clock64();
#endif
    }
};

struct Prof {
    long long last, acc[8];
    long long wave_steps;  // quad mapping: steps of the wavefront
    ILQG_DEV void start() {
        wave_steps = 0;
        for(int i = 0; i < 8; i++) acc[i] = 0;
        last = __builtin_readcyclecounter();
    }
    ILQG_DEV void probe(int i) {
#ifdef ILQG_PROFILE_SECTIONS
        const long long t = __builtin_readcyclecounter();
        acc[i] += t - last;
        last = t;
#endif
    }
};

__host__ __device__ constexpr int tri(int n) { return n * (n + 1) / 2; }
__host__ __device__ constexpr int ut(int r, int c) { return c * (c + 1) / 2 + r; }               // r <= c
__host__ __device__ constexpr int sy(int i, int j) { return i > j ? ut(j, i) : ut(i, j); }

// ---------------------------------------------------------------------------
// matMult.c
// ---------------------------------------------------------------------------
// Structural zeros of the step's record.  For CarParking 27 of the 55 entries of a record are identically 0 (fx: 7 of
// 16, fu: 4 of 8, cxx: 8 of 10, all of cxu, ...; the generated header lists them, ILQG_STRUCTURAL_ZERO, beside the list
// of the entries bp_derivsL writes).  IEEE arithmetic does not let the optimiser drop `acc + x * 0.0` even where it sees
// the constant (x may be Inf or NaN, acc may be -0.0), so the dense loops of matMult.c spend a multiply-add on every one
// of them.  A caller that knows the zeros (Z::at(i): entry i of the record, in RecLayout's numbering, is identically 0)
// has those terms left out — decided at compile time once the loops are unrolled, never at run time.  With finite
// operands the result is the reference's bit for bit but for the sign of a zero sum; an Inf or NaN in the OTHER factor no
// longer reaches the sum through a structural zero (it still does through every other term).  The FMA-free twin
// (ILQG_STRICT_FP, the bit-for-bit tests) and every caller without the list keep the dense form.
// (Tried first: `if(__builtin_constant_p(b) && b == 0.0)` per term, which needs no list — the undecided llvm.is.constant
// branches keep SROA from dissolving the step's arrays, 1 120 bytes of scratch per lane; `#pragma float_control` for the
// no-signed-zeros / no-NaN flags — "not supported on this target".)
struct NoZeros {
    static constexpr bool at(int) { return false; }
};
// term a * b of a sum, where b is entry I of the record (I < 0: not a record entry)
template <class Z>
ILQG_DEV double mad(const double acc, const double a, const double b, const int I) {
    if(I >= 0 && Z::at(I)) return acc;
    return acc + a * b;
}
// base + v, where base started as entry I of the record
template <class Z>
ILQG_DEV double plus(const double base, const double v, const int I) {
    if(I >= 0 && Z::at(I)) return v;
    return base + v;
}

// base[c] += sum_r a[r] * b[r + c*NR]                 (BO: b is the record from entry BO on, or -1)
template <int NR, int NC, class Z = NoZeros, int BO = -1>
ILQG_DEV void add_mul_vec(double *base, const double *a, const double *b) {
#pragma unroll
    for(int c = 0; c < NC; c++)
#pragma unroll
        for(int r = 0; r < NR; r++) base[c] = mad<Z>(base[c], a[r], b[r + c * NR], BO < 0 ? -1 : BO + r + c * NR);
}

// packed-upper base (NC x NC) += A' B A, B packed symmetric NR x NR, A is NR x NC
// (AO: A is the record from entry AO on; SO: base started as the record's entries from SO on; -1: neither)
template <int NR, int NC, class Z = NoZeros, int AO = -1, int SO = -1>
ILQG_DEV void add_square_tri(double *base, const double *B, const double *A) {
    double ba[NR * NC];
#pragma unroll
    for(int r = 0; r < NR; r++)
#pragma unroll
        for(int c = 0; c < NC; c++) {
            double acc = 0.0;
#pragma unroll
            for(int s = 0; s < NR; s++) acc = mad<Z>(acc, B[sy(r, s)], A[s + c * NR], AO < 0 ? -1 : AO + s + c * NR);
            ba[r + c * NR] = acc;
        }
#pragma unroll
    for(int c = 0; c < NC; c++)
#pragma unroll
        for(int r = 0; r <= c; r++) {
            double acc = 0.0;
#pragma unroll
            for(int s = 0; s < NR; s++) acc = mad<Z>(acc, ba[s + c * NR], A[s + r * NR], AO < 0 ? -1 : AO + s + r * NR);
            if(r != c) {
#pragma unroll
                for(int s = 0; s < NR; s++) acc = mad<Z>(acc, ba[s + r * NR], A[s + c * NR], AO < 0 ? -1 : AO + s + c * NR);
                acc *= 0.5;
            }
            base[ut(r, c)] = plus<Z>(base[ut(r, c)], acc, SO < 0 ? -1 : SO + ut(r, c));
        }
}

// full base (NCA x NCC) += A' B C, B packed symmetric NRA x NRA, A is NRA x NCA, C is NRA x NCC
template <int NRA, int NCA, int NCC, class Z = NoZeros, int AO = -1, int CO = -1, int SO = -1>
ILQG_DEV void add_mul2_tri(double *base, const double *B, const double *A, const double *C) {
    double bc[NRA * NCC];
#pragma unroll
    for(int r = 0; r < NRA; r++)
#pragma unroll
        for(int q = 0; q < NCC; q++) {
            double acc = 0.0;
#pragma unroll
            for(int s = 0; s < NRA; s++) acc = mad<Z>(acc, B[sy(r, s)], C[s + q * NRA], CO < 0 ? -1 : CO + s + q * NRA);
            bc[r + q * NRA] = acc;
        }
#pragma unroll
    for(int r = 0; r < NCA; r++)
#pragma unroll
        for(int q = 0; q < NCC; q++) {
            double acc = 0.0;
#pragma unroll
            for(int s = 0; s < NRA; s++) acc = mad<Z>(acc, bc[s + q * NRA], A[s + r * NRA], AO < 0 ? -1 : AO + s + r * NRA);
            base[r + q * NCA] = plus<Z>(base[r + q * NCA], acc, SO < 0 ? -1 : SO + r + q * NCA);
        }
}

// ---------------------------------------------------------------------------
// Square root, reciprocal and quotient for operands in a "plain" range, bit for bit what sqrt(x), 1.0 / d and v / d
// give.  The compiler expands a double-precision sqrt into 18 instructions and a division into 11; 8 resp. 3 of them
// scale very small or large operands and patch up 0, Inf and NaN.  With the operand known to lie in [2^-200, 2^200]
// (the box QP's pivots, tested as they come) what is left is the same arithmetic, i.e. the same
// bits.  The quotient by a divisor whose correctly rounded reciprocal rd is at hand: q = v rd, q' = q + (v - d q) rd
// (two fused operations; Markstein's theorem: q' is the correctly rounded v / d provided nothing underflows —
// |v| >= 2^-970 or v == 0 with d in the range above; checked against v / d on 2.7e8 random pairs, tools/ubench/quotient_check.c).
// The box QP factorises once or twice per backward step with 8 square roots and 24 divisions each time (cholesky.c:6-74).
// ---------------------------------------------------------------------------
ILQG_DEV double sqrt_plain(const double x) {
    const double y0 = __builtin_amdgcn_rsq(x);
    const double g0 = x * y0, h0 = y0 * 0.5;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0);
    const double d0 = __builtin_fma(-g1, g1, x);
    const double h1 = __builtin_fma(h0, r0, h0);
    const double g2 = __builtin_fma(d0, h1, g1);
    const double d1 = __builtin_fma(-g2, g2, x);
    return __builtin_fma(d1, h1, g2);
}
ILQG_DEV double rcp_plain(const double d) {
    const double y0 = __builtin_amdgcn_rcp(d);
    const double y1 = __builtin_fma(y0, __builtin_fma(-d, y0, 1.0), y0);
    const double y2 = __builtin_fma(y1, __builtin_fma(-d, y1, 1.0), y1);
    return __builtin_fma(__builtin_fma(-d, y2, 1.0), y2, y2);
}
ILQG_DEV double div_plain(const double v, const double d, const double rd) {
    const double q = v * rd;
    return __builtin_fma(__builtin_fma(-d, q, v), rd, q);
}
// x in [2^-200, 2^200]?  (per lane)
ILQG_DEV bool plain_range_lane(const double x) { return (unsigned)__double2hiint(x) - 0x33700000u < 0x4c700000u - 0x33700000u; }

// ---------------------------------------------------------------------------
// cholesky.c (plain part)
// ---------------------------------------------------------------------------
// A = U'U on packed upper triangles.  false as soon as a pivot is <= 0.
template <int M>
ILQG_DEV bool chol_factor(const double *A, double *U) {
    bool ok = true;
#pragma unroll
    for(int i = 0; i < M; i++)
#pragma unroll
        for(int j = 0; j <= i; j++) {
            double dot = 0.0;
#pragma unroll
            for(int k = 0; k < j; k++) dot += U[ut(k, i)] * U[ut(k, j)];
            const double s = A[ut(j, i)] - dot;
            if(i == j) {
                if(s <= 0.0) ok = false;
                U[ut(j, i)] = sqrt(s);
            } else {
                U[ut(j, i)] = 1.0 / U[ut(j, j)] * s;
            }
        }
    return ok;
}

// explicit packed inverse of U'U
template <int M>
ILQG_DEV void chol_inverse(const double *U, double *inv) {
#pragma unroll
    for(int l = 0; l < M; l++) {
        double x[M];
#pragma unroll
        for(int k = 0; k < M; k++) x[k] = (k == l) ? 1.0 : 0.0;
#pragma unroll
        for(int k = l; k < M; k++) {
#pragma unroll
            for(int i = l; i < k; i++) x[k] -= x[i] * U[ut(i, k)];
            x[k] /= U[ut(k, k)];
        }
#pragma unroll
        for(int k = M - 1; k >= l; k--) {
#pragma unroll
            for(int i = k + 1; i < M; i++) x[k] -= x[i] * U[ut(k, i)];
            x[k] /= U[ut(k, k)];
            inv[ut(l, k)] = x[k];
        }
    }
}

// chol_factor + chol_inverse (what the box QP does with the free block, boxQP.c:131-146) with the short forms above
// while every pivot of every active lane lies in their range — the same bits — and once more in the general form if
// one does not (a pivot <= 0 among them: `false` as from chol_factor).  A quotient is then a chain of 3 dependent
// instructions instead of 11, a square root of 10 instead of 18: with one wavefront per SIMD the chain is what counts.
template <int M>
ILQG_DEV bool chol_factor_inverse(const double *A, double *inv) {
    {
        double U[tri(M)], rd[M];
        bool plain = true;
#pragma unroll
        for(int i = 0; i < M; i++)
#pragma unroll
            for(int j = 0; j <= i; j++) {
                double dot = 0.0;
#pragma unroll
                for(int k = 0; k < j; k++) dot += U[ut(k, i)] * U[ut(k, j)];
                const double s = A[ut(j, i)] - dot;
                if(i == j) {
                    plain = plain && plain_range_lane(s);
                    U[ut(j, i)] = sqrt_plain(s);
                    rd[j] = rcp_plain(U[ut(j, i)]);
                } else {
                    U[ut(j, i)] = rd[j] * s;
                }
            }
        if(__builtin_amdgcn_ballot_w64(!plain) == 0ull) {
#pragma unroll
            for(int l = 0; l < M; l++) {
                double x[M];
#pragma unroll
                for(int k = 0; k < M; k++) x[k] = (k == l) ? 1.0 : 0.0;
#pragma unroll
                for(int k = l; k < M; k++) {
#pragma unroll
                    for(int i = l; i < k; i++) x[k] -= x[i] * U[ut(i, k)];
                    x[k] = div_plain(x[k], U[ut(k, k)], rd[k]);
                }
#pragma unroll
                for(int k = M - 1; k >= l; k--) {
#pragma unroll
                    for(int i = k + 1; i < M; i++) x[k] -= x[i] * U[ut(k, i)];
                    x[k] = div_plain(x[k], U[ut(k, k)], rd[k]);
                    inv[ut(l, k)] = x[k];
                }
            }
            return true;
        }
    }
    double U[tri(M)];
    const bool pd = chol_factor<M>(A, U);
    chol_inverse<M>(U, inv);
    return pd;
}

// ---------------------------------------------------------------------------
// Factor and inverse of the free block for EVERY pattern of clamped variables at once (M <= 3: 2^M - 1 patterns with
// at least one free variable).  What boxQP.c:129-146 computes when the free set changes is a pure function of H and
// the pattern — the reference compacts the free rows / columns and runs cholesky_tri / cholesky_tri_inv on the small
// matrix — so it is evaluated here ONCE per call for each pattern, on the compacted block with the reference's
// operations for that size (the same bits), and an outer iteration of the box QP SELECTS by its pattern.  In the lane
// mapping a factorisation under `if(free set changed)` runs whenever ANY of the 64 lanes needs it, i.e. in nearly
// every outer iteration of the slowest lane (measured: 1 234 -> 1 672 cycles per step between iteration 1 and 20 of
// the benchmark); the patterns of one call are independent instruction streams that overlap instead.
// Pattern p = bit mask of the CLAMPED variables, p in [0, 2^M - 2].  inv[p]: the inverse in full index form (free
// block at the original indices, identity on the clamped ones — what the embedded factorisation of box_qp gives),
// pd[p]: every pivot > 0.  Short forms of sqrt / reciprocal / quotient while every pivot of every pattern and lane is in
// their range, else everything once more in the general form (a pattern the iteration never visits may have any
// pivots: nothing of it is used).
// ---------------------------------------------------------------------------
template <int M, unsigned P, bool PLAIN>
ILQG_DEV void chol_pattern(const double *H, double *inv, bool &pd, bool &plain) {
    constexpr int NF = M - __builtin_popcount(P);
    int idx[NF > 0 ? NF : 1];  // the free variables, ascending (compile-time after unrolling)
    {
        int q = 0;
#pragma unroll
        for(int i = 0; i < M; i++)
            if(!((P >> i) & 1u)) idx[q++] = i;
    }
    double A[tri(NF)], U[tri(NF)], iv[tri(NF)], rd[NF];
#pragma unroll
    for(int c = 0; c < NF; c++)
#pragma unroll
        for(int r = 0; r <= c; r++) A[ut(r, c)] = H[ut(idx[r], idx[c])];
    if constexpr(PLAIN) {
        pd = true;  // (a pivot in the plain range is positive; out of range: the general form decides)
#pragma unroll
        for(int i = 0; i < NF; i++)
#pragma unroll
            for(int j = 0; j <= i; j++) {
                double dot = 0.0;
#pragma unroll
                for(int k = 0; k < j; k++) dot += U[ut(k, i)] * U[ut(k, j)];
                const double sv = A[ut(j, i)] - dot;
                if(i == j) {
                    plain = plain && plain_range_lane(sv);
                    U[ut(j, i)] = sqrt_plain(sv);
                    rd[j] = rcp_plain(U[ut(j, i)]);
                } else {
                    U[ut(j, i)] = rd[j] * sv;
                }
            }
#pragma unroll
        for(int l = 0; l < NF; l++) {
            double x[NF];
#pragma unroll
            for(int k = 0; k < NF; k++) x[k] = (k == l) ? 1.0 : 0.0;
#pragma unroll
            for(int k = l; k < NF; k++) {
#pragma unroll
                for(int i = l; i < k; i++) x[k] -= x[i] * U[ut(i, k)];
                x[k] = div_plain(x[k], U[ut(k, k)], rd[k]);
            }
#pragma unroll
            for(int k = NF - 1; k >= l; k--) {
#pragma unroll
                for(int i = k + 1; i < NF; i++) x[k] -= x[i] * U[ut(k, i)];
                x[k] = div_plain(x[k], U[ut(k, k)], rd[k]);
                iv[ut(l, k)] = x[k];
            }
        }
    } else {
        pd = chol_factor<NF>(A, U);
        chol_inverse<NF>(U, iv);
    }
#pragma unroll
    for(int c = 0; c < M; c++)
#pragma unroll
        for(int r = 0; r <= c; r++) inv[ut(r, c)] = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for(int c = 0; c < NF; c++)
#pragma unroll
        for(int r = 0; r <= c; r++) inv[ut(idx[r], idx[c])] = iv[ut(r, c)];
}

template <int M, bool PLAIN, unsigned P = 0>
ILQG_DEV void chol_pattern_all(const double *H, double (*inv)[tri(M)], bool *pd, bool &plain) {
    if constexpr(P < (1u << M) - 1u) {
        chol_pattern<M, P, PLAIN>(H, inv[P], pd[P], plain);
        chol_pattern_all<M, PLAIN, P + 1>(H, inv, pd, plain);
    }
}

template <int M>
ILQG_DEV void chol_pattern_table(const double *H, double (*inv)[tri(M)], bool *pd) {
    bool plain = true;
    chol_pattern_all<M, true>(H, inv, pd, plain);
    if(__builtin_amdgcn_ballot_w64(!plain) != 0ull) chol_pattern_all<M, false>(H, inv, pd, plain);
}
// MEASURED (round 4, CarParking headline, same box): no gain.  The section profile of the fused backward kernel at
// iteration 20 moves 680 cycles per step out of "factorisation + inverse" (1 242 -> 565) and 715 into the section that
// now holds the table and the selects (2 976 -> 3 691); the step stays at 13 035 cycles, the headline 179-180 against
// 181-182 it/s.  A wavefront factorised ~2 times per step before (once per outer iteration in which some lane's free set
// changed), each time ONE pattern per lane; the table is three patterns per lane, every step.
// profiles/r4_sections_qp_table.txt.  Off by default (-DILQG_QP_TABLE=1 turns it on); the unit-test kernel runs both forms.
#ifndef ILQG_QP_TABLE
#define ILQG_QP_TABLE 0
#endif

// ---------------------------------------------------------------------------
// boxQP.c
// ---------------------------------------------------------------------------
template <int M>
ILQG_DEV double qp_value(const double *H, const double *g, const double *x) {
    double v = 0.0;
#pragma unroll
    for(int i = 0; i < M; i++) {
        double hx = 0.0;
#pragma unroll
        for(int j = 0; j < M; j++) hx += H[sy(i, j)] * x[j];
        v += x[i] * (g[i] + 0.5 * hx);
    }
    return v;
}

// The acceptance test of the backtracking line search, boxQP.c:219:
//     (vc - oldvalue) / (step * sdotg) >= armijo          (step > 0, sdotg < 0)
// decided without the division wherever the outcome is beyond doubt: with n = vc - oldvalue, d = step*sdotg < 0
// the quotient is >= armijo iff n <= armijo*d.  Both sides carry a few rounding errors of relative size 2^-53
// each, so the comparison is trusted only outside a band of relative width 1e-15 (nine times that) around
// armijo*d; inside the band — and for NaN, where every comparison fails — the reference's own expression is
// evaluated.  The result is therefore always the reference's.
ILQG_DEV bool armijo_passes(double vc, double oldvalue, double step, double sdotg, double armijo) {
    const double n = vc - oldvalue, d = step * sdotg;
    const double t = armijo * d, m = fabs(t) * 1e-15;
    if(n < t - m) return true;
    if(n > t + m) return false;
    return (n / d) >= armijo;
}

// The same test in two parts, for callers that evaluate several trials before branching: the quick verdict
// (+1 passes, -1 fails, 0 too close to call) without any control flow, and the reference's expression for the rest.
#ifndef ARMIJO_TRIALS
#define ARMIJO_TRIALS 2
#endif
#ifndef ARMIJO_TRIALS_LATE
#define ARMIJO_TRIALS_LATE 4
#endif
ILQG_DEV int armijo_quick(double vc, double oldvalue, double step, double sdotg, double armijo) {
    const double n = vc - oldvalue, d = step * sdotg;
    const double t = armijo * d, m = fabs(t) * 1e-15;
    return (n < t - m) ? 1 : ((n > t + m) ? -1 : 0);
}
// number of the last Armijo trial the reference can reach: trial k uses step_k = step_{k-1} * stepDec (step_0 = 1), and
// after a failed trial k the loop returns 2 if step_{k+1} < minStep (boxQP.c:221-224).  The same IEEE products as at
// run time, evaluated by the compiler.
constexpr int armijo_last_trial(double step_dec, double min_step) {
    double s = 1.0;
    for(int k = 0; k < 4096; k++) {
        s = s * step_dec;
        if(s < min_step) return k;
    }
    return 4096;
}
constexpr int ARMIJO_LAST = armijo_last_trial(0.6, 1e-22);
static_assert(ARMIJO_LAST > 8 && ARMIJO_LAST < 200, "0.6^k falls below 1e-22 near k = 99");
ILQG_DEV bool armijo_exact(double vc, double oldvalue, double step, double sdotg, double armijo) {
    return ((vc - oldvalue) / (step * sdotg)) >= armijo;
}

// Projected-Newton box QP.  Same iteration, constants and return codes as
// boxQP.c:39-238.  One representational difference: the reference compacts
// the free rows/columns into a smaller matrix (which needs runtime indices);
// here the clamped rows/columns of a full-size copy are replaced by identity
// before factorising.  The factor and inverse of that matrix are the free
// block's factor and inverse embedded at the original indices (all extra terms
// are exact zeros), so `invH` is returned in FULL index form: invH[sy(i,j)]
// for free i,j equals the reference's invHfree[sy(i_free,j_free)].
template <int M, bool WITH_TABLE = (ILQG_QP_TABLE != 0)>
ILQG_DEV int box_qp(const double *H, const double *g, const double *lower, const double *upper, double *x,
                    int *clamp, int &n_free_out, double *invH, Prof *pf = nullptr) {
    // Control flow.  The reference leaves its loop through seven `return`s; compiled literally for 64 lanes in
    // lock step each of them is a divergent branch (save/restore of the execution mask, a VALU->SALU round
    // trip, a fetch bubble), and with one wavefront per SIMD nothing hides them: measured, the branches cost
    // more than the arithmetic between them.  Here a lane that has reached a return only RECORDS its code in
    // `rc` and stops committing results (everything below is predicated on rc == 0 through selects); it leaves
    // at the single test at the end of the iteration.  The lanes of a wavefront wait for the slowest one
    // anyway, so the work a finished lane still steps through costs no time.  Results, codes and the state
    // left in x / clamp / invH at each exit are the reference's.
    constexpr int T = tri(M);
    const int max_iter = 100;
    const double min_grad = 1e-8, min_rel_improve = 1e-8, step_dec = 0.6, min_step = 1e-22, armijo = 0.1;
    double grad[M], search[M];
    double value, oldvalue = 0.0;
    int rc = 0;  // 0: iterating

#pragma unroll
    for(int i = 0; i < M; i++) {
        if(x[i] > upper[i]) x[i] = upper[i];
        if(x[i] < lower[i]) x[i] = lower[i];
        clamp[i] = 0;
    }
#pragma unroll
    for(int i = 0; i < T; i++) invH[i] = 0.0;
    n_free_out = 0;
    value = qp_value<M>(H, g, x);
    // factor and inverse of the free block for every pattern of clamped variables (see chol_pattern_table)
    constexpr bool TABLE = WITH_TABLE && M <= 3;
    constexpr int NPAT = TABLE ? (1 << M) - 1 : 1;
    double inv_of[NPAT][T];
    bool pd_of[NPAT];
    if constexpr(TABLE) chol_pattern_table<M>(H, inv_of, pd_of);

    for(int iter = 0; iter < max_iter; iter++) {
        if(iter > 0 && (oldvalue - value) < min_rel_improve * fabs(oldvalue)) rc = 4;  // boxQP.c:85-86
        const bool live0 = (rc == 0);
        oldvalue = live0 ? value : oldvalue;

        // gradient and clamp flags (boxQP.c:95-117)
        bool all_clamped = true, changed = false;
        int n_free = 0;
        double gnorm = 0.0;
#pragma unroll
        for(int i = 0; i < M; i++) {
            double hx = 0.0;
#pragma unroll
            for(int j = 0; j < M; j++) hx += H[sy(i, j)] * x[j];
            grad[i] = g[i] + hx;
            const int was = clamp[i];
            int now;
            if(x[i] <= lower[i] && grad[i] > 0)
                now = 1;
            else if(x[i] >= upper[i] && grad[i] < 0)
                now = 2;
            else {
                now = 0;
                all_clamped = false;
                gnorm += grad[i] * grad[i];
                n_free++;
            }
            if((!was) != (!now)) changed = true;
            clamp[i] = live0 ? now : was;
        }
        n_free_out = live0 ? n_free : n_free_out;
        if(live0 && all_clamped) rc = 6;  // boxQP.c:124-126

        // factor + explicit inverse of the free block when the free set changed (boxQP.c:129-146)
        if(pf) pf->probe(2);
        if constexpr(TABLE) {
            // The reference factorises when the free set changed (or in the first iteration); between two such
            // iterations the pattern — and with it the inverse — stays what it is, so the selection may run in every
            // iteration: no branch, a handful of selects.
            int pat = 0;
#pragma unroll
            for(int i = 0; i < M; i++) pat |= clamp[i] ? (1 << i) : 0;
            double inv[T];
            bool pd = true;
#pragma unroll
            for(int i = 0; i < T; i++) inv[i] = inv_of[0][i];
            pd = pd_of[0];
#pragma unroll
            for(int p = 1; p < NPAT; p++) {
                const bool is = (pat == p);
#pragma unroll
                for(int i = 0; i < T; i++) inv[i] = is ? inv_of[p][i] : inv[i];
                pd = is ? pd_of[p] : pd;
            }
            const bool fresh = (rc == 0) & ((iter == 0) | changed);
            if(fresh & !pd) rc = -1;
#pragma unroll
            for(int i = 0; i < T; i++) invH[i] = (fresh & pd) ? inv[i] : invH[i];
        } else if(rc == 0 && (iter == 0 || changed)) {
            double Hm[T], inv[T];
#pragma unroll
            for(int j = 0; j < M; j++)
#pragma unroll
                for(int i = 0; i <= j; i++)
                    Hm[ut(i, j)] = (clamp[i] || clamp[j]) ? ((i == j) ? 1.0 : 0.0) : H[ut(i, j)];
            const bool pd = chol_factor_inverse<M>(Hm, inv);
            if(!pd) rc = -1;
#pragma unroll
            for(int i = 0; i < T; i++) invH[i] = pd ? inv[i] : invH[i];
        }
        if(pf) pf->probe(3);

        if(rc == 0 && gnorm < min_grad * min_grad) rc = 5;  // boxQP.c:149-150

        // search(free) = -invH(free,free) * (g + H x_clamped)(free) - x(free); search(clamped) = 0
        double gc[M];
#pragma unroll
        for(int i = 0; i < M; i++) {
            double hc = 0.0;
#pragma unroll
            for(int j = 0; j < M; j++)
                if(clamp[j]) hc += H[sy(i, j)] * x[j];
            gc[i] = g[i] + hc;
        }
#pragma unroll
        for(int i = 0; i < M; i++) {
            double sd = -x[i];
#pragma unroll
            for(int j = 0; j < M; j++)
                if(!clamp[j]) sd -= invH[sy(i, j)] * gc[j];
            search[i] = clamp[i] ? 0.0 : sd;
        }
        double sdotg = 0.0;
#pragma unroll
        for(int i = 0; i < M; i++) sdotg += search[i] * grad[i];
        if(rc == 0 && sdotg >= 0.0) rc = -2;  // boxQP.c:189-196

        // Armijo backtracking (boxQP.c:199-227): step = 1, 0.6, 0.6*0.6, ... until the candidate passes.
        // Most calls pass at once, but the lanes of a wavefront wait for the slowest one (measured: 1.9 trials
        // per lane, 15 per wavefront by iteration 20 of the benchmark: a third of the backward step), so the loop
        // is built for the long case and every trial is as few instructions as the reference's arithmetic allows:
        //  * several consecutive step sizes per trip as independent instruction streams (2 in the first trip,
        //    ARMIJO_TRIALS_LATE in the later ones, when only the slow lanes are left);
        //  * the acceptance test without the division wherever the outcome is beyond doubt: with
        //    n = vc - oldvalue and d = step*sdotg < 0 the reference's (n / d) >= armijo holds iff
        //    n <= armijo*d (up to the rounding of the quotient).  The thresholds step*c_lo / step*c_hi are
        //    armijo*sdotg*step widened by a relative 2e-15 (18 rounding errors; the three roundings that separate
        //    them from the exact product cost 4) — below the lower one the trial passes, above the upper one it
        //    fails, in between (and only there) the reference's own expression is evaluated;
        //  * the box by min / max.  They differ from the reference's two comparisons only for NaN, and a NaN
        //    anywhere in x, H, g or the factor makes sdotg NaN: then every trial fails as in the reference (its
        //    test is false for NaN) — `sane` — and the loop runs out at minStep with rc = 2;
        //  * a trip only records WHICH step size passed; the candidate and its value are evaluated once more
        //    from it behind the loop (the same expressions: the same bits) instead of being carried through a
        //    chain of selects per trial.
        // The sequence of step sizes, the order of the exits and every value are the reference's.
        double step = 1.0, st_take = 0.0;
        bool searching = (rc == 0), took = false;
        const bool sane = sdotg < 0.0;
        const double thr = armijo * sdotg;
        const double c_lo = thr * (1.0 + 2e-15), c_hi = thr * (1.0 - 2e-15);
        int k0 = 0;  // number of the trip's first trial; the lanes still searching share it (and `step`)
        if(pf) pf->probe(2);
        // (conditions are combined with & and |, not && and ||: lane masks in scalar registers, no short-circuit
        // branches and no per-lane integers)
        auto trip = [&](auto nt_tag) {
            constexpr int NT = decltype(nt_tag)::value;
            double st[NT + 1], vt[NT];
            bool below[NT], doubt[NT], unsure = false;
            st[0] = step;
#pragma unroll
            for(int j = 0; j < NT; j++) st[j + 1] = st[j] * step_dec;
#pragma unroll
            for(int j = 0; j < NT; j++) {
                double xt[M];
#pragma unroll
                for(int i = 0; i < M; i++) xt[i] = __builtin_fmax(__builtin_fmin(x[i] + st[j] * search[i], upper[i]), lower[i]);
                vt[j] = qp_value<M>(H, g, xt);
                const double n = vt[j] - oldvalue;
                below[j] = n < st[j] * c_lo;
                doubt[j] = sane & !(below[j] | (n > st[j] * c_hi));
                unsure = unsure | doubt[j];
            }
            // ONE (rarely taken) branch for the exact expression: the trials stay in one basic block and overlap
            if(unsure) {
#pragma unroll
                for(int j = 0; j < NT; j++)
                    if(doubt[j]) below[j] = armijo_exact(vt[j], oldvalue, st[j], sdotg, armijo);
            }
            // The reference's loop reaches trial k if all before it failed and k <= ARMIJO_LAST: a failed trial k is
            // followed by step * stepDec < minStep -> return 2 exactly for k = ARMIJO_LAST (boxQP.c:222-224; the
            // step sizes are the same numbers in every call).  So: the first passing trial of this trip, if reached.
            int first = NT;
            double st_first = st[0];
#pragma unroll
            for(int j = NT - 1; j >= 0; j--) {
                const bool p = sane & below[j];
                first = p ? j : first;
                st_first = p ? st[j] : st_first;
            }
            const bool hit = searching & (first < NT) & (k0 + first <= ARMIJO_LAST);
            st_take = hit ? st_first : st_take;
            took = took | hit;
            const bool out = searching & !hit & (k0 + NT - 1 >= ARMIJO_LAST);
            rc = out ? 2 : rc;
            searching = searching & !hit & !out;
            k0 += NT;
            step = st[NT];
        };
        if(searching) trip(std::integral_constant<int, ARMIJO_TRIALS>());
        while(searching) trip(std::integral_constant<int, ARMIJO_TRIALS_LATE>());
        double xc[M], vc = value;
#pragma unroll
        for(int i = 0; i < M; i++) {
            const double xi = __builtin_fmax(__builtin_fmin(x[i] + st_take * search[i], upper[i]), lower[i]);
            xc[i] = took ? xi : x[i];
        }
        {
            const double v = qp_value<M>(H, g, xc);
            vc = took ? v : vc;
        }
        if(pf) pf->probe(4);
        const bool accepted = (rc == 0);
#pragma unroll
        for(int i = 0; i < M; i++) x[i] = accepted ? xc[i] : x[i];
        value = accepted ? vc : value;
        if(rc != 0) break;
    }
    return rc ? rc : 1;  // 1: max_iter iterations (boxQP.c:237)
}

// The same algorithm in the reference's own control flow (early returns, one trial per trip of the Armijo loop):
// for callers whose lanes all solve the SAME problem (wave mapping: the box QP of a step is evaluated redundantly
// by every lane), where nothing diverges and the predicated form above would only add work.
template <int M>
ILQG_DEV int box_qp_uniform(const double *H, const double *g, const double *lower, const double *upper, double *x,
                    int *clamp, int &n_free_out, double *invH) {
    constexpr int T = tri(M);
    const int max_iter = 100;
    const double min_grad = 1e-8, min_rel_improve = 1e-8, step_dec = 0.6, min_step = 1e-22, armijo = 0.1;
    double grad[M], search[M], xc[M];
    double value, oldvalue = 0.0;

#pragma unroll
    for(int i = 0; i < M; i++) {
        if(x[i] > upper[i]) x[i] = upper[i];
        if(x[i] < lower[i]) x[i] = lower[i];
        clamp[i] = 0;
    }
#pragma unroll
    for(int i = 0; i < T; i++) invH[i] = 0.0;
    n_free_out = 0;
    value = qp_value<M>(H, g, x);

    for(int iter = 0; iter < max_iter; iter++) {
        if(iter > 0 && (oldvalue - value) < min_rel_improve * fabs(oldvalue)) return 4;
        oldvalue = value;

        bool all_clamped = true, changed = false;
        int n_free = 0;
        double gnorm = 0.0;
#pragma unroll
        for(int i = 0; i < M; i++) {
            double hx = 0.0;
#pragma unroll
            for(int j = 0; j < M; j++) hx += H[sy(i, j)] * x[j];
            grad[i] = g[i] + hx;
            const int was = clamp[i];
            if(x[i] <= lower[i] && grad[i] > 0)
                clamp[i] = 1;
            else if(x[i] >= upper[i] && grad[i] < 0)
                clamp[i] = 2;
            else {
                clamp[i] = 0;
                all_clamped = false;
                gnorm += grad[i] * grad[i];
                n_free++;
            }
            if((!was) != (!clamp[i])) changed = true;
        }
        n_free_out = n_free;
        if(all_clamped) return 6;

        if(iter == 0 || changed) {
            double Hm[T], U[T];
#pragma unroll
            for(int j = 0; j < M; j++)
#pragma unroll
                for(int i = 0; i <= j; i++)
                    Hm[ut(i, j)] = (clamp[i] || clamp[j]) ? ((i == j) ? 1.0 : 0.0) : H[ut(i, j)];
            if(!chol_factor<M>(Hm, U)) return -1;
            chol_inverse<M>(U, invH);
        }

        if(gnorm < min_grad * min_grad) return 5;

        // search(free) = -invH(free,free) * (g + H x_clamped)(free) - x(free); search(clamped) = 0
        double gc[M];
#pragma unroll
        for(int i = 0; i < M; i++) {
            double hc = 0.0;
#pragma unroll
            for(int j = 0; j < M; j++)
                if(clamp[j]) hc += H[sy(i, j)] * x[j];
            gc[i] = g[i] + hc;
        }
#pragma unroll
        for(int i = 0; i < M; i++) {
            double s = -x[i];
#pragma unroll
            for(int j = 0; j < M; j++)
                if(!clamp[j]) s -= invH[sy(i, j)] * gc[j];
            search[i] = clamp[i] ? 0.0 : s;
        }

        double sdotg = 0.0;
#pragma unroll
        for(int i = 0; i < M; i++) sdotg += search[i] * grad[i];
        if(sdotg >= 0.0) return -2;

        double step = 1.0, vc;
        for(;;) {
#pragma unroll
            for(int i = 0; i < M; i++) {
                xc[i] = x[i] + step * search[i];
                if(xc[i] > upper[i]) xc[i] = upper[i];
                if(xc[i] < lower[i]) xc[i] = lower[i];
            }
            vc = qp_value<M>(H, g, xc);
            if(((vc - oldvalue) / (step * sdotg)) >= armijo) break;
            step = step * step_dec;
            if(step < min_step) return 2;
        }
#pragma unroll
        for(int i = 0; i < M; i++) x[i] = xc[i];
        value = vc;
    }
    return 1;
}

// ---------------------------------------------------------------------------
// one time step of back_pass.c
// ---------------------------------------------------------------------------
// Offsets of the packed per-step derivative record (doubles).  The order is
// the read order of back_pass.c; optional blocks sit at the end so that the
// device record is always a prefix of the full host record.
template <int NX, int NU, bool FULL, bool HX>
struct RecLayout {
    static constexpr int SXX = tri(NX), SUU = tri(NU), NXU = NX * NU;
    static constexpr int CX = 0;
    static constexpr int CXX = CX + NX;
    static constexpr int CU = CXX + SXX;
    static constexpr int CUU = CU + NU;
    static constexpr int CXU = CUU + SUU;
    static constexpr int FX = CXU + NXU;
    static constexpr int FU = FX + NX * NX;
    static constexpr int LOWER = FU + NXU;
    static constexpr int UPPER = LOWER + NU;
    static constexpr int BASE_END = UPPER + NU;
    static constexpr int FXX = BASE_END;
    static constexpr int FUU = FXX + (FULL ? NX * SXX : 0);
    static constexpr int FXU = FUU + (FULL ? NX * SUU : 0);
    static constexpr int FULL_END = FXU + (FULL ? NX * NXU : 0);
    static constexpr int LSIGN = FULL_END;
    static constexpr int USIGN = LSIGN + NU;
    static constexpr int LHX = USIGN + NU;
    static constexpr int UHX = LHX + NXU;
    static constexpr int HOST_SIZE = UHX + NXU;              // what the host always exchanges
    static constexpr int SIZE = HX ? HOST_SIZE : FULL_END;   // what the device stores per step
    // (by member name of trajEl_t, for lists X(member, index) of the generated header)
    static constexpr int off_cx = CX, off_cxx = CXX, off_cu = CU, off_cuu = CUU, off_cxu = CXU, off_fx = FX, off_fu = FU, off_fxx = FXX,
                         off_fuu = FUU, off_fxu = FXU;
};

// r: derivative record of this step (registers), uk: nominal control,
// Vx/Vxx: value function of step k+1 in, of step k out, l: warm start in /
// feed-forward out, K: feedback gains out (NU x NX column-major).
// Returns the box-QP code (< 1 means the sweep must be abandoned: the outputs are then meaningless).
template <int NX, int NU, bool FULL, bool HX, class Z = NoZeros>
ILQG_DEV int back_step(const double *r, const double *uk, double *Vx, double *Vxx, double *l, double *K,
                       const double lambda, const int regType, double &dV0, double &dV1, double &gsum,
                       Prof *pf = nullptr) {
    using R = RecLayout<NX, NU, FULL, HX>;
    constexpr int SXX = R::SXX, SUU = R::SUU, NXU = R::NXU;
    const double *cx = r + R::CX, *cxx = r + R::CXX, *cu = r + R::CU, *cuu = r + R::CUU, *cxu = r + R::CXU;
    const double *fx = r + R::FX, *fu = r + R::FU, *lower = r + R::LOWER, *upper = r + R::UPPER;

    double Qu[NU], Qx[NX], Qxu[NXU], Quu[SUU], Qxx[SXX];
#pragma unroll
    for(int i = 0; i < NU; i++) Qu[i] = cu[i];
    add_mul_vec<NX, NU, Z, R::FU>(Qu, Vx, fu);
#pragma unroll
    for(int i = 0; i < NX; i++) Qx[i] = cx[i];
    add_mul_vec<NX, NX, Z, R::FX>(Qx, Vx, fx);

#pragma unroll
    for(int i = 0; i < NXU; i++) Qxu[i] = cxu[i];
    add_mul2_tri<NX, NX, NU, Z, R::FX, R::FU, R::CXU>(Qxu, Vxx, fx, fu);
    if(FULL) {
        const double *fxu = r + R::FXU;
#pragma unroll
        for(int j = 0; j < NXU; j++) {
            double acc = 0.0;
#pragma unroll
            for(int i = 0; i < NX; i++) acc = mad<Z>(acc, Vx[i], fxu[j + i * NXU], R::FXU + j + i * NXU);
            Qxu[j] += acc;
        }
    }
#pragma unroll
    for(int i = 0; i < SUU; i++) Quu[i] = cuu[i];
    add_square_tri<NX, NU, Z, R::FU, R::CUU>(Quu, Vxx, fu);
    if(FULL) {
        const double *fuu = r + R::FUU;
#pragma unroll
        for(int j = 0; j < SUU; j++) {
            double acc = 0.0;
#pragma unroll
            for(int i = 0; i < NX; i++) acc = mad<Z>(acc, Vx[i], fuu[j + i * SUU], R::FUU + j + i * SUU);
            Quu[j] += acc;
        }
    }
#pragma unroll
    for(int i = 0; i < SXX; i++) Qxx[i] = cxx[i];
    add_square_tri<NX, NX, Z, R::FX, R::CXX>(Qxx, Vxx, fx);
    if(FULL) {
        const double *fxx = r + R::FXX;
#pragma unroll
        for(int j = 0; j < SXX; j++) {
            double acc = 0.0;
#pragma unroll
            for(int i = 0; i < NX; i++) acc = mad<Z>(acc, Vx[i], fxx[j + i * SXX], R::FXX + j + i * SXX);
            Qxx[j] += acc;
        }
    }

    // regularisation (back_pass.c:134-159); regType 2 keeps the reference's
    // literal index expressions
    double QuuF[SUU], Qxu_reg[NXU];
#pragma unroll
    for(int i = 0; i < SUU; i++) QuuF[i] = Quu[i];
#pragma unroll
    for(int i = 0; i < NXU; i++) Qxu_reg[i] = Qxu[i];
    if(regType == 2) {
#pragma unroll
        for(int j = 0; j < NU; j++)
#pragma unroll
            for(int i = 0; i <= j; i++) {
                double acc = 0.0;
#pragma unroll
                for(int q = 0; q < NU; q++) acc += fu[sy(q, i)] * fu[sy(q, j)];
                QuuF[ut(i, j)] += acc * lambda;
            }
#pragma unroll
        for(int i = 0; i < NX; i++)
#pragma unroll
            for(int j = 0; j < NU; j++) {
                double acc = 0.0;
#pragma unroll
                for(int q = 0; q < NX; q++) acc += fx[q + i * NX] * fu[q + j * NU];
                Qxu_reg[i + j * NX] += acc * lambda;
            }
    }
    if(regType == 1) {
#pragma unroll
        for(int i = 0; i < NU; i++) QuuF[ut(i, i)] += lambda;
    }

    int clamp[NU], n_free;
    double invH[SUU];
    if(pf) pf->probe(1);
    const int rc = box_qp<NU>(QuuF, Qu, lower, upper, l, clamp, n_free, invH, pf);
    if(pf) pf->probe(2);
    // No early return on rc < 1 (back_pass.c:168-171 abandons the sweep there): the rest of the step is
    // evaluated for every lane and the CALLER drops the lanes with rc < 1 afterwards.  A divergent return
    // here would put everything below under its own exec-mask region, cut off from the code the caller
    // places behind this step (the derivative evaluation of the next step, which overlaps with it).

    // feedback gains (back_pass.c:175-201)
#pragma unroll
    for(int i = 0; i < NU; i++) {
        double row[NX];
#pragma unroll
        for(int q = 0; q < NX; q++) row[q] = 0.0;
        if(clamp[i]) {
            if(HX) {
                const double sg = (clamp[i] == 1) ? r[R::LSIGN + i] : r[R::USIGN + i];
#pragma unroll
                for(int q = 0; q < NX; q++) row[q] -= sg * ((clamp[i] == 1) ? r[R::LHX + q + i * NX] : r[R::UHX + q + i * NX]);
            }
        } else {
#pragma unroll
            for(int j = 0; j < NU; j++) {
                if(!clamp[j]) {
#pragma unroll
                    for(int q = 0; q < NX; q++) row[q] -= invH[sy(i, j)] * Qxu_reg[q + j * NX];
                } else if(HX) {
                    double w = 0.0;
#pragma unroll
                    for(int s = 0; s < NU; s++)
                        if(!clamp[s]) w -= invH[sy(i, s)] * QuuF[sy(s, j)];
                    const double sg = (clamp[j] == 1) ? r[R::LSIGN + j] : r[R::USIGN + j];
#pragma unroll
                    for(int q = 0; q < NX; q++)
                        row[q] -= w * (sg * ((clamp[j] == 1) ? r[R::LHX + q + j * NX] : r[R::UHX + q + j * NX]));
                }
            }
        }
#pragma unroll
        for(int q = 0; q < NX; q++) K[i + q * NU] = row[q];
    }

    // expected cost change (back_pass.c:205-214)
#pragma unroll
    for(int i = 0; i < NU; i++) dV0 += Qu[i] * l[i];
#pragma unroll
    for(int i = 0; i < NU; i++) {
        double acc = 0.0;
#pragma unroll
        for(int j = 0; j < NU; j++) acc += l[j] * Quu[sy(j, i)];
        dV1 += 0.5 * l[i] * acc;
    }

    // value function with the unregularised Quu / Qxu (back_pass.c:219-241)
#pragma unroll
    for(int i = 0; i < NX; i++) Vx[i] = Qx[i];
    add_mul2_tri<NU, NX, 1>(Vx, Quu, K, l);
#pragma unroll
    for(int i = 0; i < NX; i++)
#pragma unroll
        for(int j = 0; j < NU; j++) Vx[i] += K[j + i * NU] * Qu[j];
#pragma unroll
    for(int i = 0; i < NX; i++)
#pragma unroll
        for(int j = 0; j < NU; j++) Vx[i] += Qxu[i + j * NX] * l[j];

#pragma unroll
    for(int i = 0; i < SXX; i++) Vxx[i] = Qxx[i];
    add_square_tri<NU, NX>(Vxx, Quu, K);
#pragma unroll
    for(int i = 0; i < NX; i++)
#pragma unroll
        for(int j = 0; j < NX; j++)
#pragma unroll
            for(int q = 0; q < NU; q++) {
                double term = K[q + i * NU] * Qxu[j + q * NX];
                if(i == j) term *= 2.0;
                Vxx[sy(i, j)] += term;
            }

    // gradient-norm summand (back_pass.c:246-251)
    double gmax = 0.0;
#pragma unroll
    for(int i = 0; i < NU; i++) {
        const double gi = fabs(l[i]) / (fabs(uk[i]) + 1.0);
        if(gi > gmax) gmax = gi;
    }
    gsum += gmax;
    if(pf) pf->probe(5);
    return rc;
}

}  // namespace ilqg
