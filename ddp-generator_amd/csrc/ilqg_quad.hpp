// "Quad mapping" of the backward step: 16 LANES PER TRAJECTORY, four trajectories per wavefront (N_X <= 16, N_U <= 16;
// round 4).
//
// The row mapping (ilqg_row.hpp) gives a trajectory a whole wavefront.  Its matrix products keep all 64 lanes busy, but a
// third of a step is not products: the box QP — N_U variables solved by 64 lanes in unison — the loads, the selections,
// the address arithmetic.  The kernel is bound by the NUMBER of vector instructions it issues, so what 64 lanes do in
// unison for one trajectory should serve four.  Here every 16-lane DPP row works on a trajectory of ITS OWN: its own
// step index, its own sweep, its own lambda retries, its own place in the queue of the piece.  `row_newbcast` never
// crosses a row, so the four share nothing but the instruction stream; what the wavefront pays in common is the longest
// of four box-QP iteration counts.
//
// Lane c of a row holds COLUMN c of every matrix of its trajectory in registers (Vxx, fx, Vxx fx, Qxx: N_X doubles each;
// the N_U-wide ones in the first N_U lanes, or one ROW per lane where the state is the long side: Qxu, Vxx fu, the
// gains).  A product C = A B is
//         C[r, c] += (A[r, s] of lane s, by row_newbcast:s) * B[s, c]            s ascending: the reference's order
// i.e. one v_fmac_f64_dpp per multiply-add with both operands in registers; A' B the same with the broadcast taken from
// lane r.  LDS is where a result changes shape: the sums of the tensor contraction (entry-major -> columns), Vxx fu (rows
// -> columns), and the two symmetric results whose lower triangle lives in other lanes' registers (Quu, Vxx: every lane
// writes its column and reads its row).  Diagonal entries of the symmetric products involve a lane's own columns only and
// are plain multiply-adds; no selection by lane number is needed for them.
//
// The box QP is box_qp_row's data path (lane j of the row owns variable j; every exchange a row broadcast inside the
// consuming multiply-add) under the control flow of the lane mapping's box_qp: a row that reaches one of the reference's
// exits records its code and stops committing, the wavefront leaves when all four have.  Sums over the free (or clamped)
// variables take zeros from the others — masked at the SOURCE lane, so the consuming lanes need no condition.
//
// Summation order and temporaries are the reference's (back_pass.c:80-241, boxQP.c:39-238, cholesky.c:6-74): results
// equal the row mapping's and the CPU's up to FMA contraction, bit for bit in the -ffp-contract=off build.
// Not here (the row mapping takes them): limits that depend on the state (HX), regType 2, tensors stored in the records.
#pragma once
#include "ilqg_row.hpp"
#include "ilqg_dpp_blocks.hpp"

namespace ilqg {

// ILQG_QUAD_LEAN (round 5; default 0): the step written for TWO wavefronts per SIMD — at most 256 registers, eight
// wavefronts per workgroup.  Less is kept alive: a row without a step to do gets zeros for its value function instead of
// keeping the old one through the step, the first-order entries of the record and cxx / cxu / cuu are requested where they
// are used, the box QP's diagonal lies in LDS, the contraction may run with one register set — and, what decided it, the
// sums Qxx / Qxu / Quu are PINNED in front of the box QP: the optimiser otherwise sinks these cheap additions to their
// use behind it and keeps their three operands alive instead (430 -> 254 registers, no spills with the pins; 126 spilled
// without).  The arithmetic is untouched: bit-identical to the default layout (test_quad_lean_layout_equals_the_default).
// MEASURED (config 5, profiles/r5_quad_lean.txt): two co-resident wavefronts take 37 500 cycles per step against 26 400 for
// one alone — 1.41x the throughput per SIMD while every row is busy — but a single wavefront's step takes 17 us instead
// of 12.4 (every wait the other wavefront covers is exposed to the one that waits), and the kernel is NOT bound by
// throughput at this batch size: the sweeps of a trajectory are a serial chain, the longest trajectories walk 4 000-5 000
// steps (mean 1 800: lambda retries), a worker that takes one of them late finishes at start + 4 500 x 17 us.  106-109 ms
// against 101-104 ms for the default; with 4 096 trajectories (one per row, pure chain) 62 against 42 ms.  Kept as a build
// option (bits below; -DILQG_QUAD_LEAN=63 -> eight wavefronts per workgroup) and as the `_lean` library of the tests.
#ifndef ILQG_QUAD_LEAN
#define ILQG_QUAD_LEAN 0
#endif
// the measures one by one (bits of ILQG_QUAD_LEAN)
#define ILQG_LEAN_ZERO ((ILQG_QUAD_LEAN & 1) != 0)    // a row without a step gets zeros for its value function
#define ILQG_LEAN_FIRST ((ILQG_QUAD_LEAN & 2) != 0)   // first-order entries requested behind the contraction
#define ILQG_LEAN_SINGLE ((ILQG_QUAD_LEAN & 4) != 0)  // one register set in the contraction
#define ILQG_LEAN_DG ((ILQG_QUAD_LEAN & 8) != 0)      // the factor's diagonal in LDS
#define ILQG_LEAN_LATE ((ILQG_QUAD_LEAN & 16) != 0)   // cxx / cxu / cuu requested where they are used
#define ILQG_LEAN_HERE ((ILQG_QUAD_LEAN & 32) != 0)   // sums pinned in front of the box QP

// "this value exists HERE": keeps the optimiser from sinking a cheap sum to its use behind the box QP — which keeps the
// sum's operands alive instead (three values for one)
ILQG_DEV void here(double &v) {
    if(ILQG_LEAN_HERE) asm volatile("" : "+v"(v));
}

// acc += a * b with the contraction of the DPP forms: fused in the product build, two roundings in the strict one
ILQG_DEV void mac(double &acc, const double a, const double b) {
#ifdef ILQG_STRICT_FP
    acc = acc + a * b;
#else
    acc = __builtin_fma(a, b, acc);
#endif
}

// the bits of a 64-lane mask that belong to this lane's 16-lane row
ILQG_DEV unsigned row_bits(const unsigned long long m, const int lane) { return (unsigned)(m >> (lane & 48)) & 0xffffu; }
ILQG_DEV bool any_lane(const bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// lanes (any row) whose column c = lane % 16 is > r
constexpr unsigned long long col_above(int r) {
    unsigned long long m = 0;
    for(int l = 0; l < 64; l++)
        if((l & 15) > r) m |= 1ull << l;
    return m;
}

// ---------------------------------------------------------------------------
// boxQP.c:39-238, one problem per 16-lane row.  Lane j of a row owns variable j = (lane % 16) % M: x, g, limits, clamp
// flag, row j of H (Hrow: H[me][.]) and of the inverse, column j of the Cholesky factor.
//   x        in: warm start, out: the solution (of a row that is not `active`: unchanged)
//   inv_at   LDS address of (M + 1) x M doubles of this row: exchange of the inverse's rows
//   clamp, invrow, n_free: as box_qp_row hands them back
// Returns the reference's code per row (the same in all lanes of a row); rows that are not active: 0.
// ---------------------------------------------------------------------------
template <int M>
ILQG_DEV int box_qp_quad(const double (&Hrow)[M], const double g, const double lower, const double upper, double &x, const bool active,
                         const unsigned inv_at, int &clamp_out, double (&invrow)[M], const unsigned dg_at = 0) {
    static_assert(M <= 16, "one 16-lane row holds all variables");
    constexpr int LD = M + 1;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    lane &= 63;
    const int me = (lane & 15) % M;
    const unsigned all = (1u << M) - 1u;
    const int max_iter = 100;
    const double min_grad = 1e-8, min_rel_improve = 1e-8, step_dec = 0.6, min_step = 1e-22, armijo = 0.1;

    double Ucol[M];
#pragma unroll
    for(int j = 0; j < M; j++) {
        invrow[j] = 0.0;
        Ucol[j] = 0.0;
    }
    double xs = x;  // warm start, into the box
    if(xs > upper) xs = upper;
    if(xs < lower) xs = lower;
    int clamp = 0;

    // value(y) = sum_i y_i (g_i + 0.5 (H y)_i), boxQP.c:17-37
    auto qp_value = [&](double y) {
        double hx = 0.0;
        bc_dot<M>(hx, y, Hrow);
        const double w = g + 0.5 * hx;
        double v = 0.0;
        bc_dot2<M>(v, y, w);
        return v;
    };

    double value = qp_value(xs), oldvalue = 0.0;
    int rc = active ? 0 : 99;  // 0: iterating
    for(int iter = 0; iter < max_iter; iter++) {
        if(iter > 0 && rc == 0 && (oldvalue - value) < min_rel_improve * fabs(oldvalue)) rc = 4;  // boxQP.c:85-86
        if(!any_lane(rc == 0)) break;
        const bool live0 = rc == 0;
        oldvalue = live0 ? value : oldvalue;

        // gradient and clamped set (boxQP.c:95-124)
        double hx = 0.0;
        bc_dot<M>(hx, xs, Hrow);
        const double grad = g + hx;
        const int was = clamp;
        int now = 0;
        if(xs <= lower && grad > 0)
            now = 1;
        else if(xs >= upper && grad < 0)
            now = 2;
        clamp = live0 ? now : was;
        const unsigned cm = row_bits(__builtin_amdgcn_ballot_w64(clamp != 0), lane) & all;  // this row's clamped variables
        const bool changed = (row_bits(__builtin_amdgcn_ballot_w64((!was) != (!clamp)), lane) & all) != 0u;
        if(live0 && cm == all) rc = 6;  // boxQP.c:124-126
        double gradm = clamp ? 0.0 : grad;  // (free variables only: the others add zeros)
        double gnorm = 0.0;
        bc_dot2<M>(gnorm, gradm, gradm);

        // factor + explicit inverse of the free block when the free set changed (boxQP.c:129-146, cholesky.c:6-74): the
        // Hessian with clamped rows and columns replaced by identity; lane i computes column i of U, row j in step j
        const bool fresh = rc == 0 && (iter == 0 || changed);
        if(any_lane(fresh)) {
            double Hm[M];
            static_for<0, M>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const bool masked = clamp != 0 || ((cm >> j) & 1u) != 0u;
                Hm[j] = masked ? lane_unit<lanes_of<M>(j, 0)>() : Hrow[j];  // (me == j) ? 1 : 0
            });
            double y[M];
            bool pd = true;
            auto factor = [&](auto plain_c) {
                constexpr bool PLAIN = decltype(plain_c)::value;
                bool plain = true;
                // the factor's diagonal and its reciprocals (the same in every lane of the row).  Lean build: 2 M doubles of
                // the row's LDS block instead of 4 M registers held through the inverse (every lane writes the same value)
                double dg_r[ILQG_LEAN_DG ? 1 : M], rdg_r[ILQG_LEAN_DG ? 1 : M];
                const LdsBase pdg = lds_base(dg_at);
                auto put_dg = [&](auto jc, const double d, const double r) {
                    constexpr int j = decltype(jc)::value;
                    if constexpr(ILQG_LEAN_DG) {
                        pdg[2 * j] = d;
                        pdg[2 * j + 1] = r;
                    } else {
                        dg_r[j] = d;
                        rdg_r[j] = r;
                    }
                };
                auto get_dg = [&](auto jc, double &d, double &r) {
                    constexpr int j = decltype(jc)::value;
                    if constexpr(ILQG_LEAN_DG) {
                        d = pdg.fetch(2 * j);
                        r = pdg.fetch(2 * j + 1);
                    } else {
                        d = dg_r[j];
                        r = rdg_r[j];
                    }
                };
                static_for<0, M>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    double dot = 0.0;
                    bc_dotj<j, j>(dot, Ucol, Ucol);  // sum_{k < j} U[k, j] * U[k, me]
                    const double sv = Hm[j] - dot;
                    const double piv = bc_get<j>(sv);
                    double dj, rj;
                    if constexpr(PLAIN) {
                        plain = plain && plain_range_lane(piv);
                        dj = sqrt_plain(piv);
                        rj = rcp_plain(dj);
                    } else {
                        if(piv <= 0.0) pd = false;
                        dj = sqrt(piv);
                        rj = 1.0 / dj;
                    }
                    put_dg(jc, dj, rj);
                    // (me == j) ? d : ((me > j) ? 1.0 / d * sv : 0.0)
                    Ucol[j] = lane_pick<lanes_of<M>(j, 0)>(dj, lane_pick<lanes_of<M>(j, 1)>(rj * sv, 0.0));
                });
                if constexpr(ILQG_LEAN_DG) wave_sync();
                // explicit inverse: lane l solves U'U y = e_l; y[k] for k >= l is row l of the inverse
                static_for<0, M>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    double v = lane_unit<lanes_of<M>(k, 0)>();
                    bc_dotjn<k, k>(v, Ucol, y);  // v -= sum_{i < k} y[i] * U[i, k]    (y[i] = 0 for i < l: exact zeros)
                    double dk, rk;
                    get_dg(kc, dk, rk);
                    y[k] = PLAIN ? div_plain(v, dk, rk) : v / dk;
                });
                static_for<0, M>([&](auto kr) {
                    constexpr int k = M - 1 - decltype(kr)::value;
                    double v = y[k];
                    bc_dotn<M - 1 - k, k + 1>(v, Ucol[k], y + k + 1);  // v -= sum_{i > k} y[i] * U[k, i]
                    double dk, rk;
                    get_dg(std::integral_constant<int, k>{}, dk, rk);
                    y[k] = PLAIN ? div_plain(v, dk, rk) : v / dk;
                });
                return plain;
            };
            // short forms of sqrt / reciprocal / quotient while every pivot of every row that factorises is in their range
            // (the same bits), else once more in the general form (which also finds a pivot <= 0)
            const bool plain = factor(std::true_type{});
            if(any_lane(fresh && !plain)) factor(std::false_type{});
            wave_sync();
            // Row `me` of the (symmetric) inverse: entries j >= me are the lane's own y[j]; entry j < me is y[me] of
            // lane j, through LDS — every lane lays down its y (what lies left of the diagonal is never read)
            {
                const LdsBase pr = lds_base(inv_at + me * (LD * 8));
#pragma unroll
                for(int k = 0; k < M; k++) pr[k] = y[k];
            }
            wave_sync();
            const LdsBase pc = lds_base(inv_at + me * 8);
            const bool take = fresh && pd;
            static_for<0, M>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const double other = pc[j * LD];
                const double v = lane_pick<lanes_of<M>(j, 1)>(other, y[j]);  // (me > j) ? lane j's : own
                invrow[j] = take ? v : invrow[j];
            });
            wave_sync();
            if(fresh && !pd) rc = -1;
        }

        if(rc == 0 && gnorm < min_grad * min_grad) rc = 5;  // boxQP.c:149-150

        // search(free) = -invH(free,free) (g + H x_clamped)(free) - x(free); search(clamped) = 0 (boxQP.c:170-196)
        double xcl = clamp ? xs : 0.0;
        double hc = 0.0;
        bc_dot<M>(hc, xcl, Hrow);
        double gcm = clamp ? 0.0 : g + hc;
        double sr = -xs;
        bc_dotn<M>(sr, gcm, invrow);
        double search = clamp ? 0.0 : sr;
        double sdotg = 0.0;
        bc_dot2<M>(sdotg, search, grad);
        if(rc == 0 && sdotg >= 0.0) rc = -2;  // boxQP.c:189-196

        // Armijo backtracking (boxQP.c:199-227): every row walks the same sequence of step sizes; a row that has passed
        // keeps its candidate, a row still searching when the step falls below minStep leaves with 2
        double step = 1.0, xn = xs, vn = value;
        bool searching = rc == 0;
        while(any_lane(searching)) {
            double xc = xs + step * search;
            if(xc > upper) xc = upper;
            if(xc < lower) xc = lower;
            const double vc = qp_value(xc);
            const bool pass = ((vc - oldvalue) / (step * sdotg)) >= armijo;
            const bool hit = searching && pass;
            xn = hit ? xc : xn;
            vn = hit ? vc : vn;
            searching = searching && !pass;
            step = step * step_dec;
            if(step < min_step) {  // (the same in every lane)
                rc = searching ? 2 : rc;
                searching = false;
            }
        }
        const bool accepted = rc == 0;
        xs = accepted ? xn : xs;
        value = accepted ? vn : value;
    }
    if(rc == 0) rc = 1;  // max_iter iterations (boxQP.c:237)
    if(rc == 99) rc = 0;
    x = active ? xs : x;
    clamp_out = clamp;
    return rc;
}

// LDS of one trajectory (one 16-lane row) in the quad mapping.  The arrays of the early part of a step (the sums of the
// tensor contraction) give their space to what comes later: Vxx fu on its way from rows to columns and Quu on its way
// to full rows (in dxx's, once Qxx has taken its sums), the box QP's exchange of the inverse (in dxu's), and at the end
// of the step the whole block to Vxx on its way to full columns.
template <int NX, int NU>
struct QuadRow {
    static constexpr int SXX = tri(NX), SUU = tri(NU), NXU = NX * NU;
    static constexpr int LDX = NX + 1, LDU = NU + 1;
    static constexpr int p16(int n) { return (n + 31) / 32 * 32; }  // a lane takes entries 2c, 2c + 1 of every 32
    // entries e = 2 c + 32 q, + 1 of a lane (16-byte LDS accesses); + NX: reads of rows beyond the diagonal of the last columns stay inside
    static constexpr int DXX = p16(SXX) + NX, DUU = p16(SUU) + NU, DXU = p16(NXU);
    static constexpr int EARLY = DXX + DUU + DXU, T2N = LDX * NU, QN = LDU * NU, VN = LDX * NX;
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    // offsets (doubles).  Order of use within a step: dxx, duu, dxu written (contraction); dxx read (Qxx); dxu read (Qxu);
    // duu read; t2 written / read; quu written / read; inv (box QP); vxx (end of the step).
    static constexpr int dxx = 0, duu = DXX, dxu = DXX + DUU;
    static constexpr int t2 = 0, quu = T2N;                 // behind each other, from the start of the block
    static_assert(T2N + QN <= dxu, "Vxx fu and Quu on their way must not reach dxu, which Qxu may still have to read");
    static constexpr int inv = dxu;
    static constexpr int vxx = 0;
    static constexpr int BODY = cmax(cmax(EARLY, VN), cmax(quu + QN, inv + QN));
    static constexpr int basis = BODY;                      // 64 doubles: the step's products
    static constexpr int SIZE = BODY + 64;                  // doubles
};

// One backward step of four trajectories.  Per lane (the values of its row's trajectory):
//   rb       LDS address of the row's QuadRow block;  table: LDS address of the coefficient tables (FACT)
//   rec      the step's derivative record;  nom_u: the step's nominal inputs;  lout / Kout: where its gains go
//   live     the row has a step to do (else it computes along on whatever rec points at and commits nothing)
//   vx, vxx  Vx[c], column c of Vxx of step k+1 in, of step k out;  lcur: l[me] (warm start in, solution out)
// Returns the box-QP code of the row (< 1: the sweep is abandoned, back_pass.c:168-171; nothing of the row's state is
// meaningful then).
template <int NX, int NU, bool FULL, bool FACT, class R, class Tab>
__device__ __forceinline__ int back_step_quad(const unsigned rb, const unsigned table, const char *rec, const double *nom_u, double *lout,
                                              double *Kout, const bool live, double &vx, double (&vxx)[NX], double &lcur,
                                              const double lambda, double &dV0, double &dV1, double &gsum, Prof *pf = nullptr) {
    static_assert(NX <= 16 && NU <= 16 && NU <= NX, "one 16-lane row per trajectory");
    using Q = QuadRow<NX, NU>;
    constexpr int SXX = tri(NX), SUU = tri(NU), NXU = NX * NU;
    constexpr int LDX = NX + 1, LDU = NU + 1;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    lane &= 63;
    const int c = lane & 15;
    const int cx_ = (c < NX) ? c : 0;  // the state column / row this lane holds
    const int me = c % NU;             // the input column / variable this lane holds
    auto ldd = [&](unsigned off) { return *reinterpret_cast<const double *>(rec + off); };

    // ---- the step's record
    double fxc[NX], fuc[NX];  // column cx_ of fx, column me of fu
    double cxl, cul, lo_k, up_k, u_l;
    auto load_first_order = [&]() {
#pragma unroll
        for(int s = 0; s < NX; s++) fxc[s] = ldd(R::fx + (unsigned)(s + cx_ * NX) * 8u);
#pragma unroll
        for(int s = 0; s < NX; s++) fuc[s] = ldd(R::fu + (unsigned)(s + me * NX) * 8u);
        cxl = ldd(R::cx + (unsigned)cx_ * 8u);
        cul = ldd(R::cu + (unsigned)me * 8u);
        lo_k = ldd(R::lower + (unsigned)me * 8u);
        up_k = ldd(R::upper + (unsigned)me * 8u);
        u_l = nom_u[me];
    };
    // (lean build, two wavefronts per SIMD: the first-order entries are requested BEHIND the contraction — 37 doubles less
    // alive through it; the other wavefront of the SIMD covers the wait)
    if(!(ILQG_LEAN_FIRST && FULL && FACT)) load_first_order();
    const int bxx = cx_ * (cx_ + 1) / 2, buu = me * (me + 1) / 2;  // packed column starts: entry (r, c) = b + r, r <= c
    // (The vector ALU addresses 256 registers = 128 doubles; what a wavefront holds beyond that sits in accumulation
    // registers and costs a copy per use.  So every block below loads what it needs itself, right in front of its
    // arithmetic, and scheduling barriers keep the blocks apart: the other wavefront of the SIMD covers the latency.)
    if(pf) pf->probe(0);

    // T1 = Vxx fx (column cx_), T2 = Vxx fu (ROW cx_: T2[cx_, j]), term s of every sum.  (Issued between the slices of the
    // contraction they hide its LDS reads — and need 139 doubles alive at once: measured slower than one after the other.)
    double t1[NX], t2r[NU];
#pragma unroll
    for(int r = 0; r < NX; r++) t1[r] = 0.0;
#pragma unroll
    for(int j = 0; j < NU; j++) t2r[j] = 0.0;
    auto products_of = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        bc_rows<s, NX>(t1, vxx, fxc[s]);       // Vxx[r, s] fx[s, c], all r
        bc_cols<NU>(t2r, fuc[s], vxx[s]);      // fu[s, j] Vxx[s, c] = Vxx[c, s] fu[s, j], all j
    };

    // ---- second-order terms of the dynamics (back_pass.c:95-131): d[e] = sum_i Vx[i] * (coefficient_i[e] * product), i
    // ascending; a lane takes the entries e = c, c + 16, ... of every array and hands the sums over through LDS
    if constexpr(FULL && FACT) {
        // a lane takes the entries e = 2 c + 32 q and e + 1 of every array (pairs: 16-byte LDS reads of the coefficients,
        // half as many LDS instructions; an LDS instruction costs a wavefront about two multiply-adds of issue time)
        constexpr int NTX = (SXX + 31) / 32 * 2, NTU = (SUU + 31) / 32 * 2, NTC = (NXU + 31) / 32 * 2;
        constexpr int NB = Tab::NBASIS, NBL = (NB + 15) / 16;
        {   // the step's products, for all lanes of the row to read
            const LdsBase pb = lds_base(rb + (Q::basis + c) * 8);
#pragma unroll
            for(int q = 0; q < NBL; q++) pb[16 * q] = ldd(R::fxx + (unsigned)((c + 16 * q < NB) ? c + 16 * q : 0) * 8u);
        }
        wave_sync();
        double dxx[NTX], duu[NTU], dxu[NTC];
#pragma unroll
        for(int q = 0; q < NTX; q++) dxx[q] = 0.0;
#pragma unroll
        for(int q = 0; q < NTU; q++) duu[q] = 0.0;
#pragma unroll
        for(int q = 0; q < NTC; q++) dxu[q] = 0.0;
        const LdsBase pg = lds_base(rb + Q::basis * 8);
        unsigned pt2 = table + c * 16;  // the lane's pair of every 32 entries
        asm("" : "+v"(pt2));
        typedef double dpair __attribute__((ext_vector_type(2)));
        using lds_pair = __attribute__((address_space(3))) volatile dpair;
        auto pair_at = [&](int doubles) { return ((lds_pair *)(uintptr_t)pt2)[doubles / 2]; };  // (doubles: even)
        constexpr int PER = NTX + NTU + NTC;
        // what this lane multiplies of slice i: its entries of xx, uu, xu and the three products
        auto fetch = [&](auto ic, double (&t)[PER], double (&gg)[3]) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for(int q = 0; q < NTX; q += 2) {
                const dpair v = pair_at(i * Tab::SLICE + 16 * q);
                t[q] = v.x;
                t[q + 1] = v.y;
            }
#pragma unroll
            for(int q = 0; q < NTU; q += 2) {
                const dpair v = pair_at(i * Tab::SLICE + SXX + 16 * q);
                t[NTX + q] = v.x;
                t[NTX + q + 1] = v.y;
            }
#pragma unroll
            for(int q = 0; q < NTC; q += 2) {
                const dpair v = pair_at(i * Tab::SLICE + SXX + SUU + 16 * q);
                t[NTX + NTU + q] = v.x;
                t[NTX + NTU + q + 1] = v.y;
            }
            const int sxx = Tab::slice_xx(i), suu = Tab::slice_uu(i), sxu = Tab::slice_xu(i);  // (constants once unrolled)
            gg[0] = pg.fetch(sxx);
            gg[1] = (suu == sxx) ? gg[0] : pg.fetch(suu);
            gg[2] = (sxu == sxx) ? gg[0] : ((sxu == suu) ? gg[1] : pg.fetch(sxu));
        };
        // A stage: the reads of the NEXT slice first (into the other of two register sets), then this slice's coefficients
        // times their product — which waits for reads issued a whole stage ago, with the ones just issued still in flight
        // (the wait counter of LDS operations holds 15: a slice's 11 + 3 reads fit) — then the multiply-adds.
        double ca[PER], cb[PER], ga[3], gb[3], m[PER];
        if(!ILQG_LEAN_SINGLE) fetch(std::integral_constant<int, 0>{}, ca, ga);
        static_for<0, NX>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            double(&cur)[PER] = (i % 2 && !ILQG_LEAN_SINGLE) ? cb : ca;
            double(&nxt)[PER] = (i % 2) ? ca : cb;
            double(&gc)[3] = (i % 2 && !ILQG_LEAN_SINGLE) ? gb : ga;
            double(&gn)[3] = (i % 2) ? ga : gb;
            if constexpr(ILQG_LEAN_SINGLE) {
                // (lean build: ONE register set — the slice's reads, then its arithmetic; the other wavefront of the SIMD
                // issues meanwhile)
                __builtin_amdgcn_sched_barrier(0);
                fetch(std::integral_constant<int, i>{}, ca, ga);
            } else if constexpr(i + 1 < NX) fetch(std::integral_constant<int, i + 1>{}, nxt, gn);
#ifdef ILQG_STRICT_FP
            // the reference's two roundings: t->fxx[e] = coefficient * product, then d += Vx[i] * t->fxx[e]
#pragma unroll
            for(int q = 0; q < NTX; q++) m[q] = cur[q] * gc[0];
#pragma unroll
            for(int q = 0; q < NTU; q++) m[NTX + q] = cur[NTX + q] * gc[1];
#pragma unroll
            for(int q = 0; q < NTC; q++) m[NTX + NTU + q] = cur[NTX + NTU + q] * gc[2];
            bc_vecs<i, NTC>(dxu, vx, m + NTX + NTU);
            bc_vecs<i, NTU>(duu, vx, m + NTX);
            bc_vecs<i, NTX>(dxx, vx, m);
#else
            // product build (FMA contraction anyway): d += coefficient * (Vx[i] * product) — three broadcasts per slice
            // instead of one per entry, the entries plain multiply-adds: 28 instead of 44 vector instructions per slice
            double w[3] = {0.0, 0.0, 0.0};
            bc_vecs<i, 3>(w, vx, gc);
#pragma unroll
            for(int q = 0; q < NTC; q++) dxu[q] = __builtin_fma(cur[NTX + NTU + q], w[2], dxu[q]);
#pragma unroll
            for(int q = 0; q < NTU; q++) duu[q] = __builtin_fma(cur[NTX + q], w[1], duu[q]);
#pragma unroll
            for(int q = 0; q < NTX; q++) dxx[q] = __builtin_fma(cur[q], w[0], dxx[q]);
#endif
        });
        using lds_pair_w = __attribute__((address_space(3))) dpair;
        lds_pair_w *const pd2 = (lds_pair_w *)(uintptr_t)(rb + c * 16);
#pragma unroll
        for(int q = 0; q < NTC; q += 2) pd2[(Q::dxu + 16 * q) / 2] = dpair{dxu[q], dxu[q + 1]};
#pragma unroll
        for(int q = 0; q < NTU; q += 2) pd2[(Q::duu + 16 * q) / 2] = dpair{duu[q], duu[q + 1]};
#pragma unroll
        for(int q = 0; q < NTX; q += 2) pd2[(Q::dxx + 16 * q) / 2] = dpair{dxx[q], dxx[q + 1]};
        wave_sync();
        if(ILQG_LEAN_FIRST) {
            __builtin_amdgcn_sched_barrier(0);
            load_first_order();
        }
    } else if constexpr(FULL) {
        // STORED tensors (a pair without factored tables: the record carries fxx / fuu / fxu as the generated bp_derivsL
        // wrote them, back_pass.c:95-131 reads them back): the same split of the entries over the row's 16 lanes — a lane
        // takes the pairs e = 2 c + 32 q', e + 1 of every slice, one 16-byte load each — but the operand comes from HBM,
        // so the loads run DEPTH slices ahead of the multiply-adds (4 trajectories x 11 loads of 256 bytes per slice).
        // d += Vx[i] * t->f??[e] with the reference's one rounding per term in either build.
        constexpr int NTX = (SXX + 31) / 32 * 2, NTU = (SUU + 31) / 32 * 2, NTC = (NXU + 31) / 32 * 2;
        constexpr int PER = NTX + NTU + NTC;
        typedef double dpair __attribute__((ext_vector_type(2)));
        using gpair = const __attribute__((address_space(1))) dpair;
        double dxx[NTX], duu[NTU], dxu[NTC];
#pragma unroll
        for(int q = 0; q < NTX; q++) dxx[q] = 0.0;
#pragma unroll
        for(int q = 0; q < NTU; q++) duu[q] = 0.0;
#pragma unroll
        for(int q = 0; q < NTC; q++) dxu[q] = 0.0;
        auto fetch = [&](auto ic, double (&t)[PER]) {
            constexpr int i = decltype(ic)::value;
            // the lane's pair e = 2 c + 16 q, e + 1 of slice i of a tensor with `size` entries per slice (pairs beyond the slice —
            // its length is not a multiple of 32 — read the slice's first pair and count as zeros)
            auto pair_of = [&](unsigned member, int size, int q, double &a, double &b) {
                const bool whole = 16 * q + 32 <= size;  // (every lane's pair is inside: known once q is unrolled)
                const bool in = whole || 2 * c + 16 * q < size;
                const int e = in ? 2 * c + 16 * q : 0;
                const dpair v = *(gpair *)(rec + member + (unsigned)(i * size + e) * 8u);
                a = in ? v.x : 0.0;
                b = in ? v.y : 0.0;
            };
#pragma unroll
            for(int q = 0; q < NTX; q += 2) pair_of(R::fxx, SXX, q, t[q], t[q + 1]);
#pragma unroll
            for(int q = 0; q < NTU; q += 2) pair_of(R::fuu, SUU, q, t[NTX + q], t[NTX + q + 1]);
#pragma unroll
            for(int q = 0; q < NTC; q += 2) pair_of(R::fxu, NXU, q, t[NTX + NTU + q], t[NTX + NTU + q + 1]);
        };
        static_assert(SXX % 2 == 0 && SUU % 2 == 0 && NXU % 2 == 0, "a pair of entries does not straddle two slices");
#ifndef ILQG_QUAD_STORED_DEPTH
#define ILQG_QUAD_STORED_DEPTH 4
#endif
        constexpr int DEPTH = ILQG_QUAD_STORED_DEPTH < NX ? ILQG_QUAD_STORED_DEPTH : NX;
        double buf[DEPTH][PER];
        static_for<0, DEPTH>([&](auto ic) { fetch(ic, buf[decltype(ic)::value]); });
        static_for<0, NX>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            double m[PER];
#pragma unroll
            for(int q = 0; q < PER; q++) m[q] = buf[i % DEPTH][q];
            if constexpr(i + DEPTH < NX) fetch(std::integral_constant<int, i + DEPTH>{}, buf[i % DEPTH]);
            bc_vecs<i, NTC>(dxu, vx, m + NTX + NTU);
            bc_vecs<i, NTU>(duu, vx, m + NTX);
            bc_vecs<i, NTX>(dxx, vx, m);
        });
        using lds_pair_w = __attribute__((address_space(3))) dpair;
        lds_pair_w *const pd2 = (lds_pair_w *)(uintptr_t)(rb + c * 16);
#pragma unroll
        for(int q = 0; q < NTC; q += 2) pd2[(Q::dxu + 16 * q) / 2] = dpair{dxu[q], dxu[q + 1]};
#pragma unroll
        for(int q = 0; q < NTU; q += 2) pd2[(Q::duu + 16 * q) / 2] = dpair{duu[q], duu[q + 1]};
#pragma unroll
        for(int q = 0; q < NTX; q += 2) pd2[(Q::dxx + 16 * q) / 2] = dpair{dxx[q], dxx[q + 1]};
        wave_sync();
    }

    if(pf) pf->probe(1);
    // ---- Qx = cx + fx'Vx, Qu = cu + fu'Vx
    double qxl = cxl, qul = cul;
    bc_dot<NX>(qxl, vx, fxc);
    bc_dot<NX>(qul, vx, fuc);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NX>([&](auto sc) { products_of(sc); });

    if(pf) pf->probe(2);
    // ---- Qxx = cxx + fx'T1 symmetrised (+ contraction): rows r < c of column c, and the diagonal entry apart
    double qxx[NX], qxx_d;
    __builtin_amdgcn_sched_barrier(0);
    {
        double cxx_c[NX], cxx_d;  // cxx[r, cx_] (r <= cx_; rows beyond the diagonal: inside the array, unused)
        auto load_cxx = [&]() {
#pragma unroll
            for(int r = 0; r < NX; r++) cxx_c[r] = ldd(R::cxx + (unsigned)(bxx + r) * 8u);
            cxx_d = ldd(R::cxx + (unsigned)(bxx + cx_) * 8u);
        };
        if(!ILQG_LEAN_LATE) load_cxx();  // (lean build: behind the products, where they are used)
        double a[NX];
#pragma unroll
        for(int r = 0; r < NX; r++) a[r] = 0.0;
#pragma unroll
        for(int s = 0; s < NX; s++) bc_cols<NX>(a, fxc[s], t1[s]);  // fx[s, r] T1[s, c], all r
        double dsum = 0.0;
#pragma unroll
        for(int s = 0; s < NX; s++) mac(dsum, fxc[s], t1[s]);  // fx[s, c] T1[s, c]
#ifdef ILQG_STRICT_FP
#pragma unroll
        for(int s = 0; s < NX; s++) bc_cols<NX>(a, t1[s], fxc[s]);  // + fx[s, c] T1[s, r]
        constexpr double HALF = 0.5;
#else
        // (product build: fx'(Vxx fx) is symmetric but for rounding — the reference's second half sum, the same entry of
        // the transpose, and the halving are left out; the FMA-free twin keeps the reference's arithmetic.  Likewise Quu
        // and K'(Quu K) below: 450 of a wavefront step's 4 350 vector instructions)
        constexpr double HALF = 1.0;
#endif
        if(ILQG_LEAN_LATE) {
            __builtin_amdgcn_sched_barrier(0);
            load_cxx();
        }
        const LdsBase pd = lds_base(rb + (Q::dxx + bxx) * 8);
#pragma unroll
        for(int r = 0; r < NX; r++) {
            double v = cxx_c[r] + a[r] * HALF;
            if(FULL) v += pd[r];
            here(v);
            qxx[r] = v;
        }
        qxx_d = cxx_d + dsum;
        if(FULL) qxx_d += pd[cx_];
        here(qxx_d);
    }
    wave_sync();

    if(pf) pf->probe(3);
    // ---- Qxu (row cx_): cxu + fx'T2 (+ contraction)
    double qxu[NU];
    __builtin_amdgcn_sched_barrier(0);
    {
        double cxu_r[NU];  // cxu[cx_, j]
        auto load_cxu = [&]() {
#pragma unroll
            for(int j = 0; j < NU; j++) cxu_r[j] = ldd(R::cxu + (unsigned)(cx_ + j * NX) * 8u);
        };
        if(!ILQG_LEAN_LATE) load_cxu();
        double a[NU];
#pragma unroll
        for(int j = 0; j < NU; j++) a[j] = 0.0;
        static_for<0, NX>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            bc_rows<s, NU>(a, t2r, fxc[s]);  // T2[s, j] fx[s, r], all j
        });
        if(ILQG_LEAN_LATE) {
            __builtin_amdgcn_sched_barrier(0);
            load_cxu();
        }
        const LdsBase pd = lds_base(rb + (Q::dxu + cx_) * 8);
#pragma unroll
        for(int j = 0; j < NU; j++) {
            double v = cxu_r[j] + a[j];
            if(FULL) v += pd[j * NX];
            here(v);
            qxu[j] = v;
        }
    }

    // ---- T2 from rows to columns (through what was dxx), then Quu = cuu + fu'T2 symmetrised (+ contraction)
    double quu[NU], hrow[NU];  // Quu[me, .], the regularised one
    __builtin_amdgcn_sched_barrier(0);
    {
        double cuu_c[NU], cuu_d;  // cuu[i, me] (i <= me)
        auto load_cuu = [&]() {
#pragma unroll
            for(int i = 0; i < NU; i++) cuu_c[i] = ldd(R::cuu + (unsigned)(buu + i) * 8u);
            cuu_d = ldd(R::cuu + (unsigned)(buu + me) * 8u);
        };
        if(!ILQG_LEAN_LATE) load_cuu();
        double duu_c[NU], duu_d = 0.0;
        if(FULL) {
            const LdsBase pd = lds_base(rb + (Q::duu + buu) * 8);
#pragma unroll
            for(int i = 0; i < NU; i++) duu_c[i] = pd[i];
            duu_d = pd[me];
        }
        wave_sync();
        {
            const LdsBase w = lds_base(rb + (Q::t2 + cx_) * 8);
            if(c < NX) {
#pragma unroll
                for(int j = 0; j < NU; j++) w[j * LDX] = t2r[j];
            }
        }
        wave_sync();
        double t2c[NX];
        {
            const LdsBase p = lds_base(rb + (Q::t2 + me * LDX) * 8);
#pragma unroll
            for(int s = 0; s < NX; s++) t2c[s] = p[s];
        }
        double a[NU];
#pragma unroll
        for(int i = 0; i < NU; i++) a[i] = 0.0;
#pragma unroll
        for(int s = 0; s < NX; s++) bc_cols<NU>(a, fuc[s], t2c[s]);  // fu[s, i] T2[s, me], all i
        double dsum = 0.0;
#pragma unroll
        for(int s = 0; s < NX; s++) mac(dsum, fuc[s], t2c[s]);
#ifdef ILQG_STRICT_FP
#pragma unroll
        for(int s = 0; s < NX; s++) bc_cols<NU>(a, t2c[s], fuc[s]);  // + fu[s, me] T2[s, i]
        constexpr double HALF = 0.5;
#else
        constexpr double HALF = 1.0;
#endif
        if(ILQG_LEAN_LATE) {
            __builtin_amdgcn_sched_barrier(0);
            load_cuu();
        }
        double col[NU];  // Quu[i, me], i < me
#pragma unroll
        for(int i = 0; i < NU; i++) {
            double v = cuu_c[i] + a[i] * HALF;
            if(FULL) v += duu_c[i];
            here(v);
            col[i] = v;
        }
        double qd = cuu_d + dsum;
        if(FULL) qd += duu_d;
        here(qd);
        // every lane lays down its column (rows above the diagonal are good, the diagonal entry apart) and reads its row
        wave_sync();
        {
            const LdsBase w = lds_base(rb + (Q::quu + me * LDU) * 8);
#pragma unroll
            for(int i = 0; i < NU; i++) w[i] = col[i];
            w[me] = qd;
        }
        wave_sync();
        const LdsBase p = lds_base(rb + (Q::quu + me) * 8);
        static_for<0, NU>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const double other = p[j * LDU];                                      // Quu[me, j] of column j: good for j >= me
            quu[j] = lane_pick<lanes_of<NU>(j, 1)>(col[j], other);                // (me > j) ? own : column j's
            hrow[j] = lane_pick<lanes_of<NU>(j, 0)>(quu[j] + lambda, quu[j]);     // regType 1: + lambda on the diagonal
        });
        wave_sync();
    }

    if(pf) pf->probe(4);
    __builtin_amdgcn_sched_barrier(0);
    // ---- box QP of the row; warm start: the later step's solution (back_pass.c:163-166)
    int mine;
    double ih[NU];  // invH[me, .]
    double lsol = lcur;
    const int rc = box_qp_quad<NU>(hrow, qul, lo_k, up_k, lsol, live, rb + Q::inv * 8, mine, ih, rb + Q::basis * 8);
    const bool ok = live && rc >= 1;
    if(pf) pf->probe(5);
    __builtin_amdgcn_sched_barrier(0);

    // ---- feedback gains (back_pass.c:175-201), column cx_ of K: K[i, cx_] = - sum_{j free} invH[i, j] Qxu(reg)[cx_, j];
    // rows of clamped inputs are zero (their entries of the embedded inverse are)
    double kt[NU];
    {
        double ihm[NU];  // row me of the inverse, zero if input me is clamped (masked at the source)
#pragma unroll
        for(int i = 0; i < NU; i++) ihm[i] = mine ? 0.0 : ih[i];
#pragma unroll
        for(int i = 0; i < NU; i++) kt[i] = 0.0;
        static_for<0, NU>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            bc_rowsn<j, NU>(kt, ihm, qxu[j]);  // - invH[j, i] Qxu[cx_, j], all i
        });
        if(ok && c < NX) {
            double *const ko = Kout + (unsigned)(cx_ * NU);
#pragma unroll
            for(int i = 0; i < NU; i++) ko[i] = kt[i];
        }
        if(ok && c < NU) lout[c] = lsol;
    }

    // ---- Quu l, Quu K; expected cost change (back_pass.c:205-214)
    double ba[NU];  // (Quu K)[i, cx_]
    double bcl = 0.0, ll = lsol;
    {
        bc_dot<NU>(bcl, ll, quu);  // (Quu l)[me]
#pragma unroll
        for(int i = 0; i < NU; i++) ba[i] = 0.0;
        static_for<0, NU>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            bc_rows<j, NU>(ba, quu, kt[j]);  // Quu[i, j] K[j, cx_], all i
        });
        const double hl = 0.5 * ll;
        double d0 = dV0, d1 = dV1;
        bc_dot2<NU>(d0, qul, ll);
        bc_dot2<NU>(d1, hl, bcl);
        dV0 = ok ? d0 : dV0;
        dV1 = ok ? d1 : dV1;
    }

    // ---- Vx, Vxx with the unregularised Quu / Qxu (back_pass.c:219-241)
    {
        // Vx[c] = Qx[c] + K[:, c]'(Quu l) + K[:, c]'Qu + Qxu[c, :] l
#ifdef ILQG_STRICT_FP
        double d = 0.0;
        bc_dot<NU>(d, bcl, kt);
        double vxn = qxl + d;
        bc_dot<NU>(vxn, qul, kt);
#else
        double vxn = qxl;  // (product build: K'(Quu l + Qu) in one pass)
        const double bq = bcl + qul;
        bc_dot<NU>(vxn, bq, kt);
#endif
        bc_dot<NU>(vxn, ll, qxu);

        // Vxx[r, c], r < c, and the diagonal entry apart
        double vd = qxx_d;
        double vv[NX];
#ifdef ILQG_STRICT_FP
        double a[NX];
#pragma unroll
        for(int r = 0; r < NX; r++) a[r] = 0.0;
#pragma unroll
        for(int s = 0; s < NU; s++) bc_cols<NX>(a, kt[s], ba[s]);  // K[s, r] (Quu K)[s, c], all r
        {
            double dsum = 0.0;
#pragma unroll
            for(int s = 0; s < NU; s++) mac(dsum, kt[s], ba[s]);
            vd = vd + dsum;
        }
#pragma unroll
        for(int s = 0; s < NU; s++) bc_cols<NX>(a, ba[s], kt[s]);  // + K[s, c] (Quu K)[s, r]
#pragma unroll
        for(int r = 0; r < NX; r++) vv[r] = qxx[r] + a[r] * 0.5;
        // the reference's loop nest touches packed entry (r, c) first as (i = r, j = c), then as (i = c, j = r); a
        // diagonal entry once, with the term doubled
#pragma unroll
        for(int q = 0; q < NU; q++) bc_cols<NX>(vv, kt[q], qxu[q]);  // K[q, r] Qxu[c, q]
#pragma unroll
        for(int q = 0; q < NU; q++) bc_cols<NX>(vv, qxu[q], kt[q]);  // K[q, c] Qxu[r, q]
#pragma unroll
        for(int q = 0; q < NU; q++) mac(vd, kt[q], qxu[q] * 2.0);
#else
        // product build: K'(Quu K) + K'Qxu' + Qxu K = K'(Quu K + Qxu') + Qxu K — two passes over the inputs instead of
        // the reference's four (the first half sum of the symmetric term alone, like Qxx and Quu)
        double wq[NU];  // (Quu K + Qxu')[q, c]
#pragma unroll
        for(int q = 0; q < NU; q++) wq[q] = ba[q] + qxu[q];
#pragma unroll
        for(int r = 0; r < NX; r++) vv[r] = qxx[r];
#pragma unroll
        for(int q = 0; q < NU; q++) bc_cols<NX>(vv, kt[q], wq[q]);   // K[q, r] (Quu K + Qxu')[q, c]
#pragma unroll
        for(int q = 0; q < NU; q++) bc_cols<NX>(vv, qxu[q], kt[q]);  // Qxu[r, q] K[q, c]
#pragma unroll
        for(int q = 0; q < NU; q++) mac(vd, kt[q], wq[q] + qxu[q]);  // K[q, c] (Quu K)[q, c] + 2 K[q, c] Qxu[c, q]
#endif
        // columns -> full columns: every lane lays down its column, the diagonal entry on top, and reads its row
        wave_sync();
        {
            const LdsBase w = lds_base(rb + (Q::vxx + cx_ * LDX) * 8);
            if(c < NX) {
#pragma unroll
                for(int r = 0; r < NX; r++) w[r] = vv[r];
                w[cx_] = vd;
            }
        }
        wave_sync();
        const LdsBase p = lds_base(rb + (Q::vxx + cx_) * 8);
        static_for<0, NX>([&](auto rc_) {
            constexpr int r = decltype(rc_)::value;
            const double other = p[r * LDX];                                // Vxx[c, r] of column r: good for r >= c
            const double v = lane_pick<col_above(r)>(vv[r], other);       // (c > r) ? own : column r's
            vxx[r] = ok ? v : (ILQG_LEAN_ZERO ? 0.0 : vxx[r]);
        });
        vx = ok ? vxn : (ILQG_LEAN_ZERO ? 0.0 : vx);
        wave_sync();
    }
    lcur = ok ? lsol : lcur;

    // gradient-norm summand (back_pass.c:246-251)
    {
        double gl = fabs(ll) / (fabs(u_l) + 1.0);
        double gmax = 0.0;
        static_for<0, NU>([&](auto ic) { gmax = __builtin_fmax(gmax, bc_get<decltype(ic)::value>(gl)); });
        gsum = ok ? gsum + gmax : gsum;
    }
    if(pf) pf->probe(6);
    return live ? rc : 1;
}

}  // namespace ilqg
