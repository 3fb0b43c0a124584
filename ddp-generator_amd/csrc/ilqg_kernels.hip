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
// wavefront; all small matrices of a trajectory live in that lane's VGPRs
// (ilqg_device.hpp).  Data layout: the line search reads packed per-step
// records (nomp), the roll-outs store tiled arrays (cur_x); DESIGN.md §2.
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

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include <rccl/rccl.h>

#include "mex.h"

// NaN/Inf guards in generated code still `return 0`; their printing is dropped on the device
#define PRNT(...) ((void)0)

// ---------------------------------------------------------------------------
// Hooks of the generated code on the device.
//
// Every non-constant assignment of a generated callback is followed by
//     if(isNANorINF(v)) { PRNT(...); return 0; }              (genenerator_main.mac:193-198)
// with isNANorINF(v) = (mxIsNaN(v) || mxIsInf(v))               (iLQG_problem.tem:11).
// Compiled literally that is one branch per assignment (52 in the CarParking file): the inlined
// callbacks fall apart into dozens of small basic blocks, each a VALU->SALU round trip, and the
// scheduler cannot overlap anything across them.  mex.h is ours, so on the device mxIsNaN()
// RECORDS a non-finite value in a per-lane sticky flag and evaluates to 0: the callbacks become
// straight-line code that always returns 1, and the caller tests the flag once per time step.
// The flag is a double that stays 0.0 until the first NaN/Inf is folded in (v*0.0 is NaN for
// both), i.e. exactly "some guarded value was NaN or Inf" — the condition under which the
// reference's callback returns 0.  What differs is only that the remaining assignments of a failed
// step are still evaluated (their results are discarded with the step).
//
// The flag has to be reachable from inside the generated functions without changing them: every
// guard sits in a function that has the parameter table `double **p` in scope
// (iLQG_func.tem:43-467), and on the device that table is a private array owned by the kernel,
// so the slots in front of it, p[-1], p[-2], p[-3], carry pointers to the lane's hook variables.
// After inlining and SROA they are plain registers.
// ---------------------------------------------------------------------------
struct ilqg_hooks {
    double nonfinite;  // p[-1]: 0.0, or NaN once a guarded value was NaN/Inf
    double huge;       // p[-2]: != 0 once sin/cos saw an argument the straight-line path cannot reduce
    double slow;       // p[-3]: != 0: sin/cos go to the device library (the rarely taken re-evaluation)
    double limgrad;    // p[-4]: != 0 (the default): limitsU() also stores the limits' signs and gradients
};
#include "ilqg_param_layout.h"  // generated at build time from the problem's paramdesc[]: ILQG_NP, sizes, offsets
#define ILQG_HOOK_SLOTS 4
#define ILQG_HOOK_NONFINITE (ILQG_NP)
#define ILQG_HOOK_HUGE (ILQG_NP + 1)
#define ILQG_HOOK_SLOW (ILQG_NP + 2)
#define ILQG_HOOK_LIMGRAD (ILQG_NP + 3)

__device__ __forceinline__ int ilqg_note_nonfinite(double **p, double v) {
    double *f = p[ILQG_HOOK_NONFINITE];
    *f = v * 0.0 + *f;
    return 0;
}
// Large generated files (-DILQG_SINCOS_CALL: thousands of guarded assignments in functions too big to inline, e.g.
// the tensors of an n = 16 problem) keep the plain guards: there the parameter table and the hooks stay in scratch
// memory, and a recorded guard would cost memory operations instead of a compare and a branch (measured 4x slower).
#ifndef ILQG_SINCOS_CALL
#undef mxIsNaN
#undef mxIsInf
#define mxIsNaN(v) ilqg_note_nonfinite(p, (v))
#define mxIsInf(v) 0
#define ILQG_UNIFORM_GUARDS 0
#else
// What those files get instead is a guard whose condition is WAVE-UNIFORM: "some active lane's value is NaN or
// Inf".  A literal per-lane guard is a divergent early return; 5 000 of them in one function leave the compiler
// with more saved exec masks than scalar registers, and it spills them through vector-register lanes that are
// spilled themselves: a scratch load, a wait and a scratch store around EVERY assignment (measured: the device
// copy of bp_derivsL of the n = 16 problem was 223 000 lines of ISA).  A uniform condition is a compare and a
// scalar branch.  When it fires, all active lanes leave the callback together; the caller then repeats the call
// lane by lane (run_alone below: one active lane, so the condition is exactly that lane's), which restores the
// per-lane result of the reference.  Lanes that have failed stay out of later calls.
#undef mxIsNaN
#undef mxIsInf
#define mxIsNaN(v) (__builtin_amdgcn_ballot_w64(!(__builtin_fabs(v) < __builtin_inf())) != 0ull)
#define mxIsInf(v) 0
#define ILQG_UNIFORM_GUARDS 1
#endif

// The generated callbacks call sin(x) and cos(x) of the same few arguments many times, spread
// over several functions (calcXUVariableAux, ddpf, bp_derivsL, ...).  The device math
// library's sin/cos contain branches (huge-argument reduction), so once the callbacks are
// inlined into a kernel every call is a separate ~130-instruction body the optimiser cannot
// merge; and a non-inlined helper would stall on the function-call ABI's `s_waitcnt vmcnt(0)`.
// ilqg_sincos below is STRAIGHT-LINE code, exact to < 1 ulp for |x| < 8e5, so value numbering
// merges all evaluations of the same argument: one argument reduction + one sine and one cosine
// polynomial per distinct argument and loop iteration.  Anything it cannot reduce (|x| >= 8e5;
// NaN and Inf give NaN here as in the library) raises the lane's `huge` hook; the kernel then
// evaluates that time step again with the `slow` hook set, which routes every sin/cos of the
// step to the device library.
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

__device__ __forceinline__ static ilqg_sc ilqg_sincos_fast(double x) {
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
    return out;
}

// stand-alone form (unit test, large generated files): library for what the fast path cannot reduce
__device__ __forceinline__ static ilqg_sc ilqg_sincos(double x) {
    ilqg_sc out = ilqg_sincos_fast(x);
    if(!(fabs(x) < 8.0e5)) out = ilqg_sincos_slow(x);
    return out;
}

// form used by the generated code of small problems: no branch, hooks instead (see above)
__device__ __forceinline__ static ilqg_sc ilqg_sincos_hooked(double **p, double x) {
    if(*p[ILQG_HOOK_SLOW] != 0.0) return ilqg_sincos_slow(x);  // compile-time constant per copy of the step
    double *h = p[ILQG_HOOK_HUGE];
    *h = (fabs(x) < 8.0e5) ? *h : 1.0;
    return ilqg_sincos_fast(x);
}
#if defined(ILQG_SINCOS_CALL)
// Large generated files (thousands of sin/cos call sites, e.g. the tensors of an n = 16 problem): keep
// every evaluation a CALL to a side-effect-free function.  Calls with equal arguments are merged before
// anything is inlined, which also keeps the compile time bounded.
__device__ __attribute__((noinline, const)) static ilqg_sc ilqg_sincos_call(double x) { return ilqg_sincos(x); }
#define sin(x) (ilqg_sincos_call(x).s)
#define cos(x) (ilqg_sincos_call(x).c)
// ... except in the parts of a roll-out step (ilqg_step_part, a few dozen call sites): there they are inline, the
// library behind a branch for what the straight-line path cannot reduce.  A call makes the caller wait for every memory
// operation in flight — the operands of the NEXT step, requested a step ahead — once per step (ILQG_PART_INLINE_SINCOS=0:
// calls there too).
#ifndef ILQG_PART_INLINE_SINCOS
#define ILQG_PART_INLINE_SINCOS 1
#endif
#if ILQG_PART_INLINE_SINCOS
#define ILQG_PART_SIN(v) (ilqg_sincos(v).s)
#define ILQG_PART_COS(v) (ilqg_sincos(v).c)
#endif
#else
#define sin(x) (ilqg_sincos_hooked(p, (x)).s)
#define cos(x) (ilqg_sincos_hooked(p, (x)).c)
#endif
#endif

// sin and cos of one argument at once for the derivative record in parts (ilqg_deriv_prepare)
#ifndef ILQG_NO_SHARED_SINCOS
// (straight-line form with the hooks, see ilqg_sincos_hooked: a call to the library on the spot would make every value
// alive at that point travel through scratch memory around it — 32 call sites in ilqg_deriv_prepare)
#define ILQG_DERIV_SINCOS(ARG_, SIN_, COS_)               \
    do {                                                  \
        const ilqg_sc r_ = ilqg_sincos_hooked(p, (ARG_)); \
        (SIN_) = r_.s;                                    \
        (COS_) = r_.c;                                    \
    } while(0)
#endif

// Both functions of the derivative record in parts are inlined (what they hand over stays in registers), and every part
// begins with a statement the optimiser must take as having an effect: a switch over 30 cheap side-effect-free cases is
// otherwise flattened into selects — every part evaluated in every trip, every part's literals alive at once (measured:
// 512 registers, 430 spilled).  (Called instead of inlined, a part's inputs and outputs travel through the caller's frame
// in scratch memory: 17 KB per step for a 4 KB record, 68 ms per iteration against 44 for k_derivs_wave.)
#define ILQG_DERIV_PREPARE_FN static __attribute__((always_inline))
#define ILQG_DERIV_PART_FN static __attribute__((always_inline))
#define ILQG_DERIV_CASE(q) asm volatile("" ::: "memory");

// The parts of a roll-out step belong INTO the kernel: as a called function they get x and u through scratch memory.
// (The kernel has two instantiations; with two callers the inliner leaves a function of this size alone.)
#define ILQG_PART_FN static __attribute__((always_inline))

extern "C" {
#pragma clang attribute push(__attribute__((device)), apply_to = function)
#pragma clang attribute push(__attribute__((internal_linkage)), apply_to = variable(is_global))
#include "iLQG.h"
#include "matMult.h"
// Limits that do not depend on the state (the header's hint): their gradients are zeros and their signs constants that
// nothing on the device reads (the backward steps' HX = false) — but limitsU() stores them into the element for every
// step, 2 N_X N_U + 2 N_U doubles: half as many bytes again as the time-varying entries of the n = 16 problem's record
// (measured: 26 of k_derivs_wave's 106 ms per iteration of config 5).  The generated file asks this condition; the hook
// is on unless a kernel clears it for records nobody but the backward pass will read (k_derivs_wave, `transient`).
#if defined(ILQG_STATE_DEPENDENT_LIMITS) && !ILQG_STATE_DEPENDENT_LIMITS
#define ILQG_LIMIT_GRADIENTS_WANTED (*p[ILQG_HOOK_LIMGRAD] != 0.0)
#endif
#include "iLQG_func.c"
#pragma clang attribute pop
#pragma clang attribute pop
}
#undef sin
#undef cos
// The function file's own macros end here: the reference's template leaves `mcond`, `sec`, `csc` and one `aux_<name>` /
// `daux_<name>` / `mu_<kind>_<i>` per auxiliary and multiplier defined to the end of the translation unit
// (iLQG_func.tem:5-30), which on the host is the end of the file and here would be the kernels.  The list is made from
// the file at build time (csrc/Makefile); ILQG_* names, the additive surface the kernels ask for, stay.
#include "ilqg_problem_undefs.h"

// Mapping: lane mapping (one lane per trajectory, everything in registers) for small problems,
// wave mapping (one wavefront per trajectory, matrices in LDS) when a lane's registers cannot
// hold the matrices.  -DILQG_WAVE_MAP=1 forces the wave mapping for a small problem.
#ifndef ILQG_WAVE_MAP
#define ILQG_WAVE_MAP (N_X > 8)
#endif

#include "ilqg_device.hpp"
#include "ilqg_wave.hpp"
#include "ilqg_row.hpp"
#include "ilqg_quad.hpp"
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
// the backward step of the wave mapping: row mapping (ilqg_row.hpp: products by row broadcast, no LDS operands) for
// everything that fits one 16-lane row per block of matrix rows, else one output element per lane (ilqg_wave.hpp)
#ifndef ILQG_ROW_STEP
#define ILQG_ROW_STEP (N_X <= 16 && N_U <= 16)
#endif
constexpr bool ROW_STEP = ILQG_ROW_STEP;
// Factored tensors (additive tables of the generated file, tools/gen_problem.py: every entry of fxx / fuu / fxu is a
// number times one product shared by its slice): the records then carry the products of the step instead of the
// tensors (k_derivs_wave writes 0.25 KB instead of 38 KB per step for the n = 16 problem) and the backward step
// multiplies them out on the fly from coefficient tables in LDS.
#if defined(ILQG_TENSOR_NBASIS) && FULL_DDP && ILQG_WAVE_MAP
#define ILQG_FACTORED (ILQG_TENSOR_NBASIS > 0 && ILQG_ROW_STEP)
#else
#define ILQG_FACTORED 0
#endif
constexpr bool FACTORED = ILQG_FACTORED;
#if ILQG_FACTORED
constexpr int NBASIS = ILQG_TENSOR_NBASIS;
#else
constexpr int NBASIS = 0;
#endif
static_assert(NBASIS <= 64 && (!FACTORED || NBASIS <= NX * SXX), "the products of a step travel in the record's fxx member, one per lane");
// wavefronts (= trajectories) per workgroup of the backward kernel that shares the coefficient tables
#ifndef ILQG_FACT_WAVES
#define ILQG_FACT_WAVES 8
#endif
// Augmented-Lagrangian multipliers (iLQG_problem.tem:70-89): the generated structs hold nothing but doubles
// (mu and the constraint value of the last update per constraint); empty for a problem without hle/hli/hfe/hfi.
constexpr int ME = std::is_empty<multipliersEl_t>::value ? 0 : (int)(sizeof(multipliersEl_t) / sizeof(double));
constexpr int MF = std::is_empty<multipliersFin_t>::value ? 0 : (int)(sizeof(multipliersFin_t) / sizeof(double));
constexpr bool HAS_MUL = ME + MF > 0;
constexpr int MEW = ME > 0 ? ME : 1, MFW = MF > 0 ? MF : 1;  // widths of the (possibly unused) device fields
static_assert(!ILQG_UNIFORM_GUARDS || WAVE_MAP, "wave-uniform guards (large generated files): the lane mapping's "
              "derivative and backward kernels have no lane-by-lane repetition");

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
    double *xpl, *upl;   // lane mapping, ls_keep = 2: the kept roll-outs of the first line-search stage, 2 x plane_n planes
    size_t xplane, uplane; //   of the layout and size of X resp. U (doubles per plane); see cur_x
    int plane_n;         //   planes per set = step sizes of the first stage (<= PLANE_A)
    double *cand1;       // wave mapping, ls_keep = 2: what the FIRST stage's lanes roll out, [step size][step 0..N][x u][trajectory]
    double *cand;        // second line-search stage: the trajectories its lanes roll out, [step size][step 0..N][x u]
                         //   [entry of pending] (entry fastest: a wavefront stores whole rows) — the accepted one is
                         //   copied, not rolled out again (k_adopt)
    trajEl_t *work;      // wave mapping: derivative records of one chunk of trajectories, [chunk][N], work_stride apart
    size_t work_stride;  //   sizeof(trajEl_t), or less for factored records (see FACT_STRIDE)
    int *queue;          // wave mapping: next trajectory of the chunk to be taken by a wavefront of the backward kernel
    double *nom;         // packed trajectory records, see nomp()
    double **p;
    int B, Bp, N;
};

// Nothing may be in flight when a prefetching loop is entered: the compiler's wait-count pass
// merges the state of the loop entry with that of the back edge, and a load still pending from
// before the loop makes it wait for EVERYTHING (vmcnt(0)) at the first use inside the loop, i.e.
// right behind the prefetch of the next step, which would then never overlap with the arithmetic.
__device__ __forceinline__ void drain_memory_ops() { __builtin_amdgcn_s_waitcnt(0); }

// "this value is needed HERE": keeps the optimiser from sinking its computation into a later block
__device__ __forceinline__ void pin(double &v) { asm volatile("" : "+v"(v)); }

// Builds with wave-uniform guards (ILQG_UNIFORM_GUARDS): f() for all active lanes together; if that fails (for all
// of them, see the guard), once more with one lane active at a time.  ONE call site for both, so that the repetition
// is the same machine code and a healthy lane gets the same bits either way.
template <class Fn>
__device__ __forceinline__ int run_guarded(Fn &&f) {
    int r = 1;
    for(int a = 0; a <= 64; a++) {
        if(a == 0 || (int)(threadIdx.x & 63) == a - 1) r = f();
        if(a == 0 && __builtin_amdgcn_ballot_w64(r == 0) == 0ull) break;
    }
    return r;
}

// Layout of the derivative-record fields (DER, FIN) of width W (doubles per step and trajectory):
//   lane mapping: [step][tile of 64 trajectories][component][trajectory in tile] — k_derivs, lane = (trajectory,
//                 step), writes and the backward kernel, lane = trajectory, reads one component as 512 contiguous
//                 bytes per wavefront; the components of a step sit at fixed distances of 512 bytes;
//   wave mapping: trajEl_t structs / [trajectory][component] (FIN).
constexpr int SI = WAVE_MAP ? 1 : WAVE;  // distance between components
__device__ __forceinline__ size_t step_stride(const DevPtrs &P, int W) { return WAVE_MAP ? (size_t)W : (size_t)W * P.Bp; }
__device__ __forceinline__ size_t traj_off(const DevPtrs &P, int W, int steps, int b) {
    return WAVE_MAP ? (size_t)b * steps * W : (size_t)(b >> 6) * (W * WAVE) + (b & 63);
}
// component i of a width-W, single-step field that is tiled in BOTH mappings (per-step-size costs)
__device__ __forceinline__ size_t tile_ix(int W, int i, int b) { return (size_t)(b >> 6) * (W * WAVE) + (size_t)i * WAVE + (b & 63); }
// The trajectory fields X, U, l, L live in ONE packed record per trajectory and time step,
//     nom[trajectory][step 0..N][ x (N_X) | u (N_U) | l (N_U) | L (N_U*N_X) ]        (step N holds x_N only)
// i.e. what a lane of the roll-out reads of the nominal trajectory per step is one contiguous piece (128 bytes =
// one cache line for CarParking), the step stride is a compile-time constant, and a lane's accesses do not depend
// on which other trajectories share its wavefront.  Measured (tools/ubench/layout_gather.hip): with the next step
// prefetched this costs the same as a [step][tile][component][lane] layout when the lanes are consecutive
// trajectories, and — unlike it — nothing extra when they are an arbitrary subset (the compacted second stage of
// the line search: up to 2x there, 6-13x for a full shuffle).  Four separate trajectory-major arrays, on the other
// hand, fetch four partly used cache lines per lane and step and were measured 2x slower.
constexpr int NOM_X = 0, NOM_U = NX, NOM_L = NX + NU, NOM_K = NX + 2 * NU;
constexpr int RN = (NX + 2 * NU + NXU + 1) / 2 * 2;  // doubles per record (even: 16-byte aligned pieces)
__device__ __forceinline__ double *nomp(const DevPtrs &P, int k, int b) {
    return P.nom + ((size_t)b * (P.N + 1) + k) * RN;
}
// element (step k, component i of W) of trajectory b of a derivative-record field (DER, FIN)
__device__ __forceinline__ size_t ix(const DevPtrs &P, int W, int steps, int k, int i, int b) {
    return traj_off(P, W, steps, b) + (size_t)k * step_stride(P, W) + (size_t)i * SI;
}

// In the lane mapping the trajectory (x, u) exists twice.  The packed records above are what the line search reads
// (lane = any trajectory).  Writing them from a roll-out is slow, though: a lane-per-trajectory store touches 64
// different cache lines, and a step of the winner pass is short (measured: the pass doubles from 1.1 to 2.1 ms).
// So the roll-outs STORE into tiled arrays X and U, [step][tile of 64][component][trajectory in tile], 512
// contiguous bytes per component and wavefront, which the passes over consecutive trajectories also read
// (derivatives, backward pass, cost sweep, host copies); and the backward pass, whose steps are long enough to
// absorb scattered stores, copies each (x_k, u_k) it reads into the record it completes with the gains of step k.
// The records are therefore current whenever a line search starts.  (Wave mapping: records only.)
//
// With ls_keep = 2 the tiled representation exists in 1 + 2 PLANE_A copies of identical layout: the arrays X / U
// ("home") and two sets of PLANE_A planes that the first stage of the line search rolls its candidates out into (set
// by set in turn).  ILQG_I_LOC says where the CURRENT trajectory of b is: accepting the roll-out of step size a of a
// search that wrote set s is `loc = 1 + s plane_n + a` (plane_n = step sizes of the first stage) — no second roll-out of the winner and no copy (both were
// measured: the winner pass is a chain of N dependent steps and 376 vector instructions per step and trajectory; a
// copy out of per-step-size planes reads four lines for every one it needs).  Everything that reads or writes "the
// current (x, u)" goes through cur_x / cur_u and follows: a lane's base address is chosen once, the strides are
// those of X / U.  A wavefront of 64 consecutive trajectories then reads pieces of up to PLANE_A + 1 rows per load
// instead of one whole row — the backward pass moves 48 of these bytes per step and is nowhere near the memory system.
constexpr int PLANE_A = 4;                // step sizes of a first stage that keeps its roll-outs (ls_split <= PLANE_A)
constexpr int XSI = WAVE_MAP ? 1 : WAVE;  // distance between components of x / u in that representation
__device__ __forceinline__ double *cur_x(const DevPtrs &P, int k, int b) {
    if(WAVE_MAP) return nomp(P, k, b) + NOM_X;
    const int loc = P.i[ILQG_I_LOC][b];
    return (loc ? P.xpl + (size_t)(loc - 1) * P.xplane : P.f[ILQG_F_X]) + ix(P, NX, P.N + 1, k, 0, b);
}
__device__ __forceinline__ double *cur_u(const DevPtrs &P, int k, int b) {
    if(WAVE_MAP) return nomp(P, k, b) + NOM_U;
    const int loc = P.i[ILQG_I_LOC][b];
    return (loc ? P.upl + (size_t)(loc - 1) * P.uplane : P.f[ILQG_F_U]) + ix(P, NU, P.N, k, 0, b);
}
// the arrays X / U themselves (host copies, the initial roll-out, trajectories adopted from the second stage)
__device__ __forceinline__ double *home_x(const DevPtrs &P, int k, int b) { return P.f[ILQG_F_X] + ix(P, NX, P.N + 1, k, 0, b); }
__device__ __forceinline__ double *home_u(const DevPtrs &P, int k, int b) { return P.f[ILQG_F_U] + ix(P, NU, P.N, k, 0, b); }
__device__ __forceinline__ size_t cur_xstride(const DevPtrs &P) { return WAVE_MAP ? (size_t)RN : (size_t)NX * P.Bp; }
__device__ __forceinline__ size_t cur_ustride(const DevPtrs &P) { return WAVE_MAP ? (size_t)RN : (size_t)NU * P.Bp; }

// Multipliers of (trajectory b, step k) resp. the final ones: [step][tile of 64][component][trajectory in tile]
// in both mappings (what the host copies produce).  The structs travel as arrays of doubles.
__device__ __forceinline__ size_t mul_ix(const DevPtrs &P, int W, int k, int b) {
    return (size_t)k * W * P.Bp + (size_t)(b >> 6) * (W * WAVE) + (b & 63);
}
__device__ __forceinline__ void load_mul(const DevPtrs &P, int k, int b, multipliersEl_t &m) {
    if(ME > 0) {
        double v[MEW];
        const double *s = P.f[ILQG_F_MUL] + mul_ix(P, ME, k, b);
#pragma unroll
        for(int i = 0; i < ME; i++) v[i] = s[i * WAVE];
        __builtin_memcpy(&m, v, sizeof(double) * ME);
    }
}
__device__ __forceinline__ void store_mul(const DevPtrs &P, int k, int b, const multipliersEl_t &m) {
    if(ME > 0) {
        double v[MEW];
        __builtin_memcpy(v, &m, sizeof(double) * ME);
        double *s = P.f[ILQG_F_MUL] + mul_ix(P, ME, k, b);
#pragma unroll
        for(int i = 0; i < ME; i++) s[i * WAVE] = v[i];
    }
}
__device__ __forceinline__ void load_mul_fin(const DevPtrs &P, int b, multipliersFin_t &m) {
    if(MF > 0) {
        double v[MFW];
        const double *s = P.f[ILQG_F_MULF] + mul_ix(P, MF, 0, b);
#pragma unroll
        for(int i = 0; i < MF; i++) v[i] = s[i * WAVE];
        __builtin_memcpy(&m, v, sizeof(double) * MF);
    }
}
__device__ __forceinline__ void store_mul_fin(const DevPtrs &P, int b, const multipliersFin_t &m) {
    if(MF > 0) {
        double v[MFW];
        __builtin_memcpy(v, &m, sizeof(double) * MF);
        double *s = P.f[ILQG_F_MULF] + mul_ix(P, MF, 0, b);
#pragma unroll
        for(int i = 0; i < MF; i++) s[i * WAVE] = v[i];
    }
}

// Per-lane snapshot of the problem parameters.  The generated callbacks read parameters as
// p[i][j] through a `double **`; read from global memory, every such value would have to be
// re-loaded after each store of the kernel (the compiler cannot prove that the parameter
// arrays do not alias the output arrays), which costs two dependent memory round trips per
// use.  Fixed-size parameters are therefore passed BY VALUE as a kernel argument (ParamValues)
// and copied into a private array: after SROA every p[i][j] is a wave-uniform value that was
// loaded once from the kernel-argument segment by a scalar load, i.e. it lives in an SGPR and
// costs no vector register.  Per-time-step parameters (size -1) stay in global memory.
struct ParamValues {
    double v[ILQG_PTOTAL];
};
struct ParamTable {
    double *ptr[ILQG_NP + ILQG_HOOK_SLOTS];  // the problem's parameters, then the hooks (see ilqg_hooks)
};
__device__ __forceinline__ void load_params(ParamValues &V, ParamTable &T, ilqg_hooks &H, const ParamValues &A,
                                            double **p) {
    constexpr int sizes[ILQG_NP > 0 ? ILQG_NP : 1] = ILQG_PSIZES;
    constexpr int offs[ILQG_NP > 0 ? ILQG_NP : 1] = ILQG_POFFSETS;
#pragma unroll
    for(int i = 0; i < ILQG_NP; i++) {
        if(sizes[i] > 0) {
#pragma unroll
            for(int j = 0; j < sizes[i]; j++) V.v[offs[i] + j] = A.v[offs[i] + j];
            T.ptr[i] = &V.v[offs[i]];
        } else {
            T.ptr[i] = p[i];
        }
    }
    H.nonfinite = 0.0;
    H.huge = 0.0;
    H.slow = 0.0;
    H.limgrad = 1.0;
    T.ptr[ILQG_HOOK_NONFINITE] = &H.nonfinite;
    T.ptr[ILQG_HOOK_HUGE] = &H.huge;
    T.ptr[ILQG_HOOK_SLOW] = &H.slow;
    T.ptr[ILQG_HOOK_LIMGRAD] = &H.limgrad;
}

// What a kernel needs to call the generated callbacks.  Four separate private objects, each pointing
// only at the next (o -> table -> values, hooks): the optimiser dissolves them one level at a time into
// registers; a single struct holding pointers into itself would stay in scratch memory.
struct Callbacks {
    tOptSet o;   // what the callbacks see as `o` (p, n_hor, penalty weights)
    tOptSet o1;  // the same with n_hor = 1: init_running loops over n_hor elements
};
__device__ __forceinline__ void make_callbacks(Callbacks &C, ParamTable &T, ParamValues &V, ilqg_hooks &H,
                                               const DevPtrs &P, const ilqg_dev_opts_t &O, const ParamValues &A) {
    load_params(V, T, H, A, P.p);
    tOptSet &o = C.o;
    o.p = T.ptr;
    o.n_hor = P.N;
    o.w_pen_l = O.w_pen_init_l;
    o.w_pen_f = O.w_pen_init_f;
    o.tolConstraint = O.tolConstraint;
    o.w_pen_fact1 = O.w_pen_fact1;
    o.w_pen_fact2 = O.w_pen_fact2;
    o.w_pen_max_l = O.w_pen_max_l;
    o.w_pen_max_f = O.w_pen_max_f;
    C.o1 = o;
    C.o1.n_hor = 1;
}
// problems with multipliers: the penalty weights are per trajectory (they grow with the constraint violation)
__device__ __forceinline__ void set_penalty_weights(Callbacks &C, double w_l, double w_f) {
    C.o.w_pen_l = C.o1.w_pen_l = w_l;
    C.o.w_pen_f = C.o1.w_pen_f = w_f;
}
__device__ __forceinline__ void load_penalty_weights(Callbacks &C, const DevPtrs &P, int b) {
    if(HAS_MUL) set_penalty_weights(C, P.f[ILQG_F_WPEN_L][b], P.f[ILQG_F_WPEN_F][b]);
}
// for the derivatives: the weights of the last accepted step.  A rejected step may raise the current ones
// (iLQG.c:345-349) while the reference sweeps again over the derivatives it has; kernels that re-evaluate
// derivatives instead of keeping them (fused backward pass, wave mapping) reproduce those with these weights.
__device__ __forceinline__ void load_penalty_weights_der(Callbacks &C, const DevPtrs &P, int b) {
    if(HAS_MUL) set_penalty_weights(C, P.f[ILQG_F_WPEN_L_DER][b], P.f[ILQG_F_WPEN_F_DER][b]);
}
// declares the callback context C and the lane's hooks H of a kernel with arguments (P, O, A)
#define ILQG_CALLBACKS(C, H) \
    ParamValues C##_values;  \
    ParamTable C##_table;    \
    ilqg_hooks H;            \
    Callbacks C;             \
    make_callbacks(C, C##_table, C##_values, H, P, O, A)

// ---------------------------------------------------------------------------
// host layout [b][k][f]  <->  device layout [k][b/64][f][b%64] of the lane mapping (the wave mapping's
// trajectory-major fields are copied without a kernel)
// ---------------------------------------------------------------------------
__global__ void k_to_dev(const double *__restrict__ host, double *__restrict__ dev, int B, int Bp, int steps, int wh,
                         int wd) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * steps * wd;
    if(i >= total) return;
    const int fcol = (int)(i % wd);
    const int k = (int)((i / wd) % steps);
    const int b = (int)(i / ((size_t)wd * steps));
    dev[(size_t)k * wd * Bp + (size_t)(b >> 6) * (wd * WAVE) + (size_t)fcol * WAVE + (b & 63)] =
        host[((size_t)b * steps + k) * wh + fcol];
}

// host [b][steps][w]  <->  columns [col0, col0 + w) of the packed trajectory records nom[b][0..N][RN]
__global__ void k_nom_io(double *__restrict__ nom, double *__restrict__ host, int B, int N, int steps, int w, int col0,
                         int to_device) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * steps * w;
    if(i >= total) return;
    const int c = (int)(i % w);
    const int k = (int)((i / w) % steps);
    const int b = (int)(i / ((size_t)w * steps));
    double *rec = nom + ((size_t)b * (N + 1) + k) * RN + col0 + c;
    if(to_device)
        *rec = host[i];
    else
        host[i] = *rec;
}

__global__ void k_from_dev(const double *__restrict__ dev, double *__restrict__ host, int B, int Bp, int steps, int wh,
                           int wd) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * steps * wh;
    if(i >= total) return;
    const int fcol = (int)(i % wh);
    const int k = (int)((i / wh) % steps);
    const int b = (int)(i / ((size_t)wh * steps));
    host[i] = (fcol < wd) ? dev[(size_t)k * wd * Bp + (size_t)(b >> 6) * (wd * WAVE) + (size_t)fcol * WAVE + (b & 63)] : 0.0;
}

#if !ILQG_WAVE_MAP
// ---------------------------------------------------------------------------
// The reference's calc_derivs body for one time step (iLQG_func.tem:207-211), from (x_k, u_k) in t.
// Auxiliaries are not kept in HBM: they are recomputed from (x,u) exactly as forward_pass did
// (iLQG_func.tem:160-164).  Straight-line code (see ilqg_hooks); evaluated a second time through the
// device library's sin/cos in the rare case that an argument was beyond the fast reduction.
// ---------------------------------------------------------------------------
// `overlapped` is called between the evaluation and the test for the rare second evaluation: a caller
// that has independent work in flight pins its results there, so that both share one basic block.
template <class F>
__device__ __forceinline__ int derivs_step(trajEl_t &t, multipliersEl_t *m, Callbacks &C, ilqg_hooks &H, int k, int N,
                                           F &&overlapped) {
    double x[NX], u[NU];
#pragma unroll
    for(int i = 0; i < NX; i++) x[i] = t.x[i];
#pragma unroll
    for(int i = 0; i < NU; i++) u[i] = t.u[i];
    const double nf0 = H.nonfinite;
    H.huge = 0.0;
    int ok = 1;
    auto body = [&]() {
        ok = calcXVariableAux(&t, m, k, &C.o);
        ok &= calcXUVariableAux(&t, m, k, &C.o);
        ok &= calcLAuxDeriv(&t, m, k, &C.o);
        ok &= bp_derivsL(&t, k, C.o.p);
        limitsU(&t, k, C.o.p, N);
    };
    body();
    overlapped();
    if(H.huge != 0.0) {
#pragma unroll
        for(int i = 0; i < NX; i++) t.x[i] = x[i];
#pragma unroll
        for(int i = 0; i < NU; i++) t.u[i] = u[i];
        H.nonfinite = nf0;
        H.slow = 1.0;
        body();
        H.slow = 0.0;
    }
    return ok;
}

__device__ __forceinline__ int derivs_final(trajFin_t &fin, multipliersFin_t *m, Callbacks &C, ilqg_hooks &H, int N) {
    double x[NX];
#pragma unroll
    for(int i = 0; i < NX; i++) x[i] = fin.x[i];
    const double nf0 = H.nonfinite;
    H.huge = 0.0;
    int ok = 1;
    auto body = [&]() {
        ok = calcFVariableAux(&fin, m, &C.o);
        ok &= calcFAuxDeriv(&fin, m, &C.o);
        ok &= bp_derivsF(&fin, N, C.o.p);
    };
    body();
    if(H.huge != 0.0) {
#pragma unroll
        for(int i = 0; i < NX; i++) fin.x[i] = x[i];
        H.nonfinite = nf0;
        H.slow = 1.0;
        body();
        H.slow = 0.0;
    }
    return ok;
}

// record of one step in the order of RecLayout
#define REC_COPY(OP, t)                                                                                 \
    OP(RL::CX, (t).cx, NX) OP(RL::CXX, (t).cxx, SXX) OP(RL::CU, (t).cu, NU) OP(RL::CUU, (t).cuu, SUU)  \
    OP(RL::CXU, (t).cxu, NXU) OP(RL::FX, (t).fx, NX * NX) OP(RL::FU, (t).fu, NXU)                      \
    OP(RL::LOWER, (t).lower, NU) OP(RL::UPPER, (t).upper, NU) REC_COPY_FULL(OP, t)                      \
    if(HX) {                                                                                            \
        OP(RL::LSIGN, (t).lower_sign, NU) OP(RL::USIGN, (t).upper_sign, NU)                             \
        OP(RL::LHX, (t).lower_hx, NXU) OP(RL::UHX, (t).upper_hx, NXU)                                   \
    }
#if FULL_DDP
#define REC_COPY_FULL(OP, t) OP(RL::FXX, (t).fxx, NX * SXX) OP(RL::FUU, (t).fuu, NX * SUU) OP(RL::FXU, (t).fxu, NX * NXU)
#else
#define REC_COPY_FULL(OP, t)
#endif

// ---------------------------------------------------------------------------
// calc_derivs: one lane per (trajectory, time step); step N is the final record
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_derivs(DevPtrs P, ilqg_dev_opts_t O, ParamValues A) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = (int)(tid % P.Bp);
    const int k = (int)(tid / P.Bp);
    if(k > P.N || b >= P.B) return;
    if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE || !P.i[ILQG_I_NEED_DERIVS][b]) return;

    ILQG_CALLBACKS(C, H);
    load_penalty_weights_der(C, P, b);
    int ok = 1;
    if(k < P.N) {
        trajEl_t t;
        multipliersEl_t m;
        load_mul(P, k, b, m);
        init_running(&t, &C.o1);
        const double *xs = cur_x(P, k, b), *us = cur_u(P, k, b);
#pragma unroll
        for(int i = 0; i < NX; i++) t.x[i] = xs[i * XSI];
#pragma unroll
        for(int i = 0; i < NU; i++) t.u[i] = us[i * XSI];

        ok = derivs_step(t, HAS_MUL ? &m : nullptr, C, H, k, P.N, [] {});
        double *out = P.f[ILQG_F_DER] + ix(P, REC, P.N, k, 0, b);
#define PUT(off, arr, cnt) _Pragma("unroll") for(int i = 0; i < (cnt); i++) out[((off) + i) * SI] = (arr)[i];
        REC_COPY(PUT, t)
    } else {
        trajFin_t fin;
        multipliersFin_t mf;
        load_mul_fin(P, b, mf);
        init_final(&fin, &C.o);
        const double *xs = cur_x(P, P.N, b);
#pragma unroll
        for(int i = 0; i < NX; i++) fin.x[i] = xs[i * XSI];

        ok = derivs_final(fin, HAS_MUL ? &mf : nullptr, C, H, P.N);
        double *out = P.f[ILQG_F_FIN] + ix(P, FIN, 1, 0, 0, b);
        PUT(0, fin.cx, NX)
        PUT(NX, fin.cxx, SXX)
#undef PUT
    }
    if(!ok || H.nonfinite != 0.0) P.derivs_failed[b] = 1;
}

// ---------------------------------------------------------------------------
// back_pass: one lane per trajectory, sequential in time, next record prefetched
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load_record(double *dst, double *udst, const double *src, const double *us) {
#pragma unroll
    for(int i = 0; i < REC; i++) dst[i] = src[i * SI];
#pragma unroll
    for(int i = 0; i < NU; i++) udst[i] = us[i * XSI];
}

// The gains of a step are stored right behind its arithmetic, i.e. BEHIND the prefetch of the next step's inputs
// in issue order: the memory counter retires in issue order, so the wait for the prefetched values (vmcnt = number
// of younger operations) leaves the stores in flight.  They are stored for every lane, also for one whose box QP
// has just failed (back_pass.c:168-171 returns before writing L): that lane leaves the sweep right after, and a
// retry rewrites all gains.
template <int CS = 1>
__device__ __forceinline__ void store_gains(const double *l, const double *K, double *lo, double *ko) {
#pragma unroll
    for(int i = 0; i < NU; i++) lo[i * CS] = l[i];
#pragma unroll
    for(int i = 0; i < NXU; i++) ko[i * CS] = K[i];
}

// one sweep k = N-1..0; returns 0 ok, 1 box-QP failed (back_pass.c:168-171)
__device__ __forceinline__ int backward_sweep(const DevPtrs &P, int b, double lambda, int regType, double &dV0,
                                              double &dV1, double &g_norm) {
    const int N = P.N;
    double Vx[NX], Vxx[SXX], l[NU], K[NXU];
    const double *fin = P.f[ILQG_F_FIN] + ix(P, FIN, 1, 0, 0, b);
#pragma unroll
    for(int i = 0; i < NX; i++) Vx[i] = fin[i * SI];
#pragma unroll
    for(int i = 0; i < SXX; i++) Vxx[i] = fin[(NX + i) * SI];
#pragma unroll
    for(int i = 0; i < NU; i++) l[i] = 0.0;  // warm start of the last step (back_pass.c:163-164)
    dV0 = 0.0;
    dV1 = 0.0;
    double gsum = 0.0;

    // pointers to step k of this lane's trajectory, walked backwards
    const double *rp = P.f[ILQG_F_DER] + ix(P, REC, N, N - 1, 0, b);
    // This sweep streams the stored records; its gains go to the tiled arrays l / L (512 contiguous bytes per
    // component and wavefront) and k_pack_records turns them into the line search's records afterwards.
    const double *up = cur_u(P, N - 1, b);
    double *lo = WAVE_MAP ? nomp(P, N - 1, b) + NOM_L : P.f[ILQG_F_LG] + ix(P, NU, N, N - 1, 0, b);
    double *ko = WAVE_MAP ? nomp(P, N - 1, b) + NOM_K : P.f[ILQG_F_KG] + ix(P, NXU, N, N - 1, 0, b);
    const size_t rs = step_stride(P, REC), us = cur_ustride(P);
    const size_t ls = WAVE_MAP ? (size_t)RN : step_stride(P, NU), ks = WAVE_MAP ? (size_t)RN : step_stride(P, NXU);

    double cur[REC], ucur[NU];
    load_record(cur, ucur, rp, up);
    int failed = 0;
    drain_memory_ops();
    for(int k = N - 1; k >= 0; k--) {
        double nxt[REC], unxt[NU];
        if(k > 0) load_record(nxt, unxt, rp - rs, up - us);  // in flight while this step computes
        // l still holds the solution of step k+1: the warm start (back_pass.c:165-166)
        const int rc = back_step<NX, NU, FULL, HX>(cur, ucur, Vx, Vxx, l, K, lambda, regType, dV0, dV1, gsum);
        store_gains<XSI>(l, K, lo, ko);
        if(rc < 1) {
            failed = 1;
            break;
        }
        rp -= rs;
        up -= us;
        lo -= ls;
        ko -= ks;
#pragma unroll
        for(int i = 0; i < REC; i++) cur[i] = nxt[i];
#pragma unroll
        for(int i = 0; i < NU; i++) ucur[i] = unxt[i];
    }
    if(!failed) g_norm = gsum / ((double)(N - 1));  // N summands over N-1 (back_pass.c:254)
    return failed;
}

// The same sweep with the derivative record of each step evaluated on the fly from the stored
// (x_k, u_k) by the generated callbacks instead of being read from HBM: per step 6 doubles are
// read and 10 written, instead of 57 + 10 (and k_derivs' 61 are not moved at all).  The values
// are the ones k_derivs would have stored (same callbacks, same inputs).
// Returns 0 ok, 1 box-QP failed, 2 NaN/Inf in the derivatives (iLQG.c:247-249).
__device__ __forceinline__ int backward_sweep_fused(const DevPtrs &P, Callbacks &C, ilqg_hooks &H, int b, double lambda, int regType,
                                                    double &dV0, double &dV1, double &g_norm) {
    const int N = P.N;
    H.nonfinite = 0.0;
    const double *xp = cur_x(P, N, b);
    const double *up = cur_u(P, N - 1, b);
    double *rec_k = nomp(P, N - 1, b);  // record of step k, completed here: (x_k, u_k) as read + the gains
    const size_t xs = cur_xstride(P), us = cur_ustride(P);

    double Vx[NX], Vxx[SXX], l[NU], K[NXU];
    {
        trajFin_t fin;
        init_final(&fin, &C.o);
#pragma unroll
        for(int i = 0; i < NX; i++) fin.x[i] = xp[i * XSI];
        if(!WAVE_MAP) {
            double *recN = nomp(P, N, b);
#pragma unroll
            for(int i = 0; i < NX; i++) recN[NOM_X + i] = fin.x[i];
        }
        multipliersFin_t mf;
        load_mul_fin(P, b, mf);
        const int ok = derivs_final(fin, HAS_MUL ? &mf : nullptr, C, H, N);
        if(!ok || H.nonfinite != 0.0) return 2;
#pragma unroll
        for(int i = 0; i < NX; i++) Vx[i] = fin.cx[i];
#pragma unroll
        for(int i = 0; i < SXX; i++) Vxx[i] = fin.cxx[i];
    }
    xp -= xs;  // step N-1
#pragma unroll
    for(int i = 0; i < NU; i++) l[i] = 0.0;
    dV0 = 0.0;
    dV1 = 0.0;
    double gsum = 0.0;

    // Software pipeline: the record of step k-1 is evaluated in the same loop iteration as the Riccati
    // update of step k, behind the box QP.  The two are independent chains of dependent fp64
    // operations, so with ONE wavefront per SIMD (all that 65 536 trajectories give) the scheduler can
    // fill the latency of one with the other.
    trajEl_t t;
    init_running(&t, &C.o1);  // constant entries of the record (iLQG_func.tem:312-347)
    double xk[NX], uk[NU];    // state and control of the step whose record is in `cur`
    double cur[REC];
#define GETF(off, arr, cnt) _Pragma("unroll") for(int i = 0; i < (cnt); i++) rec[(off) + i] = (arr)[i];
    auto record_of = [&](const double *xv, const double *uv, int k, double *rec, auto &&overlapped) {
#pragma unroll
        for(int i = 0; i < NX; i++) t.x[i] = xv[i];
#pragma unroll
        for(int i = 0; i < NU; i++) t.u[i] = uv[i];
        multipliersEl_t m;
        load_mul(P, k, b, m);
        const int ok = derivs_step(t, HAS_MUL ? &m : nullptr, C, H, k, N, overlapped);
        REC_COPY(GETF, t)
        return ok;
    };
    {
#pragma unroll
        for(int i = 0; i < NX; i++) xk[i] = xp[i * XSI];
#pragma unroll
        for(int i = 0; i < NU; i++) uk[i] = up[i * XSI];
        const int ok = record_of(xk, uk, N - 1, cur, [] {});
        if(!ok || H.nonfinite != 0.0) return 2;
    }
    // (x, u) of step k-1, loaded one iteration ahead of their use (N >= 2)
    double xn[NX], un[NU];
#pragma unroll
    for(int i = 0; i < NX; i++) xn[i] = (xp - xs)[i * XSI];
#pragma unroll
    for(int i = 0; i < NU; i++) un[i] = (up - us)[i * XSI];
    int result = 0;
#ifdef ILQG_PROFILE_SECTIONS
    Prof prof;
    prof.start();
    Prof *pf = &prof;
#else
    Prof *pf = nullptr;
#endif
    drain_memory_ops();
    for(int k = N - 1; k >= 0; k--) {
        if(pf) pf->probe(7);
        // step k-2 (clamped at step 0, so that the pipeline below needs no special case at its end: the
        // last iteration evaluates the record of step 0 once more and discards it)
        double xnn[NX], unn[NU];
        const int back = (k > 1) ? 2 : k;
#pragma unroll
        for(int i = 0; i < NX; i++) xnn[i] = (xp - back * xs)[i * XSI];
#pragma unroll
        for(int i = 0; i < NU; i++) unn[i] = (up - back * us)[i * XSI];
        if(pf) pf->probe(0);
        const int rc = back_step<NX, NU, FULL, HX>(cur, uk, Vx, Vxx, l, K, lambda, regType, dV0, dV1, gsum, pf);
        double nxt[REC];
        // The value-function update above is only needed by the next iteration, so the optimiser would
        // sink it behind the exit tests below, into a block of its own, where it cannot overlap with
        // the derivative evaluation.  Pinning its results right behind that evaluation keeps both in
        // one block.
        const int ok = record_of(xn, un, (k > 0) ? k - 1 : 0, nxt, [&] {
#pragma unroll
            for(int i = 0; i < NX; i++) pin(Vx[i]);
#pragma unroll
            for(int i = 0; i < SXX; i++) pin(Vxx[i]);
            pin(dV0);
            pin(dV1);
            pin(gsum);
        });
        if(!ok || H.nonfinite != 0.0) {
            result = 2;
            break;
        }
        if(pf) pf->probe(6);
        // the record of step k for the line search: gains, and (lane mapping) the (x_k, u_k) they belong to
        store_gains(l, K, rec_k + NOM_L, rec_k + NOM_K);
        if(!WAVE_MAP) {
#pragma unroll
            for(int i = 0; i < NX; i++) rec_k[NOM_X + i] = xk[i];
#pragma unroll
            for(int i = 0; i < NU; i++) rec_k[NOM_U + i] = uk[i];
        }
        if(rc < 1) {
            result = 1;
            break;
        }
#pragma unroll
        for(int i = 0; i < REC; i++) cur[i] = nxt[i];
        xp -= xs;
        up -= us;
        rec_k -= RN;
#pragma unroll
        for(int i = 0; i < NX; i++) xk[i] = xn[i];
#pragma unroll
        for(int i = 0; i < NU; i++) uk[i] = un[i];
#pragma unroll
        for(int i = 0; i < NX; i++) xn[i] = xnn[i];
#pragma unroll
        for(int i = 0; i < NU; i++) un[i] = unn[i];
    }
#undef GETF
    if(!result) g_norm = gsum / ((double)(N - 1));
#ifdef ILQG_PROFILE_SECTIONS
    if((threadIdx.x & 63) == 0)
        for(int i = 0; i < 8; i++) atomicAdd(&ilqg_prof_cycles[i], (unsigned long long)prof.acc[i]);
#endif
    return result;
}

// mode: 0 = records from HBM, lambda retry loop and gradient test (iLQG.c:261-303)
//       1 = records from HBM, ONE sweep (the drop-in back_pass(): the caller owns the retry loop)
//       2 = as 0 with the derivatives evaluated on the fly (k_derivs is not needed)
template <int mode>
__global__ __launch_bounds__(WAVE, 1) void k_backward(DevPtrs P, ilqg_dev_opts_t O, ParamValues A) {
    const int b = blockIdx.x * WAVE + threadIdx.x;
    if(b >= P.B) return;
    if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE) return;
    const int single_sweep = (mode == 1);
    P.i[ILQG_I_NEED_DERIVS][b] = 0;
    if(P.derivs_failed[b]) {  // iLQG.c:247-249
        P.i[ILQG_I_STATUS][b] = ILQG_ST_DERIVS_FAILED;
        return;
    }
    ILQG_CALLBACKS(C, H);
    load_penalty_weights_der(C, P, b);
    double lambda = P.f[ILQG_F_LAMBDA][b], dlambda = P.f[ILQG_F_DLAMBDA][b];
    double dV0 = 0.0, dV1 = 0.0, g_norm = P.f[ILQG_F_GNORM][b];
    int calls = 0, rc;
    for(;;) {
        if(mode == 2)
            rc = backward_sweep_fused(P, C, H, b, lambda, O.regType, dV0, dV1, g_norm);
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

// ---------------------------------------------------------------------------
// Mode 2 on TWO wavefronts per tile of 64 trajectories ("split"; problems whose header lists the entries bp_derivsL
// writes, no multipliers, constant limits).  65 536 trajectories are one wavefront per SIMD, and that wavefront's step is
// a chain of ~1 265 dependent-ish instructions: the derivative evaluation of step k-1 and the Riccati update of step k
// are independent, and the software pipeline of backward_sweep_fused only lets the scheduler interleave them.  Here the
// PRODUCER wavefront evaluates the record of step k-1 (same generated callbacks, same arithmetic) while the CONSUMER
// wavefront of the same workgroup — on another SIMD of the CU — does step k; the time-varying entries of a record
// (ILQG_TIME_VARYING: 14 of 55 doubles for CarParking), x_k, u_k and a failure flag cross over in LDS, two slots, one
// s_barrier per step.  The consumer fills the constant entries from init_running() as before, so what the optimiser
// folds stays folded.  Control flow is the consumer's: it publishes "some lane still sweeps" / "another sweep" and both
// wavefronts leave their loops together.  Results, memory effects and failure behaviour are those of mode 2.
// MEASURED: no gain — 2.73 ms alone against 2.65 ms for k_backward<2>, headline 130-133 against 135.6 it/s: what bounds
// a step is the chain of dependent operations of the Riccati update with its box QP (~5.3 us), behind which the
// derivative evaluation already hides in ONE instruction stream.  Off by default (option bw_split), kept with its test.
// ---------------------------------------------------------------------------
#if defined(ILQG_TIME_VARYING) && !ILQG_STATE_DEPENDENT_LIMITS
#define ILQG_HAVE_SPLIT 1
#define ILQG_COUNT_ENTRY(member, index) +1
constexpr int SPLIT_VARY = 0 ILQG_TIME_VARYING(ILQG_COUNT_ENTRY) ILQG_TIME_VARYING_FULL(ILQG_COUNT_ENTRY);
constexpr int SPLIT_SLOT = SPLIT_VARY + NX + NU + 1;  // + x_k, u_k, failure flag

__global__ __launch_bounds__(2 * WAVE) void k_backward_split(DevPtrs P, ilqg_dev_opts_t O, ParamValues A) {
    __shared__ double ring[2][SPLIT_SLOT][WAVE];
    // the consumer's word: [k & 1] "some lane goes on after step k" (two cells: the producer reads the one of step k+1
    // while the consumer may already write the one of step k), [2] "some lane needs another sweep"
    __shared__ int ctl[3];
    const int role = threadIdx.x >> 6, lane = threadIdx.x & 63;  // 0 consumer, 1 producer
    const int b = blockIdx.x * WAVE + lane;
    const int N = P.N;
    const bool mine = b < P.B && P.i[ILQG_I_STATUS][b] == ILQG_ST_ACTIVE;
    if(__builtin_amdgcn_ballot_w64(mine) == 0ull) return;  // the same for both wavefronts: they hold the same trajectories
    ILQG_CALLBACKS(C, H);
    const size_t xs = cur_xstride(P), us = cur_ustride(P);

    if(role == 1) {
        // ---------------- producer: records of steps N-1, N-2, ..., 0 of every sweep ----------------
        trajEl_t t;
        init_running(&t, &C.o1);
        for(;;) {  // sweeps
            H.nonfinite = 0.0;
            const double *xp = cur_x(P, N - 1, b), *up = cur_u(P, N - 1, b);
            double xn[NX], un[NU];
#pragma unroll
            for(int i = 0; i < NX; i++) xn[i] = xp[i * XSI];
#pragma unroll
            for(int i = 0; i < NU; i++) un[i] = up[i * XSI];
            for(int k = N - 1; k >= 0; k--) {
                // (x, u) of the step after this one, in flight while this record is evaluated
                double xnn[NX], unn[NU];
                const int back = (k > 0) ? 1 : 0;
#pragma unroll
                for(int i = 0; i < NX; i++) xnn[i] = (xp - back * xs)[i * XSI];
#pragma unroll
                for(int i = 0; i < NU; i++) unn[i] = (up - back * us)[i * XSI];
#pragma unroll
                for(int i = 0; i < NX; i++) t.x[i] = xn[i];
#pragma unroll
                for(int i = 0; i < NU; i++) t.u[i] = un[i];
                const int ok = derivs_step(t, nullptr, C, H, k, N, [] {});
                double(*slot)[WAVE] = ring[k & 1];
                int j = 0;
#define ILQG_PUT_ENTRY(member, index) slot[j++][lane] = t.member[index];
                ILQG_TIME_VARYING(ILQG_PUT_ENTRY) ILQG_TIME_VARYING_FULL(ILQG_PUT_ENTRY)
#undef ILQG_PUT_ENTRY
#pragma unroll
                for(int i = 0; i < NX; i++) slot[SPLIT_VARY + i][lane] = xn[i];
#pragma unroll
                for(int i = 0; i < NU; i++) slot[SPLIT_VARY + NX + i][lane] = un[i];
                slot[SPLIT_VARY + NX + NU][lane] = (!ok || H.nonfinite != 0.0) ? 1.0 : 0.0;
                __syncthreads();  // record k is there; the consumer is done with record k+1 (whose slot is written next)
                // The consumer's word on step k+1.  If every lane left there, the consumer has matched the barrier
                // above with one of its own and is on its way to the end of the sweep.
                if(k < N - 1 && !ctl[(k + 1) & 1]) break;
                xp -= back * xs;
                up -= back * us;
#pragma unroll
                for(int i = 0; i < NX; i++) xn[i] = xnn[i];
#pragma unroll
                for(int i = 0; i < NU; i++) un[i] = unn[i];
            }
            // the consumer's last step (0, or the one every lane left at) and its decision about another sweep
            __syncthreads();
            if(!ctl[2]) break;
        }
        return;
    }

    // ---------------- consumer: back_pass + retry loop of k_backward<2>, records from the ring ----------------
    load_penalty_weights_der(C, P, b);
    if(mine) P.i[ILQG_I_NEED_DERIVS][b] = 0;
    // iLQG.c:247-249.  (The status is written at the end: the producer reads it when it starts.)
    const bool dead = mine && P.derivs_failed[b];
    bool sweeping = mine && !dead;  // this lane wants (another) sweep
    double lambda = mine ? P.f[ILQG_F_LAMBDA][b] : 1.0, dlambda = mine ? P.f[ILQG_F_DLAMBDA][b] : 1.0;
    double dV0 = 0.0, dV1 = 0.0, g_norm = mine ? P.f[ILQG_F_GNORM][b] : 0.0;
    int calls = 0, rc = 0;
    const bool took_part = sweeping;
    trajEl_t t;
    init_running(&t, &C.o1);
    for(;;) {  // sweeps of the tile: the lanes that want one take part, the others idle through it
        bool in_sweep = sweeping;
        int result = 0;
        double Vx[NX], Vxx[SXX], l[NU], K[NXU], gsum = 0.0;
        H.nonfinite = 0.0;
        if(in_sweep) {
            const double *xp = cur_x(P, N, b);
            trajFin_t fin;
            init_final(&fin, &C.o);
#pragma unroll
            for(int i = 0; i < NX; i++) fin.x[i] = xp[i * XSI];
            double *recN = nomp(P, N, b);
#pragma unroll
            for(int i = 0; i < NX; i++) recN[NOM_X + i] = fin.x[i];
            const int ok = derivs_final(fin, nullptr, C, H, N);
            if(!ok || H.nonfinite != 0.0) {
                result = 2;
                in_sweep = false;
            }
#pragma unroll
            for(int i = 0; i < NX; i++) Vx[i] = fin.cx[i];
#pragma unroll
            for(int i = 0; i < SXX; i++) Vxx[i] = fin.cxx[i];
        }
#pragma unroll
        for(int i = 0; i < NU; i++) l[i] = 0.0;
        if(sweeping) {  // (a lane that is done keeps the results of ITS last sweep)
            dV0 = 0.0;
            dV1 = 0.0;
        }
        double *rec_k = nomp(P, N - 1, b);
        // gains of the step before (k+1), stored once the record of step k is known to be good — the order of
        // backward_sweep_fused: a failed record of step k ends the sweep before the gains of step k+1 are stored
        double lp[NU], Kp[NXU], xkp[NX], ukp[NU];
        int rcp = 1;
        bool pending = false;
        auto commit = [&]() {  // what backward_sweep_fused does behind record_of(k-1) for step k
            store_gains(lp, Kp, rec_k + NOM_L, rec_k + NOM_K);
#pragma unroll
            for(int i = 0; i < NX; i++) rec_k[NOM_X + i] = xkp[i];
#pragma unroll
            for(int i = 0; i < NU; i++) rec_k[NOM_U + i] = ukp[i];
            rec_k -= RN;
            pending = false;
            if(rcp < 1) {
                result = 1;
                in_sweep = false;
            }
        };
        for(int k = N - 1; k >= 0; k--) {
            __syncthreads();  // record k is in its slot
            if(in_sweep) {
                double(*slot)[WAVE] = ring[k & 1];
                if(slot[SPLIT_VARY + NX + NU][lane] != 0.0) {  // the record of step k failed
                    result = 2;
                    in_sweep = false;
                } else {
                    if(pending) commit();  // step k+1
                    if(in_sweep) {
                        int j = 0;
#define ILQG_GET_ENTRY(member, index) t.member[index] = slot[j++][lane];
                        ILQG_TIME_VARYING(ILQG_GET_ENTRY) ILQG_TIME_VARYING_FULL(ILQG_GET_ENTRY)
#undef ILQG_GET_ENTRY
                        double xk[NX], uk[NU];
#pragma unroll
                        for(int i = 0; i < NX; i++) xk[i] = slot[SPLIT_VARY + i][lane];
#pragma unroll
                        for(int i = 0; i < NU; i++) uk[i] = slot[SPLIT_VARY + NX + i][lane];
#pragma unroll
                        for(int i = 0; i < NX; i++) t.x[i] = xk[i];
#pragma unroll
                        for(int i = 0; i < NU; i++) t.u[i] = uk[i];
                        limitsU(&t, k, C.o.p, N);  // constant limits: parameters only
                        double cur[REC];
#define GETF(off, arr, cnt) _Pragma("unroll") for(int i = 0; i < (cnt); i++) cur[(off) + i] = (arr)[i];
                        REC_COPY(GETF, t)
#undef GETF
                        rcp = back_step<NX, NU, FULL, HX>(cur, uk, Vx, Vxx, l, K, lambda, O.regType, dV0, dV1, gsum, nullptr);
#pragma unroll
                        for(int i = 0; i < NU; i++) lp[i] = l[i];
#pragma unroll
                        for(int i = 0; i < NXU; i++) Kp[i] = K[i];
#pragma unroll
                        for(int i = 0; i < NX; i++) xkp[i] = xk[i];
#pragma unroll
                        for(int i = 0; i < NU; i++) ukp[i] = uk[i];
                        pending = true;
                    }
                }
            }
            const bool more = __builtin_amdgcn_ballot_w64(in_sweep) != 0ull;
            if(lane == 0) ctl[k & 1] = more ? 1 : 0;
            if(!more) {
                if(k > 0) __syncthreads();  // the producer is on its way to this barrier with record k-1: meet it there
                break;
            }
        }
        if(in_sweep && pending) commit();  // step 0 (its record was checked when it arrived)
        if(sweeping) {
            rc = result;
            if(!result) g_norm = gsum / ((double)(N - 1));
            calls++;
            // raise the regularisation and retry (iLQG.c:271-274)
            sweeping = false;
            if(rc == 1) {
                const double t1 = dlambda * O.lambdaFactor;
                dlambda = (t1 > O.lambdaFactor) ? t1 : O.lambdaFactor;
                const double t2 = lambda * dlambda;
                lambda = (t2 > O.lambdaMin) ? t2 : O.lambdaMin;
                sweeping = !(lambda > O.lambdaMax);
            }
        }
        const bool again = __builtin_amdgcn_ballot_w64(sweeping) != 0ull;
        if(lane == 0) ctl[2] = again ? 1 : 0;
        __syncthreads();
        if(!again) break;
    }
    if(dead) P.i[ILQG_I_STATUS][b] = ILQG_ST_DERIVS_FAILED;
    if(took_part) {
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
        P.f[ILQG_F_LAMBDA][b] = lambda;
        P.f[ILQG_F_DLAMBDA][b] = dlambda;
        P.f[ILQG_F_DV0][b] = dV0;
        P.f[ILQG_F_DV1][b] = dV1;
        P.f[ILQG_F_GNORM][b] = g_norm;
        P.i[ILQG_I_BP_CALLS][b] = calls;
        P.i[ILQG_I_BP_RC][b] = rc;
    }
}
#else
#define ILQG_HAVE_SPLIT 0
#endif

// After a backward pass over stored records (modes 0 and 1): the line search's packed records from the tiled
// X, U, l, L.  One wavefront per (tile of 64 trajectories, time step): the tile's RN values per trajectory are read
// as coalesced rows (lane = trajectory), turned through LDS, and written as whole records (16 consecutive lanes =
// one 128-byte record for CarParking) — a lane-per-trajectory write of the records would touch 64 cache lines per
// instruction (measured 8.8 ms for 65 536 x 501 records; this form is bandwidth bound).  Step N holds x_N only.
constexpr int PACK_WAVES = 4, PACK_LD = RN + 1;  // row stride padded by one double: no LDS bank conflicts
__global__ __launch_bounds__(WAVE *PACK_WAVES) void k_pack_records(DevPtrs P) {
    __shared__ double rows[PACK_WAVES][WAVE * PACK_LD];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x, k = blockIdx.y * PACK_WAVES + w;
    double *row = rows[w];
    const int b = tile * WAVE + lane;
    if(k <= P.N) {
        double v[RN];
#pragma unroll
        for(int i = 0; i < RN; i++) v[i] = 0.0;
        const double *xs = cur_x(P, k, b);
#pragma unroll
        for(int i = 0; i < NX; i++) v[NOM_X + i] = xs[i * XSI];
        if(k < P.N) {
            const double *us = cur_u(P, k, b);
            const double *ls = P.f[ILQG_F_LG] + ix(P, NU, P.N, k, 0, b), *ks = P.f[ILQG_F_KG] + ix(P, NXU, P.N, k, 0, b);
#pragma unroll
            for(int i = 0; i < NU; i++) v[NOM_U + i] = us[i * XSI];
#pragma unroll
            for(int i = 0; i < NU; i++) v[NOM_L + i] = ls[i * SI];
#pragma unroll
            for(int i = 0; i < NXU; i++) v[NOM_K + i] = ks[i * SI];
        }
#pragma unroll
        for(int i = 0; i < RN; i++) row[lane * PACK_LD + i] = v[i];
    }
    __syncthreads();
    if(k <= P.N) {
#pragma unroll
        for(int r = 0; r < RN; r++) {
            const int e = r * WAVE + lane;       // element e of the tile's WAVE x RN values, trajectory-major
            const int tt = e / RN, c = e % RN;
            if(tile * WAVE + tt < P.B) nomp(P, k, tile * WAVE + tt)[c] = row[tt * PACK_LD + c];
        }
    }
}

#else  // ILQG_WAVE_MAP
// ---------------------------------------------------------------------------
// wave mapping: calc_derivs straight into the device trajEl_t records, one lane per
// (trajectory of the chunk, time step); step N is the final record
// ---------------------------------------------------------------------------
// Factored records use the head of trajEl_t (x .. fu and the NBASIS products at the start of fxx) and its tail (the
// members behind fxu: the auxiliaries) — the 38 KB of tensors in between are never touched.  Such records are laid
// down OVERLAPPING, FACT_STRIDE < sizeof(trajEl_t) apart, the head of one record inside the unused middle of an
// earlier one: the smallest distance at which no head (offsets [0, A) modulo the distance) meets a tail (offsets
// [TAIL mod distance, + T)).  4x as many trajectories per chunk of the work buffer for the n = 16 problem
// (11 992 instead of 47 944 bytes per step).  Only when init_running() stores nothing in the tensors (generator
// hint ILQG_TENSOR_INIT_WRITES).
#if ILQG_FACTORED && defined(ILQG_TENSOR_INIT_WRITES) && !ILQG_TENSOR_INIT_WRITES
constexpr size_t fact_stride() {
    const size_t A = offsetof(trajEl_t, fxx) + NBASIS * sizeof(double);
    const size_t TAIL = offsetof(trajEl_t, fxu) + sizeof(double) * NX * NXU, T = sizeof(trajEl_t) - TAIL;
#ifndef ILQG_FACT_ALIGN
#define ILQG_FACT_ALIGN 128
#endif
    constexpr size_t AL = ILQG_FACT_ALIGN;  // whole cache lines (measured against 8: k_derivs_wave 116 -> 113 ms per iteration)
    if(T == 0) return (A + AL - 1) / AL * AL;
    for(size_t S = (A + T + AL - 1) / AL * AL; S < sizeof(trajEl_t); S += AL) {
        const size_t r = TAIL % S;
        if(r >= A && r + T <= S) return S;
    }
    return sizeof(trajEl_t);
}
constexpr size_t FACT_STRIDE = fact_stride();
#else
constexpr size_t FACT_STRIDE = sizeof(trajEl_t);
#endif
__device__ __forceinline__ trajEl_t *work_rec(const DevPtrs &P, int bw, int k) {
    return reinterpret_cast<trajEl_t *>(reinterpret_cast<char *>(P.work) + ((size_t)bw * P.N + k) * P.work_stride);
}

// factored: the first-order part of the record and, in place of the tensors, the products they are multiples of
// (NBASIS doubles at the start of the record's fxx member)
// (workgroups of four wavefronts: 105 against 112 ms per iteration of config 5 with one; 16: 128 registers, 165 ms)
#ifndef ILQG_DERIVS_BLOCK
#define ILQG_DERIVS_BLOCK 256
#endif
#ifdef ILQG_DERIVS_WAVES  // experiments: that many wavefronts per SIMD (register cap)
#define ILQG_DERIVS_ATTR __attribute__((amdgpu_waves_per_eu(ILQG_DERIVS_WAVES, ILQG_DERIVS_WAVES)))
#else
#define ILQG_DERIVS_ATTR
#endif
__global__ __launch_bounds__(ILQG_DERIVS_BLOCK) ILQG_DERIVS_ATTR void k_derivs_wave(DevPtrs P, ilqg_dev_opts_t O, ParamValues A, int chunk_first,
                                                    int chunk_count, int init_consts, int factored, int limit_gradients, int final_only) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (final_only: lane = trajectory, the final record alone — the steps' records come from k_derivs_parts)
    const int bw = final_only ? (int)tid : (int)(tid / (P.N + 1));
    const int k = final_only ? P.N : (int)(tid % (P.N + 1));
    const int b = chunk_first + bw;
    if(bw >= chunk_count || b >= P.B) return;
    ILQG_CALLBACKS(C, H);
    load_penalty_weights_der(C, P, b);
    if(!limit_gradients) H.limgrad = 0.0;  // (records only the backward pass reads: see ILQG_LIMIT_GRADIENTS_WANTED)
    tOptSet &o = C.o;
    // constant entries, once per buffer (init_opt, iLQG_func.tem:402-415) — of EVERY record of the chunk: the slot of a
    // trajectory that is finished serves another one in a later chunk
    if(init_consts && k < P.N) init_running(work_rec(P, bw, k), &C.o1);
    if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE) return;
    int ok = 1;
    if(k < P.N) {
        multipliersEl_t mk;
        load_mul(P, k, b, mk);
        multipliersEl_t *const mp = HAS_MUL ? &mk : nullptr;
        trajEl_t *t = work_rec(P, bw, k);
        auto body = [&]() {
#ifndef ILQG_ABLATE  // timing experiments: leave parts of the evaluation out (profiles/README.md)
#define ILQG_ABLATE 0
#endif
            if(!(ILQG_ABLATE & 32)) {
                for(int i = 0; i < NX; i++) t->x[i] = nomp(P, k, b)[NOM_X + i];
                for(int i = 0; i < NU; i++) t->u[i] = nomp(P, k, b)[NOM_U + i];
            }
            ok = 1;
            if(!(ILQG_ABLATE & 1)) ok = calcXVariableAux(t, mp, k, &o);
            if(!(ILQG_ABLATE & 1)) ok &= calcXUVariableAux(t, mp, k, &o);
            if(!(ILQG_ABLATE & 2)) ok &= calcLAuxDeriv(t, mp, k, &o);
#if ILQG_FACTORED
            if(factored) {
                // (the two as ONE generated function, so that the products share the first derivatives' 32 sin / cos
                // evaluations, was measured twice: all shared products first, 105 -> 169 ms; statements grouped by the
                // sin / cos they use, 83 -> 141 ms)
                if(!(ILQG_ABLATE & 4)) ok &= bp_derivsL_first(t, k, o.p);
                if(!(ILQG_ABLATE & 8)) ok &= bp_tensor_basis(t->fxx, t, k, o.p);
            } else
#endif
                ok &= bp_derivsL(t, k, o.p);
            // (the limits — which re-read u behind the 400 stores of the derivatives — moved in front of them, and the
            // products in front of the first derivatives: measured, no difference)
            if(!(ILQG_ABLATE & 16)) limitsU(t, k, o.p, P.N);
            if(ILQG_ABLATE & 64) {  // (timing experiment: the limits' 16 stores without the function)
                for(int i = 0; i < NU; i++) {
                    t->lower[i] = -1.0 - t->u[i];
                    t->upper[i] = 1.0 - t->u[i];
                }
            }
        };
#if ILQG_UNIFORM_GUARDS
        ok = run_guarded([&]() { body(); return ok; });
#else
        body();
        if(H.huge != 0.0) {  // an argument beyond the fast sin/cos reduction: once more through the library
            H.nonfinite = 0.0;
            H.slow = 1.0;
            body();
        }
#endif
    } else {
        trajFin_t fin;
        multipliersFin_t mfin;
        load_mul_fin(P, b, mfin);
        multipliersFin_t *const mfp = HAS_MUL ? &mfin : nullptr;
        init_final(&fin, &o);
        auto body = [&]() {
            for(int i = 0; i < NX; i++) fin.x[i] = nomp(P, P.N, b)[NOM_X + i];
            ok = calcFVariableAux(&fin, mfp, &o);
            ok &= calcFAuxDeriv(&fin, mfp, &o);
            ok &= bp_derivsF(&fin, P.N, o.p);
        };
#if ILQG_UNIFORM_GUARDS
        ok = run_guarded([&]() { body(); return ok; });
#else
        body();
        if(H.huge != 0.0) {
            H.nonfinite = 0.0;
            H.slow = 1.0;
            body();
        }
#endif
        double *out = P.f[ILQG_F_FIN] + (size_t)b * FIN;
        for(int i = 0; i < NX; i++) out[i] = fin.cx[i];
        for(int i = 0; i < SXX; i++) out[NX + i] = fin.cxx[i];
    }
    if(!ok || H.nonfinite != 0.0) P.derivs_failed[b] = 1;
}

// ---------------------------------------------------------------------------
// The time-varying part of the factored derivative records, assembled ON CHIP and written as whole cache lines (round 4).
// k_derivs_wave runs the generated scalar code on a struct per lane in HBM: every store instruction touches 64 records,
// the kernel spends most of its time waiting for stores (SQ_WAIT_ANY / SQ_WAVE_CYCLES 0.55, 116 GB written for 72 GB of
// payload) and evaluates the 64 sin / cos of a step twice (bp_derivsL_first, bp_tensor_basis).  Here a lane still owns a
// (trajectory, step), but the generated file offers the record's entries in PARTS of 16 outputs (tools/gen_problem.py
// _emit_deriv_parts: the expressions of bp_derivsL_first, bp_tensor_basis and limitsU as the same printer prints them;
// ilqg_deriv_prepare evaluates the auxiliaries, every sin / cos ONCE and the products made of them): a part's outputs go into a tile in LDS,
// [lane][16], and come out of it turned round — 16 consecutive lanes store the 16 outputs of ONE record: a whole line
// wherever the outputs are neighbours in the record (fx and fu column by column, cx, cu, the products, the limits).
// The constant entries of the records are written once per buffer by k_derivs_wave as before, the final record too.
//
// MEASURED (round 4, config 5): 51.3 ms per iteration against 43.8 for k_derivs_wave — a negative result so far, and why:
// the parts' arithmetic is a few thousand 64-bit literals.  As ONE block (parts unrolled) they are all materialised ahead
// (788 scalar registers spilled through vector lanes, 447 vector registers spilled); as a loop over a switch the
// optimiser first flattens the cheap side-effect-free cases into selects, then — with a statement at the head of every
// case that stops that — hoists the loop-invariant arithmetic of all 30 parts in front of the loop, then — with the inputs
// laundered in every trip — still holds ilqg_deriv_prepare's 32 auxiliaries + 64 sin / cos values + 64 products at
// once whatever order the generator prints them in: 360 vector registers spilled, 1.5 KB of scratch per lane, whose
// traffic is what the coalesced record stores had saved.  Called instead of inlined, a part's inputs and outputs travel
// through the caller's frame (68 ms).  What it would take: the auxiliaries and their products in a kernel of their own
// (lane = (step, auxiliary pair)) handing 64 + 32 doubles per step over in HBM or LDS.  The kernel is kept, tested
// (test_derivative_records_in_parts_equal_the_per_lane_ones) and off by default (ILQG_DERIV_PARTS=1 turns it on).
// ---------------------------------------------------------------------------
#if ILQG_FACTORED && defined(ILQG_DERIV_PARTS) && !defined(ILQG_NO_SHARED_SINCOS)
#define ILQG_HAVE_DERIV_PARTS 1
constexpr int DP_OUT = ILQG_DERIV_PART_OUT, DP_PARTS = ILQG_DERIV_PARTS, DP_NOUT = ILQG_DERIV_NOUT, DP_NPROD = ILQG_DERIV_NPROD;
static_assert(DP_OUT == 16, "a part's outputs are stored by the 16 lanes of a DPP row");
#define ILQG_OUT_OFFSET(member, index) (unsigned)(offsetof(trajEl_t, member) + (index) * sizeof(double)),
__device__ const unsigned deriv_out_offset[DP_PARTS * DP_OUT] = {ILQG_DERIV_OUTPUTS(ILQG_OUT_OFFSET)};
#undef ILQG_OUT_OFFSET
constexpr int DP_WAVES = 4;

__global__ __launch_bounds__(64 * DP_WAVES) void k_derivs_parts(DevPtrs P, ilqg_dev_opts_t O, ParamValues A, int chunk_first, int chunk_count) {
    __shared__ double tile[DP_WAVES][WAVE][DP_OUT + 1];  // (+1: the lanes' rows start on different banks)
    __shared__ unsigned long long recs[DP_WAVES][WAVE];  // the records of the wavefront's lanes (0: not to be written)
    __shared__ unsigned offs[DP_PARTS * DP_OUT];         // byte offset of every output in a record
    for(int i = threadIdx.x; i < DP_PARTS * DP_OUT; i += 64 * DP_WAVES) offs[i] = deriv_out_offset[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int bw = (int)(tid / P.N), k = (int)(tid % P.N);
    const int b = chunk_first + bw;
    const bool live = bw < chunk_count && b < P.B && P.i[ILQG_I_STATUS][(b < P.B) ? b : 0] == ILQG_ST_ACTIVE;
    if(__builtin_amdgcn_ballot_w64(live) == 0ull) return;
    ILQG_CALLBACKS(C, H);
    const double *nom = nomp(P, k, live ? b : chunk_first);
    double x[NX], u[NU];
#pragma unroll
    for(int i = 0; i < NX; i++) x[i] = nom[NOM_X + i];
#pragma unroll
    for(int i = 0; i < NU; i++) u[i] = nom[NOM_U + i];
    ilqg_deriv_aux_t aux;
    double prod[DP_NPROD], basis[NBASIS];
    int bad = 0;
    H.huge = 0.0;
    ilqg_deriv_prepare(&aux, prod, basis, &bad, x, u, k, C.o.p, P.N);
    if(H.huge != 0.0) {  // an argument beyond the fast sin / cos reduction: once more through the library
        H.slow = 1.0;
        bad = 0;
        ilqg_deriv_prepare(&aux, prod, basis, &bad, x, u, k, C.o.p, P.N);
        H.slow = 0.0;
    }
    recs[wave][lane] = live ? (unsigned long long)work_rec(P, bw, k) : 0ull;
    const int pr0 = lane >> 4, j = lane & 15;
    // (a LOOP over the parts, not unrolled: the parts' hundreds of 64-bit literals stay with their case of the switch; laid out
    // as one block they are all materialised ahead — 788 scalar registers spilled through vector lanes)
    const int nparts = DP_PARTS;
#pragma unroll 1
    for(int q = 0; q < nparts; q++) {
        // (every part's arithmetic is loop invariant — it depends on the step, not on q — and would be hoisted in front of
        // the loop, all 472 outputs at once; its inputs pass through a statement the optimiser cannot see through)
#pragma unroll
        for(int i = 0; i < NX; i++) asm volatile("" : "+v"(x[i]));
#pragma unroll
        for(int i = 0; i < NU; i++) asm volatile("" : "+v"(u[i]));
#pragma unroll
        for(int i = 0; i < DP_NPROD; i++) asm volatile("" : "+v"(prod[i]));
#pragma unroll
        for(int i = 0; i < NBASIS; i++) asm volatile("" : "+v"(basis[i]));
        double out[DP_OUT];
#pragma unroll
        for(int i = 0; i < DP_OUT; i++) out[i] = 0.0;
        ilqg_deriv_part(q, out, &bad, &aux, prod, basis, x, u, k, C.o.p, P.N);

#pragma unroll
        for(int i = 0; i < DP_OUT; i++) tile[wave][lane][i] = out[i];
        wave_sync();
        if(q * DP_OUT + j < DP_NOUT) {
            const unsigned off = offs[q * DP_OUT + j];
#pragma unroll
            for(int i = 0; i < WAVE / 4; i++) {
                const int pr = 4 * i + pr0;
                const unsigned long long base = recs[wave][pr];
                if(base) *reinterpret_cast<double *>(base + off) = tile[wave][pr][j];
            }
        }
        wave_sync();
    }
    if(live && bad) P.derivs_failed[b] = 1;
}
#else
#define ILQG_HAVE_DERIV_PARTS 0
#endif

using StepLds = std::conditional_t<ROW_STEP, RowLds<(ROW_STEP ? NX : 1), (ROW_STEP ? NU : 1)>, WaveLds<NX, NU>>;

// byte offsets of the members of a step's record (trajEl_t) the row-mapped backward step reads
struct RecOffsets {
    static constexpr unsigned cx = offsetof(trajEl_t, cx), cxx = offsetof(trajEl_t, cxx), cu = offsetof(trajEl_t, cu),
                              cuu = offsetof(trajEl_t, cuu), cxu = offsetof(trajEl_t, cxu), fx = offsetof(trajEl_t, fx),
                              fu = offsetof(trajEl_t, fu), lower = offsetof(trajEl_t, lower), upper = offsetof(trajEl_t, upper),
                              lower_sign = offsetof(trajEl_t, lower_sign), upper_sign = offsetof(trajEl_t, upper_sign),
                              lower_hx = offsetof(trajEl_t, lower_hx), upper_hx = offsetof(trajEl_t, upper_hx);
#if FULL_DDP
    static constexpr unsigned fxx = offsetof(trajEl_t, fxx), fuu = offsetof(trajEl_t, fuu), fxu = offsetof(trajEl_t, fxu);
#endif
};

#if ILQG_FACTORED
// The step's record with the tensors multiplied out on the fly: coefficient tables (an LDS copy of the generated
// ilqg_tensor_coef_*, shared by the workgroup) times the products bp_tensor_basis left in the record.  The LDS copy
// is slice-major, [slice][xx | uu | xu], so that what a lane reads of a slice sits within a few hundred bytes: ONE
// address register (table + 8 * lane) serves the whole contraction, the rest is immediate offsets.  A lane beyond
// the end of an array reads its neighbour's (or the next slice's) numbers; those sums are never used.
constexpr int FACT_SLICE = SXX + SUU + NXU;  // doubles per slice
struct FactoredSource : RecordSource<NX, NU, true, RecOffsets> {
    unsigned table;  // LDS address of the coefficient tables
    unsigned basis;  // LDS address of 64 doubles of this wavefront: the step's products, for all lanes to read
    double product;  // lane i: product i of this step (bp_tensor_basis)
    ILQG_DEV void contract(const double vxl, double (&dxx)[NTX], double (&duu)[NTU], double (&dxu)[NTC], const int lane) const {
        constexpr int PER = NTX + NTU + NTC;
        // The products every lane multiplies by are wave-uniform numbers held one per lane.  Through LDS (one store,
        // then reads of one address by all lanes) they cost no vector instruction; two v_readlane each otherwise.
        lds_base(basis + lane * 8)[0] = product;
        wave_sync();
        const LdsBase p = lds_base(table + lane * 8), pg = lds_base(basis);
        // what this lane multiplies of slice i: its entries of xx, uu, xu and the three products
        auto fetch = [&](auto ic, double (&t)[PER], double (&g)[3]) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for(int q = 0; q < NTX; q++) t[q] = p.fetch(i * FACT_SLICE + 64 * q);
#pragma unroll
            for(int q = 0; q < NTU; q++) t[NTX + q] = p.fetch(i * FACT_SLICE + SXX + 64 * q);
#pragma unroll
            for(int q = 0; q < NTC; q++) t[NTX + NTU + q] = p.fetch(i * FACT_SLICE + SXX + SUU + 64 * q);
            // (entries of constant tables: folded when the loop is unrolled, as are the comparisons below)
            const int sxx = ilqg_tensor_slice_xx[i], suu = ilqg_tensor_slice_uu[i], sxu = ilqg_tensor_slice_xu[i];
#ifdef ILQG_BASIS_READLANE  // (comparison build: the products by v_readlane)
            g[0] = lane_bcast(product, sxx);
            g[1] = lane_bcast(product, suu);
            g[2] = lane_bcast(product, sxu);
#else
            g[0] = pg.fetch(sxx);
            g[1] = (suu == sxx) ? g[0] : pg.fetch(suu);
            g[2] = (sxu == sxx) ? g[0] : ((sxu == suu) ? g[1] : pg.fetch(sxu));
#endif
        };
        double cur[PER], nxt[PER], gc[3], gn[3];
        fetch(std::integral_constant<int, 0>{}, cur, gc);
        static_for<0, NX>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr(i + 1 < NX) fetch(std::integral_constant<int, i + 1>{}, nxt, gn);  // (in flight during this slice's arithmetic)
            // d += Vx[i] * (coefficient * product), Vx[i] broadcast from lane i of the row
#pragma unroll
            for(int q = 0; q < NTC; q++) row_fma<i>(dxu[q], vxl, cur[NTX + NTU + q] * gc[2]);
#pragma unroll
            for(int q = 0; q < NTU; q++) row_fma<i>(duu[q], vxl, cur[NTX + q] * gc[1]);
#pragma unroll
            for(int q = 0; q < NTX; q++) row_fma<i>(dxx[q], vxl, cur[q] * gc[0]);
#pragma unroll
            for(int q = 0; q < PER; q++) cur[q] = nxt[q];
#pragma unroll
            for(int q = 0; q < 3; q++) gc[q] = gn[q];
        });
    }
};
constexpr int TABLE_DOUBLES = NX * FACT_SLICE;
#else
constexpr int TABLE_DOUBLES = 0;
#endif
constexpr int WAVE_LDS_DOUBLES = (int)((sizeof(StepLds) + 7) / 8);  // LDS of one wavefront

// one backward step in the form that goes with the LDS block (a template, so that only that form is instantiated)
template <bool FACT, class Lds>
__device__ __forceinline__ int step_of_wave(Lds &S, const double *tables, double product, const trajEl_t *t, const double *u_nom,
                                            const StepFields<NX, NU> &F, double *lout,
                                            double *Kout, double lambda, int regType, double &dV0, double &dV1, double &gsum,
                                            Prof *pf) {
    if constexpr(std::is_same<Lds, WaveLds<NX, NU>>::value) {
        return back_step_wave<NX, NU, FULL, HX>(S, F, lout, Kout, lambda, regType, dV0, dV1, gsum, pf);
    } else if constexpr(FACT) {
#if ILQG_FACTORED
        FactoredSource D;
        D.rec = reinterpret_cast<const char *>(t);
        D.table = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(tables));
        D.basis = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(S.basis));
        D.product = product;
        return back_step_row<NX, NU, FULL, HX>(S, D, u_nom, lout, Kout, lambda, regType, dV0, dV1, gsum, pf);
#else
        return 0;
#endif
    } else {
        const RecordSource<NX, NU, FULL, RecOffsets> D{reinterpret_cast<const char *>(t)};
        return back_step_row<NX, NU, FULL, HX>(S, D, u_nom, lout, Kout, lambda, regType, dV0, dV1, gsum, pf);
    }
}

// the value function behind the last step (Vx, packed Vxx of the final cost) into the LDS block, in the block's form
template <int A, int B>
__device__ __forceinline__ void set_final_value(RowLds<A, B> &S, const double *vx, const double *vxx, int lane) { S.set_value(vx, vxx, lane); }
template <int A, int B>
__device__ __forceinline__ void set_final_value(WaveLds<A, B> &S, const double *vx, const double *vxx, int lane) {
    for(int i = lane; i < A; i += 64) S.Vx[i] = vx[i];
    for(int i = lane; i < tri(A); i += 64) S.Vxx[i] = vxx[i];
}

// one sweep of one trajectory on one wave; returns 0 ok, 1 box-QP failed (wave-uniform)
template <bool FACT>
__device__ __forceinline__ int backward_sweep_wave(StepLds &S, const double *tables, const DevPtrs &P, int b, int bw, double lambda,
                                                   int regType, double &dV0, double &dV1, double &g_norm) {
    const int lane = threadIdx.x & 63;
    const int N = P.N;
    const double *fin = P.f[ILQG_F_FIN] + (size_t)b * FIN;
    set_final_value(S, fin, fin + NX, lane);
    for(int i = lane; i < NU; i += 64) S.l[i] = 0.0;
    wave_sync();
    dV0 = 0.0;
    dV1 = 0.0;
    double gsum = 0.0;
#ifdef ILQG_PROFILE_SECTIONS
    Prof prof;
    prof.start();
    Prof *pf = &prof;
#else
    Prof *pf = nullptr;
#endif
    auto fields = [&](int k) {
        const trajEl_t *t = work_rec(P, bw, k);
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
        F.u = nomp(P, k, b) + NOM_U;
        return F;
    };
    // factored records: lane i holds product i of the step (the one of step k-1 is requested at the start of step k)
    const int bl = (lane < NBASIS) ? lane : 0;
    double product = 0.0;
    if(FACT) product = fields(N - 1).fxx[bl];
    int failed = 0;
    for(int k = N - 1; k >= 0; k--) {
        if(pf) pf->probe(0);
        const StepFields<NX, NU> F = fields(k);
        double next_product = 0.0;
        if(FACT && k > 0) next_product = fields(k - 1).fxx[bl];
        const int rc = step_of_wave<FACT>(S, tables, product, work_rec(P, bw, k), nomp(P, k, b) + NOM_U, F, nomp(P, k, b) + NOM_L,
                                         nomp(P, k, b) + NOM_K, lambda, regType, dV0, dV1, gsum, pf);
#ifdef ILQG_PROFILE_SECTIONS
        prof.acc[7]++;  // steps executed (sweeps that are abandoned half way count with the steps they ran)
#endif
        if(rc < 1) {
            failed = 1;
            break;
        }
        product = next_product;
    }
#ifdef ILQG_PROFILE_SECTIONS
    if(lane == 0)
        for(int i = 0; i < 8; i++) atomicAdd(&ilqg_prof_cycles[i], (unsigned long long)prof.acc[i]);
#endif
    if(failed) return 1;
    g_norm = gsum / ((double)(N - 1));
    return 0;
}

// back_pass + retry loop, one wavefront (= one block) per trajectory of the chunk.  single_sweep: 1 = the
// drop-in back_pass() (caller owns the retry loop)
#ifndef ILQG_WAVE_OCC
#define ILQG_WAVE_OCC 1
#endif
#ifdef ILQG_WAVE_OCC_MAX
#define ILQG_WAVE_ATTR __attribute__((amdgpu_waves_per_eu(ILQG_WAVE_OCC, ILQG_WAVE_OCC_MAX)))
#else
#define ILQG_WAVE_ATTR
#endif
// FACT: factored records (see FactoredSource); ILQG_FACT_WAVES wavefronts per workgroup share the coefficient tables
// back_pass + retry loop of ONE trajectory (b; slot bw of the chunk's records) on the calling wavefront
template <bool FACT>
__device__ __forceinline__ void backward_of_trajectory(StepLds &S, const double *tables, const DevPtrs &P, const ilqg_dev_opts_t &O,
                                                       int single_sweep, int b, int bw) {
    const int lane = threadIdx.x & 63;
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
        rc = backward_sweep_wave<FACT>(S, tables, P, b, bw, lambda, O.regType, dV0, dV1, g_norm);
        calls++;
        if(single_sweep || rc != 1) break;
        const double t1 = dlambda * O.lambdaFactor;
        dlambda = (t1 > O.lambdaFactor) ? t1 : O.lambdaFactor;
        const double t2 = lambda * dlambda;
        lambda = (t2 > O.lambdaMin) ? t2 : O.lambdaMin;
        if(lambda > O.lambdaMax) break;
        wave_sync();
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

// The trajectories of a chunk need very different numbers of sweeps (lambda retries: 1 to 4 and more), so wavefronts
// are not tied to trajectories: each takes the next one of the chunk from a counter (P.queue, zeroed before the
// launch) until the chunk is used up — the grid is at most what the chip holds at once, and a wavefront whose
// trajectory was quick does not wait for the slow ones of its workgroup.  Every wavefront leaves the loop: the counter
// only grows.
template <bool FACT>
__global__ __launch_bounds__(64 * (FACT ? ILQG_FACT_WAVES : 1), FACT ? 1 : ILQG_WAVE_OCC) ILQG_WAVE_ATTR
void k_backward_wave(DevPtrs P, ilqg_dev_opts_t O, int single_sweep, int chunk_first, int chunk_count) {
    extern __shared__ double wave_lds[];  // [coefficient tables][per wavefront: step block, products]
    constexpr int WAVES = FACT ? ILQG_FACT_WAVES : 1;
#if ILQG_FACTORED
    if(FACT) {
        for(int i = threadIdx.x; i < NX * SXX; i += 64 * WAVES) wave_lds[i / SXX * FACT_SLICE + i % SXX] = ilqg_tensor_coef_xx[i];
        for(int i = threadIdx.x; i < NX * SUU; i += 64 * WAVES) wave_lds[i / SUU * FACT_SLICE + SXX + i % SUU] = ilqg_tensor_coef_uu[i];
        for(int i = threadIdx.x; i < NX * NXU; i += 64 * WAVES) wave_lds[i / NXU * FACT_SLICE + SXX + SUU + i % NXU] = ilqg_tensor_coef_xu[i];
        __syncthreads();  // the only meeting of the workgroup's wavefronts
    }
#endif
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    StepLds &S = *reinterpret_cast<StepLds *>(wave_lds + (FACT ? TABLE_DOUBLES : 0) + wave * WAVE_LDS_DOUBLES);
#ifdef ILQG_LDS_POISON  // debugging: whatever reads LDS it has not written gets this (a NaN, or any other pattern)
    {
        double *w = wave_lds + (FACT ? TABLE_DOUBLES : 0) + wave * WAVE_LDS_DOUBLES;
        for(int i = lane; i < WAVE_LDS_DOUBLES; i += 64) w[i] = __longlong_as_double(ILQG_LDS_POISON);
        wave_sync();
    }
#endif
    for(;;) {
        int bw = 0;
        if(lane == 0) bw = atomicAdd(P.queue, 1);
        bw = __builtin_amdgcn_readfirstlane(bw);
        if(bw >= chunk_count || chunk_first + bw >= P.B) break;
        backward_of_trajectory<FACT>(S, wave_lds, P, O, single_sweep, chunk_first + bw, bw);
        wave_sync();  // the LDS block goes to the next trajectory
    }
}
// ---------------------------------------------------------------------------
// Quad mapping (ilqg_quad.hpp): 16 lanes per trajectory, four trajectories per wavefront, each 16-lane row a worker of its
// own — it takes the next trajectory of the piece from the queue, walks its sweeps (lambda retries included) at its own
// pace and goes back to the queue.  Nothing in a step depends on the other rows of the wavefront but the instruction
// stream, so there is no hand-over and no waiting between trajectories; a row without work computes along on a valid
// record and commits nothing.  Used where the records are small (factored tensors, or no tensors), the limits do not
// depend on the state and regType is 1; the row mapping above takes the rest.
// ---------------------------------------------------------------------------
constexpr bool QUAD_STEP = ROW_STEP && !HX && (FACTORED || !FULL) && NU <= NX;
#ifndef ILQG_QUAD_WAVES  // wavefronts per workgroup (one workgroup per CU: the coefficient tables are shared)
#define ILQG_QUAD_WAVES 4
#endif
constexpr int QUAD_WAVES = ILQG_QUAD_WAVES;
using QRow = QuadRow<(QUAD_STEP ? NX : 1), (QUAD_STEP ? NU : 1)>;
struct QuadTab {
    static constexpr int NBASIS = ::NBASIS > 0 ? ::NBASIS : 1;
#if ILQG_FACTORED
    static constexpr int SLICE = FACT_SLICE;
    static __device__ __forceinline__ int slice_xx(int i) { return ilqg_tensor_slice_xx[i]; }
    static __device__ __forceinline__ int slice_uu(int i) { return ilqg_tensor_slice_uu[i]; }
    static __device__ __forceinline__ int slice_xu(int i) { return ilqg_tensor_slice_xu[i]; }
#else
    static constexpr int SLICE = 1;
    static __device__ __forceinline__ int slice_xx(int) { return 0; }
    static __device__ __forceinline__ int slice_uu(int) { return 0; }
    static __device__ __forceinline__ int slice_xu(int) { return 0; }
#endif
};

template <bool FACT>
__global__ __launch_bounds__(64 * QUAD_WAVES) void k_backward_quad(DevPtrs P, ilqg_dev_opts_t O, int single_sweep, int chunk_first, int chunk_count) {
    extern __shared__ double quad_lds[];  // [coefficient tables][per wavefront: four QuadRow blocks]
    if constexpr(QUAD_STEP) {
#if ILQG_FACTORED
        if(FACT) {
            for(int i = threadIdx.x; i < NX * SXX; i += 64 * QUAD_WAVES) quad_lds[i / SXX * FACT_SLICE + i % SXX] = ilqg_tensor_coef_xx[i];
            for(int i = threadIdx.x; i < NX * SUU; i += 64 * QUAD_WAVES) quad_lds[i / SUU * FACT_SLICE + SXX + i % SUU] = ilqg_tensor_coef_uu[i];
            for(int i = threadIdx.x; i < NX * NXU; i += 64 * QUAD_WAVES) quad_lds[i / NXU * FACT_SLICE + SXX + SUU + i % NXU] = ilqg_tensor_coef_xu[i];
            __syncthreads();  // the only meeting of the workgroup's wavefronts
        }
#endif
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
        const int cx_ = (c < NX) ? c : 0;
        const int N = P.N;
        const unsigned table = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(quad_lds));
        const unsigned rb = (unsigned)lds_addr(quad_lds + (FACT ? TABLE_DOUBLES : 0) + (wave * 4 + g) * QRow::SIZE);
        // state of this row's trajectory (the same in its 16 lanes)
        bool busy = false, start = false, drained = false;
        int b = 0, bw = 0, k = 0, calls = 0;
        double lambda = 1.0, dlambda = 1.0, dV0 = 0.0, dV1 = 0.0, gsum = 0.0, g_norm = 0.0;
        double vx = 0.0, vxx[NX], lcur = 0.0;
#pragma unroll
        for(int r = 0; r < NX; r++) vxx[r] = 0.0;

#ifdef ILQG_PROFILE_SECTIONS
        Prof prof;
        prof.start();
        Prof *pf = &prof;
#else
        Prof *pf = nullptr;
#endif
        for(;;) {
            if(pf) pf->probe(7);
            // ---- rows without a trajectory take the next ones of the piece: one atomic per wavefront and round
            while(any_lane(!busy && !drained)) {
                const bool want = !busy && !drained;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(want && c == 0);
                const int first = __builtin_ctzll(m);
                int base = 0;
                if(lane == first) base = atomicAdd(P.queue, __builtin_popcountll(m));
                base = __shfl(base, first);
                int mine = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));  // (the row's first lane)
                mine = __shfl(mine, lane & 48);
                if(want) {
                    if(mine >= chunk_count || chunk_first + mine >= P.B) {
                        drained = true;
                    } else {
                        bw = mine;
                        b = chunk_first + mine;
                        if(P.i[ILQG_I_STATUS][b] == ILQG_ST_ACTIVE) {
                            if(P.derivs_failed[b]) {
                                if(c == 0) {
                                    P.i[ILQG_I_NEED_DERIVS][b] = 0;
                                    P.i[ILQG_I_STATUS][b] = ILQG_ST_DERIVS_FAILED;
                                }
                            } else {
                                lambda = P.f[ILQG_F_LAMBDA][b];
                                dlambda = P.f[ILQG_F_DLAMBDA][b];
                                g_norm = P.f[ILQG_F_GNORM][b];
                                dV0 = dV1 = 0.0;
                                calls = 0;
                                busy = start = true;
                            }
                        }
                    }
                }
            }
            if(!any_lane(busy)) break;  // every row is out of work and the queue is used up
#ifdef ILQG_PROFILE_SECTIONS
            prof.wave_steps++;
#endif
            // ---- a sweep begins: the value function behind the last step (the final cost's), back_pass.c:60-67
            if(any_lane(start)) {
                if(start) {
                    const double *fin = P.f[ILQG_F_FIN] + (size_t)b * FIN;
                    vx = fin[cx_];
#pragma unroll
                    for(int r = 0; r < NX; r++) vxx[r] = fin[NX + ((r <= cx_) ? ut(r, cx_) : ut(cx_, r))];
                    lcur = 0.0;  // warm start of the last step (back_pass.c:163-164)
                    dV0 = dV1 = gsum = 0.0;
                    k = N - 1;
                    start = false;
                }
            }
            // ---- one step of every busy row
            const int kk = busy ? k : 0, bb = busy ? b : ((chunk_first < P.B) ? chunk_first : 0), ww = busy ? bw : 0;
            const char *rec = reinterpret_cast<const char *>(work_rec(P, ww, kk));
            double *nom = nomp(P, kk, bb);
            const int rc = back_step_quad<NX, NU, FULL, FACT, RecOffsets, QuadTab>(rb, table, rec, nom + NOM_U, nom + NOM_L, nom + NOM_K, busy, vx, vxx,
                                                                                  lcur, lambda, dV0, dV1, gsum, pf);
#ifdef ILQG_PROFILE_SECTIONS
            prof.acc[7] = 0;  // (queue, sweep starts, row transitions: not charged to a section)
#endif
            // ---- what the row does next
            if(busy) {
                bool done = false;
                int bp_rc = 0;
                if(rc < 1) {  // the sweep is abandoned (back_pass.c:168-171): raise lambda and sweep again (iLQG.c:267-275)
                    calls++;
                    bp_rc = 1;
                    done = true;
                    if(!single_sweep) {
                        const double t1 = dlambda * O.lambdaFactor;
                        dlambda = (t1 > O.lambdaFactor) ? t1 : O.lambdaFactor;
                        const double t2 = lambda * dlambda;
                        lambda = (t2 > O.lambdaMin) ? t2 : O.lambdaMin;
                        if(!(lambda > O.lambdaMax)) {
                            done = false;
                            start = true;
                        }
                    }
                } else if(k == 0) {
                    calls++;
                    g_norm = gsum / ((double)(N - 1));  // N summands over N-1 (back_pass.c:254)
                    done = true;
                } else {
                    k--;
                }
                if(done) {
                    int status = ILQG_ST_ACTIVE;
                    if(!single_sweep) {
                        if(bp_rc) {
                            status = ILQG_ST_NO_DESCENT;
                        } else if(g_norm < O.tolGrad && lambda < 1e-5) {  // iLQG.c:297-303
                            const double t1 = dlambda / O.lambdaFactor, t2 = 1.0 / O.lambdaFactor;
                            dlambda = (t1 < t2) ? t1 : t2;
                            lambda = lambda * dlambda * (lambda > O.lambdaMin);
                            status = ILQG_ST_CONVERGED_GRAD;
                        }
                    }
                    if(c == 0) {
                        P.i[ILQG_I_NEED_DERIVS][b] = 0;
                        P.i[ILQG_I_STATUS][b] = status;
                        P.f[ILQG_F_LAMBDA][b] = lambda;
                        P.f[ILQG_F_DLAMBDA][b] = dlambda;
                        P.f[ILQG_F_DV0][b] = dV0;
                        P.f[ILQG_F_DV1][b] = dV1;
                        P.f[ILQG_F_GNORM][b] = g_norm;
                        P.i[ILQG_I_BP_CALLS][b] = calls;
                        P.i[ILQG_I_BP_RC][b] = bp_rc;
                    }
                    busy = false;
                }
            }
        }
#ifdef ILQG_PROFILE_SECTIONS
        if(lane == 0) {
            for(int i = 0; i < 7; i++) atomicAdd(&ilqg_prof_cycles[i], (unsigned long long)prof.acc[i]);
            atomicAdd(&ilqg_prof_cycles[7], (unsigned long long)prof.wave_steps);  // steps of the wavefront (1 to 4 rows busy)
        }
#endif
    }
}
#endif  // ILQG_WAVE_MAP

// ---------------------------------------------------------------------------
// forward_pass: one lane per (trajectory, step size)
// ---------------------------------------------------------------------------
enum { ROLL_INIT = 0, ROLL_SEARCH = 1, ROLL_WINNER = 2, ROLL_COST = 3, ROLL_SEARCH_LIST = 4, ROLL_SECOND = 5,
       ROLL_LIST_KEEP = 6 };  // (k_rollout_parts only: the second stage alone, rows = step sizes, kept in P.cand)
constexpr int CAND_W = NX + NU;  // doubles per step of a kept second-stage roll-out

// nominal data of one step (what forward_pass reads of the nominal trajectory, iLQG_func.tem:145-155)
struct NomStep {
    double x[NX], u[NU], l[NU];
    double K[WAVE_MAP ? 1 : NXU];  // wave mapping: L is too large to prefetch, it is streamed (below)
};

// pointers to the current step of one trajectory in X, U, l, L
struct NomPtrs {
    const double *x, *u, *l, *K;
};

// CS: distance between the components of x and u where they are read (1 in the records, XSI in X / U)
template <bool GAINS, int CS>
__device__ __forceinline__ void load_nominal(NomStep &s, const NomPtrs &q) {
#pragma unroll
    for(int i = 0; i < NX; i++) s.x[i] = q.x[i * CS];
#pragma unroll
    for(int i = 0; i < NU; i++) s.u[i] = q.u[i * CS];
    if(GAINS) {
#pragma unroll
        for(int i = 0; i < NU; i++) s.l[i] = q.l[i];
        if(!WAVE_MAP) {
#pragma unroll
            for(int i = 0; i < NXU; i++) s.K[i] = q.K[i];
        }
    }
}

// Three instantiations of the roll-out:
//   RK_GENERAL  u = u_nom + alpha*l + L (x - x_nom): ROLL_SEARCH, ROLL_SEARCH_LIST and ROLL_WINNER.  These differ
//               by the RUN-TIME argument `mode` only, on purpose: the search passes and the winner pass must
//               execute the same machine code so that the re-rolled winner reproduces the cost its selection
//               was based on, bit for bit.
//   RK_INIT     alpha = 0: u = u_nom, no gains read (initial roll-out, iLQG_mex.c:116), stored in place
//   RK_COST     cost of the stored trajectory (forward_pass with cost_only = 1, iLQG.c:338)
// Lanes:
//   ROLL_SEARCH       lane = (trajectory blockIdx.x*64+lane, step size a0 + blockIdx.y); only the cost is kept
//   ROLL_SEARCH_LIST  as ROLL_SEARCH for the trajectories listed in P.pending (second stage)
//   ROLL_WINNER       lane = trajectory, accepted step size, rolled out again and stored in place of the nominal
//                     trajectory: accepted = overwritten, no candidate buffer and no swap (iLQG.c:381-386)
//   ROLL_INIT / ROLL_COST  lane = trajectory
//   ROLL_SECOND       both at once, by blockIdx.y: 0 = ROLL_WINNER for the trajectories the first stage settled,
//                     1.. = ROLL_SEARCH_LIST for step size a0 + blockIdx.y - 1, each lane KEEPING its trajectory in
//                     P.cand.  The two are independent (different trajectories), and each is a latency-bound chain of
//                     N steps that leaves most of the chip idle: side by side they cost one chain instead of two, and
//                     the few trajectories the second stage settles are copied from P.cand (k_adopt) instead of a
//                     third chain.
// Keeping the candidates of the search instead of re-rolling the winner was measured and dropped: a whole-batch copy
// of X and U per step size costs more HBM write time than the winner pass, and candidates of the compacted second
// stage can only be written or copied back as scattered 8-byte pieces.
// The time step itself is straight-line code: the generated callbacks' NaN/Inf guards and the
// huge-argument case of sin/cos are hooks (see ilqg_hooks), tested once per step.
enum { RK_GENERAL = 0, RK_INIT = 1, RK_COST = 2 };
#ifdef ILQG_ROLLOUT_WAVES  // experiments: force that many wavefronts of the roll-out kernels per SIMD (register cap)
#define ILQG_ROLLOUT_ATTR __attribute__((amdgpu_waves_per_eu(ILQG_ROLLOUT_WAVES)))
#else
#define ILQG_ROLLOUT_ATTR
#endif

// Wavefronts per workgroup of the roll-outs (experiments).  One: in the wave mapping a roll-out wavefront fills the
// register file of its SIMD and the backward kernel's workgroup needs a whole CU, so four per workgroup (one CU
// instead of four blocked) was tried to let the roll-outs of one group of trajectories share the chip with the
// backward pass of another — 1.86 it/s (one group) and 1.92 (two groups, 8 hardware queues) against 1.98 as is:
// the backward kernel keeps the SIMDs it runs on busy, there is little idle issue time to give away.
#ifndef ILQG_ROLL_WAVES
#define ILQG_ROLL_WAVES 1
#endif
constexpr int ROLL_BLOCK = WAVE * ILQG_ROLL_WAVES;

template <int KIND>
__global__ __launch_bounds__(ROLL_BLOCK) ILQG_ROLLOUT_ATTR void k_rollout(DevPtrs P, ilqg_dev_opts_t O, ParamValues A, int mode, int a0) {
    int b = blockIdx.x * ROLL_BLOCK + threadIdx.x;
    int ai = a0 + blockIdx.y;
    double *keep = nullptr;  // second stage: where this lane's trajectory is kept
    int second_row = -1;     //   and the row of the candidate buffer (its step size)
    if(mode == ROLL_SECOND) {
        if(blockIdx.y == 0) {
            mode = ROLL_WINNER;
        } else {
            mode = ROLL_SEARCH_LIST;
            ai = a0 + blockIdx.y - 1;
            second_row = blockIdx.y - 1;
        }
    }
    if(mode == ROLL_SEARCH_LIST) {
        // The list is short (the grid covers the worst case, most blocks return at once), and workgroups go to the CUs
        // round robin: with the entries in the same place of every row of the grid, the busy workgroups of ALL rows
        // landed on the same few CUs (measured, n = 16 problem, 2 676 entries: 1 / 2 / 4 / 7 rows 38 / 53 / 66 / 83 ms).
        // Each row starts its walk over the list somewhere else.
        const int nb = gridDim.x, rows = (int)gridDim.y - (second_row >= 0 ? 1 : 0);
        const int row = second_row >= 0 ? second_row : (int)blockIdx.y;
        const int first = (int)(((long long)row * nb) / rows);
        b = ((int)blockIdx.x + first) % nb * ROLL_BLOCK + threadIdx.x;
        if(b >= *P.n_pending) return;
        if(second_row >= 0) keep = P.cand + (size_t)second_row * (P.N + 1) * CAND_W * P.Bp + b;
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
        if(!P.i[ILQG_I_RESWEEP][b]) return;  // set by k_update
    }
    constexpr bool cost_only = (KIND == RK_COST);
    constexpr bool gains = (KIND == RK_GENERAL);
    const bool store = (mode == ROLL_INIT || mode == ROLL_WINNER);
    const bool feedback = (alpha != 0.0);  // alpha == 0.0: u = u_nom without feedback (iLQG_func.tem:156-158)

    ILQG_CALLBACKS(C, H);
    // penalty weights of this trajectory; the initial roll-out runs before the solver entry sets them, with the
    // zero-initialised option set of the MEX entry (iLQG_mex.c:23,116; iLQG.c:233-234)
    if(HAS_MUL && KIND == RK_INIT) set_penalty_weights(C, 0.0, 0.0);
    else load_penalty_weights(C, P, b);
    trajEl_t ct;
    multipliersEl_t mk;
    multipliersEl_t *const mp = HAS_MUL ? &mk : nullptr;
    init_running(&ct, &C.o1);  // constant auxiliaries of this problem (iLQG_func.tem:312-347)

    // this trajectory's step 0 in every field; all of them advance by one step per iteration
    // The line search reads the packed records; the initial roll-out and the cost sweep read the current (x, u)
    // where every roll-out stores them (see cur_x).
    constexpr int CS = (KIND == RK_GENERAL) ? 1 : XSI;
    NomPtrs q;
    q.x = (KIND == RK_GENERAL) ? nomp(P, 0, b) + NOM_X : cur_x(P, 0, b);
    q.u = (KIND == RK_GENERAL) ? nomp(P, 0, b) + NOM_U : cur_u(P, 0, b);
    q.l = nomp(P, 0, b) + NOM_L;
    q.K = nomp(P, 0, b) + NOM_K;
    const size_t xs = (KIND == RK_GENERAL) ? (size_t)RN : cur_xstride(P), us = (KIND == RK_GENERAL) ? (size_t)RN : cur_ustride(P);
    constexpr int ks = RN;
    // a stored roll-out (initial, winner) replaces the current trajectory
    double *xo = cur_x(P, 0, b), *uo = cur_u(P, 0, b);
    const size_t xos = cur_xstride(P), uos = cur_ustride(P);
    double *ro = nomp(P, 0, b);  // initial roll-out: the record's copy as well

    double xc[NX];
#pragma unroll
    for(int i = 0; i < NX; i++) xc[i] = q.x[i * CS];  // x0 (iLQG_func.tem:141-142)
    double csum = 0.0;
    int okc = 1;
    NomStep cur;
    load_nominal<gains, CS>(cur, q);
    drain_memory_ops();
    for(int k = 0; k < N; k++) {
        NomPtrs qn;
        qn.x = q.x + xs;
        qn.u = q.u + us;
        qn.l = q.l + ks;
        qn.K = q.K + ks;
        // inputs of the step
        double xin[NX], uin[NU];
        if(cost_only) {
#pragma unroll
            for(int i = 0; i < NX; i++) xin[i] = cur.x[i];
#pragma unroll
            for(int i = 0; i < NU; i++) uin[i] = cur.u[i];
        } else {
#pragma unroll
            for(int i = 0; i < NX; i++) xin[i] = xc[i];
            if(gains) {
                // u = u_nom + alpha*l + L (x - x_nom), state by state (iLQG_func.tem:146-155)
                double uf[NU];
#pragma unroll
                for(int j = 0; j < NU; j++) uf[j] = cur.u[j] + cur.l[j] * alpha;
#pragma unroll
                for(int i = 0; i < NX; i++) {
                    const double dx = xin[i] - cur.x[i];
                    if(WAVE_MAP) {
                        const double *Kk = q.K + i * NU;
#pragma unroll
                        for(int j = 0; j < NU; j++) uf[j] += Kk[j] * dx;
                    } else {
#pragma unroll
                        for(int j = 0; j < NU; j++) uf[j] += cur.K[j + i * NU] * dx;
                    }
                }
#pragma unroll
                for(int j = 0; j < NU; j++) uin[j] = feedback ? uf[j] : cur.u[j];
            } else {
#pragma unroll
                for(int j = 0; j < NU; j++) uin[j] = cur.u[j];
            }
        }

        // The nominal data of this step have been consumed: the next step's are loaded into the same variables
        // right here and are in flight while the step computes (no second buffer, no hand-over copies).
        // (unconditionally: the records have a step N; the tiled U has not, its pointer stays on the last step)
        if(KIND != RK_GENERAL && k + 1 >= N) qn.u = q.u;
        load_nominal<gains, CS>(cur, qn);

        if(HAS_MUL) {
            if(KIND == RK_INIT) {  // init_opt -> init_multipliers (iLQG_func.tem:371-402), element by element
                C.o1.multipliers.t = &mk;
                init_multipliers_running(&C.o1);
                store_mul(P, k, b, mk);
            } else {
                load_mul(P, k, b, mk);
            }
        }
        // the step (iLQG_func.tem:160-176)
        double xnext[NX];
        const double nf0 = H.nonfinite;
        H.huge = 0.0;
        auto step = [&]() {
#pragma unroll
            for(int i = 0; i < NX; i++) ct.x[i] = xin[i];
#pragma unroll
            for(int j = 0; j < NU; j++) ct.u[j] = uin[j];
            int r = calcXVariableAux(&ct, mp, k, &C.o);
            if(!cost_only) clampU(ct.u, &ct, k, C.o.p, N);
            r &= calcXUVariableAux(&ct, mp, k, &C.o);
            if(!cost_only) r &= ddpf(xnext, &ct, k, C.o.p, N);
            r &= ddpL(&ct, k, &C.o);
            return r;
        };
        int r = 1;
#if ILQG_UNIFORM_GUARDS
        if(okc) r = run_guarded(step);  // a lane that has failed stays out: its guards would fail the wavefront again
#else
        r = step();
        if(H.huge != 0.0) {  // an argument beyond the fast sin/cos reduction: once more through the library
            H.nonfinite = nf0;
            H.slow = 1.0;
            r = step();
            H.slow = 0.0;
        }
#endif
        okc &= r;
        csum += ct.c;
        // The step's results are stored right away, i.e. BEHIND the prefetch of the next step in issue order: the
        // memory counter retires in issue order, so the wait for the prefetched values at the end of this
        // iteration (vmcnt = number of younger operations) leaves these stores in flight.
        if(store) {
#pragma unroll
            for(int i = 0; i < NX; i++) xo[i * XSI] = ct.x[i];
#pragma unroll
            for(int i = 0; i < NU; i++) uo[i * XSI] = ct.u[i];
            if(KIND == RK_INIT && !WAVE_MAP) {
#pragma unroll
                for(int i = 0; i < NX; i++) ro[NOM_X + i] = ct.x[i];
#pragma unroll
                for(int i = 0; i < NU; i++) ro[NOM_U + i] = ct.u[i];
            }
        } else if(KIND == RK_GENERAL && keep) {
#pragma unroll
            for(int i = 0; i < NX; i++) keep[(size_t)i * P.Bp] = ct.x[i];
#pragma unroll
            for(int i = 0; i < NU; i++) keep[(size_t)(NX + i) * P.Bp] = ct.u[i];
            keep += (size_t)CAND_W * P.Bp;
        }
        if(!cost_only) {
#pragma unroll
            for(int i = 0; i < NX; i++) xc[i] = xnext[i];
        }
        q = qn;
        xo += xos;
        uo += uos;
        ro += RN;
    }
    // final cost (iLQG_func.tem:179-182); q.x now points at step N
    {
        trajFin_t cf;
        multipliersFin_t mf;
        if(HAS_MUL) {
            if(KIND == RK_INIT) {
                init_multipliers_final(&C.o);
                mf = C.o.multipliers.f;
                store_mul_fin(P, b, mf);
            } else {
                load_mul_fin(P, b, mf);
            }
        }
        init_final(&cf, &C.o);
        double xin[NX];
#pragma unroll
        for(int i = 0; i < NX; i++) xin[i] = cost_only ? q.x[i * CS] : xc[i];
        const double nf0 = H.nonfinite;
        H.huge = 0.0;
        auto fin = [&]() {
#pragma unroll
            for(int i = 0; i < NX; i++) cf.x[i] = xin[i];
            int r = calcFVariableAux(&cf, HAS_MUL ? &mf : nullptr, &C.o);
            r &= ddpF(&cf, &C.o);
            return r;
        };
        int r = 1;
#if ILQG_UNIFORM_GUARDS
        if(okc) r = run_guarded(fin);
#else
        r = fin();
        if(H.huge != 0.0) {
            H.nonfinite = nf0;
            H.slow = 1.0;
            r = fin();
            H.slow = 0.0;
        }
#endif
        okc &= r;
        csum += cf.c;
        if(store) {
#pragma unroll
            for(int i = 0; i < NX; i++) xo[i * XSI] = cf.x[i];  // xo points at step N now
            if(KIND == RK_INIT && !WAVE_MAP) {
#pragma unroll
                for(int i = 0; i < NX; i++) ro[NOM_X + i] = cf.x[i];
            }
        } else if(KIND == RK_GENERAL && keep) {
#pragma unroll
            for(int i = 0; i < NX; i++) keep[(size_t)i * P.Bp] = cf.x[i];
        }
    }
    // forward_pass returns 0 as soon as a guarded value is NaN or Inf (genenerator_main.mac:193-198)
    const int ok = (okc && H.nonfinite == 0.0) ? 1 : 0;

    if(mode == ROLL_SEARCH || mode == ROLL_SEARCH_LIST) {
        P.f[ILQG_F_ALPHA_COST][tile_ix(ILQG_MAX_ALPHA, ai, b)] = csum;
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

#if !ILQG_WAVE_MAP
// ---------------------------------------------------------------------------
// Line search that KEEPS what it rolls out (lane mapping, option ls_keep = 2)
// ---------------------------------------------------------------------------
// A wavefront takes T = 64 / n trajectories and all n step sizes of the stage, lane = a * T + t (step size a,
// trajectory t of the wavefront).  What that buys:
//  * everything the selection of line_search.c:37-75 needs is in the wavefront when the roll-outs end, so it happens
//    right there, by wavefront shuffles (no k_select launch, no per-alpha round trip through HBM);
//  * the nominal record of a step is ONE line request per trajectory, shared by its n lanes — which is what leaves
//    the memory system room for the stores: every lane keeps its roll-out (measured: the first stage with one step
//    size per ROW of the grid fetches the records once per row, 11 GB per iteration, and slows from 2.15 to 3.26 ms
//    when it also stores; in this mapping 2.80 -> 2.87 ms);
//  * with T = 16 (n = 4) the 16 lanes of a step size store 16 consecutive doubles: one whole line.
// Stage 0 (all trajectories, step sizes [0, n)): the roll-outs go to the planes of set `set` (see cur_x), in the layout
// of X / U, and the accepted one BECOMES the current trajectory when k_commit changes the trajectory's location index:
// no second roll-out of the winner (a chain of N dependent steps, 376 vector instructions per step and trajectory) and
// no copy.  Trajectories without an acceptable step size are appended to P.pending.
// Stage 1 (the entries of P.pending, step sizes [a0, a0 + n)): kept in P.cand by entry; k_adopt_home copies the
// accepted ones (few) into X / U.
// The scan state (last cnew / dcost / expected) is carried from stage to stage exactly as k_select does, so the
// accepted index and the values left behind are those of one sequential scan (line_search.c:37-75).
#ifndef ILQG_SEARCH_OCC  // wavefronts of k_search per SIMD the register allocation must allow
#define ILQG_SEARCH_OCC 1
#endif
// DMA: the nominal records of the wavefront's T trajectories reach the lanes through LDS.  Every lane needs the whole
// record of its trajectory (x, u, l, L: 128 bytes for CarParking) in every step, and with per-lane loads the n lanes of
// a trajectory each pull it through the CU's vector-memory path: 8 KB per wavefront and step for 2 KB of data, on the
// path that also carries the stores of the kept roll-outs — measured, the stores alone cost 0.65 of the kernel's 3.07
// ms, so that path and not the arithmetic bounds the kernel.  Instead the records are fetched ONCE, by
// global_load_lds_dwordx4 straight into LDS (no staging registers): lane i of load m moves the 16-byte piece
// q = 64 m + i, and the pieces lie piece-major, q = j T + r (piece j of record r), so that the T lanes of a step size
// read T consecutive 16-byte pieces (no bank conflicts).  Two buffers: the records of step k+2 are on their way while
// step k+1's are read.  Used when a step's pieces fit DMA_LOADS loads (T * RN / 2 <= 64 DMA_LOADS).
constexpr int DMA_LOADS = 3;
template <int stage, bool DMA>
__global__ __launch_bounds__(WAVE, ILQG_SEARCH_OCC) void k_search(DevPtrs P, ilqg_dev_opts_t O, ParamValues A, int a0, int n, int set) {
    extern __shared__ __attribute__((aligned(16))) double s_rec[];  // DMA: [2][T * RN] doubles
    const int lane = threadIdx.x;
    const int T = WAVE / n;
    int a = lane / T;
    const int t = lane - a * T;
    const bool used = a < n;  // (64 - n * T) lanes have nothing of their own to do: they repeat the last step size
    if(!used) a = n - 1;
    const int count = stage ? *P.n_pending : P.B;
    if((int)blockIdx.x * T >= count) return;
    const int e = blockIdx.x * T + t;                   // trajectory (stage 0) or entry of the pending list
    const int ee = e < count ? e : count - 1;           // lanes beyond the end repeat the last one (results dropped)
    const int b = stage ? P.pending[ee] : ee;
    const bool live = used && e < count && P.i[ILQG_I_STATUS][b] == ILQG_ST_ACTIVE;
    if(__builtin_amdgcn_ballot_w64(live) == 0ull) return;
    const int N = P.N;
    const int ai = a0 + a;
    const double alpha = O.alpha[ai];
    const bool feedback = (alpha != 0.0);  // alpha == 0.0: u = u_nom without feedback (iLQG_func.tem:156-158)

    ILQG_CALLBACKS(C, H);
    load_penalty_weights(C, P, b);
    trajEl_t ct;
    multipliersEl_t mk;
    multipliersEl_t *const mp = HAS_MUL ? &mk : nullptr;
    init_running(&ct, &C.o1);

    NomPtrs q;
    q.x = nomp(P, 0, b) + NOM_X;
    q.u = nomp(P, 0, b) + NOM_U;
    q.l = nomp(P, 0, b) + NOM_L;
    q.K = nomp(P, 0, b) + NOM_K;
    // DMA: this lane's pieces (see above) — source of load m at step 0, and whether the lane takes part in it
    constexpr int PIECES = RN / 2;
    const double *dsrc[DMA_LOADS];
    bool dact[DMA_LOADS];
    if(DMA) {
#pragma unroll
        for(int m = 0; m < DMA_LOADS; m++) {
            const int qq = m * WAVE + lane;
            const int j = qq / T, r = qq - j * T;
            dact[m] = qq < T * PIECES;
            const int br = __shfl(b, r < T ? r : 0);  // trajectory of record r: held by lane r (step size 0)
            dsrc[m] = nomp(P, 0, br) + 2 * (j < PIECES ? j : 0);
        }
    }
    auto dma_issue = [&](int buf) {  // the records of the step dsrc points at -> LDS buffer buf; dsrc moves on one step
#pragma unroll
        for(int m = 0; m < DMA_LOADS; m++) {
            if(dact[m])
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)dsrc[m],
                                                 (__attribute__((address_space(3))) void *)(s_rec + (size_t)buf * T * RN + m * WAVE * 2),
                                                 16, 0, 0);
            dsrc[m] += RN;
        }
    };
    auto dma_read = [&](NomStep &c, int buf) {  // this lane's record out of LDS buffer buf
        double rec[RN];
        const double *base = s_rec + (size_t)buf * T * RN + 2 * t;
#pragma unroll
        for(int j = 0; j < PIECES; j++) {
            rec[2 * j] = base[(size_t)j * T * 2];
            rec[2 * j + 1] = base[(size_t)j * T * 2 + 1];
        }
#pragma unroll
        for(int i = 0; i < NX; i++) c.x[i] = rec[NOM_X + i];
#pragma unroll
        for(int i = 0; i < NU; i++) c.u[i] = rec[NOM_U + i];
#pragma unroll
        for(int i = 0; i < NU; i++) c.l[i] = rec[NOM_L + i];
#pragma unroll
        for(int i = 0; i < NXU; i++) c.K[i] = rec[NOM_K + i];
    };
    // where this lane keeps its roll-out: stage 0 in plane (set, a) in the layout of X / U, stage 1 in P.cand by entry
    double *kx, *ku;
    const size_t kxs = stage ? (size_t)CAND_W * P.Bp : cur_xstride(P), kus = stage ? kxs : cur_ustride(P);  // between steps
    const size_t kcs = stage ? (size_t)P.Bp : (size_t)XSI;                                                 // between components
    if(stage) {
        kx = P.cand + (size_t)a * (N + 1) * CAND_W * P.Bp + ee;
        ku = kx + (size_t)NX * P.Bp;
    } else {
        const int plane = set * P.plane_n + a;
        kx = P.xpl + (size_t)plane * P.xplane + ix(P, NX, N + 1, 0, 0, b);
        ku = P.upl + (size_t)plane * P.uplane + ix(P, NU, N, 0, 0, b);
    }

    double xc[NX];
#pragma unroll
    for(int i = 0; i < NX; i++) xc[i] = q.x[i];  // x0 (iLQG_func.tem:141-142)
    double csum = 0.0;
    int okc = 1;
    NomStep cur;
    if(DMA) {
        dma_issue(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dma_issue(1);  // step 1 (N >= 2)
    } else {
        load_nominal<true, 1>(cur, q);
    }
    drain_memory_ops();
    for(int k = 0; k < N; k++) {
        NomPtrs qn;
        qn.x = q.x + RN;
        qn.u = q.u + RN;
        qn.l = q.l + RN;
        qn.K = q.K + RN;
        // DMA: the record is read out of LDS HERE, where it is consumed, not a step ahead: its 32 registers are live for
        // the first few instructions of the step only, not across the sin / cos evaluations (145 -> fewer registers)
        if(DMA) dma_read(cur, k & 1);
        double xin[NX], uin[NU];
#pragma unroll
        for(int i = 0; i < NX; i++) xin[i] = xc[i];
        {   // u = u_nom + alpha*l + L (x - x_nom), state by state (iLQG_func.tem:146-155)
            double uf[NU];
#pragma unroll
            for(int j = 0; j < NU; j++) uf[j] = cur.u[j] + cur.l[j] * alpha;
#pragma unroll
            for(int i = 0; i < NX; i++) {
                const double dx = xin[i] - cur.x[i];
#pragma unroll
                for(int j = 0; j < NU; j++) uf[j] += cur.K[j + i * NU] * dx;
            }
#pragma unroll
            for(int j = 0; j < NU; j++) uin[j] = feedback ? uf[j] : cur.u[j];
        }
        if(!DMA) load_nominal<true, 1>(cur, qn);  // the next step's record is in flight while this one computes
        if(HAS_MUL) load_mul(P, k, b, mk);
        double xnext[NX];
        const double nf0 = H.nonfinite;
        H.huge = 0.0;
        auto step = [&]() {
#pragma unroll
            for(int i = 0; i < NX; i++) ct.x[i] = xin[i];
#pragma unroll
            for(int j = 0; j < NU; j++) ct.u[j] = uin[j];
            int r = calcXVariableAux(&ct, mp, k, &C.o);
            clampU(ct.u, &ct, k, C.o.p, N);
            r &= calcXUVariableAux(&ct, mp, k, &C.o);
            r &= ddpf(xnext, &ct, k, C.o.p, N);
            r &= ddpL(&ct, k, &C.o);
            return r;
        };
        int r = step();
        if(H.huge != 0.0) {  // an argument beyond the fast sin/cos reduction: once more through the library
            H.nonfinite = nf0;
            H.slow = 1.0;
            r = step();
            H.slow = 0.0;
        }
        okc &= r;
        csum += ct.c;
        if(DMA) {
            // Outstanding here: the kept roll-out of step k-1 and the records of step k+1, both issued a whole step
            // ago.  The wait stands BEFORE this step's stores so that it never waits for a store just issued.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if(live) {  // (!DMA: behind the prefetch in issue order: the wait for the prefetched values leaves these in flight)
#pragma unroll
            for(int i = 0; i < NX; i++) kx[i * kcs] = ct.x[i];
#pragma unroll
            for(int i = 0; i < NU; i++) ku[i * kcs] = ct.u[i];
        }
        if(DMA && k + 2 <= N) dma_issue(k & 1);  // (that buffer held step k: read at the end of step k-1)
        kx += kxs;
        ku += kus;
#pragma unroll
        for(int i = 0; i < NX; i++) xc[i] = xnext[i];
        q = qn;
    }
    {   // final cost (iLQG_func.tem:179-182)
        trajFin_t cf;
        multipliersFin_t mf;
        if(HAS_MUL) load_mul_fin(P, b, mf);
        init_final(&cf, &C.o);
        const double nf0 = H.nonfinite;
        H.huge = 0.0;
        auto fin = [&]() {
#pragma unroll
            for(int i = 0; i < NX; i++) cf.x[i] = xc[i];
            int r = calcFVariableAux(&cf, HAS_MUL ? &mf : nullptr, &C.o);
            r &= ddpF(&cf, &C.o);
            return r;
        };
        int r = fin();
        if(H.huge != 0.0) {
            H.nonfinite = nf0;
            H.slow = 1.0;
            r = fin();
            H.slow = 0.0;
        }
        okc &= r;
        csum += cf.c;
        if(live) {
#pragma unroll
            for(int i = 0; i < NX; i++) kx[i * kcs] = cf.x[i];
        }
    }
    const int ok = (okc && H.nonfinite == 0.0) ? 1 : 0;
    if(live) {
        P.f[ILQG_F_ALPHA_COST][tile_ix(ILQG_MAX_ALPHA, ai, b)] = csum;
        P.i[ILQG_I_ALPHA_OK][(size_t)ai * P.Bp + b] = ok;
    }

    // Selection, as k_select does it (line_search.c:37-75), by the lane of the trajectory's first step size; the
    // other lanes' values come by wavefront shuffles.  Every lane computes the test of its own step size.
    const double cost = P.f[ILQG_F_COST][b], dV0 = P.f[ILQG_F_DV0][b], dV1 = P.f[ILQG_F_DV1][b];
    const double my_dcost = cost - csum;
    const double my_expected = -alpha * (dV0 + alpha * dV1);
    const double my_z = (my_expected > 0) ? my_dcost / my_expected : 0.0;
    const int my_pass = (ok && my_z > O.zMin) ? 1 : 0;
    double cnew = (a0 > 0) ? P.f[ILQG_F_NEW_COST][b] : 0.0;
    double dcost = P.f[ILQG_F_DCOST][b], expected = P.f[ILQG_F_EXPECTED][b];
    int win = -1;
    for(int i = 0; i < n; i++) {
        const int src = i * T + t;
        const int ok_i = __shfl(ok, src);
        const int pass_i = __shfl(my_pass, src);
        const double cnew_i = __shfl(csum, src);
        const double dcost_i = __shfl(my_dcost, src);
        const double expected_i = __shfl(my_expected, src);
        if(win < 0) {
            cnew = cnew_i;
            if(ok_i) {
                dcost = dcost_i;
                expected = expected_i;
                if(pass_i) win = i;
            }
        }
    }
    const bool leader = live && lane < T;  // a == 0
    if(leader) {
        P.i[ILQG_I_ALPHA_IDX][b] = (win >= 0 ? a0 + win : a0 + n) + 1;
        P.i[ILQG_I_ACCEPTED][b] = win >= 0 ? 1 : 0;
        P.f[ILQG_F_NEW_COST][b] = cnew;
        P.f[ILQG_F_DCOST][b] = dcost;
        P.f[ILQG_F_EXPECTED][b] = expected;
    }
    // to the second stage: one atomicAdd per wavefront
    const bool more = leader && win < 0 && stage == 0 && a0 + n < O.n_alpha;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(more);
    if(m != 0ull) {
        int base = 0;
        const int first = __builtin_ctzll(m);
        if(lane == first) base = atomicAdd(P.n_pending_next, __builtin_popcountll(m));
        base = __shfl(base, first);
        if(more) P.pending[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = b;
    }
}

// After k_search: what has to be COPIED into X / U, one thread per (step, trajectory or entry).
//  * the trajectories the second stage settled, from P.cand (by entry);
//  * the trajectories that found no acceptable step size at all while their current trajectory lives in a plane: the
//    next search writes the set of planes it is in (the sets take turns), so it moves to X / U (rare: 0.5 % of the
//    trajectories of the benchmark window).
// The location indices change afterwards (k_commit), when nothing reads the old ones any more.  s1 = step sizes of
// the first stage.
__global__ void k_adopt_home(DevPtrs P, int s1) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int np = *P.n_pending, N1 = P.N + 1;
    const size_t total2 = (size_t)N1 * np;
    for(size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total2; w += stride) {
        const int e = (int)(w % np), k = (int)(w / np);
        const int b = P.pending[e];
        if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE) continue;
        double *xo = home_x(P, k, b);
        if(P.i[ILQG_I_ACCEPTED][b]) {
            const int a = P.i[ILQG_I_ALPHA_IDX][b] - 1 - s1;
            const double *src = P.cand + ((size_t)a * N1 + k) * CAND_W * P.Bp + e;
#pragma unroll
            for(int i = 0; i < NX; i++) xo[i * XSI] = src[(size_t)i * P.Bp];
            if(k < P.N) {
                double *uo = home_u(P, k, b);
#pragma unroll
                for(int i = 0; i < NU; i++) uo[i * XSI] = src[(size_t)(NX + i) * P.Bp];
            }
        } else if(P.i[ILQG_I_LOC][b]) {
            const double *xs = cur_x(P, k, b);
#pragma unroll
            for(int i = 0; i < NX; i++) xo[i * XSI] = xs[i * XSI];
            if(k < P.N) {
                const double *us = cur_u(P, k, b);
                double *uo = home_u(P, k, b);
#pragma unroll
                for(int i = 0; i < NU; i++) uo[i * XSI] = us[i * XSI];
            }
        }
    }
}

// ... and the same for a search without a second stage, where no list of the undecided exists: rejected trajectories
// whose current trajectory lives in a plane move to X / U.  One thread per (step, trajectory).
__global__ void k_rejected_home(DevPtrs P) {
    const size_t total = (size_t)(P.N + 1) * P.Bp, stride = (size_t)gridDim.x * blockDim.x;
    for(size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += stride) {
        const int b = (int)(w % P.Bp), k = (int)(w / P.Bp);
        if(b >= P.B || P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE || P.i[ILQG_I_ACCEPTED][b] || !P.i[ILQG_I_LOC][b]) continue;
        const double *xs = cur_x(P, k, b);
        double *xo = home_x(P, k, b);
#pragma unroll
        for(int i = 0; i < NX; i++) xo[i * XSI] = xs[i * XSI];
        if(k < P.N) {
            const double *us = cur_u(P, k, b);
            double *uo = home_u(P, k, b);
#pragma unroll
            for(int i = 0; i < NU; i++) uo[i * XSI] = us[i * XSI];
        }
    }
}

// The new location of every trajectory that took part in the search: the plane of the accepted step size of the first
// stage, else X / U (adopted from the second stage, or moved there by the kernels above, or there already).
__global__ void k_commit(DevPtrs P, int s1, int set) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if(b >= P.B || P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE) return;
    const int a = P.i[ILQG_I_ALPHA_IDX][b] - 1;
    if(P.i[ILQG_I_ACCEPTED][b])
        P.i[ILQG_I_LOC][b] = (a < s1) ? 1 + set * P.plane_n + a : 0;
    else
        P.i[ILQG_I_LOC][b] = 0;
}

// every current trajectory into X / U (before the host reads or writes them, before an initial roll-out): copy, then
// the caller clears the location indices
__global__ void k_all_home(DevPtrs P) {
    const size_t total = (size_t)(P.N + 1) * P.Bp, stride = (size_t)gridDim.x * blockDim.x;
    for(size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += stride) {
        const int b = (int)(w % P.Bp), k = (int)(w / P.Bp);
        if(b >= P.B || !P.i[ILQG_I_LOC][b]) continue;
        const double *xs = cur_x(P, k, b);
        double *xo = home_x(P, k, b);
#pragma unroll
        for(int i = 0; i < NX; i++) xo[i * XSI] = xs[i * XSI];
        if(k < P.N) {
            const double *us = cur_u(P, k, b);
            double *uo = home_u(P, k, b);
#pragma unroll
            for(int i = 0; i < NU; i++) uo[i * XSI] = us[i * XSI];
        }
    }
}
#endif  // !ILQG_WAVE_MAP

#if ILQG_WAVE_MAP && defined(ILQG_ROLLOUT_PARTS)
// ---------------------------------------------------------------------------
// Roll-outs of the wave mapping on ILQG_ROLLOUT_PARTS wavefronts per 64 trajectories (round 3)
// ---------------------------------------------------------------------------
// k_rollout gives a trajectory one lane, and a lane walks the generated scalar code of a step alone: for the n = 16
// problem 32 sin / cos evaluations and ~800 multiply-adds, 35 us per step, on a `trajEl_t` in scratch memory — with
// 16 384 trajectories that is 256 wavefronts on 1 024 SIMDs, each a chain of 1 000 such steps (35 + 56 ms of a 370 ms
// iteration, the chip three quarters idle).  The generated file now offers the step in N_X independent PARTS
// (ilqg_step_part, tools/gen_problem.py: part r = component r of the dynamics with the auxiliaries it needs and every
// N_X-th summand of the running cost; the assignments are the ones of calcX*VariableAux / ddpf / ddpL, unchanged).
// Here a workgroup of RW wavefronts (N_X, or N_X / 2 above 8) takes 64 trajectories, wavefront = part(s), lane = trajectory:
//   phase 1  wavefront j < N_U: input j, u_j = u_nom_j + alpha l_j + sum_i L(j,i) (x_i - x_nom_i) (the order of
//            iLQG_func.tem:146-155), its operands prefetched one step ahead;            -> LDS, barrier
//   phase 2  every wavefront: all inputs out of LDS, clampU, its part of the step;      -> LDS, barrier
//   then     every wavefront reads the new state; the first one adds the cost summands in the order of ddpL's sum.
// The scalar code path per wavefront shrinks to 2/16 of the step and no `trajEl_t` lives in scratch.  Same expressions,
// same order: the -ffp-contract=off build gives the bits of k_rollout (the product build differs from it by contractions
// across the cost summands, which are added one by one here).  Measured (config 5): first stage 34.9 -> 13.8 ms, second
// launch 56 -> 50 ms, 2.69 -> 2.89 it/s.  A wavefront still issues ~1 400 instructions per step (625 per component: 66
// 64-bit literals = 132 scalar moves, two sin / cos calls that each evaluate both functions): instruction bound, not
// memory bound (prefetch placement, an LDS-only barrier instead of __syncthreads: no change).  Tried: one part per
// wavefront (16 wavefronts, 128 registers: 164 spilled); sin / cos inlined instead of called, with the huge-argument
// case repeated by the scalar kernel (no calls, 22 spills, but 10 000 instructions of straight-line code for the 16
// cases — more than the instruction cache: 34 / 91 ms, slower than the calls).
// Modes as k_rollout's general instantiation: ROLL_SEARCH (row = step size), ROLL_WINNER, ROLL_SEARCH_LIST,
// ROLL_SECOND (row 0 = winners, rows 1.. = second stage, kept in P.cand).
constexpr int RP = ILQG_ROLLOUT_PARTS, RT = ILQG_ROLLOUT_TERMS;
constexpr int RPX = (NX + 2 + 1) / 2 * 2, RPU = (NU + 2 + 1) / 2 * 2, RPT = (RT + 2 + 1) / 2 * 2;  // LDS rows per lane, padded
// wavefronts per workgroup: one per part up to 8 (512 threads leave a wavefront 256 registers: the phase 1 operands
// of the next step and the state are ~100 of them; with 16 wavefronts, 128 registers each, 164 were spilled), else two
// parts per wavefront
constexpr int RW = RP > 8 ? (RP + 1) / 2 : RP;
constexpr int PPW = (RP + RW - 1) / RW;  // parts per wavefront
constexpr int JPW = (NU + RW - 1) / RW;  // inputs per wavefront in phase 1
// Workgroup barrier for data handed over in LDS only: __syncthreads() also waits for every global load and store in
// flight (vmcnt(0)), i.e. for the operands prefetched for the next step and for the roll-out's stores — twice per step
// (measured: 14 us per step with it).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ... and the same behind the step's records on their way into LDS (DMA below)
__device__ __forceinline__ void lds_barrier_after_loads() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// DMA: the nominal records of the workgroup's 64 trajectories reach the wavefronts through LDS.  With per-lane loads every
// wavefront pulled its operands of phase 1 (x_nom, one row of L, u_nom, l: 34 loads of 8 bytes, lane = trajectory, i.e.
// 64 different cache lines per load instruction) through the CU's vector-memory path: 17 000 line requests per step
// and workgroup for 1 280 distinct lines — that path, not the arithmetic, set the 10 us a step took.  Now a record
// (RN doubles, contiguous) is fetched ONCE per step by global_load_lds_dwordx4, 64 consecutive 16-byte pieces per
// instruction (lanes on consecutive addresses), into s_nom[trajectory][RN + 2] (the two doubles of padding keep a
// record 16-byte aligned and put the lanes' reads of one entry on different banks); wavefront w fetches the records of
// trajectories w, w + RW, ...  ONE buffer: a step's record is consumed in phase 1, so the next step's is requested
// behind the first barrier and has landed at the second (which waits for it).
constexpr int RNP = RN + 2;
constexpr int DMA_PIECES = RN / 2, DMA_PER_REC = (DMA_PIECES + WAVE - 1) / WAVE;
template <bool DMA>
__global__ __launch_bounds__(WAVE *RW) void k_rollout_parts(DevPtrs P, ilqg_dev_opts_t O, ParamValues A, int mode, int a0) {
    extern __shared__ __attribute__((aligned(16))) double s_nom[];  // DMA: [WAVE][RNP]
    __shared__ __attribute__((aligned(16))) double s_x[WAVE][RPX];
    __shared__ __attribute__((aligned(16))) double s_u[WAVE][RPU];
    __shared__ __attribute__((aligned(16))) double s_t[WAVE][RPT];
    __shared__ int s_bad[RW][WAVE];
    const int part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    int b = blockIdx.x * WAVE + lane;
    int ai = a0 + blockIdx.y;
    double *keep = nullptr;
    int second_row = -1;
    bool live = true;
    if(mode == ROLL_SECOND) {
        if(blockIdx.y == 0) {
            mode = ROLL_WINNER;
        } else {
            mode = ROLL_SEARCH_LIST;
            ai = a0 + blockIdx.y - 1;
            second_row = blockIdx.y - 1;
        }
    } else if(mode == ROLL_LIST_KEEP) {
        mode = ROLL_SEARCH_LIST;
        second_row = blockIdx.y;
    } else if(mode == ROLL_SEARCH && P.cand1) {  // the first stage keeps what it rolls out, by trajectory
        keep = P.cand1 + (size_t)blockIdx.y * (P.N + 1) * CAND_W * P.Bp + b;
    }
    if(mode == ROLL_SEARCH_LIST) {
        // (the rows start their walk over the list at different workgroups, see k_rollout)
        const int nb = gridDim.x, rows = (int)gridDim.y - ((second_row >= 0 && second_row != (int)blockIdx.y) ? 1 : 0);
        const int row = second_row >= 0 ? second_row : (int)blockIdx.y;
        const int first = (int)(((long long)row * nb) / rows);
        const int e0 = ((int)blockIdx.x + first) % nb * WAVE, np = *P.n_pending;
        if(e0 >= np) return;  // the whole workgroup
        const int e = e0 + lane;
        live = e < np;
        if(second_row >= 0) keep = P.cand + (size_t)second_row * (P.N + 1) * CAND_W * P.Bp + e;
        b = P.pending[live ? e : e0];
    }
    if(b >= P.B) {
        live = false;
        b = P.B - 1;
    }
    const int N = P.N;
    double alpha = 0.0;
    if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE) live = false;
    if(mode == ROLL_WINNER) {
        if(!P.i[ILQG_I_ACCEPTED][b]) live = false;
        const int idx = P.i[ILQG_I_ALPHA_IDX][b] - 1;
        alpha = O.alpha[(live && idx >= 0 && idx < ILQG_MAX_ALPHA) ? idx : 0];
    } else {
        alpha = O.alpha[ai];
    }
    if(__builtin_amdgcn_ballot_w64(live) == 0ull) return;  // (the same lanes in every wavefront: the whole workgroup)
    const bool store = (mode == ROLL_WINNER) && live;
    const bool feedback = (alpha != 0.0);

    ILQG_CALLBACKS(C, H);
    const double *rec = nomp(P, 0, b);  // this trajectory's record of the current step
    double x[NX];
#pragma unroll
    for(int i = 0; i < NX; i++) x[i] = rec[NOM_X + i];
    // DMA: the records of this wavefront's trajectories (held as uniform addresses; they move on one step per request)
    constexpr int TPW = (WAVE + RW - 1) / RW;  // trajectories per wavefront
    const char *drec[DMA ? TPW : 1];
    if(DMA) {
#pragma unroll
        for(int q = 0; q < TPW; q++) {
            const int t = part + q * RW;  // (the same lane -> trajectory map in every wavefront)
            const unsigned long long a = (unsigned long long)rec;
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)a, t < WAVE ? t : 0), hi = __builtin_amdgcn_readlane((unsigned)(a >> 32), t < WAVE ? t : 0);
            drec[q] = (const char *)(((unsigned long long)hi << 32) | lo);
        }
    }
    auto dma_issue = [&]() {
#pragma unroll
        for(int q = 0; q < TPW; q++) {
            if(part + q * RW < WAVE) {
#pragma unroll
                for(int m = 0; m < DMA_PER_REC; m++) {
                    if(m * WAVE + lane < DMA_PIECES)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(drec[q] + (size_t)(m * WAVE + lane) * 16),
                                                         (__attribute__((address_space(3))) void *)(s_nom + (size_t)(part + q * RW) * RNP + m * WAVE * 2),
                                                         16, 0, 0);
                }
            }
            drec[q] += (size_t)RN * sizeof(double);
        }
    };
    // phase 1 operands of this wavefront's input(s) part, part + RW, ...: one step ahead in registers, or out of LDS
    double nx[NX], nk[JPW][NX], nu_nom[JPW], nl[JPW];
    auto fetch = [&](const double *r) {
        if(part < NU) {
#pragma unroll
            for(int i = 0; i < NX; i++) nx[i] = r[NOM_X + i];
#pragma unroll
            for(int q = 0; q < JPW; q++) {
                const int ju = (part + q * RW < NU) ? part + q * RW : part;
#pragma unroll
                for(int i = 0; i < NX; i++) nk[q][i] = r[NOM_K + ju + i * NU];
                nu_nom[q] = r[NOM_U + ju];
                nl[q] = r[NOM_L + ju];
            }
        }
    };
    if(DMA) {
        dma_issue();
        lds_barrier_after_loads();
    } else {
        fetch(rec);
    }
    double csum = 0.0;
    int bad = 0;
    double *xo = cur_x(P, 0, b), *uo = cur_u(P, 0, b);
    drain_memory_ops();
    for(int k = 0; k < N; k++) {
        // ---- phase 1: this wavefront's input
        if(DMA) fetch(s_nom + (size_t)lane * RNP);  // this step's record, out of LDS
        if(part < NU) {
#pragma unroll
            for(int q = 0; q < JPW; q++) {
                if(part + q * RW < NU) {
                    double uf = nu_nom[q] + nl[q] * alpha;
#pragma unroll
                    for(int i = 0; i < NX; i++) uf += nk[q][i] * (x[i] - nx[i]);
                    s_u[lane][part + q * RW] = feedback ? uf : nu_nom[q];
                }
            }
        }
        rec += RN;
        if(!DMA) fetch(rec);  // step k+1 (the records have a step N: its gains are not used)
        lds_barrier();
        if(DMA) dma_issue();  // step k+1 into the buffer every wavefront has just finished reading
        // ---- phase 2: all inputs, the box, this wavefront's part of the step
        double u[NU];
#pragma unroll
        for(int j = 0; j < NU; j++) u[j] = s_u[lane][j];
        {
            double xb[NX];  // clampU reads t->x, the first member of the element (iLQG_problem.tem:24)
#pragma unroll
            for(int i = 0; i < NX; i++) xb[i] = x[i];
            clampU(u, reinterpret_cast<trajEl_t *>(xb), k, C.o.p, N);
        }
        int bad_step = 0;
        // this wavefront's part(s) of the step.  (sin / cos stay calls in the large generated files, which handle their
        // huge arguments themselves; small files go through the hooks as everywhere else)
        auto parts = [&]() {
#pragma unroll  // (the wavefront's number is known to be below RW: every call site keeps the cases it can reach)
            for(int q = 0; q < PPW; q++)
                if(part + q * RW < RP) ilqg_step_part(part + q * RW, &s_x[lane][0], &s_t[lane][0], &bad_step, x, u, k, C.o.p, N);
        };
#if ILQG_UNIFORM_GUARDS
        parts();
#else
        H.huge = 0.0;
        parts();
        if(H.huge != 0.0) {  // an argument beyond the fast sin/cos reduction: once more through the library
            H.slow = 1.0;
            bad_step = 0;
            parts();
            H.slow = 0.0;
        }
#endif
        bad |= bad_step;
        // what the roll-out stores of step k: by the last wavefront (it holds x_k and the clamped u_k like all others)
        if(part == RW - 1) {
            if(store) {
#pragma unroll
                for(int i = 0; i < NX; i++) xo[i] = x[i];
#pragma unroll
                for(int i = 0; i < NU; i++) uo[i] = u[i];
            } else if(keep && live) {
#pragma unroll
                for(int i = 0; i < NX; i++) keep[(size_t)i * P.Bp] = x[i];
#pragma unroll
                for(int i = 0; i < NU; i++) keep[(size_t)(NX + i) * P.Bp] = u[i];
            }
        }
        xo += RN;
        uo += RN;
        if(keep) keep += (size_t)CAND_W * P.Bp;
        if(DMA)
            lds_barrier_after_loads();
        else
            lds_barrier();
        // ---- the new state; the cost of the step in the order of ddpL's sum
#pragma unroll
        for(int i = 0; i < NX; i++) x[i] = s_x[lane][i];
        if(part == 0) {
            double c = s_t[lane][0];
#pragma unroll
            for(int m = 1; m < RT; m++) c = c + s_t[lane][m];
            if(!(c - c == 0.0)) bad = 1;  // ddpL's guard on t->c (NaN or Inf: forward_pass returns 0, iLQG_func.tem:175)
            csum += c;
        }
        // (the next step's phase 1 writes s_u, read before this barrier pair's second barrier by everybody; its
        // phase 2 writes s_x / s_t only behind the next first barrier, which every wavefront reaches after these reads)
    }
    s_bad[part][lane] = bad;
    __syncthreads();
    if(part != 0) return;
    int okc = 1;
#pragma unroll
    for(int q = 0; q < RW; q++) okc &= (s_bad[q][lane] == 0);
    {   // final cost (iLQG_func.tem:179-182)
        trajFin_t cf;
        init_final(&cf, &C.o);
        auto fin = [&]() {
#pragma unroll
            for(int i = 0; i < NX; i++) cf.x[i] = x[i];
            int r = calcFVariableAux(&cf, nullptr, &C.o);
            r &= ddpF(&cf, &C.o);
            return r;
        };
        int r = 1;
#if ILQG_UNIFORM_GUARDS
        if(okc) r = run_guarded(fin);
#else
        const double nf0 = H.nonfinite;
        H.huge = 0.0;
        r = fin();
        if(H.huge != 0.0) {
            H.nonfinite = nf0;
            H.slow = 1.0;
            r = fin();
            H.slow = 0.0;
        }
#endif
        okc &= r;
        csum += cf.c;
        if(store) {
#pragma unroll
            for(int i = 0; i < NX; i++) xo[i] = cf.x[i];
        } else if(keep && live) {
#pragma unroll
            for(int i = 0; i < NX; i++) keep[(size_t)i * P.Bp] = cf.x[i];
        }
    }
    const int ok = (okc && H.nonfinite == 0.0) ? 1 : 0;
    if(!live) return;
    if(mode == ROLL_WINNER) {
        P.f[ILQG_F_NEW_COST][b] = csum;
    } else {
        P.f[ILQG_F_ALPHA_COST][tile_ix(ILQG_MAX_ALPHA, ai, b)] = csum;
        P.i[ILQG_I_ALPHA_OK][(size_t)ai * P.Bp + b] = ok;
    }
}
#endif  // ILQG_WAVE_MAP && ILQG_ROLLOUT_PARTS

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
        cnew = P.f[ILQG_F_ALPHA_COST][tile_ix(ILQG_MAX_ALPHA, i, b)];
        if(!ok) continue;
        dcost = cost - cnew;
        expected = -a * (dV0 + a * dV1);
        const double z = (expected > 0) ? dcost / expected : 0.0;
        if(z > O.zMin) break;
        ok = 0;
    }
    // To the second stage: one atomicAdd per wavefront, its trajectories in order behind each other (the entries of
    // one wavefront are trajectories of one 64-trajectory tile: what k_adopt copies for them lands in the same rows of
    // X and U).  No measurable gain over one atomicAdd per trajectory; kept for the 64x fewer atomics.
    if(!ok && a1 < O.n_alpha) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(true);  // the lanes in here
        const int rank = __builtin_popcountll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
        int base = 0;
        if(rank == 0) base = atomicAdd(P.n_pending_next, __builtin_popcountll(m));
        base = __builtin_amdgcn_readfirstlane(base);  // the first lane in here is the one of rank 0
        P.pending[base + rank] = b;
    }
    P.i[ILQG_I_ALPHA_IDX][b] = i + 1;
    P.i[ILQG_I_ACCEPTED][b] = ok;
    P.f[ILQG_F_NEW_COST][b] = cnew;
    P.f[ILQG_F_DCOST][b] = dcost;
    P.f[ILQG_F_EXPECTED][b] = expected;
}

// After the second stage's selection: the trajectory of the accepted step size, kept by the lane that rolled it out
// (ROLL_SECOND), becomes the current one.  One thread per (entry of P.pending, step).  n2 = step sizes of the stage,
// a0 = its first.
__global__ void k_adopt(DevPtrs P, int a0, int n2) {
    const size_t n = (size_t)*P.n_pending, total = n * (P.N + 1), stride = (size_t)gridDim.x * blockDim.x;
    // a fixed grid walks over (step, entry), entry fastest: consecutive threads read consecutive entries of one row
    for(size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += stride) {
        const int e = (int)(w % n), k = (int)(w / n);
        const int b = P.pending[e];
        if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE || !P.i[ILQG_I_ACCEPTED][b]) continue;
        const int a = P.i[ILQG_I_ALPHA_IDX][b] - 1 - a0;
        const double *src = P.cand + ((size_t)a * (P.N + 1) + k) * CAND_W * P.Bp + e;
        double *xo = cur_x(P, k, b);
#pragma unroll
        for(int i = 0; i < NX; i++) xo[i * XSI] = src[(size_t)i * P.Bp];
        if(k < P.N) {
            double *uo = cur_u(P, k, b);
#pragma unroll
            for(int i = 0; i < NU; i++) uo[i * XSI] = src[(size_t)(NX + i) * P.Bp];
        }
    }
}

// Wave mapping, ls_keep = 2: the roll-outs the first stage accepted, kept by trajectory in P.cand1, become the current
// trajectory (the records' x and u).  Replaces the winner pass: a copy at the speed of the memory system instead of
// 16 384 chains of N steps beside the second stage's.  One wavefront per (step, tile of 64 trajectories): it reads the
// step's NX + NU components lane = trajectory (512 contiguous bytes per load), turns the tile round in LDS and writes
// lane = (trajectory, component): each trajectory's x | u of the step is one run of 192 contiguous bytes in its record
// (with one thread per (step, trajectory) writing its own run: 64 cache lines per store instruction, 5.1 ms for config 5).
__global__ __launch_bounds__(256) void k_adopt_first(DevPtrs P, int s1) {
    constexpr int TW = CAND_W + 1;  // (padded rows: the lanes' stores fall on different banks)
    __shared__ double tile[4][WAVE][TW];
    __shared__ int take[4][WAVE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tiles = P.Bp / WAVE;
    const size_t total = (size_t)(P.N + 1) * tiles, stride = (size_t)gridDim.x * 4;
    for(size_t w = (size_t)blockIdx.x * 4 + wave; w < total; w += stride) {
        const int k = (int)(w / tiles), b0 = (int)(w % tiles) * WAVE, b = b0 + lane;
        int a = -1;
        if(b < P.B && P.i[ILQG_I_STATUS][b] == ILQG_ST_ACTIVE && P.i[ILQG_I_ACCEPTED][b]) a = P.i[ILQG_I_ALPHA_IDX][b] - 1;
        if(a >= s1) a = -1;
        if(__builtin_amdgcn_ballot_w64(a >= 0) == 0ull) continue;
        take[wave][lane] = a >= 0;
        if(a >= 0) {
            const double *src = P.cand1 + ((size_t)a * (P.N + 1) + k) * CAND_W * P.Bp + b;
#pragma unroll
            for(int i = 0; i < CAND_W; i++) tile[wave][lane][i] = src[(size_t)i * P.Bp];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int W = (k < P.N) ? CAND_W : NX;  // (the last step has a state only)
        for(int e = lane; e < WAVE * CAND_W; e += WAVE) {
            const int t = e / CAND_W, c = e - t * CAND_W;
            if(c < W && take[wave][t]) (c < NX ? cur_x(P, k, b0 + t) + c * XSI : cur_u(P, k, b0 + t) + (c - NX) * XSI)[0] = tile[wave][t][c];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// iLQG.c:311-361 and the loop bookkeeping of iLQG.c:239,365-378
// commit_s1 >= 0: also what k_commit does (the search of this iteration left the change of location to this kernel);
// reset_pending: the counter of the pending list is cleared for the next search (nothing reads it any more)
__global__ void k_update(DevPtrs P, ilqg_dev_opts_t O, int commit_s1, int commit_set, int reset_pending) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if(b == 0 && reset_pending) *P.n_pending = 0;
    if(b >= P.B) return;
    if(!WAVE_MAP && commit_s1 >= 0 && P.i[ILQG_I_STATUS][b] == ILQG_ST_ACTIVE) {
        const int a = P.i[ILQG_I_ALPHA_IDX][b] - 1;
        P.i[ILQG_I_LOC][b] = (P.i[ILQG_I_ACCEPTED][b] && a < commit_s1) ? 1 + commit_set * P.plane_n + a : 0;
    }
    if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE) {  // finished, possibly in this iteration's backward pass
        P.i[ILQG_I_RESWEEP][b] = 0;
        return;
    }
    double lambda = P.f[ILQG_F_LAMBDA][b], dlambda = P.f[ILQG_F_DLAMBDA][b];
    int iter = P.i[ILQG_I_ITER][b];
    int status = ILQG_ST_ACTIVE;
    int resweep = 0;
    if(P.i[ILQG_I_ACCEPTED][b]) {
        const double t1 = dlambda / O.lambdaFactor, t2 = 1.0 / O.lambdaFactor;
        dlambda = (t1 < t2) ? t1 : t2;
        lambda = lambda * dlambda * (lambda > O.lambdaMin);
        P.f[ILQG_F_COST][b] = P.f[ILQG_F_NEW_COST][b];
        P.i[ILQG_I_NEED_DERIVS][b] = 1;
        if(P.f[ILQG_F_DCOST][b] < O.tolFun) status = ILQG_ST_CONVERGED_FUN;
        else resweep = 2;  // update_multipliers + cost-only sweep (iLQG.c:337-338)
    } else {
        if(HAS_MUL && O.w_pen_fact2 > 1.0) {  // iLQG.c:345-349, before the lambdaMax exit
            const double wl = P.f[ILQG_F_WPEN_L][b] * O.w_pen_fact2, wf = P.f[ILQG_F_WPEN_F][b] * O.w_pen_fact2;
            P.f[ILQG_F_WPEN_L][b] = (O.w_pen_max_l < wl) ? O.w_pen_max_l : wl;
            P.f[ILQG_F_WPEN_F][b] = (O.w_pen_max_f < wf) ? O.w_pen_max_f : wf;
            resweep = 1;
        }
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
    // without multipliers the sweep reproduces the cost bit for bit: only where the solve goes on (option resweep)
    if(!HAS_MUL && status != ILQG_ST_ACTIVE) resweep = 0;
    P.i[ILQG_I_RESWEEP][b] = resweep;
}

// update_multipliers (iLQG_func.tem:419-521) for the trajectories k_update flagged, or with init != 0 at the solver
// entry (iLQG.c:236) for all live ones.  The generated functions walk o->nominal->t and o->multipliers.t over
// o->n_hor elements; here they see a one-element view per time step (the constraint values they read are the
// auxiliaries of that element, recomputed from the stored (x, u) with the multipliers and weights of the roll-out
// that stored them).  What they do across steps is restated around the calls: one raise of w_pen_l after the
// walk if any step asked for it; with init != 0 the running part returns inside its loop after the first element
// (iLQG_func.tem:447), so only step 0 is visited.
__global__ __launch_bounds__(WAVE) void k_multipliers(DevPtrs P, ilqg_dev_opts_t O, ParamValues A, int init) {
    const int b = blockIdx.x * WAVE + threadIdx.x;
    if(b >= P.B) return;
    if(init) {
        if(P.i[ILQG_I_STATUS][b] != ILQG_ST_ACTIVE && P.i[ILQG_I_STATUS][b] != ILQG_ST_MAX_ITER) return;
    } else if(P.i[ILQG_I_RESWEEP][b] != 2) {
        return;
    }
    const int N = P.N;
    ILQG_CALLBACKS(C, H);
    load_penalty_weights(C, P, b);
    const double wl = C.o.w_pen_l;
    traj_t view;
    trajEl_t ct;
    multipliersEl_t mk;
    init_running(&ct, &C.o1);
    view.t = &ct;
    C.o.nominal = C.o1.nominal = &view;
    C.o1.multipliers.t = &mk;
    bool raise = false;
    if(ME > 0) {
        const int steps = init ? 1 : N;
        for(int k = 0; k < steps; k++) {
            const double *xs = cur_x(P, k, b), *us = cur_u(P, k, b);
            load_mul(P, k, b, mk);
            H.huge = 0.0;
            auto aux = [&]() {
#pragma unroll
                for(int i = 0; i < NX; i++) ct.x[i] = xs[i * XSI];
#pragma unroll
                for(int i = 0; i < NU; i++) ct.u[i] = us[i * XSI];
                calcXVariableAux(&ct, &mk, k, &C.o);
                calcXUVariableAux(&ct, &mk, k, &C.o);
            };
            aux();
            if(H.huge != 0.0) {
                H.slow = 1.0;
                aux();
                H.slow = 0.0;
            }
            C.o1.w_pen_l = wl;
            update_multipliers_running(&C.o1, init);
            raise |= (C.o1.w_pen_l != wl);
            store_mul(P, k, b, mk);
        }
    }
    if(MF > 0) {
        const double *xs = cur_x(P, N, b);
        load_mul_fin(P, b, C.o.multipliers.f);
        init_final(&view.f, &C.o);
        H.huge = 0.0;
        auto aux = [&]() {
#pragma unroll
            for(int i = 0; i < NX; i++) view.f.x[i] = xs[i * XSI];
            calcFVariableAux(&view.f, &C.o.multipliers.f, &C.o);
        };
        aux();
        if(H.huge != 0.0) {
            H.slow = 1.0;
            aux();
            H.slow = 0.0;
        }
        update_multipliers_final(&C.o, init);
        store_mul_fin(P, b, C.o.multipliers.f);
        P.f[ILQG_F_WPEN_F][b] = C.o.w_pen_f;
    }
    double wl_new = wl;
    if(raise) {
        const double w = wl * O.w_pen_fact1;
        wl_new = (O.w_pen_max_l < w) ? O.w_pen_max_l : w;
        P.f[ILQG_F_WPEN_L][b] = wl_new;
    }
    // the derivatives evaluated next (this trajectory was accepted, or the solve starts) see these weights
    P.f[ILQG_F_WPEN_L_DER][b] = wl_new;
    P.f[ILQG_F_WPEN_F_DER][b] = (MF > 0) ? C.o.w_pen_f : P.f[ILQG_F_WPEN_F][b];
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
    P.i[ILQG_I_RESWEEP][b] = 0;
    P.f[ILQG_F_WPEN_L][b] = O.w_pen_init_l;
    P.f[ILQG_F_WPEN_F][b] = O.w_pen_init_f;
    P.f[ILQG_F_WPEN_L_DER][b] = O.w_pen_init_l;
    P.f[ILQG_F_WPEN_F_DER][b] = O.w_pen_init_f;
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

// unit-test kernel for box_qp<M>, one problem per lane (TABLE: the pattern-table form, see chol_pattern_table)
template <int M, bool TABLE = false>
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
    rc[t] = box_qp<M, TABLE>(h, gg, lo, up, xx, cl, nf, inv);
    n_free[t] = nf;
#pragma unroll
    for(int i = 0; i < M; i++) {
        x[(size_t)t * M + i] = xx[i];
        clamp[t * M + i] = cl[i];
    }
#pragma unroll
    for(int i = 0; i < T; i++) invH[(size_t)t * T + i] = inv[i];
}

// unit-test kernel for box_qp_rows<M> (wave mapping): one problem per wavefront
template <int M>
__global__ __launch_bounds__(64) void k_boxqp_rows_test(int count, const double *H, const double *g, const double *lower,
                                                        const double *upper, double *x, int *clamp, int *n_free, double *invH,
                                                        int *rc) {
    constexpr int T = tri(M);
    __shared__ double sH[T], sl[M], sinv[T];
    __shared__ int scl[M];
    const int t = blockIdx.x, lane = threadIdx.x;
    if(t >= count) return;
    for(int i = lane; i < T; i += 64) sH[i] = H[(size_t)t * T + i];
    if(lane < M) sl[lane] = x[(size_t)t * M + lane];
    wave_sync();
    const int me = (M <= 16) ? (lane & 15) % M : lane % M;
    int nf, r;
    if constexpr(M <= 16)  // the form the row-mapped backward step uses
        r = box_qp_row<M>(sH, g[(size_t)t * M + me], lower[(size_t)t * M + me], upper[(size_t)t * M + me], sl, scl, sinv, nf);
    else
        r = box_qp_rows<M>(sH, g[(size_t)t * M + me], lower[(size_t)t * M + me], upper[(size_t)t * M + me], sl, scl, sinv, nf);
    wave_sync();
    if(lane == 0) {
        rc[t] = r;
        n_free[t] = nf;
    }
    if(lane < M) {
        x[(size_t)t * M + lane] = sl[lane];
        clamp[t * M + lane] = scl[lane];
    }
    for(int i = lane; i < T; i += 64) invH[(size_t)t * T + i] = sinv[i];
}

}  // namespace

// ===========================================================================
// shim
// ===========================================================================
#if ILQG_WAVE_MAP
// Wave mapping: the derivative records of a chunk of trajectories (trajEl_t structs, 48 KB each with the tensors of the
// n = 16 problem) live in ONE work buffer per device, shared by all solver contexts on it.  A context owns the buffer
// from the start of a derivatives + backward pass to its end; the hand-over is an event on the streams, so a second
// context's pass starts when the first one's has finished — which is also the schedule that pays: the backward pass
// fills the chip, the roll-outs (one lane per trajectory and step size) cannot, so with the batch advancing as two
// groups of trajectories the roll-outs of one run beside the backward pass of the other.
struct SharedWork {
    trajEl_t *buf;
    size_t bytes;
    hipEvent_t free_ev;  // recorded when the current owner is done with the buffer
    int refs;
    // which constant record entries (init_running) the buffer holds: for whom, and where
    bool whole, half[2], factored, pv_set;
    int N, B, part, half_cap;
    ParamValues pv;
};
static SharedWork g_work[64];
#else
struct SharedWork;
#endif

constexpr int QUEUE_CELL = 32;  // ints between two counters: a cache line each

struct ilqg_dev {
    int device, B, Bp, N;
    hipStream_t stream;
    DevPtrs P;
    ilqg_dev_opts_t O;
    std::vector<double *> param_bufs;
    ParamValues pv;       // fixed-size parameters, passed to the kernels by value
    double *staging;
    size_t staging_bytes;
    // Deferred transfers (ilqg_dev_io_begin .. ilqg_dev_io_end): every write / read takes its own slice of the device
    // staging buffer and of a pinned host buffer and nothing waits; the stream is synchronised once, at the end, and
    // the reads are handed to the caller then.  (The drop-in back_pass() / line_search() move two dozen small arrays
    // per call: one wait instead of one per array.)
    bool io_deferred;
    size_t io_off;
    char *pinned;
    size_t pinned_bytes;
    struct PendingRead { void *dst; const void *src; size_t bytes; int transpose_w; };
    std::vector<PendingRead> pending;
    int *counter;
    size_t cand_bytes;    // size of P.cand
    size_t cand1_bytes;   // size of P.cand1
    bool keep_first;      // the roll-out launch in progress keeps the first stage's roll-outs in P.cand1
    size_t xpl_bytes, upl_bytes;  // sizes of P.xpl / P.upl (ls_keep = 2)
    size_t roll_lds;              // wave mapping: dynamic LDS asked for by the second-stage roll-outs (one workgroup per CU)
    size_t roll_pad;              //   ... on top of the records' buffer when the records go through LDS
    bool roll_dma;                // wave mapping: k_rollout_parts fetches the nominal records through LDS
    int loc_set;          // the set of planes current trajectories may live in (-1: none, all in X / U)
    bool defer_commit, commit_pending, pending_zero;  // ls_keep = 2: k_update commits / clears the pending counter
    int commit_s1, commit_set;
    bool winner_done;     // the last search ended with the accepted trajectories in place (two-stage search)
    int *queues;          // wave mapping: the backward kernel's trajectory counters (DevPtrs::queue), QUEUE_CELL ints apart
    int cus;              // compute units of the device
    int chunk;            // wave mapping: trajectories whose derivative records fit the work buffer
    bool work_consts;     // wave mapping: constant entries of the records written (init_running)
    bool work_factored;   // wave mapping: the records in the work buffer are factored ones (see FACTORED)
    bool half_consts[2];  // wave mapping: the same per half of the work buffer (chunks alternate between the halves)
    hipStream_t stream2;  // wave mapping: second stream of the chunk pipeline, fork / join events
    hipEvent_t fork, join;
    // Wave mapping: the roll-outs run through the generated callbacks with a trajEl_t per lane in scratch memory
    // (48 KB per lane for the n = 16 problem with its tensors) and the runtime reserves scratch per queue for every
    // wavefront the chip can hold — one queue works, a second one runs out of resources.  All roll-outs of all
    // contexts (groups) of a device therefore share ONE stream, tied to each context's own stream by events; the
    // backward passes of the other groups overlap with them.
    hipStream_t roll;
    hipEvent_t roll_in, roll_out;
    struct SharedWork *shared;  // wave mapping: the device's derivative work buffer (see SharedWork)
    int own_chunk;              // trajectories whose records fit the context's private buffer P.work (stage-by-stage calls)
    bool per_step_params;       // some problem parameter has one value per time step
    bool timing;
    struct Span { int kernel; hipEvent_t a, b; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;  // events of drained spans, reused: none is created while a window is timed
    double t_ms[ILQG_K_COUNT];
    int t_n[ILQG_K_COUNT];
    // launches of one kernel on the context's two streams overlap in the event clock (the later one waits for the chip):
    // t_busy is the length of the UNION of a kernel's launch intervals, measured against an event recorded when timing
    // was switched on
    hipEvent_t epoch;
    double t_busy[ILQG_K_COUNT];
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
        case ILQG_F_MUL: return {0, MEW, MEW};
        case ILQG_F_MULF: return {-1, MFW, MFW};
        case ILQG_F_ALPHA_COST: return {-1, ILQG_MAX_ALPHA, ILQG_MAX_ALPHA};
        default: return {-1, 1, 1};
    }
}

// columns of the packed trajectory records (nomp): first column, or -1 for any other field
int nom_column(int field) {
    switch(field) {
        case ILQG_F_X: return NOM_X;
        case ILQG_F_U: return NOM_U;
        case ILQG_F_LG: return NOM_L;
        case ILQG_F_KG: return NOM_K;
        default: return -1;
    }
}
// lane mapping: X and U also exist as tiled arrays, the representation the host reads (see cur_x)
bool has_tiled_copy(int field) { return !WAVE_MAP && (field == ILQG_F_X || field == ILQG_F_U); }
// lane mapping: tiled l / L are the output of the backward pass over stored records only (see k_pack_records);
// the host reads and writes gains in the records
bool has_tiled_scratch(int field) { return !WAVE_MAP && (field == ILQG_F_LG || field == ILQG_F_KG); }
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

// end of a deferred batch (or of a part of it): wait once, deliver the reads
int io_flush(ilqg_dev *d) {
    HIP_TRY(hipStreamSynchronize(d->stream));
    for(const auto &r : d->pending) {
        if(r.transpose_w) {  // int field: device [width][Bp] -> host [B][width]
            const int *src = (const int *)r.src;
            int *dst = (int *)r.dst;
            for(int b = 0; b < d->B; b++)
                for(int j = 0; j < r.transpose_w; j++) dst[(size_t)b * r.transpose_w + j] = src[(size_t)j * d->Bp + b];
        } else {
            memcpy(r.dst, r.src, r.bytes);
        }
    }
    d->pending.clear();
    d->io_off = 0;
    return 0;
}

// Staging for one transfer of `bytes`: *dev in the device staging buffer and, while transfers are deferred, *pin in the
// pinned host buffer (nullptr otherwise: the transfer then uses the caller's memory and is waited for at once).
int stage(ilqg_dev *d, size_t bytes, void **dev, void **pin) {
    if(!d->io_deferred) {
        if(ensure_staging(d, bytes)) return 1;
        *dev = d->staging;
        *pin = nullptr;
        return 0;
    }
    bytes = (bytes + 255) & ~(size_t)255;
    if(d->io_off + bytes > d->staging_bytes || d->io_off + bytes > d->pinned_bytes) {
        if(io_flush(d)) return 1;  // what is in flight uses the old buffers
        const size_t want = (bytes > (1u << 20) ? bytes : (1u << 20)) + 2 * d->pinned_bytes;
        if(ensure_staging(d, want)) return 1;
        if(d->pinned) HIP_TRY(hipHostFree(d->pinned));
        d->pinned = nullptr;
        d->pinned_bytes = 0;
        HIP_TRY(hipHostMalloc((void **)&d->pinned, want, hipHostMallocDefault));
        d->pinned_bytes = want;
    }
    *dev = (char *)d->staging + d->io_off;
    *pin = d->pinned + d->io_off;
    d->io_off += bytes;
    return 0;
}
// host -> staging slice / staging slice -> host, according to the mode
int stage_in(ilqg_dev *d, void *dev, void *pin, const void *host, size_t bytes) {
    if(pin) {
        memcpy(pin, host, bytes);
        host = pin;
    }
    HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, d->stream));
    return 0;
}
int stage_out(ilqg_dev *d, const void *dev, void *pin, void *host, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(pin ? pin : host, dev, bytes, hipMemcpyDeviceToHost, d->stream));
    if(pin) d->pending.push_back({host, pin, bytes, 0});
    else HIP_TRY(hipStreamSynchronize(d->stream));
    return 0;
}
int io_done(ilqg_dev *d) {  // end of a single write
    if(!d->io_deferred) HIP_TRY(hipStreamSynchronize(d->stream));
    return 0;
}

struct Timed {
    ilqg_dev *d;
    int kernel;
    hipEvent_t a, b;
    hipStream_t st;
    static hipEvent_t take(ilqg_dev *d) {
        hipEvent_t e = nullptr;
        if(!d->event_pool.empty()) {
            e = d->event_pool.back();
            d->event_pool.pop_back();
        } else {
            hipEventCreate(&e);
        }
        return e;
    }
    Timed(ilqg_dev *d_, int k, hipStream_t stream = nullptr) : d(d_), kernel(k), a(nullptr), b(nullptr) {
        st = stream ? stream : d->stream;
        if(d->timing) {
            a = take(d);
            b = take(d);
            hipEventRecord(a, st);
        }
    }
    ~Timed() {
        if(d->timing) {
            hipEventRecord(b, st);
            d->spans.push_back({kernel, a, b});
        }
    }
};

int drain_spans(ilqg_dev *d) {
    if(d->spans.empty()) return 0;
    HIP_TRY(hipStreamSynchronize(d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream2));
    if(d->roll) HIP_TRY(hipStreamSynchronize(d->roll));
    std::vector<std::pair<float, float>> iv[ILQG_K_COUNT];
    for(auto &s : d->spans) {
        float ms = 0.f, t0 = 0.f;
        hipEventElapsedTime(&ms, s.a, s.b);
        d->t_ms[s.kernel] += ms;
        d->t_n[s.kernel]++;
        if(d->epoch && hipEventElapsedTime(&t0, d->epoch, s.a) == hipSuccess) iv[s.kernel].push_back({t0, t0 + ms});
        d->event_pool.push_back(s.a);
        d->event_pool.push_back(s.b);
    }
    (void)hipGetLastError();
    for(int k = 0; k < ILQG_K_COUNT; k++) {  // union of the intervals of this batch of spans
        std::sort(iv[k].begin(), iv[k].end());
        float end = -1.f;
        for(auto &q : iv[k]) {
            if(q.second <= end) continue;
            d->t_busy[k] += q.second - (q.first > end ? q.first : end);
            end = q.second;
        }
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

void ilqg_dev_multiplier_dims(int *out) {
    out[0] = ME;
    out[1] = MF;
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
                                              "layout kernels", "k_backward[fused derivs]", "k_rollout[stage 2 | winner]",
                                              "k_multipliers", "k_search[stage 1]", "k_search[stage 2]", "k_adopt_home + k_commit"};
    return (k >= 0 && k < ILQG_K_COUNT) ? names[k] : "?";
}

static int dev_fill(ilqg_dev *d, int device, int batch, int n_hor);
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
    ilqg_dev *d = new ilqg_dev();  // value-initialised: every pointer and handle is null until it is acquired
    if(dev_fill(d, device, batch, n_hor)) {
        const std::string why = g_err;
        ilqg_dev_destroy(d);  // releases whatever the failed attempt had acquired
        g_err = why;
        return 1;
    }
    *out = d;
    return 0;
}

static int dev_fill(ilqg_dev *d, int device, int batch, int n_hor) {
    d->device = device;
    d->B = batch;
    d->Bp = (batch + WAVE - 1) / WAVE * WAVE;
    d->N = n_hor;
    d->staging = nullptr;
    d->staging_bytes = 0;
    d->io_deferred = false;
    d->io_off = 0;
    d->pinned = nullptr;
    d->pinned_bytes = 0;
    d->timing = false;
    d->loc_set = -1;
    memset(d->t_ms, 0, sizeof(d->t_ms));
    memset(d->t_n, 0, sizeof(d->t_n));
    memset(d->t_busy, 0, sizeof(d->t_busy));
    memset(&d->P, 0, sizeof(d->P));
    memset(&d->O, 0, sizeof(d->O));
    memset(&d->pv, 0, sizeof(d->pv));
    d->P.B = d->B;
    d->P.Bp = d->Bp;
    d->P.N = d->N;
    HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
#if ILQG_WAVE_MAP && defined(ILQG_ROLLOUT_PARTS)
    {
        // measured (config 5, ~300 busy workgroups in the second stage): 32.9 / 40.3 ms without, 33.0 / 28.5 ms with;
        // with fewer than 256 busy workgroups every one of them gets a CU to itself (ILQG_ROLL_LDS_KB=0: off)
        const char *e = getenv("ILQG_ROLL_LDS_KB");
        d->roll_lds = (size_t)(e ? atoi(e) : 84) * 1024;
        if(d->roll_lds)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rollout_parts<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)d->roll_lds));
        // records through LDS (see k_rollout_parts; ILQG_NO_DMA=1: per-lane loads).  With them a workgroup takes more than
        // half of a CU's LDS whenever a record is 640 bytes or more; below that the same is asked for explicitly
        d->roll_dma = !getenv("ILQG_NO_DMA");
        const size_t need = (size_t)WAVE * RNP * sizeof(double);
        d->roll_pad = (d->roll_lds > need + 40 * 1024) ? d->roll_lds - need - 40 * 1024 : 0;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rollout_parts<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(need + d->roll_pad)));
    }
#endif
#if ILQG_WAVE_MAP
    if(FACTORED) {  // the workgroups that share the coefficient tables need more than the default 64 KB of LDS
        const int lds = (int)((TABLE_DOUBLES + ILQG_FACT_WAVES * WAVE_LDS_DOUBLES) * sizeof(double));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_backward_wave<FACTORED>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    if(QUAD_STEP) {
        const int lds = (int)((TABLE_DOUBLES + QUAD_WAVES * 4 * QRow::SIZE) * sizeof(double));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_backward_quad<FACTORED>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_backward_quad<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
#endif
    d->chunk = 0;
    d->own_chunk = 0;
    d->shared = nullptr;
    d->per_step_params = false;
    d->work_consts = false;
    d->work_factored = false;
    d->half_consts[0] = d->half_consts[1] = false;
    HIP_TRY(hipStreamCreateWithFlags(&d->stream2, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&d->fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&d->join, hipEventDisableTiming));
    if(WAVE_MAP) {
        static std::vector<hipStream_t> shared_roll(64, nullptr);  // one per device, for the life of the process
        if(!shared_roll[device]) HIP_TRY(hipStreamCreateWithFlags(&shared_roll[device], hipStreamNonBlocking));
        d->roll = shared_roll[device];
        HIP_TRY(hipEventCreateWithFlags(&d->roll_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&d->roll_out, hipEventDisableTiming));
    }
    for(int f = 0; f < ILQG_F_COUNT; f++) {
        if(WAVE_MAP && f == ILQG_F_DER) {
#if ILQG_WAVE_MAP
            // derivative records = device trajEl_t structs for as many trajectories as fit the device's work buffer
            // (allocated by the first context: ILQG_WORK_GB if set, else half of the free device memory, at most what
            // this context needs); a private buffer for stage-by-stage calls is allocated when one is made
            SharedWork &W = g_work[device];
            const size_t per_traj = (size_t)d->N * sizeof(trajEl_t);
            if(!W.buf) {
                const char *e = getenv("ILQG_WORK_GB");
                size_t free_b = 0, total_b = 0;
                HIP_TRY(hipMemGetInfo(&free_b, &total_b));
                // Without ILQG_WORK_GB: what the whole batch needs in the record form ilqg_dev_iterate uses, if the device
                // has it once this context's other arrays (the packed records, the second line-search stage's kept
                // roll-outs, and a margin) are counted; else what is left then, but at least half of what is free.
                const double need = (double)d->B * (double)d->N * (double)FACT_STRIDE;
                const double others = (double)d->Bp * (d->N + 1) * (RN + (ILQG_MAX_ALPHA - 1) * CAND_W) * sizeof(double) + 6e9;
                double budget = (double)free_b - others;
                if(budget < 0.5 * (double)free_b) budget = 0.5 * (double)free_b;
                if(budget > need) budget = need;
                if(e) budget = atof(e) * 1e9;
                size_t bytes = (size_t)budget;
                if(bytes < per_traj) bytes = per_traj;
                if(bytes > (size_t)d->B * per_traj) bytes = (size_t)d->B * per_traj;
                memset(&W, 0, sizeof(W));
                // (+ one record: records may lie closer together than their size, see FACT_STRIDE, and the last one
                // still reaches sizeof(trajEl_t) beyond its start)
                bytes += sizeof(trajEl_t);
                // (what hipMemGetInfo calls free is not always available in one piece: take less rather than fail)
                while(hipMalloc((void **)&W.buf, bytes) != hipSuccess) {
                    (void)hipGetLastError();
                    W.buf = nullptr;
                    if(bytes / 2 < per_traj + sizeof(trajEl_t)) {
                        g_err = "ilqg_dev_create: no device memory for the derivative work buffer";
                        return 1;
                    }
                    bytes = bytes / 2 + sizeof(trajEl_t);
                }
                W.bytes = bytes;
                HIP_TRY(hipMemsetAsync(W.buf, 0, W.bytes, d->stream));
                HIP_TRY(hipStreamSynchronize(d->stream));
                HIP_TRY(hipEventCreateWithFlags(&W.free_ev, hipEventDisableTiming));
            }
            W.refs++;
            d->shared = &W;
            size_t c = (W.bytes - sizeof(trajEl_t)) / per_traj;
            if(c < 1) {
                g_err = "ilqg_dev_create: the device's derivative work buffer (made for a shorter horizon) does not hold one trajectory";
                return 1;
            }
            d->chunk = (int)(c > (size_t)d->B ? (size_t)d->B : c);
#endif
            d->P.f[f] = nullptr;
            continue;
        }
        if(nom_column(f) >= 0 && !has_tiled_copy(f) && !has_tiled_scratch(f)) {  // columns of the packed records only
            d->P.f[f] = nullptr;
            continue;
        }
        const FieldInfo fi = field_info(f);
        const size_t bytes = (size_t)field_steps(d, f) * fi.wd * d->Bp * sizeof(double);
        HIP_TRY(hipMalloc((void **)&d->P.f[f], bytes));
        HIP_TRY(hipMemsetAsync(d->P.f[f], 0, bytes, d->stream));
    }
    {
        const size_t bytes = (size_t)d->Bp * (d->N + 1) * RN * sizeof(double);
        HIP_TRY(hipMalloc((void **)&d->P.nom, bytes));
        HIP_TRY(hipMemsetAsync(d->P.nom, 0, bytes, d->stream));
    }
    for(int f = 0; f < ILQG_I_COUNT; f++) {
        const size_t bytes = (size_t)int_field_width(f) * d->Bp * sizeof(int);
        HIP_TRY(hipMalloc((void **)&d->P.i[f], bytes));
        HIP_TRY(hipMemsetAsync(d->P.i[f], 0, bytes, d->stream));
    }
    HIP_TRY(hipMalloc((void **)&d->P.derivs_failed, d->Bp * sizeof(int)));
    HIP_TRY(hipMemsetAsync(d->P.derivs_failed, 0, d->Bp * sizeof(int), d->stream));
    HIP_TRY(hipMalloc((void **)&d->counter, sizeof(int)));
    HIP_TRY(hipMalloc((void **)&d->queues, 2 * QUEUE_CELL * sizeof(int)));  // one cell per stream of the chunk pipeline
    {
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        d->cus = cus > 0 ? cus : 256;
    }
    HIP_TRY(hipMalloc((void **)&d->P.pending, d->Bp * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&d->P.n_pending, sizeof(int)));
    HIP_TRY(hipMemsetAsync(d->P.n_pending, 0, sizeof(int), d->stream));
    d->P.n_pending_next = d->P.n_pending;
    d->P.p = nullptr;
    HIP_TRY(hipStreamSynchronize(d->stream));
    return 0;
}

void ilqg_dev_destroy(ilqg_dev_t *d) {
    if(!d) return;
    hipSetDevice(d->device);
    if(d->stream) hipStreamSynchronize(d->stream);
    if(d->stream2) hipStreamSynchronize(d->stream2);
    for(auto &s : d->spans) {
        hipEventDestroy(s.a);
        hipEventDestroy(s.b);
    }
    for(hipEvent_t e : d->event_pool) hipEventDestroy(e);
    for(int f = 0; f < ILQG_F_COUNT; f++)
        if(d->P.f[f]) hipFree(d->P.f[f]);
    if(d->P.work) hipFree(d->P.work);
#if ILQG_WAVE_MAP
    if(d->shared && --d->shared->refs == 0) {  // the last context on the device: release the shared work buffer
        hipFree(d->shared->buf);
        hipEventDestroy(d->shared->free_ev);
        memset(d->shared, 0, sizeof(SharedWork));
    }
#endif
    if(d->P.nom) hipFree(d->P.nom);
    for(int f = 0; f < ILQG_I_COUNT; f++)
        if(d->P.i[f]) hipFree(d->P.i[f]);
    if(d->P.derivs_failed) hipFree(d->P.derivs_failed);
    if(d->counter) hipFree(d->counter);
    if(d->queues) hipFree(d->queues);
    if(d->P.cand) hipFree(d->P.cand);
    if(d->P.cand1) hipFree(d->P.cand1);
    if(d->P.xpl) hipFree(d->P.xpl);
    if(d->P.upl) hipFree(d->P.upl);
    if(d->P.pending) hipFree(d->P.pending);
    if(d->P.n_pending) hipFree(d->P.n_pending);
    for(double *p : d->param_bufs) hipFree(p);
    if(d->P.p) hipFree(d->P.p);
    if(d->staging) hipFree(d->staging);
    if(d->pinned) hipHostFree(d->pinned);
    if(d->stream) hipStreamDestroy(d->stream);
    if(d->stream2) hipStreamDestroy(d->stream2);
    if(d->fork) hipEventDestroy(d->fork);
    if(d->join) hipEventDestroy(d->join);
    if(d->roll) hipStreamSynchronize(d->roll);
    if(d->roll_in) hipEventDestroy(d->roll_in);
    if(d->roll_out) hipEventDestroy(d->roll_out);
    if(d->epoch) hipEventDestroy(d->epoch);
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
    {
        constexpr int want[ILQG_NP > 0 ? ILQG_NP : 1] = ILQG_PSIZES;
        constexpr int offs[ILQG_NP > 0 ? ILQG_NP : 1] = ILQG_POFFSETS;
        if(n_params != ILQG_NP) {
            g_err = "ilqg_dev_set_params: parameter count differs from the problem this library was built for";
            return 1;
        }
        for(int i = 0; i < n_params; i++) {
            if(sizes[i] != want[i]) {
                g_err = "ilqg_dev_set_params: parameter size differs from the problem's paramdesc[]";
                return 1;
            }
            for(int j = 0; j < sizes[i]; j++) d->pv.v[offs[i] + j] = values[i][j];
        }
        d->per_step_params = false;
        for(int i = 0; i < n_params; i++) d->per_step_params |= (sizes[i] == -1);
    }
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
    d->work_consts = d->half_consts[0] = d->half_consts[1] = false;  // constant record entries depend on the parameters
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

// a per-trajectory scalar field (B doubles) into device memory of the caller, on the context's stream
int ilqg_dev_copy_scalar_to(ilqg_dev_t *d, int field, void *dst_device) {
    HIP_TRY(hipSetDevice(d->device));
    if(field < ILQG_F_COST || field >= ILQG_F_ALPHA_COST) {
        g_err = "ilqg_dev_copy_scalar_to: not a per-trajectory scalar field";
        return 1;
    }
    HIP_TRY(hipMemcpyAsync(dst_device, d->P.f[field], (size_t)d->B * sizeof(double), hipMemcpyDeviceToDevice, d->stream));
    return 0;
}

int ilqg_dev_write(ilqg_dev_t *d, int field, const double *host) {
    return ilqg_dev_write_steps(d, field, host, field_steps(d, field));
}

// trajectory-major on the device, i.e. already in host layout: copied without a kernel
static bool is_traj_major(int field) { return WAVE_MAP && field == ILQG_F_FIN; }

static int nom_io(ilqg_dev *d, int field, double *host_rw, const double *host_ro, int steps) {
    const FieldInfo fi = field_info(field);
    const size_t n = (size_t)d->B * steps * fi.wh;
    void *dev, *pin;
    if(stage(d, n * sizeof(double), &dev, &pin)) return 1;
    if(host_ro && stage_in(d, dev, pin, host_ro, n * sizeof(double))) return 1;
    {
        Timed t(d, ILQG_K_TRANSPOSE);
        hipLaunchKernelGGL(k_nom_io, grid1(n, 256), dim3(256), 0, d->stream, d->P.nom, (double *)dev, d->B, d->N, steps,
                           fi.wh, nom_column(field), host_ro ? 1 : 0);
    }
    HIP_TRY(hipGetLastError());
    if(host_rw) return stage_out(d, dev, pin, host_rw, n * sizeof(double));
    return io_done(d);
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

// The context's PRIVATE record buffer, for calls that leave records behind or find them there (ilqg_dev_derivs, the
// single sweep of the drop-in back_pass(), reading / writing records): allocated when the first such call is made,
// for the whole batch (these calls exist for tests and for the single-trajectory drop-in path).
static int own_work(ilqg_dev *d) {
    if(d->P.work) return 0;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const size_t per_traj = (size_t)d->N * sizeof(trajEl_t);
    if((double)d->B * (double)per_traj > 0.5 * (double)free_b) {
        g_err = "derivative records of the whole batch do not fit the device (wave mapping): stage-by-stage calls and "
                "record transfers need a smaller batch; ilqg_dev_iterate / ilqg_dev_backward(mode 2) work in chunks";
        return 1;
    }
    HIP_TRY(hipMalloc((void **)&d->P.work, (size_t)d->B * per_traj));
    HIP_TRY(hipMemsetAsync(d->P.work, 0, (size_t)d->B * per_traj, d->stream));
    d->own_chunk = d->B;
    d->work_consts = false;
    return 0;
}

static int der_io(ilqg_dev *d, double *host_rw, const double *host_ro) {
    if(own_work(d)) return 1;
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

static int all_home(ilqg_dev_t *d);

int ilqg_dev_write_steps(ilqg_dev_t *d, int field, const double *host, int steps) {
    HIP_TRY(hipSetDevice(d->device));
    if(has_tiled_copy(field) && all_home(d)) return 1;
    const FieldInfo fi = field_info(field);
    if(steps < 1 || steps > field_steps(d, field)) {
        g_err = "ilqg_dev_write_steps: bad step count";
        return 1;
    }
#if ILQG_WAVE_MAP
    if(field == ILQG_F_DER) {
        if(io_flush(d)) return 1;
        return der_io(d, nullptr, host);
    }
#endif
    if(nom_column(field) >= 0) {
        if(nom_io(d, field, nullptr, host, steps)) return 1;
        if(!has_tiled_copy(field)) return 0;  // else: the tiled copy below as well
    }
    if(is_traj_major(field)) {
        const size_t row = (size_t)steps * fi.wd * sizeof(double);
        const size_t dpitch = (size_t)field_steps(d, field) * fi.wd * sizeof(double);
        const void *src = host;
        if(d->io_deferred) {
            void *dev, *pin;
            if(stage(d, row * d->B, &dev, &pin)) return 1;
            memcpy(pin, host, row * d->B);
            src = pin;
        }
        HIP_TRY(hipMemcpy2DAsync(d->P.f[field], dpitch, src, row, row, d->B, hipMemcpyHostToDevice, d->stream));
        return io_done(d);
    }
    const size_t n = (size_t)d->B * steps * fi.wh;
    void *dev, *pin;
    if(stage(d, n * sizeof(double), &dev, &pin)) return 1;
    if(stage_in(d, dev, pin, host, n * sizeof(double))) return 1;
    {
        Timed t(d, ILQG_K_TRANSPOSE);
        const size_t total = (size_t)d->B * steps * fi.wd;
        hipLaunchKernelGGL(k_to_dev, grid1(total, 256), dim3(256), 0, d->stream, (const double *)dev, d->P.f[field], d->B, d->Bp,
                           steps, fi.wh, fi.wd);
    }
    HIP_TRY(hipGetLastError());
    return io_done(d);
}

int ilqg_dev_read(ilqg_dev_t *d, int field, double *host) {
    HIP_TRY(hipSetDevice(d->device));
    if(has_tiled_copy(field) && all_home(d)) return 1;
    const FieldInfo fi = field_info(field);
    const int steps = field_steps(d, field);
    const size_t n = (size_t)d->B * steps * fi.wh;
#if ILQG_WAVE_MAP
    if(field == ILQG_F_DER) {
        if(io_flush(d)) return 1;
        return der_io(d, host, nullptr);
    }
#endif
    if(nom_column(field) >= 0 && !has_tiled_copy(field)) return nom_io(d, field, host, nullptr, steps);
    if(is_traj_major(field)) {
        void *dev = nullptr, *pin = nullptr;
        if(d->io_deferred && stage(d, n * sizeof(double), &dev, &pin)) return 1;
        return stage_out(d, d->P.f[field], pin, host, n * sizeof(double));
    }
    void *dev, *pin;
    if(stage(d, n * sizeof(double), &dev, &pin)) return 1;
    {
        Timed t(d, ILQG_K_TRANSPOSE);
        hipLaunchKernelGGL(k_from_dev, grid1(n, 256), dim3(256), 0, d->stream, d->P.f[field], (double *)dev, d->B, d->Bp,
                           steps, fi.wh, fi.wd);
    }
    HIP_TRY(hipGetLastError());
    return stage_out(d, dev, pin, host, n * sizeof(double));
}

// int fields are [width][Bp] on the device, [B][width] on the host
int ilqg_dev_write_int(ilqg_dev_t *d, int field, const int *host) {
    HIP_TRY(hipSetDevice(d->device));
    const int w = int_field_width(field);
    const size_t bytes = (size_t)w * d->Bp * sizeof(int);
    std::vector<int> tmp;
    int *src;
    if(d->io_deferred) {
        void *dev, *pin;
        if(stage(d, bytes, &dev, &pin)) return 1;
        src = (int *)pin;
    } else {
        tmp.resize((size_t)w * d->Bp);
        src = tmp.data();
    }
    memset(src, 0, bytes);
    for(int b = 0; b < d->B; b++)
        for(int j = 0; j < w; j++) src[(size_t)j * d->Bp + b] = host[(size_t)b * w + j];
    HIP_TRY(hipMemcpyAsync(d->P.i[field], src, bytes, hipMemcpyHostToDevice, d->stream));
    return io_done(d);
}

int ilqg_dev_read_int(ilqg_dev_t *d, int field, int *host) {
    HIP_TRY(hipSetDevice(d->device));
    const int w = int_field_width(field);
    const size_t bytes = (size_t)w * d->Bp * sizeof(int);
    if(d->io_deferred) {
        void *dev, *pin;
        if(stage(d, bytes, &dev, &pin)) return 1;
        HIP_TRY(hipMemcpyAsync(pin, d->P.i[field], bytes, hipMemcpyDeviceToHost, d->stream));
        d->pending.push_back({host, pin, bytes, w});
        return 0;
    }
    std::vector<int> tmp((size_t)w * d->Bp, 0);
    HIP_TRY(hipMemcpyAsync(tmp.data(), d->P.i[field], tmp.size() * sizeof(int), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    for(int b = 0; b < d->B; b++)
        for(int j = 0; j < w; j++) host[(size_t)b * w + j] = tmp[(size_t)j * d->Bp + b];
    return 0;
}

int ilqg_dev_io_begin(ilqg_dev_t *d) {
    d->io_deferred = true;
    d->io_off = 0;
    return 0;
}

int ilqg_dev_io_end(ilqg_dev_t *d) {
    HIP_TRY(hipSetDevice(d->device));
    d->io_deferred = false;
    return io_flush(d);
}

#define NEED_PARAMS(d)                                                   \
    if(!(d)->P.p) {                                                      \
        g_err = "problem parameters not set (ilqg_dev_set_params)";      \
        return 1;                                                        \
    }

// (re)allocates a device buffer that must hold `need` bytes; what is queued on `rs` may still use the old one
static int ensure_buffer(ilqg_dev_t *d, double **buf, size_t *have, size_t need, hipStream_t rs) {
    if(*have >= need) return 0;
    HIP_TRY(hipStreamSynchronize(rs));
    if(*buf) HIP_TRY(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc((void **)buf, need));
    *have = need;
    return 0;
}

// the same for a buffer the caller can do without: 2 = the device does not have the memory (the error is cleared)
static int try_buffer(ilqg_dev_t *d, double **buf, size_t *have, size_t need, hipStream_t rs) {
    if(*have >= need) return 0;
    HIP_TRY(hipStreamSynchronize(rs));
    if(*buf) HIP_TRY(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    // (ILQG_TEST_NO_PLANES: the tests' way to a device that is out of memory)
    if(getenv("ILQG_TEST_NO_PLANES") || hipMalloc((void **)buf, need) != hipSuccess) {
        (void)hipGetLastError();
        *buf = nullptr;
        return 2;
    }
    *have = need;
    return 0;
}

// ls_keep = 2: every current trajectory back into the arrays X / U (the host is about to read or write them, an
// initial roll-out is about to store there, or a search that does not keep its roll-outs follows)
static int all_home(ilqg_dev_t *d) {
#if !ILQG_WAVE_MAP
    if(d->loc_set < 0) return 0;
    hipLaunchKernelGGL(k_all_home, dim3(8 * d->cus), dim3(256), 0, d->stream, d->P);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(d->P.i[ILQG_I_LOC], 0, d->Bp * sizeof(int), d->stream));
    d->loc_set = -1;
#endif
    return 0;
}

int ilqg_dev_reset(ilqg_dev_t *d) {
    HIP_TRY(hipSetDevice(d->device));
    hipLaunchKernelGGL(k_reset, grid1(d->Bp, 256), dim3(256), 0, d->stream, d->P, d->O);
    if(HAS_MUL) {  // update_multipliers(o, 1) of the solver entry (iLQG.c:236)
        NEED_PARAMS(d);
        Timed t(d, ILQG_K_MULTIPLIERS);
        hipLaunchKernelGGL(k_multipliers, dim3(d->Bp / WAVE), dim3(WAVE), 0, d->stream, d->P, d->O, d->pv, 1);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// the stream the roll-out family of a context runs on, and the two hand-overs with the context's own stream
static hipStream_t roll_stream(ilqg_dev_t *d) { return d->roll ? d->roll : d->stream; }
static int roll_enter(ilqg_dev_t *d) {  // what is queued on the context's stream happens before the roll-outs
    if(!d->roll) return 0;
    HIP_TRY(hipEventRecord(d->roll_in, d->stream));
    HIP_TRY(hipStreamWaitEvent(d->roll, d->roll_in, 0));
    return 0;
}
static int roll_leave(ilqg_dev_t *d) {  // ... and the roll-outs before whatever the context's stream gets next
    if(!d->roll) return 0;
    HIP_TRY(hipEventRecord(d->roll_out, d->roll));
    HIP_TRY(hipStreamWaitEvent(d->stream, d->roll_out, 0));
    return 0;
}

// (launch errors are picked up by the caller's hipGetLastError())
static void launch_rollout(ilqg_dev_t *d, int mode, int kernel_id, int a0, int n_alpha, hipStream_t stream = nullptr) {
    if(!stream) stream = d->stream;
    Timed t(d, kernel_id, stream);
    const dim3 grid((d->Bp + ROLL_BLOCK - 1) / ROLL_BLOCK, n_alpha), block(ROLL_BLOCK);
    if(mode == ROLL_INIT)
        hipLaunchKernelGGL(k_rollout<RK_INIT>, grid, block, 0, stream, d->P, d->O, d->pv, mode, a0);
    else if(mode == ROLL_COST)
        hipLaunchKernelGGL(k_rollout<RK_COST>, grid, block, 0, stream, d->P, d->O, d->pv, mode, a0);
#if ILQG_WAVE_MAP && defined(ILQG_ROLLOUT_PARTS)
    else if(!HAS_MUL && !getenv("ILQG_NO_ROLLOUT_PARTS")) {  // the generated file offers the step in parts: several wavefronts per 64 trajectories
        DevPtrs Q = d->P;
        if(!d->keep_first) Q.cand1 = nullptr;
        // (second stage: a busy workgroup is a chain of N steps that keeps its CU's SIMDs issuing; two of them on one CU
        // take twice as long, and the dispatcher puts two on one CU while others idle.  Asking for more than half of
        // the CU's LDS leaves room for one.)
        if(d->roll_dma) {
            hipLaunchKernelGGL(k_rollout_parts<true>, dim3(d->Bp / WAVE, n_alpha), dim3(WAVE * RW), (size_t)WAVE * RNP * sizeof(double) + d->roll_pad,
                               stream, Q, d->O, d->pv, mode, a0);
        } else {
            const size_t lds = (n_alpha > 1) ? d->roll_lds : 0;
            hipLaunchKernelGGL(k_rollout_parts<false>, dim3(d->Bp / WAVE, n_alpha), dim3(WAVE * RW), lds, stream, Q, d->O, d->pv, mode, a0);
        }
    }
#endif
    else
        hipLaunchKernelGGL(k_rollout<RK_GENERAL>, grid, block, 0, stream, d->P, d->O, d->pv, mode, a0);
}

int ilqg_dev_rollout_init(ilqg_dev_t *d) {
    NEED_PARAMS(d);
    HIP_TRY(hipSetDevice(d->device));
    if(all_home(d)) return 1;  // (the controls the roll-out starts from are read where they currently are)
    if(roll_enter(d)) return 1;
    HIP_TRY(hipMemsetAsync(d->P.i[ILQG_I_STATUS], 0, d->Bp * sizeof(int), roll_stream(d)));
    launch_rollout(d, ROLL_INIT, ILQG_K_ROLLOUT_INIT, 0, 1, roll_stream(d));
    HIP_TRY(hipGetLastError());
    return roll_leave(d);
}

#if ILQG_WAVE_MAP
// wave mapping: derivative records are evaluated chunk by chunk into the work buffer and consumed by
// the backward kernel of the same chunk.  do_derivs = 0 uses the records already in the buffer.
static int wave_backward(ilqg_dev_t *d, int single_sweep, int do_derivs, int do_backward) {
    // Records produced and consumed in one go use the device's shared work buffer, chunk by chunk; records that are
    // left behind or found (ilqg_dev_derivs, the drop-in back_pass()) the context's private one, whole batch.
    const bool transient = do_derivs && do_backward;
    // factored records (FACTORED builds, option fuse_derivs): only in the transient case — records the caller reads or
    // writes are always the complete ones
    const bool fact = FACTORED && d->O.fuse_derivs && transient;
    SharedWork *W = transient ? d->shared : nullptr;
    const size_t stride = fact ? FACT_STRIDE : sizeof(trajEl_t);
    char *work;
    int chunk;
    if(transient) {
        work = reinterpret_cast<char *>(W->buf);
        // (the last record reaches sizeof(trajEl_t) beyond its start whatever the distance between records)
        const size_t fit = (W->bytes - sizeof(trajEl_t)) / ((size_t)d->N * stride);
        chunk = (int)(fit > (size_t)d->B ? (size_t)d->B : fit);
        if(chunk < 1) {
            g_err = "the device's derivative work buffer does not hold one trajectory of this horizon";
            return 1;
        }
        HIP_TRY(hipStreamWaitEvent(d->stream, W->free_ev, 0));  // the previous owner's pass has finished
    } else {
        if(own_work(d)) return 1;
        work = reinterpret_cast<char *>(d->P.work);
        chunk = d->own_chunk;
    }
    // A batch that needs several chunks alternates between the two halves of the work buffer on two streams, so that
    // the derivatives of one chunk are evaluated while the other's backward pass runs and the end of one backward
    // kernel (few wavefronts still busy) is filled by the next.  The pieces are of equal size.
    // (a batch that fits as a whole goes in two pieces as well, if it is large enough to fill the chip twice)
    // (Quad mapping: a batch whose records fit goes as ONE piece.  Its rows are workers that take trajectories from a queue,
    // four to a wavefront: the more trajectories a queue holds per worker, the less the workers' last ones — 1 to 4 sweeps
    // each — stick out at the end; measured, config 5: 197.8 ms per iteration in two pieces, 181.6 in one.
    // ILQG_TWO_PIECES=1 restores the halves.)
    const bool quad_here = QUAD_STEP && d->O.regType == 1 && (fact || !FULL) && !getenv("ILQG_NO_QUAD");
    const bool split = transient && chunk >= 2 && (d->B > chunk || (d->B >= 16 * d->cus && (!quad_here || getenv("ILQG_TWO_PIECES"))));
    const int half_cap = split ? chunk / 2 : chunk;              // trajectories a half of the buffer holds
    const int pieces = (d->B + half_cap - 1) / half_cap;
    int part = chunk;
    if(split) {
        part = ((d->B + pieces - 1) / pieces + 7) / 8 * 8;  // whole workgroups of the factored backward kernel
        if(part > half_cap) part = half_cap;
    }
    // which constant entries (init_running) the buffer already holds for this context
    bool *whole = &d->work_consts, *half = d->half_consts;
    if(transient) {
        const bool same = W->pv_set && W->N == d->N && W->B == d->B && W->part == part && W->half_cap == half_cap &&
                          W->factored == fact && !d->per_step_params && memcmp(&W->pv, &d->pv, sizeof(ParamValues)) == 0;
        if(!same) {
            W->whole = W->half[0] = W->half[1] = false;
            W->pv = d->pv;
            W->pv_set = !d->per_step_params;
            W->N = d->N;
            W->B = d->B;
            W->part = part;
            W->half_cap = half_cap;
            W->factored = fact;
        }
        whole = &W->whole;
        half = W->half;
    }
    if(split) {
        HIP_TRY(hipEventRecord(d->fork, d->stream));
        HIP_TRY(hipStreamWaitEvent(d->stream2, d->fork, 0));
    }
    int piece = 0;
    for(int c0 = 0; c0 < d->B; c0 += part, piece++) {
        const int cnt = (d->B - c0 < part) ? d->B - c0 : part;
        const int h = split ? (piece & 1) : 0;
        hipStream_t st = h ? d->stream2 : d->stream;
        DevPtrs P = d->P;
        P.work = reinterpret_cast<trajEl_t *>(work + (size_t)h * half_cap * d->N * stride);
        P.work_stride = stride;
        P.queue = d->queues + h * QUEUE_CELL;
        if(do_derivs) {
            Timed t(d, ILQG_K_DERIVS, st);
            const size_t total = (size_t)cnt * (d->N + 1);
            const bool have_consts = *whole || (split && half[h]);
            // (transient records are read by the backward pass alone: the limits' signs and gradients, which it does not
            // use unless the limits depend on the state, are left out of them)
#if ILQG_HAVE_DERIV_PARTS
            // records of the steps assembled on chip, whole lines out (k_derivs_parts) once the buffer holds the constant
            // entries.  MEASURED SLOWER than the generated code on a struct per lane (51 against 44 ms per iteration of
            // config 5, see the kernel): off unless ILQG_DERIV_PARTS=1
            if(fact && transient && have_consts && getenv("ILQG_DERIV_PARTS")) {
                hipLaunchKernelGGL(k_derivs_parts, grid1((size_t)cnt * d->N, 64 * DP_WAVES), dim3(64 * DP_WAVES), 0, st, P, d->O, d->pv, c0, cnt);
                hipLaunchKernelGGL(k_derivs_wave, grid1((size_t)cnt, ILQG_DERIVS_BLOCK), dim3(ILQG_DERIVS_BLOCK), 0, st, P, d->O, d->pv, c0, cnt, 0, 1, 0, 1);
            } else
#endif
            hipLaunchKernelGGL(k_derivs_wave, grid1(total, ILQG_DERIVS_BLOCK), dim3(ILQG_DERIVS_BLOCK), 0, st, P, d->O, d->pv, c0, cnt, have_consts ? 0 : 1,
                               fact ? 1 : 0, (transient && !HX) ? 0 : 1, 0);
            if(cnt == part) {  // every element of this (half of the) buffer that is ever used has its constants now
                if(split) half[h] = true;
                else *whole = half[0] = half[1] = true;
            }
        }
        if(do_backward) {
            HIP_TRY(hipMemsetAsync(P.queue, 0, sizeof(int), st));
            Timed t(d, ILQG_K_BACKWARD, st);
            // quad mapping (16 lanes per trajectory) where it applies; ILQG_NO_QUAD=1: the row mapping (comparison)
            const bool quad = quad_here;
            if(quad) {
                const size_t lds = (size_t)((fact ? TABLE_DOUBLES : 0) + QUAD_WAVES * 4 * QRow::SIZE) * sizeof(double);
                const int per_wg = 4 * QUAD_WAVES;
                const int wgs = (cnt + per_wg - 1) / per_wg;  // one workgroup per CU at a time
                if(fact)
                    hipLaunchKernelGGL(k_backward_quad<FACTORED>, dim3(wgs < d->cus ? wgs : d->cus), dim3(64 * QUAD_WAVES), lds, st, P, d->O,
                                       single_sweep, c0, cnt);
                else
                    hipLaunchKernelGGL(k_backward_quad<false>, dim3(wgs < d->cus ? wgs : d->cus), dim3(64 * QUAD_WAVES), lds, st, P, d->O,
                                       single_sweep, c0, cnt);
            } else if(fact) {
                constexpr int WV = ILQG_FACT_WAVES;
                const size_t lds = (size_t)(TABLE_DOUBLES + WV * WAVE_LDS_DOUBLES) * sizeof(double);
                const int wgs = (cnt + WV - 1) / WV;  // one workgroup per CU at a time (LDS)
                hipLaunchKernelGGL(k_backward_wave<FACTORED>, dim3(wgs < d->cus ? wgs : d->cus), dim3(64 * WV), lds, st, P, d->O,
                                   single_sweep, c0, cnt);
            } else {
                const int cap = d->cus * 8;
                hipLaunchKernelGGL(k_backward_wave<false>, dim3(cnt < cap ? cnt : cap), dim3(64),
                                   (size_t)WAVE_LDS_DOUBLES * sizeof(double), st, P, d->O, single_sweep, c0, cnt);
            }
        }
    }
    if(do_derivs) {  // every slot a batch of this size and division uses has been visited
        if(split) half[0] = half[1] = true;
        else *whole = half[0] = half[1] = true;
    }
    if(split) {
        HIP_TRY(hipEventRecord(d->join, d->stream2));
        HIP_TRY(hipStreamWaitEvent(d->stream, d->join, 0));
    }
    if(transient) HIP_TRY(hipEventRecord(W->free_ev, d->stream));
    HIP_TRY(hipGetLastError());
    return 0;
}
#endif

int ilqg_dev_derivs(ilqg_dev_t *d) {
    NEED_PARAMS(d);
    HIP_TRY(hipSetDevice(d->device));
#if ILQG_WAVE_MAP
    return wave_backward(d, 0, 1, 0);
#else
    {
        Timed t(d, ILQG_K_DERIVS);
        const size_t total = (size_t)d->Bp * (d->N + 1);
        hipLaunchKernelGGL(k_derivs, grid1(total, 256), dim3(256), 0, d->stream, d->P, d->O, d->pv);
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
    return wave_backward(d, mode == 1, mode == 2, 1);
#else
    {
        Timed t(d, mode == 2 ? ILQG_K_BACKWARD_FUSED : ILQG_K_BACKWARD);
        const dim3 grid(d->Bp / WAVE), block(WAVE);
        if(mode == 0)
            hipLaunchKernelGGL(k_backward<0>, grid, block, 0, d->stream, d->P, d->O, d->pv);
        else if(mode == 1)
            hipLaunchKernelGGL(k_backward<1>, grid, block, 0, d->stream, d->P, d->O, d->pv);
#if ILQG_HAVE_SPLIT
        else if(d->O.bw_split && !HAS_MUL)
            hipLaunchKernelGGL(k_backward_split, grid, dim3(2 * WAVE), 0, d->stream, d->P, d->O, d->pv);
#endif
        else
            hipLaunchKernelGGL(k_backward<2>, grid, block, 0, d->stream, d->P, d->O, d->pv);
    }
    if(mode != 2) {
        Timed t(d, ILQG_K_TRANSPOSE);
        hipLaunchKernelGGL(k_pack_records, dim3(d->Bp / WAVE, (d->N + 1 + PACK_WAVES - 1) / PACK_WAVES),
                           dim3(WAVE * PACK_WAVES), 0, d->stream, d->P);
    }
    HIP_TRY(hipGetLastError());
    return 0;
#endif
}

// Line search (line_search.c:33-78) in up to two stages, see k_select / k_rollout:
//   stage 1: step sizes [0, s1) for every trajectory; selection
//   stage 2: step sizes [s1, n_alpha) for the trajectories still without an acceptable one; selection
int ilqg_dev_search(ilqg_dev_t *d) {
    NEED_PARAMS(d);
    HIP_TRY(hipSetDevice(d->device));
    const int A = d->O.n_alpha;
    const int s1 = (d->O.ls_split > 0 && d->O.ls_split < A) ? d->O.ls_split : A;
    hipStream_t rs = roll_stream(d);
#if !ILQG_WAVE_MAP
    bool keep_all = d->O.ls_keep >= 2 && s1 <= PLANE_A && WAVE / s1 >= 1 && (A == s1 || WAVE / (A - s1) >= 1);
    const size_t xplane = (size_t)(d->N + 1) * NX * d->Bp, uplane = (size_t)d->N * NU * d->Bp;
    if(keep_all) {
        // Two sets of s1 planes of the size of X resp. U (12.6 GB for the benchmark: 2 x 4 planes of 65 536 x 501 x 6
        // doubles).  The planes are laid out s1 to a set: a first stage of another size re-homes the trajectories first.
        // A device that cannot provide them (a larger batch, several contexts sharing it) searches without keeping
        // every roll-out instead of failing: the ls_keep = 1 form below needs none of this memory.
        if(d->P.plane_n != s1) {
            if(all_home(d)) return 1;
            d->P.plane_n = s1;
        }
        const int rx = try_buffer(d, &d->P.xpl, &d->xpl_bytes, 2 * (size_t)s1 * xplane * sizeof(double), rs);
        const int ru = rx ? rx : try_buffer(d, &d->P.upl, &d->upl_bytes, 2 * (size_t)s1 * uplane * sizeof(double), rs);
        if(rx == 1 || ru == 1) return 1;
        if(rx == 2 || ru == 2) {
            if(d->P.xpl) HIP_TRY(hipFree(d->P.xpl));
            d->P.xpl = nullptr;
            d->xpl_bytes = 0;
            keep_all = false;  // (all_home() below: nothing lives in a plane any more — there are none)
            d->loc_set = -1;
            HIP_TRY(hipMemsetAsync(d->P.i[ILQG_I_LOC], 0, d->Bp * sizeof(int), d->stream));
        }
    }
    if(keep_all) {
        // Every roll-out of the search is kept and the accepted one becomes the current trajectory by a change of its
        // location index (k_search, k_adopt_home, k_commit; see cur_x).
        const int n2 = A - s1, set = (d->loc_set == 0) ? 1 : 0;
        d->P.xplane = xplane;
        d->P.uplane = uplane;
        if(n2 > 0 && ensure_buffer(d, &d->P.cand, &d->cand_bytes, (size_t)d->Bp * n2 * (d->N + 1) * CAND_W * sizeof(double), rs))
            return 1;
        if(!d->pending_zero) HIP_TRY(hipMemsetAsync(d->P.n_pending, 0, sizeof(int), rs));
        d->pending_zero = false;
        {
            Timed t(d, ILQG_K_SEARCH, rs);
            const int T = WAVE / s1;
            const bool dma = T * (RN / 2) <= WAVE * DMA_LOADS && !getenv("ILQG_NO_DMA");
            if(dma)
                hipLaunchKernelGGL((k_search<0, true>), dim3((d->B + T - 1) / T), dim3(WAVE), 2 * T * RN * sizeof(double), rs, d->P, d->O, d->pv, 0, s1, set);
            else
                hipLaunchKernelGGL((k_search<0, false>), dim3((d->B + T - 1) / T), dim3(WAVE), 0, rs, d->P, d->O, d->pv, 0, s1, set);
        }
        if(n2 > 0) {  // the grid covers the worst case; wavefronts beyond the pending count return at once
            {
                Timed t(d, ILQG_K_SEARCH2, rs);
                const int T = WAVE / n2;
                const bool dma = T * (RN / 2) <= WAVE * DMA_LOADS && !getenv("ILQG_NO_DMA");
                if(dma)
                    hipLaunchKernelGGL((k_search<1, true>), dim3((d->B + T - 1) / T), dim3(WAVE), 2 * T * RN * sizeof(double), rs, d->P, d->O, d->pv, s1, n2, set);
                else
                    hipLaunchKernelGGL((k_search<1, false>), dim3((d->B + T - 1) / T), dim3(WAVE), 0, rs, d->P, d->O, d->pv, s1, n2, set);
            }
            Timed t(d, ILQG_K_ADOPT, rs);
            hipLaunchKernelGGL(k_adopt_home, dim3(8 * d->cus), dim3(256), 0, rs, d->P, s1);
        } else {
            Timed t(d, ILQG_K_ADOPT, rs);
            hipLaunchKernelGGL(k_rejected_home, dim3(8 * d->cus), dim3(256), 0, rs, d->P);
        }
        if(d->defer_commit) {  // inside ilqg_dev_iterate: the update kernel of this iteration commits
            d->commit_pending = true;
            d->commit_s1 = s1;
            d->commit_set = set;
        } else {
            Timed t(d, ILQG_K_ADOPT, rs);
            hipLaunchKernelGGL(k_commit, grid1(d->Bp, 256), dim3(256), 0, rs, d->P, s1, set);
        }
        d->loc_set = set;
        d->winner_done = true;
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if(all_home(d)) return 1;  // the searches below store accepted trajectories in place: in X / U
#endif
    if(roll_enter(d)) return 1;
    HIP_TRY(hipMemsetAsync(d->P.n_pending, 0, sizeof(int), rs));
    d->pending_zero = false;
#if ILQG_WAVE_MAP && defined(ILQG_ROLLOUT_PARTS)
    if(d->O.ls_keep >= 2 && !HAS_MUL && !getenv("ILQG_NO_ROLLOUT_PARTS")) {
        // Wave mapping with the roll-outs in parts: BOTH stages keep what they roll out and the accepted roll-outs are
        // copied into the records — no winner pass (16 384 chains of N steps beside the second stage's: the second
        // launch needed two rounds of workgroups, 50 ms; the second stage alone fits one).
        const int n2 = A - s1;
        const size_t row = (size_t)d->Bp * (d->N + 1) * CAND_W * sizeof(double);
        if(ensure_buffer(d, &d->P.cand1, &d->cand1_bytes, row * s1, rs)) return 1;
        if(n2 > 0 && ensure_buffer(d, &d->P.cand, &d->cand_bytes, row * n2, rs)) return 1;
        d->keep_first = true;
        launch_rollout(d, ROLL_SEARCH, ILQG_K_ROLLOUT_SEARCH, 0, s1, rs);
        d->keep_first = false;
        {
            Timed t(d, ILQG_K_SELECT, rs);
            hipLaunchKernelGGL(k_select, grid1(d->Bp, 256), dim3(256), 0, rs, d->P, d->O, 0, s1, 0);
        }
        if(n2 > 0) {
            launch_rollout(d, ROLL_LIST_KEEP, ILQG_K_ROLLOUT_SEARCH2, s1, n2, rs);
            {
                Timed t(d, ILQG_K_SELECT, rs);
                hipLaunchKernelGGL(k_select, grid1(d->Bp, 256), dim3(256), 0, rs, d->P, d->O, s1, A, 1);
            }
            Timed t(d, ILQG_K_ROLLOUT_WINNER, rs);
            hipLaunchKernelGGL(k_adopt, dim3(8 * d->cus), dim3(256), 0, rs, d->P, s1, n2);
        }
        {
            Timed t(d, ILQG_K_ROLLOUT_WINNER, rs);
            hipLaunchKernelGGL(k_adopt_first, dim3(16 * d->cus), dim3(256), 0, rs, d->P, s1);
        }
        d->winner_done = true;
        HIP_TRY(hipGetLastError());
        return roll_leave(d);
    }
#endif
    launch_rollout(d, ROLL_SEARCH, ILQG_K_ROLLOUT_SEARCH, 0, s1, rs);
    {
        Timed t(d, ILQG_K_SELECT, rs);
        hipLaunchKernelGGL(k_select, grid1(d->Bp, 256), dim3(256), 0, rs, d->P, d->O, 0, s1, 0);
    }
    d->winner_done = false;
    if(s1 < A && !d->O.ls_keep) {
        // the grid covers the worst case; blocks beyond the pending count return at once
        launch_rollout(d, ROLL_SEARCH_LIST, ILQG_K_ROLLOUT_SEARCH2, s1, A - s1, rs);
        Timed t(d, ILQG_K_SELECT, rs);
        hipLaunchKernelGGL(k_select, grid1(d->Bp, 256), dim3(256), 0, rs, d->P, d->O, s1, A, 1);
    } else if(s1 < A) {
        // Second stage and the winner pass of the trajectories the first stage settled in ONE launch (ROLL_SECOND);
        // the grid covers the worst case, blocks beyond the pending count return at once.
        const int n2 = A - s1;
        const size_t need = (size_t)d->Bp * n2 * (d->N + 1) * CAND_W * sizeof(double);
        if(d->cand_bytes < need) {
            HIP_TRY(hipStreamSynchronize(rs));
            if(d->P.cand) HIP_TRY(hipFree(d->P.cand));
            d->P.cand = nullptr;
            d->cand_bytes = 0;
            HIP_TRY(hipMalloc((void **)&d->P.cand, need));
            d->cand_bytes = need;
        }
        launch_rollout(d, ROLL_SECOND, ILQG_K_ROLLOUT_SEARCH2, s1, n2 + 1, rs);
        {
            Timed t(d, ILQG_K_SELECT, rs);
            hipLaunchKernelGGL(k_select, grid1(d->Bp, 256), dim3(256), 0, rs, d->P, d->O, s1, A, 1);
        }
        {
            Timed t(d, ILQG_K_ROLLOUT_WINNER, rs);
            hipLaunchKernelGGL(k_adopt, dim3(8 * d->cus), dim3(256), 0, rs, d->P, s1, n2);
        }
        d->winner_done = true;
    }
    HIP_TRY(hipGetLastError());
    return roll_leave(d);
}

int ilqg_dev_winner(ilqg_dev_t *d) {
    NEED_PARAMS(d);
    HIP_TRY(hipSetDevice(d->device));
    if(d->winner_done) {  // the two-stage search has stored the accepted trajectories already (ROLL_SECOND, k_adopt)
        d->winner_done = false;
        return 0;
    }
    if(roll_enter(d)) return 1;
    launch_rollout(d, ROLL_WINNER, ILQG_K_ROLLOUT_WINNER, 0, 1, roll_stream(d));
    HIP_TRY(hipGetLastError());
    return roll_leave(d);
}

int ilqg_dev_update(ilqg_dev_t *d) {
    NEED_PARAMS(d);
    HIP_TRY(hipSetDevice(d->device));
    hipStream_t rs = roll_stream(d);
    if(roll_enter(d)) return 1;
    {
        Timed t(d, ILQG_K_UPDATE, rs);
        const int reset = (!WAVE_MAP && d->defer_commit) ? 1 : 0;
        hipLaunchKernelGGL(k_update, grid1(d->Bp, 256), dim3(256), 0, rs, d->P, d->O, d->commit_pending ? d->commit_s1 : -1, d->commit_set, reset);
        d->commit_pending = false;
        d->pending_zero = reset != 0;
    }
    if(HAS_MUL) {
        Timed t(d, ILQG_K_MULTIPLIERS, rs);
        hipLaunchKernelGGL(k_multipliers, dim3(d->Bp / WAVE), dim3(WAVE), 0, rs, d->P, d->O, d->pv, 0);
    }
    if(d->O.resweep || HAS_MUL) launch_rollout(d, ROLL_COST, ILQG_K_ROLLOUT_COST, 0, 1, rs);
    HIP_TRY(hipGetLastError());
    return roll_leave(d);
}

int ilqg_dev_iterate(ilqg_dev_t *d, int n) {
    struct Defer {  // search and update of one iteration are launched together: the update commits for the search
        ilqg_dev_t *d;
        explicit Defer(ilqg_dev_t *d_) : d(d_) { d->defer_commit = true; }
        ~Defer() { d->defer_commit = false; }
    } defer(d);
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

// cycle accounting of -DILQG_PROFILE_SECTIONS builds: reads and clears the 8 section counters (zeros otherwise)
int ilqg_dev_section_cycles(unsigned long long *out) {
#ifdef ILQG_PROFILE_SECTIONS
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(ilqg_prof_cycles), 8 * sizeof(unsigned long long)));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ilqg_prof_cycles), z, sizeof(z)));
#else
    for(int i = 0; i < 8; i++) out[i] = 0;
#endif
    return 0;
}

int ilqg_dev_timing(ilqg_dev_t *d, int enable) {
    HIP_TRY(hipSetDevice(d->device));
    if(drain_spans(d)) return 1;
    if(enable)
        while(d->event_pool.size() < 1024) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            d->event_pool.push_back(e);
        }
    d->timing = enable != 0;
    memset(d->t_ms, 0, sizeof(d->t_ms));
    memset(d->t_n, 0, sizeof(d->t_n));
    memset(d->t_busy, 0, sizeof(d->t_busy));
    if(enable) {
        if(!d->epoch) HIP_TRY(hipEventCreate(&d->epoch));
        HIP_TRY(hipEventRecord(d->epoch, d->stream));
    }
    return 0;
}

/* length of the union of the launch intervals of a kernel since timing was switched on (ms): what the kernel occupied
 * of the wall clock, however its launches on different streams overlap in the event clock */
int ilqg_dev_get_busy(ilqg_dev_t *d, int kernel, double *busy_ms) {
    if(kernel < 0 || kernel >= ILQG_K_COUNT) {
        g_err = "ilqg_dev_get_busy: bad kernel id";
        return 1;
    }
    if(drain_spans(d)) return 1;
    *busy_ms = d->t_busy[kernel];
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

static int boxqp_batch(int rows, int device, int n, int count, const double *H, const double *g, const double *lower,
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
    const dim3 grid = rows == 1 ? dim3(count) : grid1(count, 64), block(64);
    if(rows == 2 && n == 2)
        hipLaunchKernelGGL((k_boxqp_test<2, true>), grid, block, 0, 0, count, dH, dg, dlo, dup, dx, dcl, dnf, dinv, drc);
    else if(rows == 2 && n == NU)
        hipLaunchKernelGGL((k_boxqp_test<NU, true>), grid, block, 0, 0, count, dH, dg, dlo, dup, dx, dcl, dnf, dinv, drc);
    else if(rows == 2) {
        g_err = "ilqg_dev_boxqp_table_batch: n must be 2 or N_U";
        return 1;
    } else if(rows && n == 2)
        hipLaunchKernelGGL(k_boxqp_rows_test<2>, grid, block, 0, 0, count, dH, dg, dlo, dup, dx, dcl, dnf, dinv, drc);
    else if(rows && n == 8)
        hipLaunchKernelGGL(k_boxqp_rows_test<8>, grid, block, 0, 0, count, dH, dg, dlo, dup, dx, dcl, dnf, dinv, drc);
    else if(rows)
        hipLaunchKernelGGL(k_boxqp_rows_test<NU>, grid, block, 0, 0, count, dH, dg, dlo, dup, dx, dcl, dnf, dinv, drc);
    else if(n == 2)
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

int ilqg_dev_boxqp_batch(int device, int n, int count, const double *H, const double *g, const double *lower,
                         const double *upper, double *x, int *clamp, int *n_free, double *invH, int *rc) {
    return boxqp_batch(0, device, n, count, H, g, lower, upper, x, clamp, n_free, invH, rc);
}

int ilqg_dev_boxqp_wave_batch(int device, int n, int count, const double *H, const double *g, const double *lower,
                              const double *upper, double *x, int *clamp, int *n_free, double *invH, int *rc) {
    return boxqp_batch(1, device, n, count, H, g, lower, upper, x, clamp, n_free, invH, rc);
}

int ilqg_dev_boxqp_table_batch(int device, int n, int count, const double *H, const double *g, const double *lower,
                               const double *upper, double *x, int *clamp, int *n_free, double *invH, int *rc) {
    return boxqp_batch(2, device, n, count, H, g, lower, upper, x, clamp, n_free, invH, rc);
}

// ---------------------------------------------------------------------------
// Several GPUs of one node in ONE process: the trajectory batch is sharded, every device advances its
// shard by itself, and the only exchange is one RCCL gather of a per-trajectory scalar (the costs) to a root device.
// ---------------------------------------------------------------------------
struct ilqg_comm {
    int n;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::vector<double *> send;   // per device: its shard of the scalar, padded to `per`
    double *recv;                 // root device: n * per doubles
    int per;
    bool loopback;                // every shard on ONE device (tests, rehearsals): the gather is device-to-device copies
};

#define NCCL_TRY(expr)                                                              \
    do {                                                                            \
        ncclResult_t r_ = (expr);                                                   \
        if(r_ != ncclSuccess) {                                                     \
            g_err = std::string(#expr) + ": " + ncclGetErrorString(r_);             \
            return 1;                                                               \
        }                                                                           \
    } while(0)

void ilqg_comm_destroy(ilqg_comm_t *c) {
    if(!c) return;
    for(size_t g = 0; g < c->comms.size(); g++)
        if(c->comms[g]) ncclCommDestroy(c->comms[g]);
    for(size_t g = 0; g < c->send.size(); g++) {
        hipSetDevice(c->devices[g]);
        if(c->send[g]) hipFree(c->send[g]);
    }
    if(c->recv) {
        hipSetDevice(c->devices[0]);
        hipFree(c->recv);
    }
    delete c;
}

static int comm_fill(ilqg_comm *c, int n, const int *devices, int per) {
    c->n = n;
    c->per = per;
    c->devices.assign(devices, devices + n);
    c->comms.assign(n, nullptr);
    c->send.assign(n, nullptr);
    // RCCL wants distinct devices.  Several shards on one device (n > 1, all the same id) is how the sharding, the
    // offsets and the cost hand-over are rehearsed where only one GPU is present: no communicator, the gather copies.
    c->loopback = n > 1;
    for(int g = 1; g < n; g++)
        if(devices[g] != devices[0]) c->loopback = false;
    if(!c->loopback) NCCL_TRY(ncclCommInitAll(c->comms.data(), n, devices));
    for(int g = 0; g < n; g++) {
        HIP_TRY(hipSetDevice(devices[g]));
        HIP_TRY(hipMalloc((void **)&c->send[g], (size_t)per * sizeof(double)));
        HIP_TRY(hipMemset(c->send[g], 0, (size_t)per * sizeof(double)));
    }
    HIP_TRY(hipSetDevice(devices[0]));
    HIP_TRY(hipMalloc((void **)&c->recv, (size_t)n * per * sizeof(double)));
    return 0;
}

// one communicator over `n` distinct devices; `per` = doubles every device contributes to a gather
int ilqg_comm_create(ilqg_comm_t **out, int n, const int *devices, int per) {
    *out = nullptr;
    if(n < 1 || per < 1) {
        g_err = "ilqg_comm_create: need at least one device and one value per device";
        return 1;
    }
    ilqg_comm *c = new ilqg_comm();
    if(comm_fill(c, n, devices, per)) {
        const std::string why = g_err;
        ilqg_comm_destroy(c);
        g_err = why;
        return 1;
    }
    *out = c;
    return 0;
}

void *ilqg_comm_send_buffer(ilqg_comm_t *c, int g) { return (g >= 0 && g < c->n) ? c->send[g] : nullptr; }

// The single collective of the path: the send buffers (filled by the caller, `per` doubles per device, all copies
// complete) -> device 0 by ONE ncclGather, enqueued on the stream of each device's context devs[g], and on to the
// host: host[first[g] .. first[g] + counts[g]) = what device g sent.
int ilqg_comm_gather(ilqg_comm_t *c, ilqg_dev_t *const *devs, const int *first, const int *counts, double *host) {
    for(int g = 0; g < c->n; g++)
        if(counts[g] > c->per) {
            g_err = "ilqg_comm_gather: a shard is larger than the communicator's send buffers";
            return 1;
        }
    if(c->loopback) {
        HIP_TRY(hipSetDevice(c->devices[0]));
        for(int g = 0; g < c->n; g++) {
            HIP_TRY(hipMemcpyAsync(c->recv + (size_t)g * c->per, c->send[g], (size_t)c->per * sizeof(double),
                                   hipMemcpyDeviceToDevice, devs[g]->stream));
            if(g > 0) HIP_TRY(hipStreamSynchronize(devs[g]->stream));
        }
    } else {
        NCCL_TRY(ncclGroupStart());
        for(int g = 0; g < c->n; g++)
            NCCL_TRY(ncclGather(c->send[g], c->recv, (size_t)c->per, ncclDouble, 0, c->comms[g], devs[g]->stream));
        NCCL_TRY(ncclGroupEnd());
    }
    HIP_TRY(hipSetDevice(c->devices[0]));
    std::vector<double> tmp((size_t)c->n * c->per);
    HIP_TRY(hipMemcpyAsync(tmp.data(), c->recv, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, devs[0]->stream));
    HIP_TRY(hipStreamSynchronize(devs[0]->stream));
    for(int g = 0; g < c->n; g++) memcpy(host + first[g], tmp.data() + (size_t)g * c->per, sizeof(double) * counts[g]);
    return 0;
}

}  // extern "C"

