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

// Whether the callbacks of this pair work on a private element with proxies for the derivative arrays (ILQG_DEV_EL; what that
// is: further down, in front of the function file)
#include "ilqg_record_dev.h"  // ILQG_DEV_RECORDS; member and assignment lists of this pair (build directory)
#ifndef ILQG_DEV_ELEMENT      // (-DILQG_DEV_ELEMENT=0: the callbacks on the record itself, as before — comparison)
#define ILQG_DEV_ELEMENT 1
#endif
#if ILQG_DEV_RECORDS && ILQG_DEV_ELEMENT
// the header once under other names: the record's LAYOUT (what lies in HBM, what the host sees)
#define trajEl_t trajEl_layout_t
#define traj_t traj_layout_t
#include "iLQG_problem.h"
#undef trajEl_t
#undef traj_t
#if (N_X > 8) || (defined(ILQG_WAVE_MAP) && ILQG_WAVE_MAP)
#define ILQG_DEV_EL 1
#else
#define ILQG_DEV_EL 0  // lane mapping: the element is a handful of registers anyway
typedef trajEl_layout_t trajEl_t;
typedef traj_layout_t traj_t;
#endif
#else
#define ILQG_DEV_EL 0
#endif


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
    double alone;      // p[-5]: != 0: the callbacks run with SOME lanes of the wavefront (staged record entries: see below)
    double *record;    // p[-6] points here: where staged record entries go in HBM (the lane's trajEl_t; the `t` the
                       //        callbacks get may be a private one, see k_derivs_wave)
};
#include "ilqg_param_layout.h"  // generated at build time from the problem's paramdesc[]: ILQG_NP, sizes, offsets
#define ILQG_HOOK_SLOTS 6
#define ILQG_HOOK_NONFINITE (ILQG_NP)
#define ILQG_HOOK_HUGE (ILQG_NP + 1)
#define ILQG_HOOK_SLOW (ILQG_NP + 2)
#define ILQG_HOOK_LIMGRAD (ILQG_NP + 3)
#define ILQG_HOOK_ALONE (ILQG_NP + 4)
#define ILQG_HOOK_RECORD (ILQG_NP + 5)

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
// the same as a LEAF: the library's code inside instead of behind a call.  A function that calls keeps its return
// address in a lane of a callee-saved vector register, which it first stores to and finally reloads from scratch memory
// — a memory round trip per sin / cos call, 2 000 to 3 000 cycles each with the memory system busy (measured: 64 such
// calls were 170 000 of the 320 000 cycles a wavefront of k_derivs_wave took for the n = 16 problem).
__device__ __forceinline__ static ilqg_sc ilqg_sincos_leaf(double x) {
    ilqg_sc out = ilqg_sincos_fast(x);
    if(!(fabs(x) < 8.0e5)) sincos(x, &out.s, &out.c);
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
__device__ __attribute__((noinline, const)) static ilqg_sc ilqg_sincos_call(double x) { return ilqg_sincos_leaf(x); }
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

// ---------------------------------------------------------------------------
// Record entries collected on chip (wave mapping, k_derivs_wave: one lane = one (trajectory, step), each lane fills the
// trajEl_t of its step in HBM).  A store of the generated code `t->fx[17]= ...` touches 64 records, 64 cache lines, with
// 8 or 16 bytes each; the kernel was bound by those requests (round 3: 63 % of a wavefront's cycles waiting, 116 GB
// written for 72 GB of payload).  The function file of tools/gen_problem.py writes the RUNS of neighbouring entries it
// assigns (all of fx, fu, cx, cu; the products of the factored tensors) through ILQG_REC(member, index) and closes every
// run of at most 64 entries with ILQG_REC_DONE(member, first, count) — by default the plain assignment and nothing.  Here
// an entry goes into a ring of 64 slots per lane in LDS (slot = the entry's place in the record modulo 64; lane-fastest,
// so the 64 lanes of a store hit 64 different banks), and at the end of a run the WAVEFRONT writes it out record by
// record: lane l takes entry first + l of the record of lane s, s = 0 .. 63 — one store instruction per record, 512
// contiguous bytes.  That needs all 64 lanes: k_derivs_wave lets lanes without a record of their own go along as copies
// of one that has.  Where the callbacks run with some lanes only (the lane-by-lane repetition behind a wave-uniform
// guard, the re-evaluation with the library's sin / cos), the `alone` hook is set and a lane copies its own column.
// A file without the macros (Maxima-generated) stores directly as before.
// ---------------------------------------------------------------------------
#ifndef ILQG_DERIVS_BLOCK
#if ILQG_DEV_EL
#define ILQG_DERIVS_BLOCK 64  // one wavefront per workgroup of k_derivs_wave: its ring (ilqgdev) is 33 KB of dynamic LDS
#else
#define ILQG_DERIVS_BLOCK 256
#endif
#endif
#define ILQG_STAGE_LD 65  // doubles between the slots of a lane (64 lanes + 1: the write-out reads a column, lane l slot first + l)
__shared__ double ilqg_stage[(ILQG_DERIVS_BLOCK / 64) * 64 * ILQG_STAGE_LD];
#ifdef ILQG_PROFILE_SECTIONS
// cycle accounting of k_derivs_wave (tools/section_profile_derivs.py): slot = the next probe of the wavefront
__device__ unsigned long long ilqg_dprof_cycles[32];
__shared__ unsigned long long ilqg_dprof_last[ILQG_DERIVS_BLOCK / 64];
__shared__ int ilqg_dprof_next[ILQG_DERIVS_BLOCK / 64];
__device__ __forceinline__ static void ilqg_dprobe(int set_next = -1) {
    const unsigned long long now = __builtin_readcyclecounter();
    const int w = threadIdx.x >> 6;
    if((threadIdx.x & 63) == 0) {
        if(set_next >= 0) ilqg_dprof_next[w] = set_next;
        else {
            const int i = ilqg_dprof_next[w];
            atomicAdd(&ilqg_dprof_cycles[i < 31 ? i : 31], now - ilqg_dprof_last[w]);
            ilqg_dprof_next[w] = i + 1;
        }
        ilqg_dprof_last[w] = __builtin_readcyclecounter();
    }
}
#define ILQG_DPROBE(...) ilqg_dprobe(__VA_ARGS__)
#else
#define ILQG_DPROBE(...) ((void)0)
#endif
__device__ __forceinline__ static double &ilqg_stage_at(unsigned slot) {
    return ilqg_stage[(threadIdx.x >> 6) * (64 * ILQG_STAGE_LD) + slot * ILQG_STAGE_LD + (threadIdx.x & 63)];
}
// entries [first, first + count) (doubles from the start of the record) of all lanes' records, from the ring to HBM
__device__ __forceinline__ static void ilqg_stage_flush(double **p, unsigned first, int count) {
    typedef __attribute__((address_space(1))) double gdouble;
    double *const t = *reinterpret_cast<double **>(p[ILQG_HOOK_RECORD]);
    asm volatile("" ::: "memory");
    ILQG_DPROBE();
    if(*p[ILQG_HOOK_ALONE] != 0.0) {
        double *const out = t + first;
#pragma unroll 1
        for(int i = 0; i < count; i++) out[i] = ilqg_stage_at((first + i) & 63u);
    } else {
        const unsigned lane = threadIdx.x & 63;
        const unsigned long long tb = (unsigned long long)t;
        const unsigned lo = (unsigned)tb, hi = (unsigned)(tb >> 32);
        const double *const ring = &ilqg_stage[(threadIdx.x >> 6) * (64 * ILQG_STAGE_LD)];
        if(count > 32) {
            // one record per store instruction: lane l its entry first + l; the record's address is wave-uniform
            const double *const col = ring + ((first + lane) & 63u) * ILQG_STAGE_LD;
            for(int s0 = 0; s0 < 64; s0 += 8) {
                double v[8];
#pragma unroll
                for(int j = 0; j < 8; j++) v[j] = col[s0 + j];
#pragma unroll
                for(int j = 0; j < 8; j++) {
                    const unsigned long long base = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, s0 + j) << 32) |
                                                    (unsigned)__builtin_amdgcn_readlane((int)lo, s0 + j);
                    if((int)lane < count) reinterpret_cast<gdouble *>(base)[first + lane] = v[j];
                }
            }
        } else {
            // short runs: 64 / W records per store instruction, W = 8, 16 or 32 lanes each
            const int W = count > 16 ? 32 : (count > 8 ? 16 : 8), G = 64 / W;
            const unsigned e = lane & (W - 1), g = lane / W;
            const double *const col = ring + ((first + e) & 63u) * ILQG_STAGE_LD;
            for(int s0 = 0; s0 < 64; s0 += G) {
                const int s = s0 + (int)g;
                const unsigned long long base = ((unsigned long long)(unsigned)__shfl((int)hi, s) << 32) | (unsigned)__shfl((int)lo, s);
                const double v = col[s];
                if((int)e < count) reinterpret_cast<gdouble *>(base)[first + e] = v;
            }
        }
    }
    ILQG_DPROBE();
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------
// The element the callbacks work on (wave mapping, generated pairs WITHOUT staging macros — a Maxima / gentran pair, or
// tools/gen_problem.py --plain): not the record.  A record of the n = 16 problem is 47.9 KB, 47 KB of it the derivative
// arrays cx .. fxu; the callbacks get a `trajEl_t *t` and store `t->fxx[17]= ...`.  With t = the record in HBM every such
// store touches 64 records with 8 bytes each (k_derivs_wave was bound by the L2's request rate: 64 requests per store
// instruction, profiles/r5_stored_path.txt); with t = a private copy of the struct a lane needs 47.9 KB of scratch memory,
// 3 MB per wavefront, and the dispatcher then runs a few dozen wavefronts at a time (the roll-outs of such pairs).
// So on these builds the callbacks are compiled against `trajEl_dev_t`: the members of trajEl_t in the header's order
// (tools/gen_record_dev.py reads them at build time), the derivative arrays replaced by PROXIES without storage.  An
// assignment to a proxy's entry puts the value into a ring of 64 slots per lane in LDS (slot = the entry's place in the
// record modulo 64, lane-fastest), and the assignment that completes a run of neighbouring entries — a compile-time
// table made from the ORDER in which init_running / bp_derivsL assign, same script — has the wavefront write the run out:
// one store instruction per record, 512 contiguous bytes (ilqg_stage_flush's scheme, for a function file that knows
// nothing of it).  A kernel that wants none of this (roll-outs: init_running for the constant auxiliaries) sets the
// wavefront's mode to "discard".  The function file itself is compiled as it stands.
// ---------------------------------------------------------------------------
#if ILQG_DEV_EL
namespace ilqgdev {
constexpr int LD = 65;  // doubles between the slots of a lane (64 lanes + 1: the write-out reads a column)
constexpr int ENTRIES = (int)(sizeof(trajEl_layout_t) / sizeof(double));
enum { DISCARD = 0, TOGETHER = 1, ALONE = 2 };
// STATIC LDS, in every kernel that runs callbacks which assign derivative entries (k_derivs_wave; the roll-outs and
// k_multipliers through init_running): the ring [64 slots][LD] of the workgroup's ONE wavefront, the records its lanes
// write to, and what the end of a run does.  Static, so that a slot's address is one per-lane register (made of threadIdx
// alone, no load) plus an immediate offset: an assignment is one ds_write_b64.  (Dynamic LDS with the mode looked up per
// assignment was tried first: 10-13 minutes of compilation for the n = 16 pair against 37 s.)  Kernels that discard still
// fill the ring — 33 KB of LDS per wavefront, which costs these kernels nothing: at 230-300 registers they run one
// wavefront per SIMD, four per CU.
constexpr int WAVES = 1;  // wavefronts per workgroup of the kernels that run these callbacks (asserted where they are defined)
__shared__ double ring_lds[WAVES * 64 * LD];
__shared__ unsigned long long record_lds[WAVES * 64];
__shared__ int mode_lds[WAVES];
__device__ __forceinline__ double *ring() { return ring_lds; }
__device__ __forceinline__ unsigned long long *records() { return record_lds; }
__device__ __forceinline__ int mode() { return __builtin_amdgcn_readfirstlane(mode_lds[0]); }
__device__ __forceinline__ void set_mode(int m) { mode_lds[0] = m; }
__device__ __forceinline__ void set_record(void *rec) { record_lds[threadIdx.x & 63] = (unsigned long long)rec; }
__device__ __forceinline__ double &slot(unsigned e) { return ring_lds[(e & 63u) * LD + (threadIdx.x & 63)]; }

// entries [first, first + count) of the lanes' records, from the ring to HBM (count <= 64, one aligned window of 64)
#ifndef ILQG_DEV_FLUSH_INLINE
#define ILQG_DEV_FLUSH_INLINE 1
#endif
#if ILQG_DEV_FLUSH_INLINE
#define ILQG_DEV_FLUSH_ATTR __device__ __forceinline__
#else
#define ILQG_DEV_FLUSH_ATTR __device__ __attribute__((noinline))
#endif
ILQG_DEV_FLUSH_ATTR void flush(int m, unsigned first, int count) {
    typedef __attribute__((address_space(1))) double gdouble;
    asm volatile("" ::: "memory");
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long tb = records()[lane];
    if(m == ALONE) {
        double *const out = reinterpret_cast<double *>(tb) + first;
#pragma unroll 1
        for(int i = 0; i < count; i++) out[i] = slot(first + i);
    } else {
        const unsigned lo = (unsigned)tb, hi = (unsigned)(tb >> 32);
        const double *const rg = ring();
        // (two records per store instruction, 16 bytes per lane, was measured for whole windows: 340 against 260 ms per
        // iteration of config 5 — two 512-byte segments per instruction are slower than one)
        if(count > 32) {
            // one record per store instruction: lane l its entry first + l; the record's address is wave-uniform
            const double *const col = rg + ((first + lane) & 63u) * LD;
            for(int s0 = 0; s0 < 64; s0 += 8) {
                double v[8];
#pragma unroll
                for(int j = 0; j < 8; j++) v[j] = col[s0 + j];
#pragma unroll
                for(int j = 0; j < 8; j++) {
                    const unsigned long long base = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, s0 + j) << 32) |
                                                    (unsigned)__builtin_amdgcn_readlane((int)lo, s0 + j);
                    if((int)lane < count) reinterpret_cast<gdouble *>(base)[first + lane] = v[j];
                }
            }
        } else {
            // short runs: 64 / W records per store instruction, W = 8, 16 or 32 lanes each
            const int W = count > 16 ? 32 : (count > 8 ? 16 : 8), G = 64 / W;
            const unsigned e = lane & (W - 1), g = lane / W;
            const double *const col = rg + ((first + e) & 63u) * LD;
            for(int s0 = 0; s0 < 64; s0 += G) {
                const int s = s0 + (int)g;
                const unsigned long long base = ((unsigned long long)(unsigned)__shfl((int)hi, s) << 32) | (unsigned)__shfl((int)lo, s);
                const double v = col[s];
                if((int)e < count) reinterpret_cast<gdouble *>(base)[first + e] = v;
            }
        }
    }
    asm volatile("" ::: "memory");
}

// the assignment order of the function file -> "this entry ends the run [first, first + count)"
struct Run {
    unsigned short first, count;
};
struct RunTable {
    Run r[ENTRIES];
    int collisions;
};
#define ILQG_DEV_E(member, index) (unsigned short)(offsetof(trajEl_layout_t, member) / sizeof(double) + (index)),
constexpr unsigned short seq_init[] = {ILQG_DEV_SEQ_INIT(ILQG_DEV_E) 0xffff};
constexpr unsigned short seq_derivs[] = {ILQG_DEV_SEQ_DERIVS(ILQG_DEV_E) 0xffff};
#undef ILQG_DEV_E
constexpr void add_runs(RunTable &t, const unsigned short *seq, int n) {
    int first = 0;  // position in seq where the current run began
    for(int j = 0; j < n; j++) {
        const bool last = j + 1 == n || seq[j + 1] != seq[j] + 1 || (seq[j + 1] & 63) == 0;
        if(last) {
            if(t.r[seq[j]].count) t.collisions++;
            t.r[seq[j]].first = seq[first];
            t.r[seq[j]].count = (unsigned short)(j + 1 - first);
            first = j + 1;
        }
    }
}
constexpr RunTable make_runs() {
    RunTable t{};
    add_runs(t, seq_init, (int)(sizeof(seq_init) / sizeof(seq_init[0])) - 1);
    add_runs(t, seq_derivs, (int)(sizeof(seq_derivs) / sizeof(seq_derivs[0])) - 1);
    return t;
}
__device__ constexpr RunTable runs = make_runs();
static_assert(runs.collisions == 0, "an entry ends a run of init_running and one of bp_derivsL");

template <unsigned OFF>
struct Ref {
    int i;
    __device__ __forceinline__ void operator=(double v) const {
        const unsigned e = OFF + (unsigned)i;
        slot(e) = v;
        const Run r = runs.r[e];
        if(r.count) {  // (a compile-time condition wherever the index is a literal)
            const int m = mode();
            if(m != DISCARD) flush(m, r.first, r.count);
        }
    }
    __device__ __forceinline__ operator double() const { return slot(OFF + (unsigned)i); }
};
template <unsigned OFF, int N>
struct Arr {
    __device__ __forceinline__ Ref<OFF> operator[](int i) const { return Ref<OFF>{i}; }
};
// memset(t->fxx, 0, sizeof(double) * ...) of init_running (iLQG_func.tem:338-342): the member's entries, one by one
template <unsigned OFF, int N>
__device__ __forceinline__ void *dev_memset(const Arr<OFF, N> &a, int, size_t) {
#pragma unroll 1
    for(int i = 0; i < N; i++) a[i] = 0.0;
    return nullptr;
}
__device__ __forceinline__ void *dev_memset(void *d, int v, size_t n) { return __builtin_memset(d, v, n); }
}  // namespace ilqgdev

struct trajEl_dev_t {
#define ILQG_DEV_SCALAR(name) double name;
#define ILQG_DEV_ARRAY(name, count) double name[count];
#define ILQG_DEV_PROXY(name, count) ilqgdev::Arr<(unsigned)(offsetof(trajEl_layout_t, name) / sizeof(double)), (count)> name;
    ILQG_DEV_MEMBERS(ILQG_DEV_SCALAR, ILQG_DEV_ARRAY, ILQG_DEV_PROXY)
#undef ILQG_DEV_SCALAR
#undef ILQG_DEV_ARRAY
#undef ILQG_DEV_PROXY
};
typedef struct {
    trajEl_dev_t *t;
    trajFin_t f;
} traj_dev_t;
#define trajEl_t trajEl_dev_t
#define traj_t traj_dev_t
#define memset(d, v, n) ilqgdev::dev_memset(d, v, n)
#endif

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
// Mapping: lane mapping (one lane per trajectory, everything in registers) for small problems,
// wave mapping (one wavefront per trajectory, matrices in LDS) when a lane's registers cannot
// hold the matrices.  -DILQG_WAVE_MAP=1 forces the wave mapping for a small problem.
#ifndef ILQG_WAVE_MAP
#define ILQG_WAVE_MAP (N_X > 8)
#endif
#if ILQG_WAVE_MAP && !defined(ILQG_NO_STAGED_RECORDS)
#define ILQG_REC(member, index) ilqg_stage_at((unsigned)((offsetof(trajEl_t, member) / sizeof(double) + (index)) & 63u))
#define ILQG_REC_DONE(member, first, count) ilqg_stage_flush(p, (unsigned)(offsetof(trajEl_t, member) / sizeof(double) + (first)), count);
// (the products of the factored tensors: bp_tensor_basis(t->fxx, t, ...) — the kernel passes the start of fxx)
#define ILQG_BASIS(index) ILQG_REC(fxx, index)
#define ILQG_BASIS_DONE(count) ILQG_REC_DONE(fxx, 0, count)
#define ILQG_STAGED_RECORDS 1
#else
#define ILQG_STAGED_RECORDS 0
#endif
#include "iLQG_func.c"
#pragma clang attribute pop
#pragma clang attribute pop
}
#undef sin
#undef cos
// el_t: what the kernels hand the callbacks as `trajEl_t *`; trajEl_t: the record in HBM (the header's struct) either way
#if ILQG_DEV_EL
#undef trajEl_t
#undef traj_t
#undef memset
typedef trajEl_dev_t el_t;
typedef traj_dev_t eltraj_t;
typedef trajEl_layout_t trajEl_t;
#else
typedef trajEl_t el_t;
typedef traj_t eltraj_t;
#endif
// The function file's own macros end here: the reference's template leaves `mcond`, `sec`, `csc` and one `aux_<name>` /
// `daux_<name>` / `mu_<kind>_<i>` per auxiliary and multiplier defined to the end of the translation unit
// (iLQG_func.tem:5-30), which on the host is the end of the file and here would be the kernels.  The list is made from
// the file at build time (csrc/Makefile); ILQG_* names, the additive surface the kernels ask for, stay.
#include "ilqg_problem_undefs.h"

// Measured negative results that are kept tested — the fused backward pass on two wavefronts (k_backward_split), the
// derivative record in parts (k_derivs_parts), the box QP's pattern tables (box_qp<M, true>) — are compiled into the
// -DILQG_EXPERIMENTS=1 libraries only (csrc/Makefile EXP_LIBS, lib*_exp.so): the product libraries, and every
// measure-change-measure loop on them, do not pay for their compilation.
#ifndef ILQG_EXPERIMENTS
#define ILQG_EXPERIMENTS 0
#endif
#include "ilqg_device.hpp"
#include "ilqg_wave.hpp"
#include "ilqg_row.hpp"
#if ILQG_WAVE_MAP
#include "ilqg_quad.hpp"  // 16-lane rows; only the wave mapping uses them
#endif
#include "ilqg_shim.h"

namespace {

using namespace ilqg;

constexpr int NX = N_X, NU = N_U;
constexpr bool FULL = FULL_DDP != 0;
// (a pair without the header's hint whose limitsU() stores nothing but zeros as the limits' gradients — read off the function
// file at build time, tools/gen_record_dev.py limits_state_free — is a pair with state-independent limits)
#if !defined(ILQG_STATE_DEPENDENT_LIMITS) && ILQG_DEV_LIMITS_STATE_FREE
#define ILQG_STATE_DEPENDENT_LIMITS 0
#endif
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
    // quad mapping with speculative retries (k_backward_quad<FACT, true>; null otherwise), per launch:
    unsigned *spec_word; //   [B] attempts handed out (low 16 bits) | attempts running (high 16 bits)
    unsigned *spec_best; //   [B] smallest attempt that ended with a result (0xffffffff: none yet)
    unsigned *spec_done; //   [B] 1 once the trajectory's result has been written
    unsigned *spec_out;  //   [B][SPEC_ATTEMPTS] how attempt j ended: 0 unknown, else kind | direct << 2 | row << 3
    int *spec_row_b;     //   [rows] the trajectory a row works on (-1: none)
    double *spec_res;    //   [rows][8] what a row's attempt with a result leaves: lambda, dlambda, dV0, dV1, g_norm, status, bp_rc, sweeps
    double *spec_gains;  //   [rows][N][NU + NXU] gains of a row's attempt that does not write the trajectory's records
    int spec_rows;
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
__device__ __forceinline__ int run_guarded(Fn &&f, double *alone = nullptr) {
    int r = 1;
    for(int a = 0; a <= 64; a++) {
        if(alone) *alone = (a > 0) ? 1.0 : 0.0;  // (the hook of staged record entries: ilqg_stage_flush)
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
    H.alone = 0.0;
    T.ptr[ILQG_HOOK_ALONE] = &H.alone;
    H.record = nullptr;
    T.ptr[ILQG_HOOK_RECORD] = reinterpret_cast<double *>(&H.record);
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
#include "k_lane_backward.inc"  // k_derivs, k_backward, k_backward_split, k_pack_records
#else
#include "k_wave_backward.inc"  // k_derivs_wave, k_derivs_parts, k_backward_wave, k_backward_quad
#endif

#include "k_rollout.inc"  // k_rollout

#if !ILQG_WAVE_MAP
#include "k_lane_search.inc"  // k_search and its adoption kernels (ls_keep = 2)
#endif

#if ILQG_WAVE_MAP && defined(ILQG_ROLLOUT_PARTS)
#include "k_wave_rollout.inc"  // k_rollout_parts
#endif

#include "k_common.inc"  // k_select, k_adopt, k_update, k_multipliers, unit-test kernels

}  // namespace

#ifndef ILQG_ONLY_KERNEL
#include "ilqg_shim_impl.inc"
#else
// kernel experiments (tools/one_kernel.sh): the device code of ONE kernel, e.g.
// -DILQG_ONLY_KERNEL='k_backward_quad<true>(DevPtrs, ilqg_dev_opts_t, int, int, int)' — no shim, nothing else instantiated
namespace {
template __global__ void ILQG_ONLY_KERNEL;
}
#endif
