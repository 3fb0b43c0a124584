/* Batched iLQG on MI355X — public C-ABI (additive; the reference has no batch mode).
 *
 * The library is built per problem (like the reference, whose N_X/N_U are
 * compile-time macros of the generated header, iLQG_problem.tem:16-21):
 *     libilqg_<problem>_fd<FULL_DDP>_hip.so
 * It exports
 *   (1) the reference's own link-time symbols — iLQG(), back_pass(),
 *       line_search(), boxQP(), standard_parameters(), setOptParam(),
 *       makeCandidateNominal(), printParams() (include/iLQG.h, back_pass.h,
 *       line_search.h, boxQP.h) — operating on one `tOptSet`, with the hot path
 *       executed by the HIP kernels, and
 *   (2) the batch interface below: B independent trajectories of the same
 *       problem advanced in lock step on one GPU, all state resident in HBM.
 *
 * Every entry point takes plain pointers and sizes.  Host arrays are
 * trajectory-major: x is [B][n_hor+1][N_X], u is [B][n_hor][N_U], l is
 * [B][n_hor][N_U], L is [B][n_hor][N_U*N_X] (each step an N_U x N_X
 * column-major matrix, reference iLQG_func.tem:152) — for B = 1 exactly the
 * column-major x_new(n,N), u_new(m,N-1) matrices of the reference's MEX entry
 * (iLQG_mex.c:93-97,127-137).
 *
 * Functions returning int: 0 = ok, non-zero = error, text via ilqg_batch_error().
 */
#ifndef ILQG_BATCH_H
#define ILQG_BATCH_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ilqg_batch ilqg_batch_t;

/* problem facts of this build: out[0..7] = N_X, N_U, FULL_DDP, derivative
 * record size (host view), record size stored on the device, 1 if input limits
 * depend on the state, number of problem parameters, mapping (0 = one lane per
 * trajectory, 1 = one wavefront per trajectory; chosen at build time from N_X) */
void ilqg_problem_dims(int *out);
/* problem parameters, as the generated paramdesc[] declares them (iLQG_func.tem:11-18);
 * size -1 = one value per time step (n_hor+1 values) */
const char *ilqg_problem_param_name(int i);
int ilqg_problem_param_size(int i);

/* number of HIP devices visible; 0 if none */
int ilqg_device_count(void);

ilqg_batch_t *ilqg_batch_create(int device, int batch, int n_hor); /* NULL on failure (ilqg_batch_error(NULL)) */
/* The batch is advanced as `groups` independent sets of consecutive trajectories, each on its own HIP stream,
 * so that the latency-bound kernels of one set overlap with the throughput-bound ones of another.  Results do
 * not depend on it.  groups = 0 (what ilqg_batch_create passes): ILQG_GROUPS from the environment, else 4 for
 * batches >= 8192 in the one-lane-per-trajectory mapping (measured best at 65 536 trajectories), else 1.  At most 4. */
ilqg_batch_t *ilqg_batch_create_groups(int device, int batch, int n_hor, int groups);
/* Builds with one wavefront per trajectory (N_X > 8) keep the derivative records of the trajectories in flight in ONE
 * work buffer per device, shared by all batches on it, allocated by the first and released with the last: sized to hold
 * the first batch in the record form ilqg_batch_iterate uses if the device has the memory, else what is free after
 * that batch's other arrays (at least half of what is free).  ILQG_WORK_GB in the environment sets its size. */
int ilqg_batch_groups(const ilqg_batch_t *c);
void ilqg_batch_destroy(ilqg_batch_t *c);
const char *ilqg_batch_error(const ilqg_batch_t *c);

/* options: same keys, validation and messages as the reference's setOptParam
 * (iLQG.c:91-216); additionally
 *   "resweep"     0/1: repeat the reference's cost-only sweep after each accepted step
 *                 (iLQG.c:338).  Default 1 for problems with multipliers, 0 otherwise: without
 *                 multipliers that sweep returns the cost of the accepted roll-out bit for bit.
 *   "fuse_derivs" 0/1: ilqg_batch_iterate/solve evaluate the derivatives inside the backward
 *                 kernel instead of materialising the records in HBM.  Default 1, 0 for problems
 *                 with multipliers (measured faster there); same results either way.
 *   "ls_split"    default 4 (1 for builds with one wavefront per trajectory): step sizes
 *                 alpha[0..ls_split) are rolled out for every trajectory,
 *                 the remaining ones only for trajectories that found none acceptable among
 *                 them; 0 = all step sizes for every trajectory.  The accepted step size is the
 *                 same either way (first acceptable, line_search.c:37-60).
 *   "ls_keep"     default 2.  One lane per trajectory (first stages of up to 4 step sizes): every
 *                 roll-out of the line search is kept where it is rolled out and the accepted one becomes the
 *                 current trajectory by a change of its location index — no second roll-out, no copy.  One
 *                 wavefront per trajectory, generated file with the step in parts: both stages keep their
 *                 roll-outs and the accepted ones are copied (no second roll-out); else as 1.
 *                 1: the second stage runs side by side with the
 *                 roll-out that stores the accepted trajectories of the first stage, and keeps what it rolls
 *                 out, so that its own accepted trajectories are copied instead of rolled out once more;
 *                 0: second stage, then one storing roll-out for all.  Same results in all three.
 *   "bw_split"    default 0.  1: the fused backward pass runs on two wavefronts per 64 trajectories
 *                 (derivatives of step k-1 on one, Riccati update of step k on the other, hand-over in
 *                 LDS) where the problem allows it (no multipliers, constant limits).  Same results;
 *                 measured no faster.
 *   "compact"     default 0.  n > 0: ilqg_batch_solve retires finished trajectories — between iterations, once at most
 *                 half of the slots it iterates over are still active (and at least n are), the active trajectories are
 *                 gathered into a smaller context and iterated there; they return to their own slots of this batch when
 *                 they are gathered again or the solve ends, so every getter reads this batch as before.  Same results
 *                 bit for bit (a trajectory's iterations depend on nothing but its own state); a solve whose
 *                 trajectories converge at very different iterations (CarParking: 50 to 550) does not run 64-lane
 *                 wavefronts for one live lane.  Reference: the loop exits of iLQG.c:297-303, :331, :365-378.
 * Defaults = standard_parameters() (iLQG.c:57-78). */
int ilqg_batch_set_option(ilqg_batch_t *c, const char *name, const double *value, int n);
/* problem parameter by name, shared by all trajectories (iLQG_mex.c:70-84) */
int ilqg_batch_set_param(ilqg_batch_t *c, const char *name, const double *value, int n);

/* initial conditions and initial controls, then the initial roll-out
 * (forward_pass with alpha = 0, which also clamps u; iLQG_mex.c:113-120)
 * and the solver's entry state (iLQG.c:226-237) */
int ilqg_batch_set_x0(ilqg_batch_t *c, const double *x0 /* [B][N_X] */);
int ilqg_batch_set_u(ilqg_batch_t *c, const double *u /* [B][n_hor][N_U] */);
/* overwrite the whole nominal state trajectory (tests: re-synchronise with a checker) */
int ilqg_batch_set_x(ilqg_batch_t *c, const double *x /* [B][n_hor+1][N_X] */);
int ilqg_batch_init(ilqg_batch_t *c);

/* n lock-step iterations of { calc_derivs, back_pass (+ lambda retries),
 * line_search over all alpha, accept/reject } for every still-active trajectory */
int ilqg_batch_iterate(ilqg_batch_t *c, int n);
/* iterate until no trajectory is active (at most max_iter iterations) */
int ilqg_batch_solve(ilqg_batch_t *c);
int ilqg_batch_sync(ilqg_batch_t *c);
int ilqg_batch_active(ilqg_batch_t *c, int *n_active);
/* A STREAM of `total` starts solved through this batch's slots: finished trajectories are harvested every 8 iterations and
 * their slots given to the next starts of the stream (initialised in a staging context as ilqg_batch_init does, then moved
 * in), so that the slots stay full while starts remain.  x0 [total][N_X], u0 [total][n_hor][N_U] in; per start out: cost,
 * status (exit reason, see below), iterations, and — if not NULL — x [total][n_hor+1][N_X], u [total][n_hor][N_U].
 * Every start gets the result a plain ilqg_batch_solve of a batch holding it gives, bit for bit.  Options and parameters:
 * those of c; what c held before is overwritten.  ilqg_batch_solve_trace reports the polls (its `compactions` counts the
 * refills).  No reference counterpart (the reference solves one trajectory per call, iLQG.c:224). */
int ilqg_batch_solve_stream(ilqg_batch_t *c, int total, const double *x0, const double *u0, double *cost, int *status,
                            int *iterations, double *x, double *u);
/* the last ilqg_batch_solve, poll by poll (it polls every 4 iterations): iterations done so far, trajectories still
 * active, slots the iterations ran over (the batch, or the smaller context of option "compact"); returns the number of
 * polls (at most cap entries are written; any pointer may be NULL), *compactions = how often the active set was gathered */
int ilqg_batch_solve_trace(ilqg_batch_t *c, int *iterations, int *active, int *slots, int cap, int *compactions);

/* single stages, for tests and for callers that interleave their own work */
int ilqg_batch_calc_derivs(ilqg_batch_t *c);
/* mode 0: derivative records from HBM, with the lambda retry loop and the gradient test (iLQG.c:261-303);
 *      1: records from HBM, exactly one sweep (what the drop-in back_pass() runs);
 *      2: as 0, derivatives evaluated inside the kernel from (x,u) (what ilqg_batch_iterate uses when
 *         option "fuse_derivs" is 1) */
int ilqg_batch_back_pass(ilqg_batch_t *c, int mode);
int ilqg_batch_line_search(ilqg_batch_t *c);  /* search + selection + store the winner */
int ilqg_batch_update(ilqg_batch_t *c);

/* results (copied to host, trajectory-major) */
int ilqg_batch_get_x(ilqg_batch_t *c, double *x);
int ilqg_batch_get_u(ilqg_batch_t *c, double *u);
int ilqg_batch_get_gains(ilqg_batch_t *c, double *l, double *L);
int ilqg_batch_get_derivs(ilqg_batch_t *c, double *rec /* [B][n_hor][record] */, double *fin /* [B][N_X+sizeofQxx] */);
/* Augmented-Lagrangian multipliers of problems with hle / hli / hfe / hfi constraints (reference
 * iLQG_problem.tem:70-89, iLQG_func.tem:371-521): out[0..1] = doubles in multipliersEl_t / multipliersFin_t
 * (0, 0 for a problem without such constraints).  The arrays hold the structs member by member:
 * running [B][n_hor][out[0]], final [B][out[1]]; either pointer may be NULL.  The current penalty weights
 * are the per-trajectory scalars "w_pen_l" / "w_pen_f" of ilqg_batch_get_scalar. */
void ilqg_problem_multiplier_dims(int *out);
int ilqg_batch_get_multipliers(ilqg_batch_t *c, double *running, double *final);
int ilqg_batch_set_multipliers(ilqg_batch_t *c, const double *running, const double *final);
int ilqg_batch_set_derivs(ilqg_batch_t *c, const double *rec, const double *fin);
int ilqg_batch_set_gains(ilqg_batch_t *c, const double *l, const double *L);
/* name in: cost new_cost dcost expected lambda dlambda g_norm dV0 dV1 ([B] each),
 * alpha_cost ([B][16]) */
int ilqg_batch_get_scalar(ilqg_batch_t *c, const char *name, double *out);
int ilqg_batch_set_scalar(ilqg_batch_t *c, const char *name, const double *in);
/* name in: status iterations alpha_idx accepted bp_calls bp_rc need_derivs ([B] each), alpha_ok ([B][16]) */
int ilqg_batch_get_int(ilqg_batch_t *c, const char *name, int *out);
int ilqg_batch_set_int(ilqg_batch_t *c, const char *name, const int *in);

/* Per-trajectory exit reasons ("status" of ilqg_batch_get_int): 0 active; 1 gradient test passed (iLQG.c:297-303);
 * 2 accepted step with dcost < tolFun (iLQG.c:330-335); 3 max_iter iterations done (iLQG.c:372-376); 4 backward pass:
 * lambda > lambdaMax (iLQG.c:273-274); 5 rejected step: lambda > lambdaMax (iLQG.c:356-360); 6 NaN/Inf in calc_derivs
 * (iLQG.c:247-249); 7 NaN/Inf in the initial roll-out (iLQG_mex.c:116).
 * ilqg_reference_success: what the reference's iLQG() returns for that exit (1 for 1, 2, 5; 0 for 3, 4, 7; for 6 the
 * back-pass flag of the previous iteration is still set, iLQG.c:247-249 and 365-378: 1 unless no iteration was done). */
int ilqg_reference_success(int status, int iterations);

/* For a collective over device memory: a per-trajectory scalar of the whole batch ("cost", ...) copied device to
 * device into `dst_device` (batch doubles, contiguous); synchronises.  ilqg_batch_cost_device_ptr is the address
 * of the cost vector itself while the batch is ONE group (NULL otherwise); ilqg_batch_stream the HIP stream of
 * the first group. */
int ilqg_batch_scalar_to_device(ilqg_batch_t *c, const char *name, void *dst_device);
void *ilqg_batch_cost_device_ptr(ilqg_batch_t *c);
void *ilqg_batch_stream(ilqg_batch_t *c);

/* ---- several GPUs of one node, ONE process (no reference counterpart) -----------------------------
 * The batch is sharded in contiguous blocks of ceil(batch / n_devices) trajectories; device g advances its block with
 * the interface above and shares nothing with the others.  The single exchange is ilqg_multi_gather_costs: one
 * ncclGather (RCCL, xGMI) of the per-trajectory costs to the first device, delivered to the host.  devices = NULL:
 * devices 0..n_devices-1.  Host arrays are those of the batch interface for the WHOLE batch. */
#define ILQG_MULTI_MAX 16
typedef struct ilqg_multi ilqg_multi_t;
ilqg_multi_t *ilqg_multi_create(int n_devices, const int *devices, int batch, int n_hor); /* NULL: ilqg_multi_error(NULL) */
void ilqg_multi_destroy(ilqg_multi_t *m);
const char *ilqg_multi_error(const ilqg_multi_t *m);
int ilqg_multi_devices(const ilqg_multi_t *m);
/* the batch object of device g and its block of the batch (for everything not forwarded below) */
ilqg_batch_t *ilqg_multi_shard(ilqg_multi_t *m, int g, int *first, int *count);
int ilqg_multi_set_option(ilqg_multi_t *m, const char *name, const double *value, int n);
int ilqg_multi_set_param(ilqg_multi_t *m, const char *name, const double *value, int n);
int ilqg_multi_set_x0(ilqg_multi_t *m, const double *x0);
int ilqg_multi_set_u(ilqg_multi_t *m, const double *u);
int ilqg_multi_init(ilqg_multi_t *m);
int ilqg_multi_iterate(ilqg_multi_t *m, int n);   /* asynchronous on every device; the devices are served in turn */
int ilqg_multi_solve(ilqg_multi_t *m);
int ilqg_multi_sync(ilqg_multi_t *m);
int ilqg_multi_active(ilqg_multi_t *m, int *n_active);
int ilqg_multi_get_x(ilqg_multi_t *m, double *x);
int ilqg_multi_get_u(ilqg_multi_t *m, double *u);
int ilqg_multi_get_int(ilqg_multi_t *m, const char *name, int *out);
int ilqg_multi_gather_costs(ilqg_multi_t *m, double *cost /* [batch] */);

/* The reference's MEX entry for a C caller (iLQG_mex.c:19-144):
 *     [success, x, u, cost] = iLQG<Problem>(x0, u_nom, params, opts)
 * One trajectory through the drop-in iLQG() (outer loop on the host, back_pass() / line_search() on the GPU).
 * params: every parameter of the problem by name (length checked against paramdesc[]); opts: setOptParam keys.
 * x [n_hor+1][N_X], u [n_hor][N_U], cost, iterations, seconds (wall time of iLQG() alone, iLQG_mex.c:123-126) out.
 * Returns iLQG()'s 1 / 0, or -1 with the MEX entry's message in err when an argument is refused. */
typedef struct {
    const char *name;
    const double *value;
    int n;
} ilqg_named_t;
int ilqg_solve_single(int n_hor, const double *x0, const double *u_nom, const ilqg_named_t *params, int n_params_given,
                      const ilqg_named_t *opts, int n_opts, double *x, double *u, double *cost, int *iterations,
                      double *seconds, char *err, int err_len);

/* per-kernel device time measured with HIP events on the context's stream */
int ilqg_batch_timing(ilqg_batch_t *c, int enable);
int ilqg_batch_kernel_count(void);
const char *ilqg_batch_kernel_name(int kernel);
int ilqg_batch_get_timing(ilqg_batch_t *c, int kernel, int *launches, double *total_ms);
/* ms of wall clock the kernel occupied since timing was switched on: the union of its launch intervals per group of
 * trajectories, summed over the groups (launches on a group's two streams overlap in the event clock) */
int ilqg_batch_get_busy(ilqg_batch_t *c, int kernel, double *busy_ms);

/* device box-QP on `count` independent problems of size n in {2, 8, N_U} (unit tests) */
int ilqg_boxqp_batch(int device, int n, int count, const double *H, const double *g, const double *lower,
                     const double *upper, double *x, int *clamp, int *n_free, double *invH, int *rc);

/* the same problems through the cooperative form the one-wavefront-per-trajectory mapping uses
 * (one problem per wavefront, one lane per variable); same results bit for bit */
int ilqg_boxqp_wave_batch(int device, int n, int count, const double *H, const double *g, const double *lower,
                          const double *upper, double *x, int *clamp, int *n_free, double *invH, int *rc);

/* ... and through the per-lane form that factorises every clamp pattern up front (n = 2, or N_U when N_U <= 3) */
int ilqg_boxqp_table_batch(int device, int n, int count, const double *H, const double *g, const double *lower,
                           const double *upper, double *x, int *clamp, int *n_free, double *invH, int *rc);

/* device sin/cos as the generated callbacks see them, on n arguments (unit tests) */
int ilqg_sincos_batch(int device, int n, const double *x, double *s, double *c);

#ifdef __cplusplus
}
#endif
#endif
