/* Thin extern-"C" boundary between the C host code (ilqg_host.c) and the HIP
 * kernels (ilqg_kernels.hip).  Plain pointers, ints and doubles only.
 *
 * Host-side data crosses this boundary in trajectory-major layout
 * [trajectory][time step][field], i.e. what a caller holding `trajEl_t`
 * arrays can produce with a memcpy per field (reference iLQG_mex.c:113-137
 * walks its trajectories the same way).  The device layouts (packed per-step
 * trajectory records, tiled arrays) are private to ilqg_kernels.hip, see
 * DESIGN.md section 2.
 */
#ifndef ILQG_SHIM_H
#define ILQG_SHIM_H

#ifdef __cplusplus
extern "C" {
#endif

#define ILQG_MAX_ALPHA 16

typedef struct ilqg_dev ilqg_dev_t;

/* solver options, same meaning as the fields of tOptSet (reference iLQG.h:37-76) */
typedef struct {
    int n_alpha;
    double alpha[ILQG_MAX_ALPHA];
    double tolFun, tolGrad, tolConstraint;
    double lambdaInit, dlambdaInit, lambdaFactor, lambdaMax, lambdaMin;
    double zMin;
    int regType;
    int max_iter;
    double w_pen_init_l, w_pen_init_f, w_pen_max_l, w_pen_max_f, w_pen_fact1, w_pen_fact2;
    int resweep; /* 1: repeat the reference's cost-only sweep after an accepted step (iLQG.c:338) */
    int ls_split; /* step sizes rolled out for every trajectory in the first line-search stage; the rest only
                   * for trajectories still without an acceptable one (0 or >= n_alpha: single stage) */
    int ls_keep; /* 2 (lane mapping; the default there): every roll-out of the line search is KEPT where it is rolled out
                  * and an accepted trajectory becomes the current one by a change of its location index (k_search,
                  * k_commit): no second roll-out of the winner, no copy;
                  * 1: the second stage keeps the trajectories it rolls out and runs side by side with the winner pass
                  * of the first stage's trajectories; 0: second stage, then one winner pass for all */
    int bw_split; /* 1: the fused backward pass runs on two wavefronts per 64 trajectories (derivatives of step k-1 on
                   * one, Riccati update of step k on the other) where the problem allows it; 0: on one */
    int fuse_derivs; /* 1: ilqg_dev_iterate evaluates derivatives inside the backward kernel (no records in HBM) */
} ilqg_dev_opts_t;

/* per-trajectory status of the lock-step solver */
enum {
    ILQG_ST_ACTIVE = 0,
    ILQG_ST_CONVERGED_GRAD = 1, /* g_norm < tolGrad && lambda < 1e-5      (iLQG.c:297-303) */
    ILQG_ST_CONVERGED_FUN = 2,  /* accepted step with dcost < tolFun      (iLQG.c:330-335) */
    ILQG_ST_MAX_ITER = 3,       /* max_iter iterations done               (iLQG.c:372-376) */
    ILQG_ST_NO_DESCENT = 4,     /* back pass: lambda > lambdaMax          (iLQG.c:273-274) */
    ILQG_ST_LAMBDA_MAX = 5,     /* rejected step: lambda > lambdaMax      (iLQG.c:356-360) */
    ILQG_ST_DERIVS_FAILED = 6,  /* NaN/Inf in calc_derivs                 (iLQG.c:247-249) */
    ILQG_ST_INIT_FAILED = 7     /* NaN/Inf in the initial roll-out        (iLQG_mex.c:116) */
};

/* double-valued device fields.  [N] = n_hor */
enum {
    ILQG_F_X = 0,    /* [N+1][N_X]   nominal states                     */
    ILQG_F_U,        /* [N][N_U]     nominal controls                   */
    ILQG_F_LG,       /* [N][N_U]     feed-forward gains l               */
    ILQG_F_KG,       /* [N][N_U*N_X] feedback gains L (column-major)    */
    ILQG_F_DER,      /* [N][host record] derivative records             */
    ILQG_F_FIN,      /* [N_X+sizeofQxx] final cx, cxx                   */
    ILQG_F_MUL,      /* [N][multipliersEl_t as doubles] running multipliers (problems with hle / hli) */
    ILQG_F_MULF,     /* [multipliersFin_t as doubles]   final multipliers (problems with hfe / hfi)   */
    ILQG_F_COST,     /* scalars per trajectory from here on             */
    ILQG_F_NEW_COST,
    ILQG_F_DCOST,
    ILQG_F_EXPECTED,
    ILQG_F_LAMBDA,
    ILQG_F_DLAMBDA,
    ILQG_F_GNORM,
    ILQG_F_DV0,
    ILQG_F_DV1,
    ILQG_F_WPEN_L,   /* current penalty weights o->w_pen_l / o->w_pen_f (iLQG.h:78) */
    ILQG_F_WPEN_F,
    ILQG_F_WPEN_L_DER, /* the weights the current derivatives were evaluated with: a rejected step may raise  */
    ILQG_F_WPEN_F_DER, /* w_pen_* (iLQG.c:345-349) while the derivatives stay those of the last accepted step */
    ILQG_F_ALPHA_COST, /* [n_alpha] cost of every step size of the last line search */
    ILQG_F_COUNT
};

/* int-valued device fields, one value per trajectory unless noted */
enum {
    ILQG_I_STATUS = 0,
    ILQG_I_ITER,        /* reference's o->iterations                     */
    ILQG_I_NEED_DERIVS,
    ILQG_I_ALPHA_IDX,   /* 1-based accepted step index, n_alpha+1 = none */
    ILQG_I_ACCEPTED,
    ILQG_I_BP_CALLS,    /* backward sweeps in the last iteration         */
    ILQG_I_BP_RC,       /* result of the last backward sweep: 0 ok, 1 failed */
    ILQG_I_RESWEEP,     /* set by the update: 2 = update multipliers + cost sweep (iLQG.c:337-338), 1 = cost
                         * sweep after a rejected step raised the penalty weights (iLQG.c:345-349), 0 = none */
    ILQG_I_LOC,         /* lane mapping, ls_keep = 2: where the CURRENT (x, u) of the trajectory lives: 0 = the arrays
                         * X / U, p > 0 = plane p - 1 of the kept roll-outs of the line search (see cur_x) */
    ILQG_I_ALPHA_OK,    /* [n_alpha] forward pass finite?                */
    ILQG_I_COUNT
};

/* kernels, for timing queries */
enum {
    ILQG_K_DERIVS = 0,
    ILQG_K_BACKWARD,
    ILQG_K_ROLLOUT_SEARCH,
    ILQG_K_SELECT,
    ILQG_K_ROLLOUT_WINNER,
    ILQG_K_UPDATE,
    ILQG_K_ROLLOUT_COST,
    ILQG_K_ROLLOUT_INIT,
    ILQG_K_TRANSPOSE,
    ILQG_K_BACKWARD_FUSED,
    ILQG_K_ROLLOUT_SEARCH2,
    ILQG_K_MULTIPLIERS,
    ILQG_K_SEARCH,   /* ls_keep = 2: k_search on all trajectories (first stage) */
    ILQG_K_SEARCH2,  /*             k_search on the pending list (second stage) */
    ILQG_K_ADOPT,    /*             k_adopt_home / k_rejected_home, k_commit */
    ILQG_K_COUNT
};

/* all functions return 0 on success, non-zero on error (message via ilqg_dev_error) */
const char *ilqg_dev_error(void);
int ilqg_dev_count(void);
int ilqg_dev_create(ilqg_dev_t **out, int device, int batch, int n_hor);
void ilqg_dev_destroy(ilqg_dev_t *d);

/* compile-time facts of the problem this library was built for:
 * out[0..7] = N_X, N_U, FULL_DDP, host record size, device record size, state-dependent limits, n_params,
 * mapping (0 = one lane per trajectory, 1 = one wavefront per trajectory) */
void ilqg_dev_dims(int *out);
/* out[0..1] = doubles in multipliersEl_t / multipliersFin_t (0 for a problem without such constraints) */
void ilqg_dev_multiplier_dims(int *out);

int ilqg_dev_set_params(ilqg_dev_t *d, int n_params, const int *sizes, const double *const *values);
int ilqg_dev_set_opts(ilqg_dev_t *d, const ilqg_dev_opts_t *o);

/* host <-> device, host side trajectory-major [batch][steps][width] */
int ilqg_dev_write(ilqg_dev_t *d, int field, const double *host);
int ilqg_dev_read(ilqg_dev_t *d, int field, double *host);
/* only the first `steps` time steps of a field, host [batch][steps][width] (e.g. x0 = step 0 of X) */
int ilqg_dev_write_steps(ilqg_dev_t *d, int field, const double *host, int steps);
int ilqg_dev_write_int(ilqg_dev_t *d, int field, const int *host);
int ilqg_dev_read_int(ilqg_dev_t *d, int field, int *host);
/* Deferred transfers: between begin and end the write / read calls above return without waiting (each uses its own
 * slice of a pinned staging buffer); ilqg_dev_io_end waits once for everything on the stream and only then are the
 * arrays of the read calls filled.  Kernels launched in between run in stream order with the transfers. */
int ilqg_dev_io_begin(ilqg_dev_t *d);
int ilqg_dev_io_end(ilqg_dev_t *d);
int ilqg_dev_field_width(int field);                 /* doubles per step and trajectory (host view) */
int ilqg_dev_field_steps(ilqg_dev_t *d, int field);  /* time steps stored */
/* device address of a per-trajectory scalar field (for collectives on device memory) */
void *ilqg_dev_field_ptr(ilqg_dev_t *d, int field);
void *ilqg_dev_stream(ilqg_dev_t *d);
/* a per-trajectory scalar field (batch doubles) into device memory of the caller, on the context's stream */
int ilqg_dev_copy_scalar_to(ilqg_dev_t *d, int field, void *dst_device);

/* stages (all asynchronous on the context's stream) */
int ilqg_dev_reset(ilqg_dev_t *d);            /* solver entry state (iLQG.c:226-237) */
int ilqg_dev_rollout_init(ilqg_dev_t *d);     /* forward_pass(alpha = 0) + swap (iLQG_mex.c:113-120) */
int ilqg_dev_derivs(ilqg_dev_t *d);           /* calc_derivs for trajectories that need it */
/* back_pass.  mode 0: records from HBM + lambda retry loop + gradient test; 1: records from HBM, one
 * sweep (drop-in back_pass()); 2: as 0 with the derivatives evaluated on the fly (no k_derivs needed) */
int ilqg_dev_backward(ilqg_dev_t *d, int mode);
int ilqg_dev_search(ilqg_dev_t *d);           /* all step sizes in parallel + first-acceptable selection */
int ilqg_dev_winner(ilqg_dev_t *d);           /* re-roll the accepted step size, storing the trajectory */
int ilqg_dev_update(ilqg_dev_t *d);           /* accept/reject bookkeeping (iLQG.c:311-361) */
int ilqg_dev_iterate(ilqg_dev_t *d, int n);   /* n lock-step iterations */
int ilqg_dev_sync(ilqg_dev_t *d);
int ilqg_dev_count_active(ilqg_dev_t *d, int *n_active); /* synchronises */
/* solver state of trajectories from[0..count) of src -> trajectories to[0..count) of dst (same device and horizon): current
 * x / u (they arrive in dst's arrays X / U), records, scalars, integers, multipliers; with_records: also the stored
 * derivative records (contexts iterated with fuse_derivs = 0).  Synchronises both contexts. */
int ilqg_dev_move(ilqg_dev_t *dst, ilqg_dev_t *src, int count, const int *to, const int *from, int with_records);

/* builds with -DILQG_PROFILE_SECTIONS: cycles per section of the fused backward step, summed over wavefronts */
int ilqg_dev_section_cycles(unsigned long long *out8);
int ilqg_dev_derivs_cycles(unsigned long long *out32);
/* quad mapping with speculative retries (ILQG_QUAD_SPEC=1): the protocol's words after the last backward launch (measurement) */
int ilqg_dev_spec_words(unsigned *out, size_t max_words, size_t *count);
/* builds with -DILQG_WAVE_PLACES: where and when the wavefronts of the backward and search kernels ran (3 words per entry) */
int ilqg_dev_wave_places(unsigned long long *out, int max_entries, int *count);

/* per-kernel HIP-event timing on the context's stream */
int ilqg_dev_timing(ilqg_dev_t *d, int enable);
int ilqg_dev_get_timing(ilqg_dev_t *d, int kernel, int *launches, double *total_ms);
int ilqg_dev_get_busy(ilqg_dev_t *d, int kernel, double *busy_ms);  /* union of the kernel's launch intervals */
const char *ilqg_dev_kernel_name(int kernel);

/* unit-test entry for the device box-QP (same template the backward kernel uses):
 * `count` independent problems of size n (2 or 8), arrays [count][...] */
int ilqg_dev_boxqp_batch(int device, int n, int count, const double *H, const double *g, const double *lower,
                         const double *upper, double *x, int *clamp, int *n_free, double *invH, int *rc);

/* the same for the wave mapping's cooperative form (one problem per wavefront, one lane per variable) */
int ilqg_dev_boxqp_wave_batch(int device, int n, int count, const double *H, const double *g, const double *lower,
                              const double *upper, double *x, int *clamp, int *n_free, double *invH, int *rc);

/* ... and for the per-lane form with the factorisations of all clamp patterns made up front (n = 2 or N_U <= 3;
 * ilqg_device.hpp chol_pattern_table: measured no faster, off in the kernels, kept tested) */
int ilqg_dev_boxqp_table_batch(int device, int n, int count, const double *H, const double *g, const double *lower,
                               const double *upper, double *x, int *clamp, int *n_free, double *invH, int *rc);

/* the reference's small dense helpers on the device (one problem): op 0 addMulVec, 1 addSquareTri, 2 addMul2Tri
 * (shape 0 = (N_X,N_U), 1 = (N_X,N_X), 2 = (N_U,N_X[,1])), 3 cholesky_tri, 4 cholesky_tri_inv (shape = n).
 * flag: 1 ok, 0 Cholesky pivot <= 0, -1 size not built */
int ilqg_dev_dense(int device, int op, int shape, const double *in0, int n_in0, const double *in1, int n_in1,
                   const double *in2, int n_in2, double *out, int n_out, int *flag);

/* Several GPUs in one process: an RCCL communicator over n distinct devices (ncclCommInitAll) and the path's single
 * collective: the caller fills device g's send buffer (per doubles, e.g. with ilqg_dev_copy_scalar_to) and waits for
 * the copies; ilqg_comm_gather then moves all of them to device 0 by one ncclGather and on to the host. */
typedef struct ilqg_comm ilqg_comm_t;
int ilqg_comm_create(ilqg_comm_t **out, int n, const int *devices, int per);
void ilqg_comm_destroy(ilqg_comm_t *c);
void *ilqg_comm_send_buffer(ilqg_comm_t *c, int g);  /* device g's `per` doubles, on device g */
int ilqg_comm_gather(ilqg_comm_t *c, ilqg_dev_t *const *devs, const int *first, const int *counts, double *host);

/* unit-test entry for the device sincos that the generated callbacks' sin()/cos() are routed through */
int ilqg_dev_sincos_batch(int device, int n, const double *x, double *s, double *c);

#ifdef __cplusplus
}
#endif
#endif
