/*
 * rvll.h — C-ABI of the MI355X-native RV log-likelihood engine ("rvll").
 *
 * This is the drop-in boundary for the one hot path of nicochunger/evidence:
 * the per-live-point Keplerian RV forward model + Gaussian log-likelihood and
 * the unit-cube -> theta prior transform.  Everything here is extern "C",
 * plain pointers and sizes; the caller owns every host buffer, the library
 * owns device memory behind an opaque handle.  One handle = one device + its
 * own HIP streams.  A handle is not thread-safe; distinct handles are
 * independent.
 *
 * What each entry point replaces in the reference (paths relative to the
 * reference checkout):
 *
 *   rvll_create              evidence/rvmodel/__init__.py:23-57,94-154
 *                            (BaseModel/RVModel.__init__: concatenated epoch
 *                            table, instrument ids, planet count, model flags)
 *   rvll_loglike_batch       evidence/rvmodel/__init__.py:157-219 (log_likelihood)
 *                            -> :343-385 (kep_rv) -> :388-463 (modelk)
 *                            -> :466-494 (true_anomaly) -> rvmodel/trueanomaly.c:8-41
 *                            -> :222-273 (drift) -> :59-80 (logL)
 *                            i.e. it is the batched form of the existing FFI
 *                            `int trueanomaly(double*,int,double,double*,int,double)`
 *                            (rvmodel/trueanomaly.h:4) fused with its caller.
 *   rvll_set_priors /
 *   rvll_prior_batch         evidence/priors.py:22-460 (.ppf of every
 *                            distribution) as called from
 *                            evidence/polychord/__init__.py:130-162 and
 *                            evidence/ultranest/__init__.py:125-137
 *   rvll_prior_loglike_batch prior(cube) followed by loglike(theta), one launch
 *   rvll_comm_* / rvll_allgather_logl
 *                            replaces the MPI fan-out owned by the third-party
 *                            samplers (evidence/polychord/__init__.py:21-29,
 *                            176-199) with one RCCL all-gather of per-shard log-L
 *
 * Error convention: every function returns 0 on success or a negative
 * RVLL_E_* code; rvll_last_error() returns a human-readable message for the
 * calling thread's last failure.  Model-level conditions are NOT errors and
 * follow the reference: an invalid orbit (ecc > 1 in the secos/sesin or
 * ecos/esin parametrisation) gives log-L = -1e30
 * (evidence/rvmodel/__init__.py:198-203,430-431,438-439).
 */
#ifndef RVLL_H
#define RVLL_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVLL_VERSION_MAJOR 0
#define RVLL_VERSION_MINOR 2   /* 0.2: rvll_slice_walk takes walker_base; RVLL_FLAG_WANDERED; resident live set */

/* ---- error codes ------------------------------------------------------ */
#define RVLL_OK             0
#define RVLL_E_INVALID     -1   /* bad argument / inconsistent layout        */
#define RVLL_E_NODEVICE    -2   /* no HIP device, or device index invalid    */
#define RVLL_E_HIP         -3   /* a HIP runtime call failed                 */
#define RVLL_E_NOMEM       -4   /* host or device allocation failed          */
#define RVLL_E_NOPRIORS    -5   /* prior transform requested before set_priors */
#define RVLL_E_RCCL        -6   /* RCCL missing or a collective failed       */
#define RVLL_E_UNSUPPORTED -7   /* feature not available in this build       */

/* ---- per-point flag bits written by the log-L kernels ------------------ */
#define RVLL_FLAG_INVALID_ORBIT  1  /* ecc>1 in a derived parametrisation -> logL=-1e30 */
#define RVLL_FLAG_NONCONVERGED   2  /* a Kepler solve hit itmax (trueanomaly.c:32-33 path) */
#define RVLL_FLAG_WANDERED       4  /* a Kepler solve of this point took more than 8 Newton steps.  The reference's
                                     * iteration (Newton from E = M, stop at |dE| <= 1e-4, trueanomaly.c:17-33) only does
                                     * that next to a zero of f' = 1 - e cos E at e >= ~0.97, where it is thrown far out
                                     * and wanders back: where it then stops depends on the last bit of sin / cos, so the
                                     * reference's own log-L there is only defined to ~1e-9 relative (measured on the
                                     * oracle with its libm nudged by one ulp).  The contract: every point
                                     * WITHOUT this bit agrees with the reference to <= 1e-10; a point with it to what
                                     * the reference agrees with itself.  SURVEY 0.3 / 5: the per-point flag the
                                     * reference lacks (it ignores the solver's return code, rvmodel/__init__.py:488-492).
                                     * With RVLL_FLAG_INVALID_ORBIT set, only that bit is reported. */

/* ---- parameter slot: where a model scalar comes from ------------------- */
/* idx >= 0 : free parameter, value = theta[idx]   (theta ordered as sorted(parnames),
 *            evidence/rvmodel/__init__.py:43)
 * idx <  0 : fixed parameter, value = val         (fixedpardict)              */
typedef struct rvll_slot {
    int32_t idx;
    int32_t reserved;
    double  val;
} rvll_slot;

/* parametrisation choices of modelk (evidence/rvmodel/__init__.py:412-456) */
enum { RVLL_K_K1 = 0, RVLL_K_LOGK1 = 1 };                 /* :412-415 */
enum { RVLL_P_PERIOD = 0, RVLL_P_LOGPERIOD = 1 };         /* :417-420 */
enum { RVLL_ECC_DIRECT = 0,                               /* ecc, omega      :441-447 */
       RVLL_ECC_SECOS_SESIN = 1,                          /* sqrt(e)cos/sin  :425-431 */
       RVLL_ECC_ECOS_ESIN = 2 };                          /* e cos/sin       :433-439 */
enum { RVLL_ANOM_MA0 = 0, RVLL_ANOM_ML0 = 1 };            /* :449-454 */

typedef struct rvll_planet {
    int32_t   k_kind;
    int32_t   p_kind;
    int32_t   ecc_kind;
    int32_t   anom_kind;
    rvll_slot k;        /* k1 | logk1                          */
    rvll_slot p;        /* period | logperiod                  */
    rvll_slot e1;       /* ecc   | secos | ecos                */
    rvll_slot e2;       /* omega | sesin | esin                */
    rvll_slot anom;     /* ma0 | ml0                           */
    rvll_slot epoch;    /* planet{n}_epoch                     */
} rvll_planet;

typedef struct rvll_inst {
    rvll_slot offset;   /* {inst}_offset  (rvmodel:187)                       */
    rvll_slot jitter;   /* {inst}_jitter  (rvmodel:189-190); unused if !has_jitter */
} rvll_inst;

/* RVLL_PREC_FP64  : everything fp64, the reference's arithmetic (parity <= 1e-10).
 * RVLL_PREC_MIXED : mean anomaly and its reduction to [-pi,pi] in fp64 (|M| reaches 1e4 rad, which
 *                   fp32 cannot hold), Newton iteration + Keplerian in fp32, residual / chi^2 /
 *                   accumulation in fp64.  NOT parity with the reference: BASELINE.json configs[4]
 *                   tolerance sweep (~1e-7 relative on log-L).
 * RVLL_PREC_FP32  : as MIXED, with the per-epoch residual and chi^2 term in fp32 as well (the sum
 *                   over epochs stays fp64).                                                     */
enum { RVLL_PREC_FP64 = 0, RVLL_PREC_MIXED = 1, RVLL_PREC_FP32 = 2 };

typedef struct rvll_layout {
    int32_t struct_size;       /* = sizeof(rvll_layout), ABI check            */
    int32_t ndim;              /* D = number of free parameters               */
    int32_t nplanets;          /* rvmodel:122-124                             */
    int32_t ninst;             /* number of instruments (datadict keys)       */
    int32_t has_jitter;        /* rvmodel:138-139                             */
    int32_t has_drift;         /* rvmodel:128-129                             */
    int32_t tref_from_data;    /* 1: tref = time[0] (rvmodel:259-260)         */
    int32_t nlinpar;           /* number of linear-activity series (rvmodel:210-212) */
    rvll_slot drift[4];        /* lin, quad, cub, quar (rvmodel:246-253)      */
    rvll_slot tref;            /* drift_tref (rvmodel:257-258)                */
    const rvll_planet* planets;   /* [nplanets]                               */
    const rvll_inst*   insts;     /* [ninst]                                  */
    const rvll_slot*   linpar;    /* [nlinpar] coefficients linpar_X          */
    double  tol;               /* Newton stop rule, 1e-4 (rvmodel:466)        */
    int32_t itmax;             /* 10000 (rvmodel:491)                         */
    int32_t precision;         /* RVLL_PREC_*                                 */
} rvll_layout;

/* ---- priors ------------------------------------------------------------ */
/* Kinds follow the names exported by evidence/priors.py:429-467.            */
enum {
    RVLL_PRIOR_UNIFORM = 0,          /* priors.py:41-42   args xmin,xmax     */
    RVLL_PRIOR_JEFFREYS = 1,         /* :62-63            xmin,xmax          */
    RVLL_PRIOR_MODJEFFREYS = 2,      /* :82-83            x0,xmax            */
    RVLL_PRIOR_UNIFORMFREQUENCY = 3, /* :100-101          xmin,xmax          */
    RVLL_PRIOR_NORMAL = 4,           /* :436 stats.norm   loc,scale          */
    RVLL_PRIOR_LOGNORMAL = 5,        /* :437 stats.lognorm s,loc,scale       */
    RVLL_PRIOR_TRUNCRAYLEIGH = 6,    /* :249-252          sigma,xmax         */
    RVLL_PRIOR_TABLE = 7,            /* piecewise-linear inverse CDF on a host-built
                                        grid: Binormal :118-124, AsymmetricNormal
                                        :195-202, TruncatedUNormal :223-228,
                                        PowerLaw :282-287, DoublePowerLaw :321-326,
                                        Sine :349-354, Log10Normal :138-144.
                                        args: lo,hi (values returned at q==0 / q==1),
                                        flags                                 */
    RVLL_PRIOR_BETA = 8,             /* :397-398 stats.beta.ppf   a,b         */
    RVLL_PRIOR_GAMMA = 9,            /* :424-425 stats.gamma.ppf  alpha,beta  */
    RVLL_PRIOR_ALPHA = 10,           /* :375-376 stats.alpha.ppf  a           */
    RVLL_PRIOR_SORTED_UNIFORM = 11,  /* priors.py:462-467 (pypolychord)  a,b  */
    RVLL_PRIOR_SORTED_LOGUNIFORM = 12,
    RVLL_PRIOR__COUNT
};

#define RVLL_PRIOR_NARGS 6
typedef struct rvll_prior {
    int32_t kind;
    int32_t group;             /* sorted priors: members of one group share an id >= 0 */
    double  args[RVLL_PRIOR_NARGS];
    const double* table_cdf;   /* RVLL_PRIOR_TABLE: sorted knots  [table_n]  */
    const double* table_x;     /*                   values         [table_n]  */
    int32_t table_n;
    int32_t table_post;        /* 0: y ; 1: 10**y  (Log10Normal)              */
} rvll_prior;

/* ---- timing report of the device-resident benchmark -------------------- */
typedef struct rvll_timing {
    double kernel_ms_mean;     /* mean HIP-event time per launch             */
    double kernel_ms_min;
    double kernel_ms_median;
    double total_ms;           /* first launch -> last launch complete       */
    int64_t evals;             /* live points evaluated in total             */
    int32_t launches;
    int32_t points_per_block;  /* launch geometry actually used              */
    int32_t blocks;
    int32_t threads;
} rvll_timing;

typedef struct rvll_handle rvll_handle;

/* ---- lifecycle ---------------------------------------------------------- */
/* Upload the concatenated epoch table once.  Arrays are in the reference's
 * concatenation order (instrument by instrument, evidence/rvmodel/__init__.py:50-55),
 * NOT time-sorted.  inst[j] in [0,ninst).  linpar_series is [nlinpar][Ne]
 * row-major or NULL.  device < 0 selects the current HIP device.            */
int rvll_create(const rvll_layout* layout,
                const double* time, const double* vrad, const double* svrad,
                const int32_t* inst, int32_t n_epochs,
                const double* linpar_series,
                int32_t device, rvll_handle** out);
int rvll_destroy(rvll_handle* h);

/* ---- priors -------------------------------------------------------------- */
int rvll_set_priors(rvll_handle* h, const rvll_prior* priors, int32_t ndim);
/* Beta and Gamma quantiles (scipy.stats.beta/gamma.ppf, evidence/priors.py:397-398, 424-425) have no closed
 * form: rvll_set_priors lets the device tabulate each one once and MEASURES the table's quintic interpolant
 * against the full iterative solver.  max_err = that measured error (relative, in the interpolation
 * coordinate; NaN for parameters without such a table); direct = 1 if it is small enough that elements are
 * evaluated by interpolation alone, 0 if every element still runs a Newton step on the incomplete
 * beta/gamma function.  Diagnostics only.                                                               */
int rvll_prior_table_info(rvll_handle* h, int32_t dim, double* max_err, int32_t* direct);

/* ---- the hot calls (host buffers in, host buffers out) ------------------- */
/* theta: [B, D] row-major.  logL: [B].  flags: [B] or NULL.                  */
int rvll_loglike_batch(rvll_handle* h, const double* theta, int64_t B,
                       double* logL, int32_t* flags);
/* cube: [B, D] in [0,1].  theta: [B, D].                                     */
int rvll_prior_batch(rvll_handle* h, const double* cube, int64_t B, double* theta);
/* fused: one H2D, prior + log-L launches back to back, one D2H.
 * From 24 MB of rows on (165565 points at 19 parameters) the batch is streamed: chunks of 16384 rows through pinned staging blocks, uploads, kernels and
 * downloads on three streams, and the copies between the caller's (pageable) arrays and the blocks on worker
 * threads of the library — 2 to 4, RVLL_COPY_THREADS overrides; started by the first such call of a handle, asleep
 * between calls, joined by rvll_destroy.  The caller's arrays are only touched between entry and return.           */
int rvll_prior_loglike_batch(rvll_handle* h, const double* cube, int64_t B,
                             double* theta_out, double* logL, int32_t* flags);

/* ---- sampler proposal step on the device ---------------------------------------------------------------- */
/* The callers of the path: nested sampling replaces its worst points by new points drawn
 * from the prior inside logL > lstar.  The reference leaves that to UltraNest's region slice sampler
 * (evidence/ultranest/__init__.py:159-175: RegionSliceSampler, nsteps moves per new point, circular omega / ml0);
 * evidence_amd/nested.py does the same with batched callbacks.  This entry point runs the whole walk on the GPU:
 * K walkers start at cube[K, ndim] (theta / logl hold their transformed parameters and log-L, all above lstar)
 * and take nsteps hit-and-run slice moves each: direction = chol * normal / norm (chol: [ndim, ndim] row-major
 * lower-triangular factor of the live points' covariance), chord limited by the unit-cube walls (wrapped[k] != 0:
 * circular parameter, half a turn), candidate = uniform point of the chord -> prior transform -> log-L, accepted
 * if logL > lstar, else the chord shrinks towards the current point (at most max_rounds candidates per move).
 * In/out buffers return the end points; *ncalls = likelihood evaluations spent.  Deterministic for a given
 * seed: the random numbers are counter-based on (seed, walker_base + row, move, draw), so a rank that walks rows
 * [lo, hi) of a larger set with walker_base = lo gets exactly what the unsharded walk gives those rows — a run does
 * not depend on how many GPUs share it.  Needs rvll_set_priors.                                              */
int rvll_slice_walk(rvll_handle* h, double* cube, double* theta, double* logl, int64_t K, double lstar,
                    const double* chol, const int32_t* wrapped /*[ndim] or NULL*/, int32_t nsteps,
                    int32_t max_rounds, uint64_t seed, int64_t walker_base, int64_t* ncalls);
/* A walker's moves are a chain of dependent evaluations; when walkers of a workgroup have finished, the free slots of
 * its tile evaluate, for the walkers that are left, up to max_ahead candidates of the current move per iteration:
 * candidate r+1 is the one the walker draws if candidate r is rejected (the shrunk bracket is known in advance), and
 * the results are consumed in order — so end points, log-L and *ncalls are those of the one-candidate-per-iteration
 * walk, bit for bit; only the number of iterations drops.  Default 4; 1 switches it off.
 * rvll_slice_walk_evaluated: tile slots the last rvll_slice_walk evaluated (>= its ncalls: speculative candidates
 * that went unused are work done, not likelihood calls of the sampler).                                          */
int rvll_set_walk_speculation(rvll_handle* h, int32_t max_ahead);
int rvll_slice_walk_evaluated(rvll_handle* h, int64_t* evaluated);
/* Diagnostic build only (make -C evidence_amd/csrc walktrace; all zeros otherwise): where the workgroups of the last
 * rvll_slice_walk spent their time — 100 MHz ticks summed over workgroups for [0] directions + chord limits,
 * [1] candidates, [2] prior transform + log-L tile, [3] accept / copy / bookkeeping; [4] = number of workgroups;
 * [5] = the longest workgroup life, in ticks. */
int rvll_slice_walk_phases(rvll_handle* h, uint64_t out[6]);
/* Which form the last rvll_slice_walk / rvll_live_step took: *rounds = the number of rounds of the ROUNDS form (a round = one
 * launch that proposes and accepts for a group of walkers + one launch of the batch log-L kernel over the group's candidates,
 * no host synchronisation in between; the default wherever every Beta / Gamma prior has a verified table), 0 = one of the
 * single-kernel forms walked (RVLL_WALK_ROUNDS=0, or one of their switches).  Same results either way, bit for bit. */
int rvll_slice_walk_rounds(rvll_handle* h, int32_t* rounds);

/* ---- nested sampling with the live points resident on the device ----------------------------------------------- */
/* rvll_slice_walk above takes and returns its walkers through host buffers; a sampler built on it ships three row sets
 * each way per iteration (start points up, end points down) and gathers / scatters them on the host — a fifth of the
 * end-to-end time of evidence_amd/nested.py at 32768 live points.  With these entry points the live set (unit-cube rows,
 * theta, log-L) and the points that died stay in HBM for the whole run; per iteration the host sends indices and reads
 * back log-L — what it needs for the sort and the evidence sum (the part of evidence/ultranest/__init__.py:165-185 that is
 * the sampler's bookkeeping, not its likelihood calls).
 *
 * rvll_live_init   N unit-cube rows -> prior transform -> log-L (as rvll_prior_loglike_batch); the three arrays stay resident;
 *                  logl_out [N] (may be NULL).  Starts a new run (the dead store is emptied).
 * rvll_live_step   one iteration: order [N] = the live rows by ascending log-L.  The rows order[0 .. kdead) die — their
 *                  theta and log-L are appended to the dead store, in that order; kdead walkers start from rows
 *                  start[0 .. kdead) (the sampler draws them among the survivors order[kdead .. N)), walk nsteps moves
 *                  inside logL > lstar exactly as rvll_slice_walk does (same kernels, same counter-based random numbers:
 *                  walker i is row walker_base + i), and their end points replace the dead rows (walker i -> row
 *                  order[i]).  chol: the whitening factor [ndim, ndim], or NULL to have the covariance of the surviving
 *                  rows summed on the device (two passes, fixed order) and factored by the library; chol_out (may be
 *                  NULL) receives the factor that was used.  logl_new [kdead]: the new log-L of rows order[0 .. kdead).
 * rvll_live_sort   (round 4) the order itself, on the device: the live rows by ascending log-L (stable: ties by row, as
 *                  numpy's stable argsort) are sorted there and stay there; dead_logl [kdead] receives the log-L of the kdead
 *                  lowest in that order (the sampler's evidence sums need them), *lstar the kdead-th lowest, *max_logl the
 *                  highest.  The rvll_live_step that follows is then called with order = NULL, and its start [kdead] are
 *                  RANKS among the survivors (0 .. N - kdead - 1; the sampler's random draw): walker i starts from the row
 *                  of rank start[i].  A sampler then mirrors nothing per live point on the host: per iteration kdead ranks go
 *                  up, 2 kdead log-L values come down (0.6 ms of host sort and 128 KB of order per iteration at 32768 live
 *                  points before).
 * rvll_live_get    the live set as it stands (any pointer may be NULL).
 * rvll_live_dead   *n_dead in: capacity of theta [*, ndim] / logl [*] in rows (ignored when both are NULL);
 *                  out: rows in the dead store.  Rows are in the order they died.                                     */
int rvll_live_init(rvll_handle* h, const double* cube /*[N, ndim]*/, int64_t N, double* logl_out /*[N] or NULL*/);
int rvll_live_step(rvll_handle* h, const int32_t* order /*[N], or NULL after rvll_live_sort*/, int64_t kdead, const int32_t* start /*[kdead]*/,
                   double lstar, const double* chol /*[ndim, ndim] or NULL*/, const int32_t* wrapped /*[ndim] or NULL*/,
                   int32_t nsteps, int32_t max_rounds, uint64_t seed, int64_t walker_base, int64_t* ncalls,
                   double* logl_new /*[kdead]*/, double* chol_out /*[ndim, ndim] or NULL*/);
int rvll_live_sort(rvll_handle* h, int64_t kdead, double* dead_logl /*[kdead]*/, double* lstar, double* max_logl);
int rvll_live_get(rvll_handle* h, double* cube, double* theta, double* logl);
int rvll_live_dead(rvll_handle* h, int64_t* n_dead, double* theta, double* logl);

/* ---- scalar-callback latency ------------------------------------------------------------------------- */
/* PolyChord's loglike(theta) is irreducibly scalar (evidence/polychord/__init__.py:166-171): one theta per call.
 * With the server enabled, rvll_loglike_batch(B = 1) is answered by a persistent one-workgroup kernel that polls
 * a block of host-coherent pinned memory: the call writes theta and a request number there and spins on the
 * answer — a PCIe round trip instead of a kernel launch plus a stream synchronisation, same bits.  The kernel
 * leaves by itself after 5 ms without a request and is restarted by the next scalar call; every other entry point
 * of the handle stops it first.  enable = 0 turns it off (default; RVLL_SCALAR_SERVER=1 in the environment turns
 * it on at rvll_create).  While it runs, calls that synchronise the whole device (hipMalloc / hipFree, also of
 * other handles in the process) wait until it is idle, i.e. at most the 5 ms, provided no other thread keeps
 * feeding it meanwhile.  Round 4: up to 64 parameters the request word travels beside every value of the row (keyed with the
 * value, so a torn read cannot pass for a request), the kernel sees request and row in ONE read and answers from its LDS sums:
 * 9.7 - 10.3 us a call on an MI355X (10.8 before), 15.5 us for PolyChord's prior + loglike pair as one request.            */
int rvll_scalar_server(rvll_handle* h, int32_t enable);

/* ---- device-resident forms (no PCIe inside; used by bench and multi-GPU) -- */
/* Reserve device buffers for up to B points and copy theta (or cube) in.     */
int rvll_dev_reserve(rvll_handle* h, int64_t B);
int rvll_dev_upload_theta(rvll_handle* h, const double* theta, int64_t B);
int rvll_dev_upload_cube(rvll_handle* h, const double* cube, int64_t B);
/* Fill the resident cube buffer with counter-based uniforms on the device.   */
int rvll_dev_fill_cube(rvll_handle* h, int64_t B, uint64_t seed);
/* Launch on the handle's compute stream; asynchronous.                        */
int rvll_dev_prior(rvll_handle* h, int64_t B);                 /* cube -> theta  */
int rvll_dev_loglike(rvll_handle* h, int64_t B);               /* theta -> logL  */
/* One launch: the log-L kernel's staging step applies the prior transform to the resident cube rows, keeps
 * theta in LDS for the evaluation and writes it to the resident theta buffer too (prior(cube) followed by
 * loglike(theta) of evidence/polychord/__init__.py:130-171 for a whole batch).  Results are bit-identical to
 * rvll_dev_prior followed by rvll_dev_loglike.                                                           */
int rvll_dev_prior_loglike(rvll_handle* h, int64_t B);         /* cube -> theta, logL */
int rvll_dev_download(rvll_handle* h, int64_t B, double* theta /*or NULL*/,
                      double* logL /*or NULL*/, int32_t* flags /*or NULL*/);
int rvll_dev_sync(rvll_handle* h);
/* Device-resident launches run on one of two pipeline lanes (own stream, log-L and flags buffer each).  Flipping
 * the lane between independent batches keeps two launches in flight, so one batch's ramp-up hides the previous
 * one's tail; rvll_allgather_logl advances it by itself.  Returns the lane the next launch will use.          */
int rvll_dev_flip_lane(rvll_handle* h);
/* Time `iters` log-L launches over the resident theta with HIP events on the
 * compute stream (after `warmup` untimed launches).                          */
int rvll_dev_time_loglike(rvll_handle* h, int64_t B, int32_t warmup, int32_t iters,
                          rvll_timing* out);

/* Two HIP events on the compute stream (lane 0): record which = 0 before and which = 1 after a sequence of
 * device-resident launches; rvll_dev_mark_elapsed waits for the second and returns the time between them as the
 * device saw it (what bench.py divides by its step count for the roofline's kernel duration).               */
int rvll_dev_mark(rvll_handle* h, int32_t which);
int rvll_dev_mark_elapsed(rvll_handle* h, double* ms);

/* ---- launch geometry ------------------------------------------------------ */
/* points_per_block <= 0 restores the built-in heuristic.                     */
int rvll_set_points_per_block(rvll_handle* h, int32_t points_per_block);
/* The log-L kernel has two launch forms with bit-identical results: 256-thread tiles of a few points (small
 * batches, the walk, the scalar calls) and the CU-wide form (one 1024-thread workgroup per CU walking its share of
 * the batch; large batches).  0 = choose by batch size (default), 1 = tiles only, 2 = CU-wide wherever it fits.
 * Environment RVLL_FORM=tile|cu sets the same at rvll_create.                                                  */
int rvll_set_kernel_form(rvll_handle* h, int32_t form);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ----------------------- */
/* 128-byte opaque id created on rank 0 and handed to every rank out of band.  */
#define RVLL_COMM_ID_BYTES 128
int rvll_comm_unique_id(unsigned char id[RVLL_COMM_ID_BYTES]);
int rvll_comm_init(rvll_handle* h, const unsigned char id[RVLL_COMM_ID_BYTES],
                   int32_t nranks, int32_t rank);
/* rvll_comm_init gives ONE pipeline lane (the gather runs in-stream behind its kernel).  Further lanes — a stream,
 * a communicator (ncclCommSplit of the first) and buffers each, so that the gather of step k overlaps the kernel of
 * step k+1 — are added collectively: every rank calls rvll_comm_add_lanes, the ranks agree on the minimum of the
 * counts it returned (out of band), and every rank calls rvll_comm_set_lanes with that number.  Ranks cycling
 * through different numbers of communicators would hang their collectives.                                    */
int rvll_comm_add_lanes(rvll_handle* h, int32_t want, int32_t* have);
int rvll_comm_set_lanes(rvll_handle* h, int32_t nlanes);
/* All-gather of a small host buffer over the same communicator (n_local doubles per rank up, nranks * n_local
 * back, rank-major): what a sampler that shards host-side state over the ranks exchanges per iteration.        */
int rvll_allgather_host(rvll_handle* h, const double* mine, int64_t n_local, double* all /*[nranks * n_local]*/);
/* Which libraries this process actually runs on, as a JSON object: HIP runtime version and path of libamdhip64,
 * path and version of the librccl that was loaded (RVLL_RCCL_PATH, else the one next to that libamdhip64, else
 * /opt/rocm/lib, else by soname), path of librvll itself, and GPU_MAX_HW_QUEUES as the environment has it.
 * ON THAT VARIABLE: the HIP runtime maps a process's streams onto that many hardware queues (default 4) and reads it once, when
 * it starts.  A handle has seven streams; with four queues a copy stream of the streamed host batches
 * (rvll_prior_loglike_batch from 24 MB of rows on) can share the kernels' queue, and a 262144-row call then takes 4.0 ms
 * instead of 2.85 ms (profiles/r04_hw_queues.txt).  Nothing else depends on it (results never do).  librvll therefore asks
 * for 8 ITSELF when it is loaded, unless the variable is set already (or RVLL_KEEP_HW_QUEUES is): that is in time whenever
 * librvll is what brings the HIP runtime into the process; a caller that has made HIP calls before loading librvll exports
 * GPU_MAX_HW_QUEUES=8 itself.  "gpu_max_hw_queues_set_by" in the JSON says which of the two happened.                   */
int rvll_runtime_info(char* buf, int32_t buflen);
/* One multi-GPU step is rvll_dev_loglike(B_local) followed by rvll_allgather_logl(B_local): every rank's
 * per-shard log-L (the buffer the kernel just wrote) is all-gathered on the device, rank-major, so that every
 * rank — rank 0 owns the sampler's replacement step — holds all nranks*B_local values.  Asynchronous.  Steps
 * alternate between two pipeline lanes (own stream, communicator and buffers each), so the gather of one step
 * overlaps the kernel of the next without cross-stream events; rvll_download_gathered returns the last one. */
int rvll_allgather_logl(rvll_handle* h, int64_t B_local);
int rvll_download_gathered(rvll_handle* h, int64_t B_total, double* logL_all);
/* When theta was produced on the device (rvll_dev_prior / rvll_dev_prior_loglike on a cube shard), the sampler
 * on rank 0 also needs the physical parameters of the points it keeps: one more all-gather, of the B_local x
 * ndim theta rows, rank-major like the log-L gather.  Asynchronous on the compute stream.                    */
int rvll_allgather_theta(rvll_handle* h, int64_t B_local);
int rvll_download_gathered_theta(rvll_handle* h, int64_t B_total, double* theta_all /*[B_total, ndim]*/);
int rvll_comm_destroy(rvll_handle* h);

/* ---- Keplerian curves at arbitrary times (post-processing helper) ------------------ */
/* Replaces RVModel.kep_rv(pardict, time, exclude_planet) and RVModel.modelk(pardict, time, planet)
 * (evidence/rvmodel/__init__.py:343-385, 388-463) as evidence/post_processing.py:413-428 calls them for
 * phase folds: out[b][j] = sum over planets ip with bit ip of include_mask set of the Keplerian RV of
 * theta[b] at times[j].  kep_rv(exclude_planet=k) is mask = all bits but k-1; modelk(planet=k) is
 * mask = 1 << (k-1).  An invalid orbit (the reference returns None) gives a NaN row.                  */
int rvll_kep_rv_batch(rvll_handle* h, const double* theta, int64_t B, const double* times,
                      int32_t n_times, uint32_t include_mask, double* out /*[B, n_times]*/);

/* ---- FIP periodogram accumulation (post-processing; independent of any model handle) ------------ */
/* Replaces the accumulation loop of evidence/fip_criterion.py:305-339.  The caller flattens the posterior
 * samples of all planet models of one run into rows, in the reference's loop order (kmod = 1.., then sample
 * i), NaN-padding each row to np_max periods, with contrib[row] = pky[kmod] * weights[i] / sum(weights)
 * (:315, :339); rows of run r are run_start[r] .. run_start[r+1].  nua/nub are the window edges
 * nu -/+ nu_window/2 of :233-236 (ascending).  For every row and period x:  f = 2*pi/x,
 * beg = searchsorted(nub, f, 'right'), end = searchsorted(nua, f, 'left'); every bin in the UNION of the
 * row's [beg, end) ranges gets fapnu[r][bin] -= contrib[row], in row order (numpy's fancy-index `-=` applies
 * a repeated index once).  fapnu is [n_runs, nfreq], in/out (the reference starts from ones, :307); the
 * result is bit-identical to the reference's loop.  `repeats` > 1 re-runs the two kernels from the same
 * input for timing; timing may be NULL.  device < 0 uses the current device.                            */
#define RVLL_FIP_MAX_PLANETS 8
typedef struct rvll_fip_timing {
    double  index_ms;        /* mean HIP-event time of the interval (searchsorted) kernel   */
    double  accumulate_ms;   /* mean HIP-event time of the ordered fold kernel              */
    int64_t rows;
    int32_t repeats;
    int32_t reserved;
} rvll_fip_timing;
int rvll_fip_accumulate(int32_t device, const double* nua, const double* nub, int32_t nfreq,
                        const double* periods /*[rows, np_max]*/, const double* contrib /*[rows]*/,
                        const int64_t* run_start /*[n_runs + 1]*/, int32_t n_runs, int32_t np_max,
                        double* fapnu /*[n_runs, nfreq] in/out*/, int32_t repeats, rvll_fip_timing* timing);

/* ---- diagnostics -------------------------------------------------------------- */
/* Evaluate one device math routine elementwise (tests only; no reference counterpart):
 * op 0 sin, 1 cos (rvll sincos), 2 div_exact(x,y), 3 x/y (IEEE), 4 div_fast(x,y),
 * 5 log_pos(x), 6 library log(x), 7 ndtri(x), 8/9 sin/cos after rotate_small by y, 10 div_1nr, 11 v_rcp_f64,
 * 12/13 the wave reduction tree by shuffles / by permlane-swap + DPP (lane 0 of every 64 values), 14 the Cephes
 * form of ndtri (scipy's routine; op 7 is AS241).                                                            */
int rvll_debug_eval(rvll_handle* h, int32_t op, const double* x, const double* y,
                    int64_t n, double* out);

/* One launch of a stamped twin of the fp64 log-L kernel over the resident theta (after `warmup` ordinary
 * launches): per workgroup 8 words — [0] start, [1] per-point decode done, [2..5] item loop done per wave,
 * [6] end (constant 100 MHz clock), [7] HW_ID | XCC_ID << 32.  out == NULL only returns the geometry.  The
 * stamps never execute in the product kernels (no reference counterpart; profiles/ records are made with it). */
int rvll_dev_trace_loglike(rvll_handle* h, int64_t B, int32_t warmup, uint64_t* out, int64_t out_words,
                           int32_t* blocks, int32_t* points_per_block);

/* The exact redo of wandering Kepler solves (round 4; default on; RVLL_WANDER_EXACT=0 in the environment at rvll_create turns it
 * off).  Started at E = M next to a zero of 1 - e cos E (e >= 0.97) the reference's Newton iteration wanders for 30 - 350 steps,
 * and where it stops hangs on the last bit of every sin / cos on the way.  With the switch on, a solve that takes
 * more than eight steps is done again from its start with correctly rounded sin / cos — what glibc's are nearly always — and
 * the device is within 1e-10 of the reference on all but ~0.01 % of points with a planet at e >= 0.97 (1.4 % without; the
 * points still carry RVLL_FLAG_WANDERED).  Costs nothing where no solve wanders.  The proposal walk evaluates its CANDIDATES
 * without the redo (a wandering solve would hold a whole round up) and puts the log-L of the points it ENDS on right.   */
int rvll_set_wander_exact(rvll_handle* h, int32_t on);
/* The kernels that carry a log-L tile next to the prior stage (one-launch cube -> log-L, the walk) evaluate Beta /
 * Gamma quantiles by their verified tables over |logit q| <= umax (default: the whole table, 30) and hand every
 * other element to the routines with the full solvers — same results either way.  Lowering umax (0: nothing is
 * taken by the tables) exists so that tests can drive that hand-over; no reference counterpart.               */
int rvll_set_slim_table_range(rvll_handle* h, double umax);

/* ---- housekeeping ---------------------------------------------------------- */
const char* rvll_last_error(void);
int rvll_version(int32_t* major, int32_t* minor);
int rvll_device_count(int32_t* count);
int rvll_device_name(int32_t device, char* buf, int32_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* RVLL_H */
