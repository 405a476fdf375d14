// rvll_kernels.h — launch interface between the C-ABI host code (rvll_api.hip)
// and the gfx950 kernels (rvll_kernels.hip).  Internal; not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rvll.h"

namespace rvll {

constexpr int kThreads = 256;          // 4 wave64 per workgroup, one per SIMD
constexpr int kWave    = 64;
constexpr int kPlanetFields = 8;       // per (point, planet) scalars kept in LDS
constexpr int kMaxPointsPerBlock = 32;
constexpr int kTileWindow = 4096;      // (point, epoch) contributions one 256-thread tile keeps in LDS at a time
constexpr int kCuThreads = 1024;       // the CU-wide form: one workgroup of 16 waves per CU
constexpr int kCuMaxPoints = 128;      // points per chunk of the CU-wide form
constexpr size_t kCuLdsBudget = 156 * 1024;   // of the CU's 160 KiB

// Everything the fused log-L kernel needs; passed by value (kernarg segment).
struct LoglikeArgs {
    // batch
    const double* theta;     // [B, D] row-major
    double*       logL;      // [B]
    int32_t*      flags;     // [B]
    long long     B;
    // resident epoch table (SoA, reference concatenation order)
    const double*  t;        // [Ne]
    const double*  y;        // [Ne]
    const double*  s2;       // [Ne]  svrad^2
    const int32_t* inst;     // [Ne]
    const double*  linpar;   // [nlin, Ne] or nullptr
    int Ne;
    // layout (device copies of the ABI structs)
    const rvll_planet* planets;   // [Np]
    const rvll_inst*   insts;     // [Ni]
    const rvll_slot*   linslots;  // [nlin]
    const double*      layblob;   // planets, insts, linslots, drift[4], tref back to back (staged into LDS by the kernels)
    int D, Np, Ni, nlin;
    int has_jitter, has_drift, tref_from_data;
    double tol;
    int    itmax;
    int    precision;        // RVLL_PREC_*
    int    cr_redo;          // 1: a Kepler solve that wanders is redone with correctly rounded sin / cos (rvll_tile.h, eval_item<.., CR>)
    // geometry
    int    PB;               // live points per workgroup
    int    CH;               // contribution slots in LDS (items per chunk)
    double cte;              // -0.5 * Ne * log(2*pi)
    double tmin, tmax;       // range of the epoch table (a bound on |M| per planet is decoded from it: rvll_tile.h)
    // fused cube -> theta -> log-L form (launch_prior_loglike): the staging step applies the prior transform
    const double*      cube;        // [B, D] unit-cube rows, or nullptr for the plain form
    double*            theta_out;   // [B, D] the transformed parameters are also written here
    const rvll_prior*  priors;      // [D]
    const int32_t*     heavy_dims;  // [n_heavy] parameters with an iterative quantile (Beta, Gamma)
    int                n_heavy;
    const int32_t*     light_dims;  // [D - n_heavy] the other parameters, those of one kind next to each other (costliest
                                    // kinds first): consecutive lanes of the staging step then run the same quantile code
    // slim form of that stage (kFusedSlim): Beta / Gamma quantiles are evaluated by the verified table alone; an
    // element it cannot take (|logit q| > slim_umax, q on the boundary) marks its point kFlagDeferred, sets *defer
    // (may be null) and yields NaN — the host redoes such points through the kernels that carry the full solvers
    int*               defer;
    double             slim_umax;
    // diagnostic build only (launch_loglike_trace): kTraceWords stamps per workgroup, nullptr otherwise
    unsigned long long* trace;
};
// per-workgroup record of the diagnostic kernels (s_memrealtime, 100 MHz; [7] = HW_ID | XCC_ID << 32):
//   tile form    [0] start, [1] decode done, [2..5] item loop done per wave, [6] end
//   CU-wide form [0] start, [1] theta landed (wave 0), [2] staged, [3] decoded, [4] wave 0 out of the items,
//                [5] every wave out of them, [6] end
constexpr int kTraceWords = 8;
hipError_t launch_loglike_trace(const LoglikeArgs& a, hipStream_t stream);
// The CU-wide form (rvll_kernels.hip, loglike_cu_kernel): `grid` 1024-thread workgroups (one fills a CU), each
// owning a tile of a.PB consecutive points whose a.PB * a.Ne <= a.CH contributions are all resident in LDS.
// a.trace != nullptr launches the stamped twin (fp64 only).
hipError_t launch_loglike_cu(const LoglikeArgs& a, int grid, hipStream_t stream);

size_t loglike_lds_bytes(const LoglikeArgs& a);
// prior transform inside the tile's staging step: none / verified tables only / with the full solvers inline
constexpr int kFusedNone = 0, kFusedSlim = 1, kFusedFull = 2;
constexpr int kFlagDeferred = 0x40000000;     // internal flag bit, never returned to the caller

// ---- device-resident slice-sampling walk (the proposal step of nested sampling; evidence_amd/nested.py) -------
// K walkers start at cube points u (log-L above lstar) and take nsteps hit-and-run slice moves inside the region
// logL > lstar of the unit cube: direction = chol * normal / norm, chord limited by the cube walls (circular
// `wrapped` parameters: half a turn), candidate = uniform point of the chord, prior transform + log-L, accept if
// logL > lstar, otherwise shrink the chord towards the current point — up to max_rounds times per move.
// One workgroup owns a.PB walkers for the whole walk and evaluates its active candidates with loglike_tile.
struct WalkArgs {
    double* u;                 // [K, D] in: start, out: end positions
    double* theta;             // [K, D] in: theta of the start points, out: theta of the end points
    double* logl;              // [K]    in/out likewise
    const double* chol;        // [D, D] row-major lower-triangular whitening factor
    const int32_t* wrapped;    // [D] 1 = circular parameter
    long long K;
    int nsteps, max_rounds;
    unsigned long long seed;
    double lstar;
    unsigned long long* ncalls;   // += likelihood evaluations
    int32_t* steps_done;       // [K] out: moves completed (< nsteps: the walker met a candidate the slim prior stage
                               //          deferred and stopped at the start of that move)
    const int32_t* walker_id;  // [K] or null: the walker's index in the random-number counters (default: its row)
    const int32_t* step_start; // [K] or null: move to resume at (default 0)
    long long walker_base;     // added to the row (or to walker_id) in the random-number counters: a shard of a larger
                               // set of walkers draws exactly what the unsharded walk would draw for its rows
    int spec_max;              // candidates evaluated AHEAD per walker and iteration when the workgroup's tile has free
                               // slots (1: none): candidate r+1 is the one the walker draws if candidate r is rejected
    unsigned long long* nslots;   // or null: += tile slots evaluated (>= ncalls: includes speculative ones that went unused)
    unsigned long long* queue;    // zeroed before the launch: ticket counter of the rows no workgroup started with
    const int32_t* order;         // [K] or null: the order in which rows are handed to walker slots (default: by index)
    int32_t* cost;                // [K] or null, out: candidates every row used in this launch
    int rows_per_wg;              // the "rows" form (launch_slice_walk_rows): rows every workgroup owns for the whole launch
    int32_t* wflag;               // [K] or null, out: 1 where the walker's LAST accepted candidate carried RVLL_FLAG_WANDERED (its log-L is
                                  // then put right by the host with the exact redo, which the walk's tiles leave out: walk_core)
};
constexpr int kWalkCholLds = 48;   // the walk stages a whitening factor of up to 48 x 48 (18 KB) in LDS
size_t walk_lds_bytes(const LoglikeArgs& a);
// a: fused (cube -> theta -> log-L) arguments whose cube / theta_out / logL / flags rows [0, K) are scratch
// fat = false: slim prior stage (4 waves per SIMD; deferring walkers report steps_done < nsteps);
// fat = true : full solvers inline (every walker finishes)
// max_cus > 0: launch at most as many workgroups as max_cus compute units hold at once (the rest of the rows are drawn
// from w.queue by slots whose walker has finished); 0: one workgroup per PB rows
hipError_t launch_slice_walk(const LoglikeArgs& a, const WalkArgs& w, bool fat, int max_cus, hipStream_t stream);
long long slice_walk_resident_blocks(const LoglikeArgs& a, bool fat, int cus);
// The same walk with the rows dealt to the workgroups in advance (rvll_walk.hip, slice_walk_rows_kernel): nblocks
// workgroups own w.rows_per_wg rows each (position k of w.order — the host sorts by expected cost — goes to workgroup
// k mod nblocks, every other tier reversed), park them in LDS and interleave them over their a.PB walker slots at move
// boundaries, so that all rows end together.  Same results as launch_slice_walk, bit for bit.  No queue, no w.cost needed.
size_t walk_rows_lds_bytes(const LoglikeArgs& a, int rows_per_wg);
// nt: threads per workgroup — 256 (four workgroups per compute unit), 512 (two) or 1024 (one; a.PB <= 64 slots and, for the two wide
// forms, a.CH >= a.PB * a.Ne: the tile draws wave rounds from its ticket counter); the wide forms exist for the slim stage only
hipError_t launch_slice_walk_rows(const LoglikeArgs& a, const WalkArgs& w, bool fat, int nblocks, int nt, hipStream_t stream);

// ---- the walk as ROUNDS of launches over walker state resident in HBM (rvll_rounds.h, rvll_kernels.hip) -----------------------------------
// The kernels above keep a walker's state in its workgroup's LDS for the whole walk, and a workgroup's iteration — propose,
// prior stage, decode, items, reduce, accept, bookkeeping: a dozen barrier intervals — evaluates the eight to ten candidates
// of ITS walkers: 2600 vector instructions a candidate at 75-79 % VALU-busy against the batch kernel's 1990 at 99 %
// (profiles/r03_walk_forms.txt), whatever the structure inside the workgroup.  Here a round is two launches over a GROUP of
// walkers: rounds_step_kernel consumes the previous round's results (accept / shrink, in the order the walker would have met
// them), draws directions for the walkers that start a move and writes this round's candidates — one per listed walker, more
// ahead while the round is below the chip's latency floor — into ONE compact array, prior transform included; then the batch
// log-L kernel itself (the theta -> log-L tile, rvll_tile.h) evaluates that array, every workgroup an equal share of however many
// candidates there are (the count stays on the device).  No host synchronisation between rounds; the groups run on streams of
// their own, one group's step beside another's tiles.  The random-number counters name the walker,
// so end points, theta, log-L and the call count are those of slice_walk_kernel bit for bit (tests/test_gpu_walk.py).
constexpr int kRoundsRing = 16;        // per-round counters (slots handed out, walkers listed) live in a ring of this many rounds
struct RoundsArgs {
    double* u;                 // [K, D] the group's walkers: in start, out end positions (unit cube)
    double* theta;             // [K, D] theta of the accepted positions
    double* logl;              // [K]
    double* dir;               // [K, D] unit direction of the move in progress
    double* dirnext;           // [K, D] ... of the move after it (copied from dirs when a move starts: the step that needs it
                               //         finds it at a fixed address)
    const double* dirs;        // [K, nsteps, D] the directions of all moves of the group's walkers (rounds_dirs)
    double* tmin;              // [K] bracket of the move in progress
    double* tmax;              // [K]
    int32_t* step;             // [K] moves completed (out: steps_done — < nsteps: stopped at a candidate the slim stage deferred)
    int32_t* wflag;            // [K] out: 1 where the walker's last accepted candidate carried RVLL_FLAG_WANDERED (as WalkArgs::wflag)
    int32_t* ws;               // [K, 4] state (0 starts a move, 1 in a move, 3 deferred, 4 finished), round, first slot, slots
    long long K;
    unsigned long long wid0;   // random-number counter index of the group's first walker (walker_base + its row)
    double* theta_c[2];        // [C, D] the candidates of even / odd rounds, compacted and prior-transformed: the tiles' batch
                               //         (theta -> log-L); a step still reads the previous round's rows — theta of what it accepts
    int32_t* owner;            // [C] per slot: walker * spec_max + its index among the walker's slots (where the tiles put the result)
    double* wt;                // [K, spec_max] per walker: its candidates' positions along its direction
    int32_t* wdef;             // [K, spec_max] ... 1: a coordinate's quantile is beyond the verified tables (the walker is deferred)
    const double* wres_logl;   // [K, spec_max] per walker: what the tiles made of them
    const int32_t* wres_flags; // [K, spec_max]
    const rvll_prior* priors;  // [D] the prior stage of the step (as LoglikeArgs: light kinds grouped, iterative kinds by table)
    const int32_t* light_dims; // [D - n_heavy]
    const int32_t* heavy_dims; // [n_heavy]
    int n_heavy;
    int inplace;               // 1: no prior reads other coordinates of its row (no sorted kind): transformed in place in LDS
    double slim_umax;
    int C;                     // slots (>= K)
    int c_free;                // slots a round holds at the log-L kernel's latency floor: walkers get candidates AHEAD up to here
    const int32_t* wrapped;    // [D]
    int D, W;                  // W: walkers per workgroup of the step (<= 64: one lane of a wave each)
    int nsteps, max_rounds, spec_max;
    unsigned long long seed;
    double lstar;
    int32_t* ring;             // [kRoundsRing, 2] per round (8-byte aligned pairs): slots handed out = the tiles' batch, walkers listed
    long long* calls_part;     // [workgroups] likelihood calls consumed, summed by the host (a workgroup owns its entry: no atomics)
    unsigned long long* slots_part;   // [workgroups] slots evaluated (>= calls: candidates ahead that went unused)
    unsigned long long* stamps;       // diagnostic (RVLL_ROUNDS_STAMPS) or null: per round and workgroup 4 x s_memrealtime (100 MHz)
    int stamp_rounds;
};
// the directions of every move of every walker, in front of the rounds
struct RoundsDirs {
    double* dirs;              // [K, nsteps, D]
    const double* chol;        // [D, D] whitening factor
    long long K;
    unsigned long long wid0;   // random-number counter index of the first walker
    unsigned long long seed;
    int D, nsteps;
};
hipError_t launch_rounds_dirs(const RoundsDirs& g, int max_blocks, hipStream_t stream);
// what the tiles of a round need beside their LoglikeArgs (a.theta = the group's theta_c)
struct RoundsTiles {
    const int32_t* ring_entry;        // [2] slots to evaluate, walkers listed (the step of this round filled it)
    unsigned long long* progress;     // mapped pinned host memory or null: (round + 1) << 32 | walkers listed, when the tiles start
    const int32_t* owner;             // [C]
    double* wres_logl;                // [K, spec_max]
    int32_t* wres_flags;              // [K, spec_max]
};
int rounds_walkers_per_block(int D, int spec_max, size_t lds_budget);
size_t rounds_step_lds_bytes(int W, int D, int spec_max);
int rounds_blocks_per_cu(size_t lds_bytes);
// a group's round: two launches on the group's stream
hipError_t launch_rounds_step(const RoundsArgs& g, int round, hipStream_t stream);
hipError_t launch_rounds_tiles(const LoglikeArgs& a, const RoundsTiles& out, int tiles, int round, hipStream_t stream);
// the tiles of a round alone, in the CU-wide form: `tiles` 1024-thread workgroups of at most a.PB points (a.CH >= a.PB * a.Ne)
hipError_t launch_rounds_cu(const LoglikeArgs& a, const RoundsTiles& out, int tiles, int round, hipStream_t stream);

// ---- device-resident live set (rvll_live.hip): row gather / scatter by index; mean and covariance of a subset of rows
hipError_t launch_gather_rows(const double* src, const int32_t* idx, long long n, int width, double* dst, hipStream_t st);
hipError_t launch_scatter_rows(const double* src, const int32_t* idx, long long n, int width, double* dst, hipStream_t st);
size_t moments_scratch_doubles(int D);
hipError_t launch_moments(const double* u, const int32_t* idx, long long n, int D, double* scratch, double* mean, double* cov,
                          hipStream_t st);
// the live points' order on the device: stable ascending order of logl[0..n) (rocPRIM radix sort on order-preserving keys; ties
// by row, as numpy's stable argsort has them); out[i] = order[offset + rank[i]]
size_t sort_temp_bytes(long long n);
hipError_t launch_sort_logl(const double* logl, long long n, unsigned long long* keys_in, unsigned long long* keys_out, int32_t* rows_in,
                            int32_t* order_out, void* temp, size_t temp_bytes, hipStream_t st);
hipError_t launch_compose_index(const int32_t* order, long long offset, const int32_t* rank, long long n, int32_t* out, hipStream_t st);

// ---- scalar-call server: a one-workgroup persistent kernel that answers single-point log-L requests through a
// block of host-coherent pinned memory, so a scalar callback costs a PCIe round trip instead of a kernel launch
// and a stream synchronisation.  Host and device share this layout; every field sits on its own cache line.
constexpr int kServerMaxDim = 448;
constexpr unsigned kServerRunning = 1u, kServerExited = 2u;
constexpr unsigned kServerLogLike = 0u, kServerPrior = 1u, kServerNoop = 2u, kServerQuit = 3u,
                   kServerPriorLogLike = 4u;   // prior(cube) and log-L of the result in one request (PolyChord's pair)
struct ServerAnswer { double logL; unsigned int number; int flags; };      // 16 bytes: leaves the GPU as ONE store
struct ServerCtl {
    alignas(64) unsigned long long request;   // host -> device, one 64-bit store: (op << 32) | request number;
                                              // theta (or the cube row) is complete when it changes; op kServerQuit
                                              // makes the kernel leave
    alignas(64) unsigned int state;           // kServerRunning (host, before launch) / kServerExited (device, on exit)
    alignas(64) ServerAnswer answer;          // device -> host: number == the request's number when logL/flags
                                              // (and, for the prior op, theta) are complete
    alignas(64) double theta[kServerMaxDim];  // up to kServerSlotDims parameters: device -> host only (the prior op's result)
    // Round 4: the request AND its row in one read.  Up to kServerSlotDims parameters, the host writes value i and — after all the
    // values — the request word into slot i; the kernel's first wave polls the slots with one 16-byte load a lane and takes the
    // request when every slot carries the same new word: the values are then in its registers, and the PCIe round trip that used to
    // fetch theta after the word had changed (1.2 us of a 10.8 us call) is gone.
    // The word of a slot is the request XOR-ed with the slot's own value (rotated): a slot whose two halves were not read at one
    // instant — or a word that arrived before its value — does not decode to the request the other slots carry, and the poll
    // simply comes round again; nothing rests on how the link splits a read or orders the writes.
    struct Slot { double v; unsigned long long word; };
    alignas(64) Slot in[64];
    static __host__ __device__ __forceinline__ unsigned long long slot_key(unsigned long long vbits) { return (vbits << 17) | (vbits >> 47); }
};
constexpr int kServerSlotDims = 64;           // one lane of the polling wave a parameter; larger models keep the two-step protocol
// a = arguments of a one-point launch whose theta points into *ctl (device address of the block) and whose
// logL / flags point at device-local scratch; the kernel answers requests that differ from `last` and leaves
// after idle_ticks (100 MHz) without one
hipError_t launch_scalar_server(const LoglikeArgs& a, ServerCtl* ctl, unsigned long long last,
                                unsigned long long idle_ticks, hipStream_t stream);
hipError_t launch_loglike(const LoglikeArgs& a, hipStream_t stream);
// resident 256-thread workgroups per CU for a given dynamic-LDS size (occupancy query)
int loglike_blocks_per_cu(size_t lds_bytes);

struct PriorArgs {
    const double* cube;      // [B, D]
    double*       theta;     // [B, D]
    long long     B;
    int           D;
    const rvll_prior* priors;  // [D] device copy; table pointers are device pointers
    const int32_t* heavy_dims; // [n_heavy] parameters whose quantile is an iterative solve (Beta, Gamma)
    int           n_heavy;
};
hipError_t launch_prior(const PriorArgs& a, hipStream_t stream);
// one launch: prior transform in the staging step of the log-L kernel (a.cube / theta_out / priors set)
hipError_t launch_prior_loglike(const LoglikeArgs& a, hipStream_t stream);
// device-built table of a Beta/Gamma prior: z[n], dz[2n] (slopes, then second derivatives) with
// n = prior_table_nodes(); if max_err_bits (a zeroed device word) is given, the quintic interpolant's error
// against the full solver is measured into it (bits of a double; compare with prior_table_direct_tol())
int prior_table_nodes();
double prior_table_umax();     // |logit q| the tables cover
double prior_table_direct_tol();
hipError_t launch_prior_table(int kind, const double* args, double* z, double* dz, unsigned long long* max_err_bits,
                              hipStream_t stream);

hipError_t launch_keprv(const LoglikeArgs& a, const double* times, int Nt, unsigned include_mask, double* out,
                        hipStream_t stream);
hipError_t launch_debug_eval(int op, const double* x, const double* y, long long n, double* out, hipStream_t stream);
hipError_t launch_fill_cube(double* cube, long long n, uint64_t seed, hipStream_t stream);

}  // namespace rvll
