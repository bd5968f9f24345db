/*
 * sdfs_hip.h -- C ABI of libsdfs_hip.so: the MI355X (gfx950) implementation of the
 * wealth-consumption-ratio fixed-point path of jstac/sdfs_via_autodiff.
 *
 * The reference has no FFI layer; its boundary for this path is two Python call
 * shapes (paths relative to the reference repo):
 *
 *   operator  T(w) = T_ssy(w, shapes, params, arrays)   code/ssy/discrete/ssy_wc_ratio.py:82-151
 *             T(w) = T_gcy(w, shapes, params, arrays)   code/gcy/discrete/gcy_wc_ratio.py:134-238
 *   solver    solver(f, x_init, algorithm, verbose)     code/solvers.py:154-177
 *             successive_approx / newton_solver / anderson_solver   code/solvers.py:19-124
 *
 * Each entry point below names the reference call it replaces.  Conventions:
 * opaque handle; every call returns 0 on success and a negative code on error
 * (message via sdfs_last_error); no exceptions cross the boundary; the caller
 * owns host buffers; the library owns its device buffers and one HIP stream per
 * handle; a handle is not thread-safe, distinct handles are independent.
 * Grids are C-order fp64 exactly as the reference lays them out
 * (SSY: (h_lam, h_c, h_z, z), z fastest; GCY: (z, z_pi, h_z, h_c, h_zpi, h_lam),
 * h_lam fastest).  "_dev" variants take device pointers (hipMalloc'd or
 * torch.Tensor.data_ptr()) and never touch host memory.
 */
#ifndef SDFS_HIP_H
#define SDFS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdfs_handle sdfs_handle;

enum { SDFS_MODEL_SSY = 0, SDFS_MODEL_GCY = 1 };
enum { SDFS_ALGO_SA = 0, SDFS_ALGO_NEWTON = 1, SDFS_ALGO_ANDERSON = 2 };

enum {
  SDFS_OK = 0,
  SDFS_ERR_ARG = -1,      /* bad argument (shape, count, null pointer) */
  SDFS_ERR_HIP = -2,      /* a HIP runtime call failed */
  SDFS_ERR_UNSUPPORTED = -3,
  SDFS_ERR_NUMERIC = -4   /* NaN/Inf met in the iteration */
};

/* Options of sdfs_solve*.  Defaults (sdfs_default_opts) are the reference's:
 * code/solvers.py:16-17 (tol 1e-7, max_iter 1e6), :55 (bicgstab_atol 1e-4),
 * jax.scipy.sparse.linalg.bicgstab default tol 1e-5, :104-114 (Anderson
 * history 10, mixing 4, beta 8, ridge 1e-6, max_iter 1e4). */
typedef struct sdfs_opts {
  double tol;            /* sup-norm step tolerance (SA, Newton); l2 residual (Anderson) */
  int64_t max_iter;
  double inner_rtol;     /* BiCGSTAB relative tolerance on |r|_2            */
  double inner_atol;     /* BiCGSTAB absolute tolerance on |r|_2            */
  int64_t inner_max_iter;/* 0 -> 10 * N (JAX default)                       */
  int32_t history;       /* Anderson history size m                         */
  int32_t mixing_freq;   /* Anderson mixing frequency                       */
  double beta;           /* Anderson damping                                */
  double ridge;          /* Anderson ridge.  >= 0: jaxopt's absolute ridge on the Gram matrix (the reference's 1e-6,
                          * code/solvers.py:113).  < 0 (opt-in, NOT the reference's semantics): relative,
                          * |ridge| * trace(G) / history -- on grids of 1e7 .. 1e8 points the Gram entries N r^2 sink
                          * below an absolute 1e-6 while the residual is still 1e-6 and the acceleration stalls */
  int32_t check_every;   /* 0 (default): the library's choice -- 32, or 120 where the fused small-grid Anderson loop runs.
                            k >= 1: the host polls the device residual every k iterations: SA and Anderson enqueue (or replay from
                            a hipGraph) k gated iterations per synchronisation -- rounded up to an even number (SA) or to a
                            multiple of `history` (Anderson); the iterates, counts and error trace do not depend on k */
  int32_t use_graph;     /* 1: replay the iteration chunk from a hipGraph   */
  int32_t record_errors; /* 1: keep the per-iteration error trace (sdfs_error_trace) */
  int32_t krylov_f32;    /* Newton: 1 = inner BiCGSTAB in fp32 storage (Krylov vectors, J.v streams), fp64
                          * arithmetic and reductions, fp64 outer residual and iterate -- the mixed-precision
                          * configuration of BASELINE.json (config 5).  2 = the same with every store of those fp32
                          * containers rounded to bfloat16: the NUMERICS of bf16 storage at the bytes of fp32 (evaluation
                          * mode of the config-5 sweep).  3 = fp32 storage as under 1 AND the J.v passes of the pair
                          * plan on an fp32 LDS tile with v_mfma_f32_16x16x4_f32 (fp32 products and sums inside a
                          * pass; the BiCGSTAB reductions, the outer residual and the iterate stay fp64) -- config 5's
                          * "on MFMA" (csrc/f32_kernels.hpp); plans without such kernels run as under 1.
                          * Default 0 (everything fp64).    */
  int32_t t_f32;         /* Successive approximation on the pair plan (extents 16/20/24/32, whole 16-element chunks): 1 = the
                          * applications of T keep the intermediates between their passes as scaled floats while the step
                          * is above ~64 * 2^-24 * w / |theta| (what that storage can resolve), then the loop finishes in
                          * fp64 to `tol`: config 5 for the T passes.  The iteration count is then this configuration's
                          * own.  Ignored where the plan has no fp32 forms.  Default 0.                                 */
} sdfs_opts;

/* Per-kernel counters for the roofline line of bench.py. */
#define SDFS_MAX_KERNELS 16
typedef struct sdfs_kernel_counter {
  char name[48];
  int64_t launches;       /* launches bracketed by HIP events               */
  double total_ms;        /* sum of their durations (hipEventElapsedTime)   */
  double alg_bytes;       /* algorithmic bytes of ONE launch    */
  double alg_flops;       /* algorithmic flops of ONE launch                */
} sdfs_kernel_counter;

typedef struct sdfs_counters {
  int32_t nkernels;
  int32_t reserved;
  sdfs_kernel_counter k[SDFS_MAX_KERNELS];
} sdfs_counters;

/* Build an operator handle.  Replaces the closure
 *   T = lambda w: T_ssy(w, shapes, params, arrays)    (ssy_wc_ratio.py:230)
 *   T = lambda w: T_gcy(w, shapes, params, arrays)    (gcy_wc_ratio.py:333)
 * `params`: the 13 (SSY, ssy_model.py:81) or 18 (GCY, gcy_model.py:72-75) scalars.
 * `arrays`: the 10 (discretize_ssy, ssy_wc_ratio.py:73-77) or 15 (discretize_gcy,
 * gcy_wc_ratio.py:123-128) host arrays in the reference's order and layout;
 * `array_sizes[i]` = element count of arrays[i] (checked against `shapes`). */
int sdfs_create(int model, int ndim, const int64_t* shapes,
                const double* params, int nparams,
                const double* const* arrays, const int64_t* array_sizes, int narrays,
                int device_id, sdfs_handle** out);

/* Sharded variant for one rank of a multi-GPU run: this rank owns the
 * index block [lo, lo+len) of `shard_axis` of its INPUT grid; see
 * sdfs_apply_stage_dev.  shard_axis must be an axis no transition matrix is
 * conditioned on. */
int sdfs_create_sharded(int model, int ndim, const int64_t* shapes,
                        const double* params, int nparams,
                        const double* const* arrays, const int64_t* array_sizes, int narrays,
                        int device_id, int axis_a, int64_t a_lo, int64_t a_len,
                        int axis_b, int64_t b_lo, int64_t b_len, sdfs_handle** out);

/* Continuous-state operator: replaces the closure T_fun_factory(params, method, batch_size) returns
 * (code/ssy/continuous_junnan/ssy_wc_ratio_continuous.py:156-226,
 *  code/gcy/continuous/gcy_wc_ratio_continuous.py:190-260):
 *   Tw(x) = 1 + beta * (const(x) * E_x[ exp(theta h_lambda') * lin_interp(w)(x')^theta ])^(1/theta)
 * with the expectation a weighted sum over M shock nodes (Gauss-Hermite: `weights` from
 * quantecon.quad.qnwnorm; Monte Carlo: weights == NULL, plain mean).  `grids`: ndim uniform axis
 * grids in the reference's order (SSY h_lambda, h_c, h_z, z; GCY h_lambda, h_c, h_z, h_zpi, z, z_pi),
 * grids[d] has shapes[d] >= 2 points; `nodes`: [ndim][M] row-major, as the reference passes them.
 * The handle works with every entry point below (apply, JVP, solve, counters); batching is internal
 * (the reference's batch_size / ram_free only bound JAX's temporaries). */
int sdfs_create_continuous(int model, int ndim, const int64_t* shapes,
                           const double* params, int nparams, const double* const* grids,
                           const double* nodes, const double* weights, int64_t M,
                           int device_id, sdfs_handle** out);

/* Single-index dense form (code/ssy/discrete/temp_ssy.py:109-159: H = compute_H_single_index(ssy, shapes),
 * single_index_T(w, H, params) = 1 + beta * (H @ w**theta)**(1/theta); analytic Jacobian :204-216).  The
 * reference keeps it "for cross-checking solutions produced by the multi-index code"; same role here.
 * H: N x N row-major host array (copied to the device).  The handle works with every entry point below. */
int sdfs_create_dense(int64_t N, const double* H, double beta, double theta, int device_id, sdfs_handle** out);

/* lin_interp(x, fun_vals, grids) of code/utils.py:18-23 (multilinear, indices clipped to the grid):
 * x is [ndim][nq] row-major, out[nq]; all host pointers; ndim 4 or 6. */
int sdfs_lin_interp(int device_id, int ndim, const int64_t* shapes, const double* const* grids,
                    const double* fun_vals, const double* x, int64_t nq, double* out);

void sdfs_destroy(sdfs_handle* h);
const char* sdfs_last_error(const sdfs_handle* h);   /* h may be NULL: last create error */
int sdfs_default_opts(sdfs_opts* o);

int64_t sdfs_grid_size(const sdfs_handle* h);        /* N = prod(shapes) */
/* Launch on the caller's HIP stream (hip_stream == NULL is the device's default stream, which is
 * what torch.cuda.current_stream().cuda_stream returns for torch's default stream); use_own != 0
 * switches back to the handle's private non-blocking stream. */
int sdfs_set_stream(sdfs_handle* h, void* hip_stream, int use_own);
int sdfs_synchronize(sdfs_handle* h);

/* Tw = T(w).  Replaces one call of T_ssy / T_gcy.  Host buffers of N doubles. */
int sdfs_apply_T(sdfs_handle* h, const double* w_host, double* Tw_host);
/* Same on device pointers; if resid_dev != NULL also writes max|Tw - w| there
 * (the reduction of code/solvers.py:36 fused into the operator's last kernel). */
int sdfs_apply_T_dev(sdfs_handle* h, const double* w_dev, double* Tw_dev, double* resid_dev);

/* out = dT(w)[v], the map jax.jvp(f, (w,), (v,))[1] of code/solvers.py:87
 * (without the "- v" of g = f - id). */
int sdfs_apply_jvp(sdfs_handle* h, const double* w_host, const double* v_host, double* out_host);
/* Device form: linearise once at w (caches the two diagonal scalings), then
 * apply to any number of v.  If Tw_dev != NULL the linearisation also returns T(w). */
int sdfs_linearize_dev(sdfs_handle* h, const double* w_dev, double* Tw_dev);
int sdfs_apply_jvp_dev(sdfs_handle* h, const double* v_dev, double* out_dev, int minus_identity);

/* out = dT(w)^T [u], the vector-Jacobian product jax.grad needs for the reference's "gd" solver
 * (loss = |f(x) - x|^2, code/solvers.py:127-140): gradient = 2 (dT(x)^T r - r), r = f(x) - x.  Same
 * kernels as J.v with transposed matrices and the two diagonal scalings in each other's place; available
 * when every transition tensor is unconditional (Rouwenhorst / Tauchen chains), SDFS_ERR_UNSUPPORTED
 * otherwise.  The device form uses the linearisation cached by sdfs_linearize_dev. */
int sdfs_apply_vjp(sdfs_handle* h, const double* w_host, const double* u_host, double* out_host);
int sdfs_apply_vjp_dev(sdfs_handle* h, const double* u_dev, double* out_dev, int minus_identity);

/* max|T(w) - w| of the most recent apply that computed it. */
int sdfs_residual(sdfs_handle* h, double* sup_norm);

/* Fixed-point solve, all iterations on the device.  Replaces
 * successive_approx / newton_solver / anderson_solver (code/solvers.py:19-124):
 * w_inout holds x_init on entry and x_star on return; n_iter is the value the
 * reference returns as its second tuple element; n_apply counts operator and
 * JVP applications; final_err is the last value of the solver's own error. */
int sdfs_solve(sdfs_handle* h, int algo, const sdfs_opts* opts, double* w_inout_host,
               int64_t* n_iter, int64_t* n_apply, double* final_err);
int sdfs_solve_dev(sdfs_handle* h, int algo, const sdfs_opts* opts, double* w_inout_dev,
                   int64_t* n_iter, int64_t* n_apply, double* final_err);
/* Error trace of the last solve (record_errors = 1): copies min(cap, n) values. */
int64_t sdfs_error_trace(sdfs_handle* h, double* out, int64_t cap);

/* Multi-GPU building block: run the local kernels of one operator application.
 * stage 0: contractions that need no other rank's data (input sharded on axis_a);
 * stage 1: after the grid re-shard (input sharded on axis_b): the contractions left over -- axis_a among them -- and the
 * aggregator.  Which complete axes are contracted in which stage is the library's choice (the expectation is a Kronecker
 * product, the order is free: 6-D grids of the compile-time pair plan leave axis_a's partner axis to stage 1, so that both
 * stages run that plan's kernels), so what stage 0 writes is an intermediate only the same handle's stage 1 understands.
 * `mode` 0 = T, 1 = JVP (uses the cached linearisation), 2 = T + linearise. */
int sdfs_apply_stage_dev(sdfs_handle* h, int stage, int mode, const double* in_dev,
                         double* out_dev, const double* w_old_dev, double* resid_dev);

/* The same launch behind a device-side gate (multi-GPU successive approximation without a host read per iteration:
 * sdfs_via_autodiff_amd/distributed.py; the loop it serves is code/solvers.py:34-36).  gate_dev points at a device
 * double -- the all-reduced error of the previous iteration; if *gate_dev <= gate_tol (both non-negative) every
 * kernel of the stage returns at once and `out_dev` keeps its contents, and resid_dev, if given, is left at 0, which
 * keeps every later gate on it closed.  gate_dev == NULL: ungated. */
int sdfs_apply_stage_gated_dev(sdfs_handle* h, int stage, int mode, const double* in_dev, double* out_dev,
                               const double* w_old_dev, double* resid_dev, const double* gate_dev, double gate_tol);

/* Exchange buffers of the re-shard between two stage calls (sdfs_via_autodiff_amd/distributed.py; the reference has no
 * multi-GPU path).  The grid is viewed as [outer][n_axis][inner] in C order; `packed` is the concatenation
 * over the nblocks blocks j (axis indices offs[j] .. offs[j+1], offs[0] = 0, offs[nblocks] = n_axis) of
 * [outer][size_j][inner], i.e. what every peer sends or receives is one contiguous piece.  unpack = 0: src is the
 * grid, dst the packed buffer; unpack = 1: the reverse.  elem_bytes 8 (fp64) or 4 (fp32 Krylov streams).  One launch on
 * the handle's stream. */
int sdfs_pack_blocks(sdfs_handle* h, int unpack, const void* src_dev, void* dst_dev, int64_t outer, int64_t n_axis,
                     int64_t inner, int nblocks, const int64_t* offs, int elem_bytes);

/* Multi-GPU Krylov building blocks: the fused BLAS-1 kernels of the single-GPU BiCGSTAB / Newton loops
 * (jax.scipy.sparse.linalg.bicgstab inside code/solvers.py:91-93) on caller-owned device vectors of `n` LOCAL
 * elements -- this rank's shard -- with the scalar recurrences in the handle's device block.  A step either
 * leaves the rank's partial sums in `sums_dev` (the caller all-reduces them, SUM; MAX for the Newton step) or
 * consumes the all-reduced values from it.  v = {b, r, rhat, p, q, t, x} (b fp64; the others fp64, or fp32 when
 * f32 != 0).  Sequence per solve:  INIT -> [all-reduce 1] -> INIT_FIN;  per iteration:  UPDATE_P, (q = J p - p by
 * the caller), DOT_RQ -> [1] -> ALPHA_S -> [1] -> S_FIN, (t = J s - s, s lives in r), DOT_TS -> [2] -> OMEGA_XR
 * -> [2] -> ITER_FIN; sdfs_krylov_scalars reads the block back (one synchronisation per iteration). */
enum { SDFS_KS_INIT = 0, SDFS_KS_INIT_FIN, SDFS_KS_UPDATE_P, SDFS_KS_DOT_RQ, SDFS_KS_ALPHA_S, SDFS_KS_S_FIN,
       SDFS_KS_DOT_TS, SDFS_KS_OMEGA_XR, SDFS_KS_ITER_FIN, SDFS_KS_SUB_DOT, SDFS_KS_NEWTON_UPDATE };
enum { SDFS_SC_RR = 9, SDFS_SC_BB = 10, SDFS_SC_ATOL2 = 11, SDFS_SC_BREAK = 13, SDFS_SC_ITERS = 15 };   /* indices into the scalar block */
int sdfs_krylov_step(sdfs_handle* h, int step, int64_t n, int f32, void* const* v, double* sums_dev,
                     double rtol, double atol);
int sdfs_krylov_scalars(sdfs_handle* h, double* out16_host);
/* step | SDFS_KS_GATED: the kernels of the step return at once while the handle's gate word is closed.  INIT_FIN opens it
 * when |b|^2 is above the stopping level, ITER_FIN closes it on convergence, breakdown or a non-finite |r|^2 -- the gate
 * of the single-GPU loop (csrc/vec_kernels.hpp) -- so a caller enqueues several iterations (stage launches gated on the
 * same word: sdfs_krylov_gate + sdfs_apply_stage_gated_dev with gate_tol 0, all-reduces on whatever the sums buffer
 * holds) per sdfs_krylov_scalars; the iterates and SDFS_SC_ITERS are those of the one-iteration-per-read loop. */
enum { SDFS_KS_GATED = 0x100 };
int sdfs_krylov_gate(sdfs_handle* h, const void** gate_dev);
/* sdfs_pack_blocks(unpack = 1) with the Krylov operator's "- v" folded in: grid_out = unpacked - sub_dev (sub_dev laid out
 * like grid_out; fp64 or fp32 by elem_bytes).  (J - I) v then costs the J v stages plus ONE extra stream in the pass that
 * scatters the exchanged result back, instead of a subtraction and a copy over the whole shard. */
int sdfs_unpack_blocks_sub(sdfs_handle* h, const void* packed_dev, void* grid_out_dev, const void* sub_dev, int64_t outer,
                           int64_t n_axis, int64_t inner, int nblocks, const int64_t* offs, int elem_bytes);

/* Anderson acceleration (jaxopt.AndersonAcceleration as called at code/solvers.py:104-114; semantics restated in
 * oracle/solvers.py, iterate parity unpinned) on a SHARDED grid: the single-GPU loop's large-grid kernels
 * (csrc/vec_kernels.hpp: history as Y_j = x_j + beta r_j and r_j, <r, r> per pass, the whole Gram matrix in one sweep
 * where a solve is due) on this rank's n local points, the loop's control -- Gram matrix, the (m+1)^2 solve, the
 * domain safeguard, the stopping test -- in the handle's device state, identical on every rank because it only ever sees
 * all-reduced sums.  Sequence per pass i (0, 1, ...):  x_out = T(x_in) by the caller (stage launches gated on
 * sdfs_anderson_gate, gate_tol 0);  PUSH -> [all-reduce SUM sums[0]];  if (i + 1) % mixing_freq == 0: GRAM -> [all-reduce SUM
 * sums[1 .. 1 + SDFS_AND_NPAIRS)];  STEP;  MIX (x_out becomes the next iterate: pass i + 1 reads it as x_in, so the caller
 * alternates two buffers).  Every kernel is a no-op once the loop has ended; sdfs_anderson_state reads the state back
 * (out8 = passes executed, error = |T x - x|_2 of the last pass, 1 while the loop runs, 1 if a non-finite residual ended it,
 * rejected mixing steps, ...) and the errors of passes first .. first + count (a ring of 256).  history <= 12; the history
 * buffers hold history * n doubles each and stay the caller's. */
enum { SDFS_AND_PUSH = 0, SDFS_AND_GRAM = 1, SDFS_AND_STEP = 2, SDFS_AND_MIX = 3 };
enum { SDFS_AND_NPAIRS = 78 };
int sdfs_anderson_begin(sdfs_handle* h, int64_t n, int history, double* y_hist_dev, double* r_hist_dev, double tol,
                        int64_t max_iter, double beta, double ridge, int mixing_freq);
int sdfs_anderson_gate(sdfs_handle* h, const void** gate_dev);
int sdfs_anderson_step(sdfs_handle* h, int step, int64_t pass, const double* x_in_dev, double* x_out_dev, double* sums_dev);
int sdfs_anderson_state(sdfs_handle* h, double* out8_host, double* errs_host, int64_t first_pass, int64_t count);
/* Sharded handles: fp32 storage of the J.v streams and of the linearisation (opts.krylov_f32 of the single-GPU
 * solve) for the stage calls that follow.  `w_ref`: one positive value shared by all ranks (e.g. the geometric
 * mean of the all-reduced min and max of the iterate) from which every rank and both stages derive the same
 * power-of-two scale of c1 / c2. */
int sdfs_set_krylov_f32(sdfs_handle* h, int on, double w_ref);

/* Sharded handles whose stages run the pair plan's kernels (6-D grids, sdfs_describe_plan says "pair-plan"): fp32
 * intermediates for the plain applications of T that follow (mode 0; opts.t_f32 of the single-GPU successive
 * approximation, code/solvers.py:19-48 is the loop).  Stage 0 then WRITES scaled floats -- the re-shard between the stages
 * moves half the bytes -- and stage 1 reads them; w, T w and the residual stay fp64.  One stored float carries 2^-24
 * relative, ~ w 2^-24 / |theta| on T w: the caller runs this form while the step is well above that and finishes in fp64
 * (sdfs_via_autodiff_amd/distributed.py, successive_approx_sharded(t_f32=True)).  `w_ref` as for sdfs_set_krylov_f32: every
 * rank and both stages derive the same power-of-two scale from it.  SDFS_ERR_UNSUPPORTED on handles with generic stage plans. */
int sdfs_set_t_f32(sdfs_handle* h, int on, double w_ref);

/* Profiling: when enabled every kernel launch is bracketed by HIP events on the
 * handle's stream; sdfs_get_counters synchronises and sums them. */
int sdfs_set_profiling(sdfs_handle* h, int on);
int sdfs_reset_counters(sdfs_handle* h);
int sdfs_get_counters(sdfs_handle* h, sdfs_counters* out);

/* Test hook: out[i] = x[i]^y through the kernels' own device power routine (the
 * replacement for jnp's `**` at ssy_wc_ratio.py:145,148 / gcy_wc_ratio.py:232,235),
 * so that its accuracy can be checked in isolation. */
int sdfs_debug_pow(const double* x_host, double y, double* out_host, int64_t n, int device_id);
/* The same for the routine the kernels use when the exponent is fixed for a launch (theta in the first pass of T,
 * 1/theta in the last; csrc/pass_kernel.hpp, powy): degree 6 = the form inside the operator kernels (|y| * 1.04e-17
 * polynomial error on x^y), degree 7 = accurate for any |y| <= 64, degree 0 = sdfs_debug_pow. */
int sdfs_debug_powy(const double* x_host, double y, double* out_host, int64_t n, int degree, int device_id);

/* Measurement aid (bench.py's `copy_ceiling_GBps`; no counterpart in the reference): dst[0..n) = src[0..n) by the library's
 * own streaming copy -- one launch on the handle's stream, 16 bytes per lane, eight loads in flight per lane, non-temporal
 * loads and stores: what a pass that reads and writes every grid point once can reach on this box.  Buffers 16-byte
 * aligned, not overlapping. */
int sdfs_stream_copy_dev(sdfs_handle* h, const double* src_dev, double* dst_dev, int64_t n);

/* Human-readable description of the kernel plan (passes, tiles, grid sizes). */
int sdfs_describe_plan(const sdfs_handle* h, char* buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SDFS_HIP_H */
