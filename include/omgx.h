/* omgx.h -- C ABI of the MI355X-native batched spline-MPC solve path.
 *
 * Drop-in boundary for the one place omg-tools crosses into native numerics:
 *   - reference `problems/problem.py:113`  result = self.problem(x0=, p=, lbg=, ubg=)
 *     (CasADi nlpsol object built in `basics/optilayer.py:49-60`),
 *   - reference `problems/admm.py:390` (x-update NLP), `admm.py:424,462,505`
 *     (z / lambda / residual CasADi Functions),
 *   - reference C++ export `omg::Point2Point::update/solve`
 *     (`export/point2point/Point2Point.cpp:124-231`) and
 *     `Vehicle::sampleSplines/evalSpline` (`export/vehicles/Vehicle.cpp:112-190`).
 * The reference has no C ABI of its own (it links libcasadi from C++/Python);
 * these entry points are what a ctypes/cgo/JNI binding of that call would bind.
 *
 * Conventions: every function returns 0 on success or a negative OMGX_E_* code;
 * nothing throws across the ABI.  The caller owns every buffer it passes; the
 * library owns only the handle and its device workspace.  All floating point is
 * fp64.  A handle is bound to one HIP device and one stream and is not
 * re-entrant; different handles may be driven from different threads/processes.
 */
#ifndef OMGX_H
#define OMGX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMGX_VERSION 9
#define OMGX_TERM_VARS 4      /* variables per term (version 3: three) */

/* error codes */
#define OMGX_OK            0
#define OMGX_E_INVALID    -1   /* bad argument / inconsistent template */
#define OMGX_E_NODEVICE   -2   /* no usable HIP device */
#define OMGX_E_HIP        -3   /* a HIP runtime call failed (see omgx_last_error) */
#define OMGX_E_TOOLARGE   -4   /* even the O(n_var) vectors of one agent do not fit in LDS */

/* per-agent solver status (mirrors IPOPT's return_status strings) */
#define OMGX_SOLVE_SUCCEEDED        0
#define OMGX_MAX_ITER_EXCEEDED      1
#define OMGX_INFEASIBLE_DETECTED    2
#define OMGX_UNSUPPORTED_BOUNDS     3
#define OMGX_NUMERICAL_FAILURE      4

/* flags for pointer arguments */
#define OMGX_PTR_DEVICE     1   /* p/x0/x/lam_g/status/iters are device pointers */
#define OMGX_BOUNDS_SHARED  2   /* lbg/ubg hold n_con values shared by all agents */
#define OMGX_BOUNDS_DEVICE  4   /* lbg/ubg are device pointers */
#define OMGX_ONLY_FAILED    8   /* restart pass: agents whose status entry is Solve_Succeeded on entry are skipped
                                   (their x, lam_g, status, iters are left as they are); the others are solved from
                                   x0 -- e.g. another initial guess after a phase-I stall (needs OMGX_PTR_DEVICE, or
                                   host buffers that hold the results of the previous call) */

/* Flat NLP description (host pointers, copied by create): what the reference hands to
 * `nlpsol('solver', 'ipopt', {x, p, f, g}, ...)` (`basics/optilayer.py:54-60`) as CasADi graphs, here as
 * polynomial term lists (omgtools/template.py NLPTemplate.flat_arrays; INTEGRATION.md shows how the
 * reference's own construct code produces them).  The solver's static plan (which variables form
 * the leaves / the root of the block-arrow KKT matrix, Jacobian structure, assembly tables) is derived
 * inside the library (csrc/omgx_plan.h); `omgx_plan_describe` reports it. */
typedef struct omgx_template {
  int32_t n_var, n_par, n_con, n_atoms, n_slots, n_terms;
  int32_t n_prog, n_knots, n_pp, n_mono, n_matom;
  const int32_t* prog;      /* [n_prog*6] derived-atom program, executed in order; entry = {op, a, b, c, d, e}:
                             *   op 0  atom[c] = ppoly a / ppoly b
                             *   op 1  atoms[e ..] = the basis functions of (knots[a .. a+b), degree c) at atom d
                             *   op 2  atom[c] = cos(ppoly a)      op 3  atom[c] = sin(ppoly a) */
  const double*  knots;     /* [n_knots] */
  const int32_t* pp_ptr;    /* [n_pp+1] */
  const double*  pm_coef;   /* [n_mono] */
  const int32_t* pm_ptr;    /* [n_mono+1] */
  const int32_t* pm_atom;   /* [n_matom] */
  const int32_t* slot_pp;   /* [n_slots] */
  const int32_t* row_ptr;   /* [n_con+2], row n_con = objective */
  const double*  t_coef;    /* [n_terms] */
  const int32_t* t_slot;    /* [n_terms] */
  const int32_t* t_var;     /* [n_terms*OMGX_TERM_VARS] variable indices of the term's factors (a repeated
                             * index = a power), -1 = unused, unused entries last */
  /* row kinds: the rows with lbg == ubg (reference `optilayer.py:263-272`) */
  int32_t n_eq;
  const int32_t* eq_rows;   /* [n_eq] */
  /* optional structure hint: variables that belong to the root of the block-arrow KKT matrix (the
   * vehicle's spline coefficients, `vehicles/vehicle.py:105-120`); n_root_vars = 0: chosen automatically */
  int32_t n_root_vars;
  const int32_t* root_vars; /* [n_root_vars] */
  /* Block table (optional, n_blocks = 0: none): where every named entry of x, p and g sits -- the offsets the
   * reference's exporter hard-codes into the generated C++ (`export/export.py:302-353`, consumed by
   * `Point2Point::fillParameterDict / extractData`, `export/point2point/Point2Point.cpp:263-294`), in the order of
   * `OptiFather._var_struct / _par_struct / _con_struct` (`basics/optilayer.py:225-272`).  Entry i: name
   * "<child label>.<entry name>" (block_names: the n_blocks NUL-terminated strings one after the other), kind
   * (OMGX_BLOCK_VAR / _PAR / _CON), offset into the flat vector and shape rows x cols (column-major inside the
   * entry: spline k of a (len(basis) x n_spl) entry occupies [off + k * rows, off + (k + 1) * rows)). */
  int32_t n_blocks, block_names_len;
  const char* block_names;          /* [block_names_len] */
  const int32_t* block_kind;        /* [n_blocks] */
  const int32_t* block_off;         /* [n_blocks] */
  const int32_t* block_rows;        /* [n_blocks] */
  const int32_t* block_cols;        /* [n_blocks] */
  /* Default bounds of g (optional, has_bounds = 0: none): the LBG_DEF / UBG_DEF constants of the reference's generated
   * C++ (`export/export.py:236-262`, `Point2Point.cpp:32`), +-inf allowed; a caller without the Python front end
   * (compat/Point2Point) passes them to omgx_batch_solve. */
  int32_t has_bounds;
  const double* lbg_def;            /* [n_con] */
  const double* ubg_def;            /* [n_con] */
  /* (version 7) Lifted products and quotients: the LAST n_lift variables are auxiliary variables, the LAST n_lift rows their
   * defining equalities (row n_con - n_lift + k: linear in variable n_var - n_lift + k, whose coefficient there may be a
   * polynomial in other variables -- a quotient by a variable --, and otherwise a function of the caller's variables and of
   * auxiliaries with smaller k only).  The front end writes a product of more than OMGX_TERM_VARS variable factors or a
   * quotient by a variable expression this way (the vehicle models of `vehicles/bicycle.py:53`, `agv.py:50`, `trailer.py:28`,
   * the free end time of `examples/p2p_dubins.py`).  The solver keeps these rows satisfied exactly: every trial point of its line
   * search takes the auxiliaries from their defining rows (the Newton direction is the lifted system's, the iterates those of
   * the caller's own problem).  0: none.  lift_row0: the first of the defining rows (a front end puts them last:
   * n_con - n_lift). */
  int32_t n_lift, lift_row0;
} omgx_template;

#define OMGX_BLOCK_VAR 0
#define OMGX_BLOCK_PAR 1
#define OMGX_BLOCK_CON 2

/* What the library derived from a template (host only, needs no device). */
#define OMGX_PLAN_MAX_LEAF 16
typedef struct omgx_plan_info {
  int32_t n_leaf, n_root, n_eq, nnz_j;       /* n_root includes the phase-I variable t */
  int32_t kkt_doubles;                       /* size of the block-arrow store */
  int32_t wave_path;                         /* 1: register-resident wave-level factorisation applies */
  int32_t ws_mode;                           /* workspace placement that would be chosen (omgx_batch_workspace) */
  int64_t lds_bytes;
  int32_t leaf_size[OMGX_PLAN_MAX_LEAF], leaf_bw[OMGX_PLAN_MAX_LEAF], leaf_cpl[OMGX_PLAN_MAX_LEAF];
  int32_t n_pairs, ka_len, kh_len, kg_len;   /* assembly: Jacobian pairs; records per owner thread (pairs / Hessian / Gershgorin) */
} omgx_plan_info;

typedef struct omgx_options {
  double  tol;          /* scaled KKT tolerance ('ipopt.tol') */
  int32_t max_iter;     /* 'ipopt.max_iter' */
  double  mu_init;      /* initial barrier parameter */
  double  kappa_push;   /* rows within this distance of their bound are relaxed by t */
  double  nu_init;      /* initial weight of the phase-I variable */
  double  scale_gmax;   /* gradient-based row scaling threshold (0 = off) */
  int32_t warm_start;   /* 1: lam_g and status are in/out: an agent whose previous status was
                           Solve_Succeeded starts from x0 with the multipliers in lam_g: primal-dual warm start for receding-horizon
                           steps (the reference warm-starts IPOPT from x0 only, `problem.py:57-60,113`) */
  double  kappa_warm;   /* kappa_push used when warm_start = 1 */
  double  dw_leaf_ratio_cold;  /* cold starts weight the inertia correction of nonlinear leaf (hyperplane)
                           variables by this ratio and of root (trajectory) variables by its inverse;
                           1 = symmetric (default); 0.3 suits the Quadrotor / 3-D classes */
  double  warm_mu_factor;      /* warm starts begin at the barrier parameter clamp(warm_mu_factor * mean(s z), tol / 10,
                           mu_init).  1 (default): the average complementarity of the point handed in (safe when the
                           problem changed between the solves: ADMM x-updates); 0: at tol / 10, where the previous solve
                           of the agent ended -- receding-horizon steps of one agent then need no barrier update of
                           their own; BatchP2P sets 0.1 (tol / 10 unless the shifted point is far off that central
                           path): fewer stragglers between knot crossings */
  double  warm_z_floor;        /* multipliers handed to a warm start are lifted to warm_z_floor * tol (default 0.1; IPOPT:
                           warm_start_mult_bound_push): after a horizon shift a row may come with a small slack and a
                           vanished multiplier, which the Newton system does not see until the step runs into it
                           (hundreds of iterations at step lengths of 1e-2 observed on knot-crossing x-updates) ... */
  double  warm_z_cap;          /* ... but to no more than warm_z_cap * tol / slack (default 0.01; 0: no cap): only rows close to
                           their bound are lifted.  Without the cap the lift of all the inactive rows raises the average
                           complementarity and costs an ADMM x-update three iterations (formation bench: 4.3 instead of 1.2
                           per x-update); BatchP2P sets 0 -- its steps begin at tol / 10 whatever the complementarity
                           (warm_mu_factor 0.1), and the plain floor saves its crossing steps an iteration or two */
  int32_t max_soc;             /* (version 5; version 8: a COUNT, at most 8, as IPOPT's max_soc -- further corrections while the corrected trial still
                           leaves rows violated) 1 (default): a line search whose first trial is rejected offers the step once more with a
                           second-order correction -- one more solve with the factors of the iteration for the amount the rows
                           moved beyond their linearisation (the bilinear hyperplane rows) -- before it halves the step (IPOPT:
                           max_soc, the component replaced behind `basics/optilayer.py:60`); 0: plain backtracking.  Every template
                           class: the register-resident wave routines where the KKT panels fit them (omgx_plan_info.wave_path),
                           the blocked second solve elsewhere (spill modes included) */
  int32_t hess_approx;         /* (version 7) 0 (default): the exact Lagrangian Hessian.  1: the Hessian without the curvature of the rows (objective
                           Hessian + J' Sigma J), damped by a Levenberg-Marquardt weight that follows the accepted step length -- the
                           analogue of 'ipopt.hessian_approximation': 'limited-memory', which the reference's own examples of the
                           nonholonomic classes set (`examples/p2p_dubins.py:42`, `p2p_agv.py:43`); omgtools.backend maps that option
                           onto it.  Linear convergence (hundreds of iterations), but phase I of these classes finishes:
                           `examples/p2p_dubins.py` as shipped.  Honoured by templates on the general kernel instance (quartic terms,
                           cos / sin atoms, lifted auxiliaries); ignored by the others */
  double  compl_inf_tol;       /* (version 8) IPOPT's absolute tolerances on the unscaled problem, which stay at their documented defaults (1e-4
                           each) when the reference sets only ipopt.tol = 1e-3 (`problems/problem.py:57`): a solve then ends when the scaled
                           error is below tol AND the largest complementarity product |lam_i g_i| is below compl_inf_tol ... */
  double  constr_viol_tol;     /* ... AND the largest unscaled violation of a row is below constr_viol_tol; the barrier parameter ends at
                           min(tol, compl_inf_tol) / 10.  0 (default, rounds 1-5): neither is tested -- the scaled error alone decides.
                           omgtools.backend maps 'ipopt.compl_inf_tol' / 'ipopt.constr_viol_tol' onto them when a caller sets them */
  int32_t refine;              /* (version 9) 1: iterative refinement of a regularised Newton step (default 0).  When the Lagrangian Hessian needs an
                           inertia correction D the factors are those of K + D and the step s1 is a proximal step; (K + D) s2 = D s1 adds the
                           next term of the series towards the step of K itself -- one more solve with the factors of the iteration (IPOPT refines
                           its steps against the unperturbed system too: `min_refinement_steps`).  Taken from the second iteration of a solve
                           on, when the term is no longer than the step, not after an iteration that accepted less than a tenth of its step,
                           in cold solves once phase I is over, and only if the first trial of the line search accepts the refined step as it
                           stands (else the plain step takes the whole line search).  Templates on the wave path with the exact Hessian and no
                           lifted auxiliaries; ignored by the others and by omgx_batch_rollout (kernel instances of their own carry it: the default
                           instance stays free of its registers).  Measured: fewer iterations on every protocol of the
                           host build (tol 1e-6: 9.5 -> 6.9 per warm solve, a third of the unsolved steps), on the device +22 % / +52 % at
                           tol 1e-4 / 1e-6 but -3 % at the headline's 1e-3, where nine of ten solves end after one iteration: off by default */
} omgx_options;

typedef struct omgx_batch omgx_batch;

/* Template files.  The Python front end (omgtools.backend.save_template, after the reference's own
 * `problem.init()` built the problem) writes the template of a problem class once; a C/C++ caller reads it
 * back -- the role of the generated nlp.so in the reference's export (`export/export.py:236-262`,
 * `export/point2point/Point2Point.cpp:80-91` generateProblem).  `omgx_template_read` allocates; release with
 * `omgx_template_free` (never free() a template you filled in yourself with it). */
int  omgx_template_write(const omgx_template* tpl, const char* path);
int  omgx_template_read(const char* path, omgx_template** out);
void omgx_template_free(omgx_template* tpl);

int  omgx_version(void);
const char* omgx_last_error(void);
const char* omgx_status_string(int32_t status);
void omgx_default_options(omgx_options* o);

/* Derive the solver plan of a template without creating a batch (host only).  order [n_var+1]
 * (position -> variable, n_var = the phase-I variable) may be NULL. */
/* Block table look-up (host only): number of entries of a kind, entry i of a kind in table order, an entry by
 * name.  Return OMGX_OK, or OMGX_E_INVALID (omgx_last_error) when the template carries no such entry.  The name
 * pointer stays valid as long as the template does. */
int  omgx_template_n_blocks(const omgx_template* tpl, int32_t kind);
int  omgx_template_block_at(const omgx_template* tpl, int32_t kind, int32_t i, const char** name, int32_t* off,
                            int32_t* rows, int32_t* cols);
int  omgx_template_block(const omgx_template* tpl, int32_t kind, const char* name, int32_t* off, int32_t* rows,
                         int32_t* cols);

int  omgx_plan_describe(const omgx_template* tpl, omgx_plan_info* info, int32_t* order);

/* Create a batch of n_agents independent problems sharing one template. */
int  omgx_batch_create(const omgx_template* tpl, int32_t n_agents, int32_t device,
                       omgx_batch** out);
void omgx_batch_destroy(omgx_batch* b);
int  omgx_batch_set_options(omgx_batch* b, const omgx_options* o);
/* Launch on this hipStream_t (as void*) instead of the handle's own stream. */
int  omgx_batch_set_stream(omgx_batch* b, void* hip_stream);
/* Optional launch order for the following solves: order_device [n_agents] (device pointer, a
 * permutation of the agent indices, owned by the caller; NULL = identity).  Workgroups are handed
 * out in this order, so a caller that knows its likely stragglers (e.g. the agents that needed
 * most iterations in the previous receding-horizon step) can start them first. */
int  omgx_batch_set_order(omgx_batch* b, const int32_t* order_device);
/* Fill order_device [n_agents] from the iteration counts of the previous solve (iters_device, as
 * written by omgx_batch_solve), largest first (counts above 15 form one class; within a class the agents that carry the
 * heavier inertia correction from their previous solve come first: they are the slow ones), on the handle's stream, and
 * install it as the launch order (the receding-horizon loop calls this before every warm-started solve).  The work is
 * deferred to the next launch of the handle: an omgx_batch_predict(_ex) launch carries it as one more
 * workgroup, otherwise the next omgx_batch_solve or omgx_batch_sync runs it first -- iters_device must stay as it
 * is until then. */
int  omgx_batch_order_by_iters(omgx_batch* b, const int32_t* iters_device, int32_t* order_device);
/* Restart guesses for the following COLD solves (device-pointer solves; warm-started solves ignore them):
 * x0_alt_device [n_alt][n_agents][n_var] (device pointer, owned by the caller; n_alt = 0 / NULL switches it off).
 * An agent that does not reach Solve_Succeeded from x0 is solved again from x0_alt[0], then x0_alt[1], ... inside the
 * same launch, by the workgroup that holds it; x, lam_g, status and iters are those of its last attempt.
 * attempts_device (optional, [n_agents] int32 on the device) receives the number of restarts each agent used.
 * The reference has no such retry (`problems/problem.py:113`: its user re-initialises by hand); the results
 * equal separate OMGX_ONLY_FAILED passes from the same guesses. */
int  omgx_batch_set_restarts(omgx_batch* b, const double* x0_alt_device, int32_t n_alt, int32_t* attempts_device);
/* LDS bytes the solve kernel needs per agent (for diagnostics / DESIGN.md). */
int  omgx_batch_lds_bytes(const omgx_batch* b);
/* Workspace placement chosen at create: mode 0 = every per-agent array in LDS; 1 = KKT store in an
 * HBM slab (leaf panels by columns); 2 = + Jacobian values; 3 = + the per-row arrays (the O(n_var)
 * vectors, the matrix descriptors and a copy of the root block stay in LDS); 4 = compact KKT store and row arrays in
 * LDS, Jacobian values and row values in the slab; 5 = as 4 with the row values in LDS (4 and 5: templates on the wave
 * path, two agents per CU when the rest fits half a CU).  Every mode runs min(n_agents, n_slabs) persistent workgroups
 * that take their agents from an atomic counter. */
int  omgx_batch_workspace(const omgx_batch* b, int32_t* mode, int64_t* lds_bytes,
                          int64_t* hbm_bytes_per_slab, int32_t* n_slabs);

/* One MPC solve for every agent: the batched twin of
 *   result = solver(x0=, p=, lbg=, ubg=)  ->  x, lam_g, return_status.
 * p [B,n_par], x0 [B,n_var], lbg/ubg [B,n_con] (or [n_con] with
 * OMGX_BOUNDS_SHARED), x [B,n_var], lam_g [B,n_con], status/iters [B].
 * Asynchronous w.r.t. the host when OMGX_PTR_DEVICE is set; call omgx_batch_sync.
 * Rows: lbg == ubg is an equality row (the rows of omgx_template.eq_rows), one finite bound an inequality row, both
 * infinite a free row.  Two-sided rows lb < g < ub (`basics/optilayer.py:634-666`; version 5) are accepted where the
 * template's default bounds (has_bounds) mark them: the library solves with such a row doubled -- once per side -- and maps
 * bounds and multipliers between the caller's n_con rows and its own; lam_g of such a row is positive when the upper
 * bound is active, negative for the lower one.  A two-sided row that the default bounds do not announce ends the solve
 * with Unsupported_Bounds. */
int  omgx_batch_solve(omgx_batch* b, const double* p, const double* x0,
                      const double* lbg, const double* ubg,
                      double* x, double* lam_g, int32_t* status, int32_t* iters,
                      int32_t flags);
int  omgx_batch_sync(omgx_batch* b);

/* (version 6) The host boundary without the copy engine.  SURVEY.md 8d times "device time incl. parameter upload and
 * coefficient download": a caller that keeps p and x in host memory (`problems/problem.py:113` hands numpy arrays to the solver
 * object, C++ `Point2Point::update` std::vectors, `export/point2point/Point2Point.cpp:207-231`) would wrap every solve in
 * hipMemcpyAsync calls -- on this platform each of them is a hand-over between the compute queue and an SDMA engine, ~100 us of
 * latency per step at these sizes (0.1 - 0.7 MB).  omgx_batch_transfer moves up to OMGX_TRANSFER_MAX segments
 * dst[i] <- src[i] (bytes[i] bytes each, 8-byte aligned) with ONE small kernel on the handle's stream: ordered like every
 * other call of the handle, no engine hand-over.  Either side of a segment may be device memory or PINNED host memory
 * (hipHostMalloc / hipHostRegister, e.g. a torch pin_memory() tensor: mapped into the device's address space): the kernel
 * reads or writes it over the host link.  Pageable host memory is an error the device reports, not this call. */
#define OMGX_TRANSFER_MAX 6
int  omgx_batch_transfer(omgx_batch* b, int32_t n_seg, const void* const* src, void* const* dst, const int64_t* bytes);

/* Verification entry (host pointers, not a hot path): what the solve kernel's own tables evaluate at
 * the caller's point x [B, n_var] with parameters p [B, n_par] and multipliers lam_g [B, n_con] -- by the device code
 * of the solve (parameter stage, Jacobian items, row terms, the Hessian items of the assembly pass), unscaled:
 *   g [B, n_con] constraint values, f [B] objective, jac [B, n_con + 1, n_var] dense Jacobian of (g, f) (last row: the
 *   objective gradient), hess [B, n_var, n_var] dense symmetric Hessian of f + lam_g' g.
 * What CasADi's AD derives inside nlpsol (`basics/optilayer.py:49-60`, `expand=True`); compared with the oracle's
 * numpy restatement in tests/test_gpu_eval.py.  Any output pointer may be NULL. */
int  omgx_batch_eval(omgx_batch* b, const double* p, const double* x, const double* lam_g, double* g, double* f,
                     double* jac, double* hess);
/* Launch statistics without a host round trip: stats_device [n_slots][4] int64 on the device, zeroed and owned
 * by the caller (n_slots = 0 / NULL switches it off).  The k-th solve launch after this call adds into row
 * k % n_slots: {agents that ended with Solve_Succeeded, sum of their iteration counts over all agents it solved,
 * largest iteration count, agents it solved} -- integer atomics from inside the solve kernel.  A monitoring hook for
 * resident receding-horizon loops (the reference prints these per solve, `problems/problem.py:113-127`). */
int  omgx_batch_set_stats(omgx_batch* b, int64_t* stats_device, int32_t n_slots);
/* Attach two hipEvent_t (as void*, created by the caller with timing enabled) to the NEXT solve (or
 * omgx_batch_sample) launch only: they receive the begin and end stamps of that kernel itself (carried by its dispatch
 * packet, no extra packets on the stream).  hipEventElapsedTime(start, stop) after the launch completed = the kernel's
 * duration. */
int  omgx_batch_set_launch_events(omgx_batch* b, void* start_event, void* stop_event);
/* Device time (ms) of the last solve kernel from the handle's own event pair (attached to the dispatch like the
 * caller's pair above).  omgx_batch_set_timing(b, 0) switches it off: omgx_batch_last_kernel_ms then fails until
 * a timed solve has run again; a launch that carried the caller's pair is not timed by the handle's. */
int  omgx_batch_set_timing(omgx_batch* b, int32_t on);
int  omgx_batch_last_kernel_ms(omgx_batch* b, double* ms);
/* (version 8) What CasADi's `nlpsol` does at the head of every call behind `problems/problem.py:113` -- evaluate the constraint
 * functions and their Jacobian at x0 -- and IPOPT's own start-up (gradient-based scaling `nlp_scaling_method`, initial slacks and
 * multipliers) run as ONE launch for the whole batch ahead of the solve kernel (ipm_prepare_kernel: basis evaluation at t / T,
 * coefficient slots, Jacobian and rows at x0, row classification and scaling, warm-start multipliers; many workgroups per CU) --
 * OFF by default (measured on the 1024-agent benchmark batch, profiles/r06_prepare_ab.txt: the solve kernel loses 64 k cycles per
 * warm solve, the setup kernel costs more than that saves); omgx_batch_set_prepare(b, 1) / OMGX_PREPARE=1 switch it on.  The same
 * statements either way: the same bits (`omgx_batch_rollout` always does its setup in the kernel).  With the setup kernel on, the begin
 * stamp of the event pairs above is the setup kernel's and the end stamp the solve kernel's. */
int  omgx_batch_set_prepare(omgx_batch* b, int32_t on);
/* (version 9) The reference's stop criterion inside the solve launch.  `Simulator.run` ends a vehicle's loop at the first update
 * for which `problem.stop_criterium` holds (`execution/simulator.py:39-62`; `problems/point2point.py:98-102` ->
 * `vehicles/holonomic.py:145-151`: |state0 - poseT| <= stop_tol and |input0| <= stop_tol, Euclidean norms, stop_tol = 1e-3 by
 * default, `vehicles/vehicle.py:72`); in a batch the vehicles arrive at different updates.  under_way [n_agents] int32 on the
 * device, owned by the caller, 1 = the agent's loop is running: every solve launch after this call tests the criterion on the
 * agent's parameter vector p (offsets o_state0, o_input0, o_poseT of n_dim entries each -- the state the prediction wrote, the
 * target) before it solves an agent with under_way = 1; if it holds the flag is cleared for good.  An agent whose flag is 0 is not
 * solved: x <- x0 (it keeps its plan), lam_g and status stay as they are, iters = 0, the launch statistics and the fused
 * trajectory store skip it.  The agents under way are solved exactly as without the rule (same bits).  under_way = NULL switches
 * the rule off.  While the rule is on every solve does its own setup (omgx_batch_set_prepare is ignored).  omgx_batch_rollout applies
 * the rule at the same place of a step (after prediction, obstacle motion and knot-crossing shift): the agent's loop inside the call
 * ends there -- x, p, lam_g, status stay as they are at that step, the remaining steps log iters 0. */
int  omgx_batch_set_stop(omgx_batch* b, int32_t o_state0, int32_t o_input0, int32_t o_poseT, int32_t n_dim, double stop_tol,
                         int32_t* under_way);

/* Warm-start shift  coeffs <- T * coeffs  for the masked agents
 * (reference `point2point.py:187-198`, `optilayer.py:470-490`,
 * `spline_extra.py:165-191`).  entries: n_ent x {offset, rows, cols, T-offset};
 * Tmats: concatenated row-major (rows x rows) matrices. */
int  omgx_batch_shift(omgx_batch* b, double* x, const uint8_t* mask,
                      const int32_t* entries, int32_t n_ent,
                      const double* Tmats, int32_t n_tmat, int32_t flags);

/* Post-solve trajectory sampling (reference `vehicle.py:250-300`,
 * `spline_extra.py:406-410`, C++ `Vehicle::sampleSplines` Vehicle.cpp:112-129):
 * out[b, d, k, i] = (d-th derivative of spline k of agent b)(t0[b] + i*dt), time
 * in units of the spline domain [0,1]; coeffs [B, n_spl, L] = a slice of x.
 * out is [B, n_der, n_spl, n_samp] fp64 (or fp32 when as_f32 != 0).  knots: host pointer, n_knots <= 40.
 * With OMGX_PTR_DEVICE the call is asynchronous on the handle's stream (omgx_batch_sync to wait). */
int  omgx_batch_sample(omgx_batch* b, const double* x, int32_t coeff_off,
                       int32_t n_spl, int32_t degree, const double* knots, int32_t n_knots,
                       int32_t n_der, const double* t0, double dt, int32_t n_samp,
                       void* out, int32_t as_f32, int32_t flags);

/* Ideal prediction + horizon bookkeeping of one receding-horizon step, device-resident x [B,n_var]
 * and p [B,n_par] (reference `vehicles/vehicle.py:323-326` with `ideal_prediction`, C++
 * `Vehicle::predict` Vehicle.cpp:61-80; `problems/point2point.py:174-198`): for every agent and
 * spline k < n_spl,  p[p_state0+k] <- spline_k(tau),  p[p_input0+k] <- spline_k'(tau) * inv_T
 * (tau in the spline domain [0,1], inv_T = 1/horizon_time), and p[p_t] <- t_value (p_t < 0: skip).
 * knots: host pointer, n_knots <= 40.  Asynchronous on the handle's stream. */
int  omgx_batch_predict(omgx_batch* b, const double* x, double* p, int32_t coeff_off, int32_t n_spl,
                        int32_t degree, const double* knots, int32_t n_knots, double tau, double inv_T,
                        int32_t p_state0, int32_t p_input0, int32_t p_t, double t_value);

/* General form (device-resident x, p): for o < n_out (<= 4, <= degree + 1) and p_off[o] >= 0,
 *   p[p_off[o] + k] <- d^o/dt^o spline_k at tau = spline_k^(o)(tau) * inv_T^o
 * -- the Quadrotor's initial conditions are spl0, dspl0, ddspl0 (`vehicles/quadrotor.py:76-85`).
 * mode OMGX_PREDICT_RK4 (integrator models, `ode` = input: Holonomic, Holonomic3D): p[p_off[0] + k] is the
 * caller's current state state_in[b, k] integrated over the n_sub sample intervals of length dtau (spline
 * domain; sample time dtau / inv_T) that end at tau, with the inputs the plan holds there, by the statements
 * of `Vehicle::integrate` (`export/vehicles/Vehicle.cpp:82-110`); the derivatives o >= 1 are those at tau. */
#define OMGX_PREDICT_IDEAL 0
#define OMGX_PREDICT_RK4   1
int  omgx_batch_predict_ex(omgx_batch* b, const double* x, double* p, int32_t coeff_off, int32_t n_spl,
                           int32_t degree, const double* knots, int32_t n_knots, double tau, double inv_T,
                           int32_t n_out, const int32_t* p_off, int32_t p_t, double t_value, int32_t mode,
                           const double* state_in, int32_t n_sub, double dtau);

/* The same for the Quadrotor's own model (non-ideal prediction, `vehicles/vehicle.py:323-337` with `vehicles/quadrotor.py:
 * 149-152`: state (x, y, dx, dy, theta), inputs thrust and pitch rate, which the plan holds through its second and third
 * derivatives, `quadrotor.py:121-140`).  state_in [B, 5] (device): the vehicles' current states; they are integrated over the
 * n_sub sample intervals of length dtau (spline domain) that end at tau -- classical Runge-Kutta per interval, the inputs
 * taken linearly between the samples like the reference's interpolated odeint (`vehicle.py:412-423`) -- and written to
 * state_out [B, 5] (optional, may be state_in).  p[p_off[0] + {0, 1}] <- position of the integrated state (`quadrotor.py:
 * 110-114`: spl0 = prediction['state'][:2]), p[p_off[o] + k] <- o-th time derivative of the plan at tau for o >= 1 (dspl0,
 * ddspl0), p[p_t] <- t_value.  g: gravity. */
int  omgx_batch_predict_quadrotor(omgx_batch* b, const double* x, double* p, int32_t coeff_off, int32_t degree,
                                  const double* knots, int32_t n_knots, double tau, double inv_T, int32_t n_out,
                                  const int32_t* p_off, int32_t p_t, double t_value, const double* state_in, double* state_out,
                                  int32_t n_sub, double dtau, double g);

/* K receding-horizon steps of every agent in ONE launch.  The agents of a point-to-point batch are independent
 * (`Simulator.run` -> `Deployer.update`, `execution/deployer.py:43-79`, loops over update times for its vehicles; nothing
 * couples two problems), so a persistent workgroup runs the whole loop of an agent -- ideal prediction of the initial
 * conditions from the current plan (omgx_batch_predict_ex, ideal mode), moving obstacles advanced by
 * x += dt v + dt^2 / 2 a, v += dt a (`environment/obstacle.py:246-264`), on a knot crossing the plan shifted (omgx_batch_shift)
 * and the multipliers moved by index (lam_perm[r] = row whose multiplier warm-starts row r, -1: none), warm-started solve --
 * n_steps times before it takes the next agent: per agent the same statements, the same bits, as n_steps rounds of the
 * separate calls; what goes is the barrier between the steps of different agents.  p, x, lam_g, status, iters are updated
 * in place (device pointers: OMGX_PTR_DEVICE | OMGX_BOUNDS_DEVICE required; OMGX_BOUNDS_SHARED as for omgx_batch_solve);
 * the handle's options apply (warm_start = 1 expected), cross_options (optional) to the solves right after a crossing.
 * With omgx_batch_set_stats step k of the call fills the next free row; with omgx_batch_set_store every step writes the
 * agent's sampled trajectories as a single solve would (the last step's stay).  iters_log / status_log (optional, device,
 * [n_steps][n_agents]) keep the per-step values.  Templates of the wave path without two-sided rows only (config 1 / 2
 * class, ADMM x-update templates); others return OMGX_E_INVALID -- step with omgx_batch_solve.  Host arrays are read
 * during the call. */
typedef struct omgx_rollout_spec {
  int32_t n_steps;
  const double* tau;            /* [n_steps] host: spline-domain time of the prediction */
  const double* t_rel;          /* [n_steps] host: value written to p[p_t] (time since the last knot) */
  const uint8_t* crossed;       /* [n_steps] host: 1 = the step crosses a knot (shift before its solve) */
  int32_t coeff_off, n_spl, degree, n_knots, n_out;      /* the plan: as omgx_batch_predict_ex */
  const double* knots;          /* [n_knots] host */
  const int32_t* p_off;         /* [n_out] host: p offset of the o-th time derivative (-1: skip) */
  int32_t p_t; double inv_T;
  int32_t n_obst; const int32_t* obst;      /* [n_obst][4] host = {p_x, p_v, p_a, n_dim} of the obstacles that move (<= 8) */
  double dt;                                /* update time */
  const int32_t* shift_entries; int32_t n_ent; const double* shift_T; int32_t n_tmat;     /* as omgx_batch_shift (host) */
  const int32_t* lam_perm;      /* [n_con] host */
  const omgx_options* cross_options;        /* NULL: the handle's */
  int32_t* iters_log; int32_t* status_log;  /* device or NULL */
} omgx_rollout_spec;
int  omgx_batch_rollout(omgx_batch* b, const omgx_rollout_spec* sp, double* p, double* x, const double* lbg, const double* ubg,
                        double* lam_g, int32_t* status, int32_t* iters, int32_t flags);

/* `Vehicle.store` (reference `vehicles/vehicle.py:250-300` -> `splines2signals`, e.g.
 * `vehicles/holonomic.py:116-124`) for the whole batch, device pointers only:
 *   out[b, o, k, i] = d^o/dt^o spline_k(t0[b] + i*dt)   (o < n_der: state, input, dinput ...; time
 *                     derivatives, i.e. spline-domain derivatives * inv_T^o),
 *   v_tot[b, i]     = |first time derivative|           (optional, needs n_der >= 2).
 * knots: host pointer.  A fixed-T plan is one segment: `concat_splines` (`spline_extra.py:308-404`) is the
 * identity on it.  omgx_batch_store samples a given x; omgx_batch_set_store makes every following
 * omgx_batch_solve write the same outputs for the solution it just found, inside the solve kernel (the
 * solution is still in LDS there; sp == NULL switches it off again).  The arrays a spec points at must stay
 * valid while it is set. */
typedef struct omgx_store_spec {
  double* out;            /* [B, n_der, n_spl, n_samp] device */
  double* v_tot;          /* [B, n_samp] device, or NULL */
  const double* t0;       /* [B] device: first sample, spline domain */
  const double* knots;    /* [n_knots] host, n_knots <= 40 */
  int32_t coeff_off, n_spl, degree, n_knots, n_der, n_samp;
  double  dt, inv_T;
} omgx_store_spec;
int  omgx_batch_store(omgx_batch* b, const double* x, const omgx_store_spec* sp);
int  omgx_batch_set_store(omgx_batch* b, const omgx_store_spec* sp);

/* Same shift on any device-resident row-major array (stride doubles per row, n_rows rows):
 * used for the ADMM consensus state on a knot crossing (`problems/admm.py:477-491`).
 * The handle keeps the table sets it has seen on the device (16, found again by content): a loop that shifts with the
 * same few sets at every crossing uploads each once.  n_rows = 0 uploads the set and launches nothing (data may be
 * NULL) -- ahead of a loop that must not wait for a copy. */
int  omgx_shift_rows(omgx_batch* b, double* data, int32_t stride, int32_t n_rows, const uint8_t* mask,
                     const int32_t* entries, int32_t n_ent, const double* Tmats, int32_t n_tmat);

/* ---- Formation ADMM (reference `problems/admm.py`, `problems/formation.py`) -------------
 * Layout of one agent's consensus data inside x [B,n_var] and p [B,n_par]; shared vector
 * x_i = fleet-centre coefficients, axis-major, ns = n_dim*L doubles. */
typedef struct omgx_admm_layout {
  int32_t n_dim, L, n_nghb;
  int32_t x_spl;                    /* offset of splines_seg0 in x */
  int32_t p_rel;                    /* rel_pos_c in p */
  int32_t p_zi, p_zji, p_li, p_lji; /* z_i, z_ji, l_i, l_ji in p (`admm.py:68-72`) */
} omgx_admm_layout;

/* x_i[b] = splines_seg0[b] + rel_pos_c[b]   (`vehicles/vehicle.py:234-240`); x_i [B, ns]. */
int  omgx_admm_center(omgx_batch* b, const omgx_admm_layout* lay, const double* x, const double* p,
                      double* x_i);
/* z-update (closed-form equality QP, `admm.py:117-168, 407-445`), lambda update (`447-466`) and
 * residuals (`493-508`) of every local agent.  x_ext [B+halo, ns]: local agents first, then halo
 * rows; nbr [B, n_nghb] indexes x_ext.  M, F: (n_all x n_all) row-major, n_all = (1+n_nghb)*ns
 * (z_all = M (x_all + l_all/rho); F = forward knot transform).  z_i/l_i are read from and written
 * to p; z_ij, l_ij [B, n_nghb, ns] are updated in place; res [B,3] = per-agent (pr, dr, cr). */
int  omgx_admm_update(omgx_batch* b, const omgx_admm_layout* lay, const double* x_ext,
                      const int32_t* nbr, const double* M, const double* F, double rho,
                      double* p, double* z_ij, double* l_ij, double* res);
/* The same, and sums [3] (device) receives the fleet sums of the three residual columns of this rank's agents
 * (`admm.py:601-605`), added up in a fixed order by the workgroup that finishes last -- no second launch. */
int  omgx_admm_update_sums(omgx_batch* b, const omgx_admm_layout* lay, const double* x_ext,
                           const int32_t* nbr, const double* M, const double* F, double rho,
                           double* p, double* z_ij, double* l_ij, double* res, double* sums);
/* neighbour exchange (`admm.py:468-475`): z_ji[b,k] = z_ij_ext[nbr[b,k], slot[b,k]] (same for l),
 * written into p.  z_ij_ext / l_ij_ext [B+halo, n_nghb, ns]. */
int  omgx_admm_communicate(omgx_batch* b, const omgx_admm_layout* lay, const int32_t* nbr,
                           const int32_t* slot, const double* z_ij_ext, const double* l_ij_ext,
                           double* p);


/* ---- the same three steps for a fleet sharded over ranks (one process per GPU): no copies around the collectives -----
 * The caller keeps two exchange buffers per rank, each [B + world * rows, width] with the local agents' rows on top and
 * the all_gather result written straight behind them (nbr indexes the whole buffer):
 *   x_all  [B + world*n_max_pub, ns]           center_ex writes the local rows and, into x_send, the published ones
 *   zl_all [B + world*(n_max_pub + 1), 2 w]    w = n_nghb*ns; row = [z_ij | l_ij]; update_ex updates the local rows in
 *                                               place (z_ij = zl_all, l_ij = zl_all + w, zl_stride = 2 w) and writes the
 *                                               published rows into zl_send; `sums` may point at the extra row of
 *                                               zl_send so that the residual sums ride along
 *   communicate_ex reads the gathered rows (z_ij_ext = zl_all, l_ij_ext = zl_all + w) and adds up the sums rows of
 *   all ranks (sum_rows = first extra row, sum_stride = rows * 2 w doubles between ranks) into sums_out [3].
 * One multi-rank iteration = solve, center_ex, all_gather, update_ex, all_gather, communicate_ex. */
int  omgx_admm_center_ex(omgx_batch* b, const omgx_admm_layout* lay, const double* x, const double* p, double* x_i,
                         const int32_t* pub_rows, int32_t n_pub, double* x_send);
/* The centre step riding on the x-update: after omgx_batch_set_center every omgx_batch_solve writes x_i (and the
 * published rows into x_send) for the solution it just found, from the solve kernel's epilogue -- what
 * omgx_admm_center_ex(b, lay, x, p, x_i, pub_rows, n_pub, x_send) would write for the x and p of that solve; one launch
 * less per ADMM iteration (solve, all_gather, update_ex, all_gather, communicate_ex).  pub_rows: HOST pointer here, every
 * row at most once.  x_i / x_send must stay valid while set; lay == NULL switches it off. */
int  omgx_batch_set_center(omgx_batch* b, const omgx_admm_layout* lay, double* x_i, const int32_t* pub_rows, int32_t n_pub,
                           double* x_send);
int  omgx_admm_update_ex(omgx_batch* b, const omgx_admm_layout* lay, const double* x_ext, const int32_t* nbr,
                         const double* M, const double* F, double rho, double* p, double* z_ij, double* l_ij,
                         int32_t zl_stride, double* res, double* sums, const int32_t* pub_slot, double* zl_send,
                         int32_t send_stride);
int  omgx_admm_communicate_ex(omgx_batch* b, const omgx_admm_layout* lay, const int32_t* nbr, const int32_t* slot,
                              const double* z_ij_ext, const double* l_ij_ext, int32_t zl_stride, double* p,
                              const double* sum_rows, int32_t n_sum_rows, int32_t sum_stride, double* sums_out);

#ifdef __cplusplus
}
#endif
#endif /* OMGX_H */
