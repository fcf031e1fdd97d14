// omgx_core.h -- per-agent interior-point solve, written once for two targets:
//
//   * hipcc --offload-arch=gfx950: one 256-thread workgroup per agent, all
//     state in LDS (omgx_kernels.hip: ipm_solve_kernel);
//   * g++ -DOMGX_HOST_PORT: the same statements executed by one host thread
//     (oracle/port/omgx_port.cpp) -- used ONLY as the timed CPU baseline and as
//     a debugging aid; the product library never contains or calls it.
//
// What this replaces: IPOPT + MUMPS behind CasADi's nlpsol, called from
// reference `problems/problem.py:113` (object built in `basics/optilayer.py:
// 49-60`).  What it evaluates: the polynomial rows the front end extracted from
// the reference's define_constraint calls.
//
// Algorithm (independent numpy statement in oracle/ipm_numpy.py):
// feasible primal-dual interior point with an embedded phase I ("big-M")
//     min f(x) + nu*t   s.t.  h_i(x) - t*v_i + s_i = 0, s>0;  c_E(x) - t*c_E(x0) = 0; t>0
// gradient-based row scaling, exact Lagrangian Hessian with inertia correction,
// block-arrow LDL' (leaves = hyperplane blocks, root = trajectory block + t +
// equality multipliers), fraction-to-boundary, Armijo on the barrier function,
// monotone barrier update.
#pragma once
#include <stdint.h>
#include <math.h>

#ifdef OMGX_HOST_PORT
#define OMGX_FN inline
#define OMGX_HD inline
#else
#define OMGX_FN __device__ __forceinline__
#define OMGX_HD __host__ __device__ inline
#endif

#ifdef OMGX_HOST_PORT
#include <vector>
static thread_local std::vector<double> omgx_rbak;      // root block before its factorisation (host twin of the wave path)
#endif
#ifdef OMGX_COUNT_FACT
#include <atomic>
static std::atomic<long> omgx_dbg_nfact(0);
static std::atomic<long> omgx_dbg_cnt[12];      // 0 leaf failures, 1 root failures, 2 dw=0 attempts that failed, 3 decrease attempts that failed, 4 iterations
#endif
namespace omgx {

enum { ROW_FREE = 0, ROW_UPPER = 1, ROW_LOWER = 2, ROW_EQ = 3, ROW_BAD = 4 };
enum { OP_DIV = 0, OP_BSPL = 1, OP_COS = 2, OP_SIN = 3 };

struct Dims {
  int n_var, n_par, n_con, n_atoms, n_slots, n_terms, n_prog;
  int N;        // n_var + 1 (phase-I variable t is variable index n_var)
  int n_leaf, n_root, n_eq, nnz_j, root_off;
  int nr;       // n_root + n_eq: order of the root block
  int n_pairs;  // Jacobian entry pairs contributing to J' Sigma J
  int n_eqe;    // Jacobian entries of the equality rows
  int max_leaf, max_cpl;
  int col_doubles;   // scratch for the blocked LDL' (staging + panel buffers)
  int col_small_noroot;   // col_small without the LDS copy of the root block (mode 6)
  int col_small;     // the same for the spill modes (left-looking leaf sweep only: inverse pivots + parked diagonal blocks per leaf): kept in LDS there
  int mono_packed;   // 1: every parameter monomial has <= 4 atoms, Tables::pm_rec / sl_ell hold MonoRec; 2: <= 8 atoms, MonoRec8; 0: CSR tables only
  int n_long;        // slots with more than OMGX_SLOT_CAP monomials (the first n_long entries of Tables::sl_list)
  int n_hess;        // number of terms with >= 2 factors
  int quartic;       // 1: some term has four factors
  int general;       // 1: quartic terms, cos / sin atoms or basis rows of degree > 5: the kernel instance that carries them
  int rp_packed;     // 1: rows and positions fit 16 bits each, Tables::je_rp is valid
  int n_knots;       // total length of the knot vectors of the atoms program (copied to LDS per solve)
  int atoms_alias;   // 1: atoms and knots live in the space of gbar | sol (they are dead once the slots are evaluated, gbar and sol
                     //    are first written inside the iteration): n_atoms + n_knots <= 2 N + n_eq
  int wave_ok;       // 1: every panel of the KKT store fits one wave (register-resident factorisation, omgx_wave.h)
  int n_jv;          // Jacobian entries whose value depends on x (the others are constant over a solve)
  int n_jv4, n_ja4;  // owners of the Jacobian item tables (four entries each): x-dependent entries, all entries
  int ka_len, kh_len, kg_len;   // records per owner bin (longest bin) of the pair / Hessian / Gershgorin passes
  int n_kafix, n_kgfix;         // targets whose run was cut (fix-up records)
  int n_khfix;                  // the same for the Hessian items
  int side_off, dump_off;       // side slots / per-lane dump slots behind the KKT store
  int n_cs_own;      // owner threads of the column sums J'w (chunks of the columns)
  int kg_side_dinv;  // 1: the side sums of the Gershgorin pass live in w.dinv, 0: in the side slots behind the KKT store
  int n_owner;       // owner bins in use (<= OMGX_NBIN, the stride of the record tables): the threads of the workgroup the plan was made for
  int n_lift, lift_depth;   // auxiliary variables of lifted products / quotients (omgx_template::n_lift) and the length of their longest chain
};

// one parameter monomial coef * prod atoms[a_k] as a single 16-byte record (a_k = -1: unused): one
// load per monomial instead of the pointer chase pm_ptr -> pm_atom -> atoms of the CSR form
struct MonoRec { double coef; int16_t a0, a1, a2, a3; };
// the same with up to eight atoms (Dims::mono_packed == 2: ADMM objectives multiply rho, multipliers and basis values)
struct MonoRec8 { double coef; int16_t a0, a1, a2, a3, a4, a5, a6, a7; };

// Owner-computes records (omgx_plan.h): every sum of the solve has one owner thread and a fixed order.  A term is
// coef * slot * x_v0 x_v1 x_v2 x_v3 (OMGX_TERM_VARS = 4 factors, -1: unused; a repeated index is a power).
// JItem: one contribution coef * slot * x[va] * x[vb] * x[vc] (v = -1: factor 1) to a Jacobian entry: the term without
// one of its factors.
// HItem: one contribution of a nonlinear term to a KKT address (or, for the Gershgorin sums, to a position): the term
// without two of its factors, h = lambda_row * coef * slot * x[vthird] * x[vfourth]; kind 1 = the two factors that
// were taken out are the same variable (a diagonal entry: both orders of the pair land on it).
// (slot indices are 16 bit in the 16- and 24-byte records: the plan refuses templates with more than 32767 slots)
#define OMGX_TV 4
struct JItem { double coef; int16_t slot, va, vb, vc; };
// one term of a row for the row-value passes (24 bytes; padding records have coef 0)
struct RowTerm { double coef; int32_t slot; int16_t v0, v1, v2, v3; };
struct HItem { double coef; int32_t row, target; int16_t slot, vthird, vfourth, kind; };

struct Tables {   // read-only, shared by all agents (global memory)
  const int32_t* prog; const double* knots;
  const int32_t* pp_ptr; const double* pm_coef; const int32_t* pm_ptr; const int32_t* pm_atom;
  const int32_t* slot_pp;
  const MonoRec* pm_rec;    // [n_mono] packed form of (pm_coef, pm_ptr, pm_atom); valid if Dims::mono_packed
  const int32_t* slot_rng;  // [n_slots][2] monomial range of every slot (= pp_ptr[slot_pp[s]], pp_ptr[slot_pp[s] + 1])
  const int32_t* row_ptr; const double* t_coef; const int32_t* t_slot; const int32_t* t_var;
  const int32_t* order; const int32_t* leaf_off; const int32_t* blk;
  const int32_t* eq_rows; const int32_t* eq_index;
  const int32_t* cpl_ptr; const int32_t* cpl_idx; const int32_t* cpl_map;
  const int32_t* d_off; const int32_t* b_off;   // panel offsets / leading dimensions inside the KKT store
  // compact store of the wave path (omgx_plan.h `compact`): per leaf the offset of its carried rows (sparse diagonal
  // leaf: the number of coupling slots S), the row stride and width of its band, its kind (0 banded panel, 1 sparse
  // diagonal leaf); dl_pos[4 * leaf_off[l] + s * n_l + j]: root position coupled through slot s of variable j (-1: none)
  const int32_t* lf_w; const int32_t* lf_ldb; const int32_t* lf_band; const int32_t* lf_kind; const int32_t* dl_pos;
  // precomputed KKT addresses (HostPlan): Jacobian pairs, t-column, diagonal, Hessian terms
  const int32_t* eqe3;      // [n_eqe][3] = {Jacobian entry, KKT address, row} of the equality-row entries
  const int32_t* pair4;     // [n_pairs][4] = {Jacobian entry a, entry b, KKT address, row}: one 16-byte record per pair
  const int32_t* je_row; const int32_t* diag_addr;
  const int32_t* je_rp;     // [nnz_j] (row << 16) | position of every Jacobian entry (valid if Dims::rp_packed)
  const double* reg_w;   // [N] position order: inertia-correction class (+1 nonlinear root variable, -1 nonlinear leaf variable, else the weight itself: OMGX_DW_LINEAR)
  const int32_t* leaf_bw;   // [n_leaf] half bandwidth of the leaf block in its (reverse Cuthill-McKee) order
  const int32_t* tq_addr;   // [n_var] KKT address of (t, q)
  // owner-computes tables
  const int32_t* row_perm;  // [n_con] rows, longest term list first
  // ELL tables of the row / column / entry owners: record `step` of owner i at index step * n_owner + i
  // (coalesced across the lanes), padded with null records; *_glen[i >> 6] = steps the 64 owners of a
  // group need (rounded up to the batch the loops load at a time)
  const RowTerm* rt_ell;    // [rt_steps][n_con] terms of row row_perm[i]
  const int32_t* rt_glen;
  const int32_t* jp_ell;    // [jp_steps][n_con][2] {Jacobian entry, position} of row row_perm[i] (padding: entry nnz_j = 0.0)
  const int32_t* jp_glen;
  const int32_t* cs_ell;    // [cs_steps][n_cs_own][2] {Jacobian entry, row} of column-sum owner o (padding: entry nnz_j = 0.0)
  const int32_t* cs_glen;
  const int32_t* cs_own;    // [n_var + 1] owners of the column in slot j: cs_own[j] .. cs_own[j + 1]
  const int32_t* cs_col;    // [n_var] column (position) of slot j (columns by decreasing length)
  // Jacobian items, four entries per owner (jac_entries4): the x-dependent entries (every iteration) and all entries (setup)
  const JItem* jv_ell;      // [jv_steps][4][n_jv4] items of the owner's k-th entry (entries by decreasing item count, four in a row per owner)
  const int32_t* jv_own;    // [4][n_jv4][2] {entry, row} (-1: the owner has no k-th entry)
  const int32_t* jv_glen;   // [n_jv4 / 64] items per entry the group walks (1 or even)
  const JItem* ja_ell;      // [ja_steps][4][n_ja4]
  const int32_t* ja_own;
  const int32_t* ja_glen;
  const int32_t* sl_list;   // [n_slots] slots by decreasing monomial count
  const MonoRec* sl_ell;    // [sl_steps][n_slots] monomials of slot sl_list[i] (padding: coef 0)
  const int32_t* sl_glen;
  // lifted auxiliaries (Dims::n_lift), level by level: lift_rec[k] = {ELL slot of the defining row, the auxiliary variable}; the records of
  // level L (rows that read auxiliaries of the levels below only) are lift_lev[L] .. lift_lev[L + 1]
  const int32_t* lift_rec; const int32_t* lift_lev;
  const int32_t* cs_ptr;    // [n_var + 1] column (position) -> its entries, row order
  const int32_t* cs_rec;    // [.][2] {Jacobian entry, row}
  const int32_t* obj_ent;   // [n_var] objective-row entry of the position (-1: none)
  // ELL layout: record r of owner bin b at index r * OMGX_NBIN + b (coalesced across the lanes), a bin's
  // records sorted by target; the target is stored only in the last record of its run (-1 in the others
  // and in the padding): the owner adds every record to a running sum and stores / restarts where it sees one
  const int32_t* ka_rec;    // [ka_len][OMGX_NBIN][4] pair records {entry a, entry b, KKT address, row}
  const HItem* kh_rec;      // [kh_len][OMGX_NBIN] Hessian items (target = KKT address)
  const HItem* kg_rec;      // [kg_len][OMGX_NBIN] Gershgorin items (target = position, N + k = side slot k)
  const int32_t* kh_fix;    // [n_khfix][3] the same for cut runs of Hessian items (after hess_bin)
  const int32_t* ka_fix;    // [n_kafix][3] {KKT address, first side slot, number of side slots}: address += the slots, in order
  const int32_t* kg_fix;    // [n_kgfix][3] the same for the Gershgorin sums (position)
};

struct Opts {
  double tol; int max_iter; double mu_init, kappa_push, nu_init, scale_gmax;
  int warm_start;      // 1: lam0 holds the multipliers of the previous solve (primal-dual warm start)
  double kappa_warm;   // kappa_push used for warm starts
  double dw_leaf_ratio_cold;   // cold starts: inertia-correction weight of nonlinear leaf variables (root: its inverse)
  int prio_iter;               // device: iteration from which a solve runs at raised wave priority (0: never)
  double warm_mu_factor;       // warm starts: mu_0 = clamp(warm_mu_factor * mean(s z), tol / 10, mu_init)
  double warm_z_floor;         // warm starts: multipliers lifted to max(OMGX_WARM_ZMIN, min(warm_z_floor * tol, warm_z_cap * tol / slack))
  double warm_z_cap;           // (0: no cap)
  int max_soc;                 // > 0: a rejected first trial of the line search is answered by up to max_soc second-order corrections (every template class: kkt_solve2_wave / kkt_solve2)
  int hess_approx;             // 1: no constraint curvature in the Hessian, damping driven by the accepted step length (general instances; include/omgx.h)  // (version 8) IPOPT's absolute tolerances on the UNSCALED problem beside `tol` (its documented defaults: compl_inf_tol = constr_viol_tol = 1e-4,
  // which the reference leaves in force when it sets ipopt.tol = 1e-3, `problems/problem.py:57`); 0: not tested (rounds 1-5).  The barrier
  // parameter ends at min(tol, compl_tol) / 10.
  double compl_tol, viol_tol;
  int refine;                  // (version 9) 1: iterative refinement of a regularised Newton step, from the second iteration of a solve on (default 0)
};

// fixed constants of the iteration (same values in oracle/ipm_numpy.py DEFAULTS)
#ifndef OMGX_KAPPA_EPS
#define OMGX_KAPPA_EPS   10.0
#endif
#define OMGX_KAPPA_MU    0.2
#define OMGX_THETA_MU    1.5
#define OMGX_TAU_MIN     0.99
#define OMGX_DELTA_C     1e-8    // regularisation of the equality block: delta_c = OMGX_DELTA_C * mu^(1/4) (IPOPT's delta_c_bar mu^kappa_c)
#define OMGX_ETA         1e-4
#define OMGX_PHI_NOISE   1e-10   // predicted merit decrease (relative) below which the Armijo test is skipped
#define OMGX_DW_FIRST    1e-4
#ifndef OMGX_DW_INC
#define OMGX_DW_INC      10.0
#endif
#ifndef OMGX_DW_DEC
#define OMGX_DW_DEC      (1.0 / 3.0)
#endif
#define OMGX_DW_MAX      1e10
#define OMGX_DW_ZERO     1e-9
#define OMGX_DW_HEAVY    10.0
#ifndef OMGX_KAPPA_EPS_HEAVY
#define OMGX_KAPPA_EPS_HEAVY 100.0    // (round 4 tried 30: same-box A/B, 4096 agents: config 2 cold 91.4k vs 90.0k solves/s -- noise -- but the Quadrotor class 12.2k vs 15.7k cold, 25.2 vs 21.7 ms per warm step: kept at 100)
#endif
#ifndef OMGX_DW_BACKOFF_MAX
#define OMGX_DW_BACKOFF_MAX 8
#endif
#ifndef OMGX_EXPAND_MAX
#define OMGX_EXPAND_MAX   16.0    // longest multiple of a regularised Newton step the line search offers
#endif
#ifndef OMGX_EXPAND_DW
#define OMGX_EXPAND_DW    1e-2    // ... when the inertia correction is at least this heavy
#endif
#ifndef OMGX_EXPAND_FROM
#define OMGX_EXPAND_FROM  2       // ... after this many iterations in a row that accepted their full step at the first trial
#endif
#define OMGX_LS_RETRY    3       // line-search failures in a row that are answered by a heavier inertia correction
#define OMGX_LS_RETRY_DW 100.0
#define OMGX_DW_CAP_FLOOR 0.03  // share of dw every nonlinear variable keeps under the Gershgorin cap
#ifndef OMGX_DW_CLAMP_FROM
#define OMGX_DW_CLAMP_FROM 1.0  // an inertia correction above this is compared with the Gershgorin guarantee (round 5, see the factorisation loop)
#endif
#define OMGX_DW_LINEAR   1e-8   // relative inertia correction of variables that only appear linearly
#ifndef OMGX_FTB_ACTUAL
#define OMGX_FTB_ACTUAL  0.5     // share of the linear fraction-to-boundary bound (1 - tau) s the slack of a row must really keep at an accepted trial point
#endif
#define OMGX_S_MAX       100.0
#define OMGX_KAPPA_SIGMA 1e10
#define OMGX_MAX_BACKTRACK 25
#define OMGX_NU_MAX      1e8
#ifndef OMGX_STALL_ITERS
#define OMGX_STALL_ITERS 20
#endif
#define OMGX_WARM_ZMIN   1e-8
#define OMGX_MAX_LEAF    16
#define OMGX_BMAT_DOUBLES 5      // sizeof(BMat) / 8
#define OMGX_PAN_LD 5      // panel buffer row stride: U[4] + pad (odd: conflict-free row-per-lane access)
#define OMGX_STAGE_LD 20   // per matrix: 4x4 block rows [16] + inverse pivots of the block [4]
#define OMGX_MIN_LEAF    8       // smaller components are gathered into one leaf
#define OMGX_NBIN        512     // owner bins of the assembly passes (= threads of the solve workgroup)
#define OMGX_SLOT_CAP    16      // monomials of a parameter slot one thread sums (setup); the rest of a longer slot: one wave
#define OMGX_REC_BATCH   8       // records an owner loads at a time (all in flight together)
#define OMGX_RUN_CAP     16      // longest run of records summed by one owner (longer runs are cut, see omgx_plan.h)
#define OMGX_RUN_CAP_H   64      // the same for the Hessian items (only templates with lifted auxiliaries have such runs)
#define OMGX_WAVE_ROWS   64      // register rows of a panel the wave-level routines take (lanes)
#define OMGX_WAVE_COLS   40      // columns (registers per lane)

// Per-agent work arrays (LDS on the device, heap on the host port).
struct Work {
  double *atoms, *slots, *knots;
  double *x, *xt;                 // [N] variable order, x[n_var] = t
  double *hv, *ht;                // [n_con] scaled row values (h for inequality, c for equality)
  double *rho, *vv;               // [n_con] row scale (signed), phase-I weights
  double *z, *ds;                 // [n_con] multiplier (y for equality rows), slack step
  // (the slack of an inequality row is not stored: every iterate keeps s = t v - h exactly -- `row_slack`; the bound of a
  // row is read from the caller's lbg / ubg where the line search needs it)
  double *jval;                   // [nnz_j] scaled Jacobian entries (objective row unscaled)
  double *gbar, *sol;             // [N], [N + n_eq]   position order
  double *kkt;                    // D_l (packed) | B_l | R (packed)
  double *col;                    // [col_doubles] blocked-LDL' staging + panel buffers
  double *root;                   // spill modes: LDS copy of the packed root block (+ its right-hand-side row) from the Schur step on (inside col)
  double *dinv;                   // [N] inverse leaf pivots
  int8_t *rtype;                  // [n_con]
  double *red;                    // reduction scratch [64]
};

// Workspace placement.  Mode 0 keeps every array in LDS (the fast path: cfg 1/2/4).  Problems
// whose arrays exceed the 160 KiB of one CU spill the largest ones to a per-workgroup slab in
// HBM (L2/MALL-cached), largest first:
//   mode 1  kkt                  mode 2  + jval
//   mode 3  + the six [n_con] row arrays and rtype (only the O(n_var) vectors stay in LDS)
// Mode 4 is the other direction: templates on the register-resident wave path whose workspace then fits HALF a CU --
// two workgroups (agents) per CU -- put the Jacobian values and the row values hv into the slab and keep the
// (compact) KKT store and every array the owner passes gather from in LDS.
// Mode 5 (round 4) is mode 4 with the row values hv back in LDS, for templates that still fit half a CU then (config 2: 80,280 B):
// hv is read by a dozen row passes per iteration, each a dependent global load per pass of the workgroup (+2 % solves/s).
// Mode 6 (round 5) is mode 3 with the root block left in the slab as well: templates whose root block (its variables + ALL
// equality rows: the classes with lifted auxiliaries carry hundreds, include/omgx.h n_lift) does not fit LDS even alone
// (Bicycle: order 439 = 773 KB).  Only the O(n_var) vectors, the descriptors and the panel buffers stay on chip.
enum { WS_LDS = 0, WS_KKT_HBM = 1, WS_JAC_HBM = 2, WS_ROWS_HBM = 3, WS_JAC_ONLY = 4, WS_JAC_HV = 5, WS_ROOT_HBM = 6, WS_MODES = 7 };
OMGX_HD constexpr bool ws_kkt_hbm(int mode) { return (mode >= WS_KKT_HBM && mode <= WS_ROWS_HBM) || mode == WS_ROOT_HBM; }
OMGX_HD constexpr bool ws_jac_hbm(int mode) { return mode >= WS_JAC_HBM; }
OMGX_HD constexpr bool ws_rows_hbm(int mode) { return mode == WS_ROWS_HBM || mode == WS_ROOT_HBM; }
OMGX_HD constexpr bool ws_hv_hbm(int mode) { return mode == WS_ROWS_HBM || mode == WS_JAC_ONLY || mode == WS_ROOT_HBM; }
OMGX_HD constexpr bool ws_root_lds(int mode) { return ws_kkt_hbm(mode) && mode != WS_ROOT_HBM; }      // spill modes 1-3: the root block is copied to LDS for its factorisation

OMGX_HD size_t root_doubles(const Dims& d) { return ((size_t)(d.nr + 1) * (d.nr + 2)) / 2; }

OMGX_HD void work_split(const Dims& d, int kkt_doubles, int mode, size_t* lds, size_t* hbm) {
  size_t nl = 0, ng = 0;
  nl += d.n_slots + (d.atoms_alias ? 0 : d.n_atoms + d.n_knots);
  nl += 2 * (size_t)d.N;
  nl += d.N + (d.N + d.n_eq);
  nl += d.N;                      // dinv
  nl += 64;                       // red
  const size_t rows = 5 * (size_t)d.n_con + (d.n_con + 7) / 8;
  (ws_rows_hbm(mode) ? ng : nl) += rows;
  (ws_hv_hbm(mode) ? ng : nl) += d.n_con;             // hv: only ever read by the thread that owns the row
  (ws_jac_hbm(mode) ? ng : nl) += d.nnz_j + 1;        // + one slot that stays 0.0 (padding records point at it)
  // (the spill modes keep the matrix descriptors and the small panel scratch of the leaf sweep in LDS: every row of
  // every block column reads them)
  if (ws_kkt_hbm(mode)) { ng += (size_t)kkt_doubles; nl += ws_root_lds(mode) ? d.col_small : d.col_small_noroot; } else nl += (size_t)kkt_doubles + d.col_doubles;
  *lds = nl; *hbm = ng;
}

OMGX_HD size_t work_doubles(const Dims& d, int kkt_doubles) {
  size_t nl, ng;
  work_split(d, kkt_doubles, WS_LDS, &nl, &ng);
  return nl;
}

// MODE is a compile-time constant so that every pointer keeps a provable address space
template <int MODE>
OMGX_HD void work_carve_split(Work& w, double* lds, double* hbm, const Dims& d, int kkt_doubles) {
  double* p = lds;
  double* g = hbm;
  w.slots = p; p += d.n_slots;
  if (!d.atoms_alias) { w.atoms = p; p += d.n_atoms; w.knots = p; p += d.n_knots; }
  w.x = p; p += d.N;             w.xt = p; p += d.N;
  w.gbar = p; p += d.N;          w.sol = p; p += d.N + d.n_eq;
  if (d.atoms_alias) { w.atoms = w.gbar; w.knots = w.gbar + d.n_atoms; }
  w.dinv = p; p += d.N;
  w.red = p; p += 64;
  if (ws_rows_hbm(MODE)) {
    w.ht = g; g += d.n_con;        w.rho = g; g += d.n_con;     w.vv = g; g += d.n_con;
    w.z = g; g += d.n_con;         w.ds = g; g += d.n_con;
    w.rtype = (int8_t*)g; g += (d.n_con + 7) / 8;
  } else {
    w.ht = p; p += d.n_con;        w.rho = p; p += d.n_con;     w.vv = p; p += d.n_con;
    w.z = p; p += d.n_con;         w.ds = p; p += d.n_con;
    w.rtype = (int8_t*)p; p += (d.n_con + 7) / 8;
  }
  if (ws_hv_hbm(MODE)) { w.hv = g; g += d.n_con; } else { w.hv = p; p += d.n_con; }
  if (ws_jac_hbm(MODE)) { w.jval = g; g += d.nnz_j + 1; } else { w.jval = p; p += d.nnz_j + 1; }
  if (ws_kkt_hbm(MODE)) { w.kkt = g; g += kkt_doubles; w.col = p; p += ws_root_lds(MODE) ? d.col_small : d.col_small_noroot; }
  else { w.kkt = p; p += kkt_doubles; w.col = p; p += d.col_doubles; }
  // spill modes: the root block is copied behind the root's panel buffer before the Schur updates -- over the leaf
  // sweep's scratch, which is dead by then (Dims::col_small covers both)
  w.root = ws_root_lds(MODE) ? w.col + (OMGX_BMAT_DOUBLES + OMGX_STAGE_LD) * (OMGX_MAX_LEAF + 1) + OMGX_PAN_LD * (d.nr + 1) : nullptr;
}

OMGX_HD void work_carve(Work& w, double* base, const Dims& d, int kkt_doubles) {
  work_carve_split<WS_LDS>(w, base, nullptr, d, kkt_doubles);
}

// ---------------------------------------------------------------------------
// execution context: thread id, barrier, reductions, LDS atomic add
// ---------------------------------------------------------------------------
#ifdef OMGX_HOST_PORT
struct Ctx {
  static constexpr bool wave_only = false, hbm = false, no_wave = false, root_lds = false, general = true, prep = false, refine = true;
  double* red;
  int tid() const { return 0; }
  int nthr() const { return 1; }
  void sync() const {}
  double uni(double v) const { return v; }
  double rsum(double v) const { return v; }
  double wave_sum(double v) const { return v; }
  double rmax(double v) const { return v; }
  double rmin(double v) const { return v; }
  template <int... OPS> void reduce_ops(double (&)[sizeof...(OPS)]) const {}
  int lane() const { return 0; }
  int nlanes() const { return 1; }
  int wave() const { return 0; }
  int nwaves() const { return 1; }
  void wave_sync() const {}
};
#else
// kWaveOnly: the kernel instance for templates whose panels all fit one wave (Dims::wave_ok): the blocked LDS
// routines are not compiled into it
// kGeneral: the instance for templates with Dims::general set (terms with four factors, cos / sin atoms, basis rows of
// degree > 5): the other instance does not carry that code -- at the 256-register cap of this kernel it cost spills
// kPrep: the instance of the setup kernel (ipm_prepare_kernel): row arrays and Jacobian values in the agent's record in global
// memory, no KKT store
// kRefine: the instance that carries the refinement of regularised steps (Opts::refine): its second solve and its second pass through
// the step phase cost the instance without them 2.5 % of its cycles per solve through register pressure alone (345.8 k against 337 k)
template <bool kHbm, bool kWaveOnly = false, bool kRootLds = false, bool kGeneral = false, bool kPrep = false, bool kRefine = false>
struct CtxT {
  static constexpr bool hbm = kHbm;
  static constexpr bool prep = kPrep;
  static constexpr bool refine = kRefine;
  static constexpr bool general = kGeneral;
  static constexpr bool root_lds = kRootLds;      // Work::root holds the root block from the Schur step on (spill modes)
  static constexpr bool wave_only = kWaveOnly;
  // the wave-level routines address the KKT store as LDS: the spill-mode instances do not carry them
  static constexpr bool no_wave = kHbm;
  double* red;
  long long* prof;
  // (opaque to the optimiser: otherwise every per-thread table address `base + tid * size` of every phase is
  // computed once at kernel entry, kept alive across the whole solve and -- at the 256-register cap -- spilled to
  // scratch memory and reloaded from there in every iteration)
  __device__ int tid() const { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
  __device__ int nthr() const { return blockDim.x; }
  __device__ void sync() const { __syncthreads(); }
  // N values at once (op: 0 sum, 1 max, 2 min) for the price of one: two barriers in total.
  // Ops are template arguments so that everything unrolls into registers (no scratch).
  template <int OP> static __device__ __forceinline__ double comb(double a, double b) {
    return OP == 0 ? a + b : (OP == 1 ? fmax(a, b) : fmin(a, b));
  }
  // Reduction over the 64 lanes of a wave on the cross-lane data path (DPP), not through the LDS crossbar:
  // four rotations inside each row of 16 lanes (row_ror 8, 4, 2, 1: afterwards every lane holds the result of its
  // row), then the four rows are read out by v_readlane and combined.  The __shfl_down ladder it replaces is six
  // dependent ds_bpermute pairs per value, ~360 cycles; an interior-point iteration reduces ~25 values.
  // All lanes must be active.  The result is the same in every lane; fixed order: deterministic.
  template <int OP, int CTRL> static __device__ __forceinline__ double dpp_comb(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return comb<OP>(v, __hiloint2double(hi2, lo2));
  }
  template <int OP> static __device__ __forceinline__ double wave_reduce(double v) {
    v = dpp_comb<OP, 0x128>(v);      // row_ror:8
    v = dpp_comb<OP, 0x124>(v);      // row_ror:4
    v = dpp_comb<OP, 0x122>(v);      // row_ror:2
    v = dpp_comb<OP, 0x121>(v);      // row_ror:1
    const double r0 = rl(v, 0), r1 = rl(v, 16), r2 = rl(v, 32), r3 = rl(v, 48);
    return comb<OP>(comb<OP>(r0, r1), comb<OP>(r2, r3));
  }
  static __device__ __forceinline__ double rl(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
  }
  // a value that is the same in every lane, moved to scalar registers: the scalars of the iteration (mu, t, f, step lengths,
  // ...) come out of LDS reductions as per-lane copies, and ~30 of them live across the whole iteration -- as vector
  // registers they are what pushes the kernel over its 256
  static __device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
  }
  template <int OP> __device__ double reduce(double v) const {
    v = wave_reduce<OP>(v);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double r = red[0];
    for (int i = 1; i < nw; ++i) r = comb<OP>(r, red[i]);
    __syncthreads();
    return uni(r);
  }
  template <int I, int N, int OP0, int... OPS>
  __device__ __forceinline__ void red_wave(double (&v)[N]) const {
    v[I] = wave_reduce<OP0>(v[I]);
    if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * N + I] = v[I];
    if constexpr (sizeof...(OPS) > 0) red_wave<I + 1, N, OPS...>(v);
  }
  template <int I, int N, int OP0, int... OPS>
  __device__ __forceinline__ void red_block(double (&v)[N]) const {
    const int nw = (blockDim.x + 63) >> 6;
    double r = red[I];
    for (int k = 1; k < nw; ++k) r = comb<OP0>(r, red[k * N + I]);
    v[I] = uni(r);
    if constexpr (sizeof...(OPS) > 0) red_block<I + 1, N, OPS...>(v);
  }
  template <int... OPS> __device__ __forceinline__ void reduce_ops(double (&v)[sizeof...(OPS)]) const {
    static_assert(sizeof...(OPS) <= 8, "reduction scratch holds 8 values per wave");
    red_wave<0, sizeof...(OPS), OPS...>(v);
    __syncthreads();
    red_block<0, sizeof...(OPS), OPS...>(v);
    __syncthreads();
  }
  __device__ double rsum(double v) const { return reduce<0>(v); }
  __device__ double wave_sum(double v) const { return wave_reduce<0>(v); }      // all lanes of the wave must be active
  __device__ double rmax(double v) const { return reduce<1>(v); }
  __device__ double rmin(double v) const { return reduce<2>(v); }
  __device__ int lane() const { return threadIdx.x & 63; }
  __device__ int nlanes() const { return 64; }
  __device__ int wave() const { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }     // (provably wave-uniform: scalar branches)
  __device__ int nwaves() const { return blockDim.x >> 6; }
  // lanes of one wave run in lockstep; this only orders their memory traffic (LDS: in-order per
  // wave; spilled arrays in HBM: wait for the vector-memory counters as well)
  __device__ void wave_sync() const {
    if (kHbm) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
  }
};
typedef CtxT<false> Ctx;
#endif

#define OMGX_PFOR(i, n) for (int i = c.tid(); i < (n); i += c.nthr())
#define OMGX_PFOR_U4(i, n) _Pragma("unroll 4") for (int i = c.tid(); i < (n); i += c.nthr())

// optional per-phase cycle counters (profiling build only, -DOMGX_PROFILE)
enum { PH_JAC = 0, PH_RESID, PH_ASSEMBLE, PH_FACTOR, PH_SOLVE, PH_STEP, PH_LINESEARCH, PH_UPDATE, PH_F_LEAF, PH_F_SCHUR, PH_F_ROOT, PH_LA, PH_LS, PH_LB, PH_SETUP, PH_TOTAL,
       PH_S_DESC, PH_S_PARAMS, PH_S_JAC0, PH_S_CLASS, PH_S_INIT, PH_A_ZERO, PH_A_PAIRS, PH_A_REST, PH_A_DIAG,
       PH_K_FWD, PH_K_ROOTRHS, PH_K_ROOT, PH_K_LEAFRHS, PH_K_BWD, PH_L_TERMS, PH_L_ROWS,
       PH_F_PARK, PH_F_SWEEP, PH_F_SCALE, PH_A_TCOL, PH_A_HESS, PH_P_LOAD, PH_P_DIV, PH_P_BSPL, PH_P_SLOTS, PH_COUNT };
#if defined(OMGX_PROFILE) && !defined(OMGX_HOST_PORT)
#define OMGX_TIC() long long tic_ = (c.sync(), clock64())
#define OMGX_TOC(k) do { c.sync(); long long now_ = clock64(); if (c.tid() == 0 && c.prof) c.prof[k] += now_ - tic_; tic_ = now_; } while (0)
#else
#define OMGX_TIC() do {} while (0)
#define OMGX_TOC(k) do {} while (0)
#endif

// ---------------------------------------------------------------------------
// parameter stage: atoms (Cox-de Boor at t/T, quotients) and coefficient slots
// reference twin: symbolic evalspline `basics/spline_extra.py:28-55`, with the
// span convention of `basics/spline.py:131-136`
// ---------------------------------------------------------------------------
OMGX_FN double pp_eval(const Tables& T, int pp, const double* a) {
  double tot = 0.0;
  for (int m = T.pp_ptr[pp]; m < T.pp_ptr[pp + 1]; ++m) {
    double v = T.pm_coef[m];
    for (int q = T.pm_ptr[m]; q < T.pm_ptr[m + 1]; ++q) v *= a[T.pm_atom[q]];
    tot += v;
  }
  return tot;
}

// monomials [m0, m1) of the CSR tables
OMGX_FN double csr_range_eval(const Tables& T, int m0, int m1, const double* a) {
  double tot = 0.0;
  for (int m = m0; m < m1; ++m) {
    double v = T.pm_coef[m];
    for (int q = T.pm_ptr[m]; q < T.pm_ptr[m + 1]; ++q) v *= a[T.pm_atom[q]];
    tot += v;
  }
  return tot;
}

// (range form: the caller already knows the monomial range)
OMGX_FN double mono_range_eval(const Tables& T, int m0, int m1, const double* a) {
  double tot = 0.0;
#pragma unroll 4
  for (int m = m0; m < m1; ++m) {
    const MonoRec r = T.pm_rec[m];
    double v = r.coef;
    if (r.a0 >= 0) { v *= a[r.a0]; if (r.a1 >= 0) { v *= a[r.a1]; if (r.a2 >= 0) { v *= a[r.a2]; if (r.a3 >= 0) v *= a[r.a3]; } } }
    tot += v;
  }
  return tot;
}

// the same sum, same order of operations, from the packed records: the record loads do not depend on
// each other (the CSR form chains three global loads per monomial)
OMGX_FN double pp_eval_packed(const Tables& T, int pp, const double* a) {
  double tot = 0.0;
  const int m0 = T.pp_ptr[pp], m1 = T.pp_ptr[pp + 1];
#pragma unroll 4
  for (int m = m0; m < m1; ++m) {
    const MonoRec r = T.pm_rec[m];
    double v = r.coef;
    if (r.a0 >= 0) { v *= a[r.a0]; if (r.a1 >= 0) { v *= a[r.a1]; if (r.a2 >= 0) { v *= a[r.a2]; if (r.a3 >= 0) v *= a[r.a3]; } } }
    tot += v;
  }
  return tot;
}

// All basis functions of one basis row at u by the span algorithm (The NURBS Book A2.2, O(deg^2) with
// deg (deg + 1) / 2 divisions) instead of one Cox-de Boor triangle per function: the deg + 1 functions of
// the active span are computed together, the others are zero.  Span convention of the reference
// (`basics/spline.py:131-136`): k_j < u <= k_{j+1}, closed on the left at the first knot.  Fully unrolled
// with compile-time indices (no scratch).  out[0 .. n_fun) receives the row.
template <bool GEN>
OMGX_FN void bspl_row(const double* k, int n_knots, int deg, double u, double* out) {
  const int n_fun = n_knots - deg - 1;
  int j = deg;                                               // first non-degenerate span
  for (int q = deg + 1; q < n_fun; ++q) if (k[q] < u) j = q;
  const bool inside = (u >= k[0]) && (u <= k[n_knots - 1]);
  if (GEN && deg > 5) {
    // bases of products and integrals of splines (degree 10 in the tangent-half-angle models): the same recurrence with
    // the values kept in the output row itself (N[r] ends up at out[j - deg + r]), left / right recomputed from the knots
    for (int i = 0; i < n_fun; ++i) out[i] = 0.0;
    if (!inside) return;
    double* Nv = out + (j - deg);
    Nv[0] = 1.0;
    for (int r = 1; r <= deg; ++r) {
      double saved = 0.0;
      for (int q = 0; q < r; ++q) {
        const double rgt = k[j + q + 1] - u, lft = u - k[j + 1 - (r - q)];
        const double den = rgt + lft;
        const double temp = den != 0.0 ? Nv[q] / den : 0.0;
        Nv[q] = saved + rgt * temp;
        saved = lft * temp;
      }
      Nv[r] = saved;
    }
    return;
  }
  double N[6], left[6], right[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) { N[r] = 0.0; left[r] = 0.0; right[r] = 0.0; }
  N[0] = 1.0;
#pragma unroll
  for (int r = 1; r <= 5; ++r) {
    if (r <= deg) {
      left[r] = u - k[j + 1 - r];
      right[r] = k[j + r] - u;
      double saved = 0.0;
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        if (q < r) {
          const double den = right[q + 1] + left[r - q];
          const double temp = den != 0.0 ? N[q] / den : 0.0;
          N[q] = saved + right[q + 1] * temp;
          saved = left[r - q] * temp;
        }
      }
      N[r] = saved;
    }
  }
  for (int i = 0; i < n_fun; ++i) out[i] = 0.0;
#pragma unroll
  for (int r = 0; r < 6; ++r) if (r <= deg && inside) out[j - deg + r] = N[r];
}

// value of one packed monomial (branch-free: an unused atom reads atom 0 and multiplies by one)
OMGX_FN double mono_value(const MonoRec& r, const double* at) {
  const double a0 = at[r.a0 < 0 ? 0 : r.a0], a1 = at[r.a1 < 0 ? 0 : r.a1];
  const double a2 = at[r.a2 < 0 ? 0 : r.a2], a3 = at[r.a3 < 0 ? 0 : r.a3];
  return r.coef * (r.a0 < 0 ? 1.0 : a0) * (r.a1 < 0 ? 1.0 : a1) * (r.a2 < 0 ? 1.0 : a2) * (r.a3 < 0 ? 1.0 : a3);
}
OMGX_FN double mono_value(const MonoRec8& r, const double* at) {
  const double a0 = at[r.a0 < 0 ? 0 : r.a0], a1 = at[r.a1 < 0 ? 0 : r.a1];
  const double a2 = at[r.a2 < 0 ? 0 : r.a2], a3 = at[r.a3 < 0 ? 0 : r.a3];
  const double a4 = at[r.a4 < 0 ? 0 : r.a4], a5 = at[r.a5 < 0 ? 0 : r.a5];
  const double a6 = at[r.a6 < 0 ? 0 : r.a6], a7 = at[r.a7 < 0 ? 0 : r.a7];
  return r.coef * (r.a0 < 0 ? 1.0 : a0) * (r.a1 < 0 ? 1.0 : a1) * (r.a2 < 0 ? 1.0 : a2) * (r.a3 < 0 ? 1.0 : a3)
                * (r.a4 < 0 ? 1.0 : a4) * (r.a5 < 0 ? 1.0 : a5) * (r.a6 < 0 ? 1.0 : a6) * (r.a7 < 0 ? 1.0 : a7);
}

// Slot values from the packed tables: the first OMGX_SLOT_CAP monomials of every slot from the ELL table, one thread
// per slot, four records at a time (all loads in flight), summed in table order; the tail of a long slot by one
// wave: lanes stride over the monomials, fixed-order wave sum (a single thread walking the 314 monomials of the
// constant term of the formation objective was 302 k of the 379 k cycles of a converged ADMM x-update).
template <class Rec, class C>
OMGX_FN void slots_packed(const C& c, const Dims& d, const Tables& T, Work& w) {
  const Rec* ell = (const Rec*)T.sl_ell;
  const Rec* rec = (const Rec*)T.pm_rec;
  OMGX_PFOR(i, d.n_slots) {
    const int L = T.sl_glen[i >> 6];
    double tot = 0.0;
    for (int s0 = 0; s0 < L; s0 += 4) {
      Rec q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = ell[(s0 + k) * d.n_slots + i];
      double v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = mono_value(q[k], w.atoms);
#pragma unroll
      for (int k = 0; k < 4; ++k) tot += v[k];
    }
    w.slots[T.sl_list[i]] = tot;
  }
  if (d.n_long > 0) {
    c.sync();
    for (int q = c.wave(); q < d.n_long; q += c.nwaves()) {
      const int sl = T.sl_list[q];
      const int m0 = T.slot_rng[2 * sl] + OMGX_SLOT_CAP, m1 = T.slot_rng[2 * sl + 1];
      double part = 0.0;
      for (int m = m0 + c.lane(); m < m1; m += c.nlanes()) part += mono_value(rec[m], w.atoms);
      const double rest = c.wave_sum(part);
      if (c.lane() == 0) w.slots[sl] += rest;
    }
  }
}

template <class C>
OMGX_FN void eval_params(const C& c, const Dims& d, const Tables& T, Work& w, const double* p) {
  OMGX_PFOR(i, d.n_par) w.atoms[i] = p[i];
  OMGX_TIC();
  OMGX_PFOR(i, d.n_knots) w.knots[i] = T.knots[i];     // the Cox-de Boor triangles read them dozens of times
  c.sync();
  OMGX_TOC(PH_P_LOAD);
  for (int k = 0; k < d.n_prog;) {              // ops in order (an op may read atoms of earlier ones)
    const int32_t* op = T.prog + 6 * k;
    if (op[0] != OP_BSPL) {
      // quotient of two parameter polynomials, or cos / sin of one (orientation of a rotating obstacle)
      if (c.tid() == 0) {
        const double num = d.mono_packed == 1 ? pp_eval_packed(T, op[1], w.atoms) : pp_eval(T, op[1], w.atoms);
        if (op[0] == OP_DIV) w.atoms[op[3]] = num / (d.mono_packed == 1 ? pp_eval_packed(T, op[2], w.atoms) : pp_eval(T, op[2], w.atoms));
        else if constexpr (C::general) w.atoms[op[3]] = op[0] == OP_COS ? cos(num) : sin(num);
      }
      ++k;
      c.sync();
      OMGX_TOC(PH_P_DIV);
      continue;
    } else {
      // a run of consecutive basis-row ops is one parallel pass: their arguments are raw parameters
      // or quotients (never the output of another basis row), so they do not depend on each other
      int k2 = k;
      while (k2 < d.n_prog && T.prog[6 * k2] == OP_BSPL) ++k2;
      // one thread per basis row: the deg + 1 functions of the active span together (bspl_row)
      OMGX_PFOR(it, k2 - k) {
        const int32_t* oq = T.prog + 6 * (k + it);
        bspl_row<C::general>(w.knots + oq[1], oq[2], oq[3], w.atoms[oq[4]], w.atoms + oq[5]);
      }
      k = k2;
    }
    c.sync();
    OMGX_TOC(PH_P_BSPL);
  }
  if (d.mono_packed == 1) slots_packed<MonoRec>(c, d, T, w);
  else if (d.mono_packed == 2) slots_packed<MonoRec8>(c, d, T, w);
  else {
    // (monomials with more than eight atoms: the CSR tables, three chained loads per monomial -- the same split:
    // a thread sums the first OMGX_SLOT_CAP monomials of its slot, a wave the rest of a long one)
    OMGX_PFOR(s, d.n_slots) {
      const int m0 = T.slot_rng[2 * s], m1 = T.slot_rng[2 * s + 1];
      w.slots[s] = csr_range_eval(T, m0, m1 < m0 + OMGX_SLOT_CAP ? m1 : m0 + OMGX_SLOT_CAP, w.atoms);
    }
    if (d.n_long > 0) {
      c.sync();
      for (int q = c.wave(); q < d.n_long; q += c.nwaves()) {
        const int sl = T.sl_list[q];
        const int m0 = T.slot_rng[2 * sl] + OMGX_SLOT_CAP, m1 = T.slot_rng[2 * sl + 1];
        double part = 0.0;
        for (int m = m0 + c.lane(); m < m1; m += c.nlanes()) part += csr_range_eval(T, m, m + 1, w.atoms);
        const double rest = c.wave_sum(part);
        if (c.lane() == 0) w.slots[sl] += rest;
      }
    }
  }
  c.sync();
  OMGX_TOC(PH_P_SLOTS);
}

OMGX_FN double term_coef(const Tables& T, const Work& w, int t) {
  const int sl = T.t_slot[t];
  return sl < 0 ? T.t_coef[t] : T.t_coef[t] * w.slots[sl];
}

OMGX_FN double rec_coef(const Work& w, double coef, int slot) { return slot < 0 ? coef : coef * w.slots[slot]; }

// slack of inequality row r at phase-I variable t: every iterate keeps s = t v - h exactly (the slack reset of the
// accept step), so it is recomputed where it is needed instead of being stored
OMGX_FN double row_slack(const Work& w, int r, double t) { return t * w.vv[r] - w.hv[r]; }

// value of one Jacobian item at xv
template <bool Q4>
OMGX_FN double jitem_value(const JItem& q, const Work& w, const double* xv) {
  const double xs = w.slots[q.slot < 0 ? 0 : q.slot], xa = xv[q.va < 0 ? 0 : q.va], xb = xv[q.vb < 0 ? 0 : q.vb];
  double v = q.coef * (q.slot < 0 ? 1.0 : xs) * (q.va < 0 ? 1.0 : xa) * (q.vb < 0 ? 1.0 : xb);
  if (Q4) { const double xc = xv[q.vc < 0 ? 0 : q.vc]; v *= (q.vc < 0 ? 1.0 : xc); }
  return v;
}
// Jacobian entries from an item table, OMGX_JE_OWN = 4 entries per owner: item `step` of the owner's k-th entry at index
// (step * 4 + k) * n_own + o, {entry, row} of that entry at own[2 * (k * n_own + o)].  Most entries are one item (config 2:
// 1784 of 2636; 672 have two, 180 three): with one entry per thread the passes were a chain of single loads -- 10 passes of
// the workgroup over all entries, 5 over the x-dependent ones; here a thread has the items of four entries in flight, a
// group of 64 owners whose entries are all single items loads just those.  fn(entry, row, value); items summed in step
// order.  glen[o >> 6]: items per entry the group walks (1, or a multiple of 2).
#define OMGX_JE_OWN 4
template <class C, class F>
OMGX_FN void jac_entries4(const C& c, const JItem* ell, const int32_t* own, const int32_t* glen, int n_own, const Work& w, const double* xv, F fn) {
  OMGX_PFOR(o, n_own) {
    const int L = glen[o >> 6];
    int32_t er[OMGX_JE_OWN][2];
#pragma unroll
    for (int k = 0; k < OMGX_JE_OWN; ++k) { const int32_t* q = own + 2 * (k * n_own + o); er[k][0] = q[0]; er[k][1] = q[1]; }
    double sj[OMGX_JE_OWN];
    if (L == 1) {
      JItem q[OMGX_JE_OWN];
#pragma unroll
      for (int k = 0; k < OMGX_JE_OWN; ++k) q[k] = ell[k * n_own + o];
#pragma unroll
      for (int k = 0; k < OMGX_JE_OWN; ++k) sj[k] = jitem_value<C::general>(q[k], w, xv);
    } else {
#pragma unroll
      for (int k = 0; k < OMGX_JE_OWN; ++k) sj[k] = 0.0;
      for (int s0 = 0; s0 < L; s0 += 2) {
        JItem q[2 * OMGX_JE_OWN];
#pragma unroll
        for (int k = 0; k < 2 * OMGX_JE_OWN; ++k) q[k] = ell[(s0 * OMGX_JE_OWN + k) * n_own + o];
        double v[2 * OMGX_JE_OWN];
#pragma unroll
        for (int k = 0; k < 2 * OMGX_JE_OWN; ++k) v[k] = jitem_value<C::general>(q[k], w, xv);
#pragma unroll
        for (int k = 0; k < OMGX_JE_OWN; ++k) { sj[k] += v[k]; sj[k] += v[OMGX_JE_OWN + k]; }
      }
    }
#pragma unroll
    for (int k = 0; k < OMGX_JE_OWN; ++k) if (er[k][0] >= 0) fn(er[k][0], er[k][1], sj[k]);
  }
}

// unscaled value of the row in slot i (row T.row_perm[i]) at xv: its terms from the ELL table, eight at a
// time (all loads in flight together), summed in term order
#define OMGX_ROW_BATCH 8
template <bool Q4>
OMGX_FN double row_value_ell_t(const Tables& T, const Work& w, int i, int m, const double* xv) {
  const int L = T.rt_glen[i >> 6];
  double g = 0.0;
  for (int s0 = 0; s0 < L; s0 += OMGX_ROW_BATCH) {
    RowTerm q[OMGX_ROW_BATCH];
#pragma unroll
    for (int k = 0; k < OMGX_ROW_BATCH; ++k) q[k] = T.rt_ell[(s0 + k) * m + i];
    double v[OMGX_ROW_BATCH];
#pragma unroll
    for (int k = 0; k < OMGX_ROW_BATCH; ++k) {
      const double xs = w.slots[q[k].slot < 0 ? 0 : q[k].slot];
      const double x0 = xv[q[k].v0 < 0 ? 0 : q[k].v0], x1 = xv[q[k].v1 < 0 ? 0 : q[k].v1], x2 = xv[q[k].v2 < 0 ? 0 : q[k].v2];
      v[k] = q[k].coef * (q[k].slot < 0 ? 1.0 : xs) * (q[k].v0 < 0 ? 1.0 : x0) * (q[k].v1 < 0 ? 1.0 : x1) * (q[k].v2 < 0 ? 1.0 : x2);
      if (Q4) { const double x3 = xv[q[k].v3 < 0 ? 0 : q[k].v3]; v[k] *= (q[k].v3 < 0 ? 1.0 : x3); }
    }
#pragma unroll
    for (int k = 0; k < OMGX_ROW_BATCH; ++k) g += v[k];
  }
  return g;
}
template <class C>
OMGX_FN double row_value_ell(const Dims& d, const Tables& T, const Work& w, int i, int m, const double* xv) {
  return row_value_ell_t<C::general>(T, w, i, m, xv);
}

// Lifted auxiliaries (omgx_template::n_lift: products of more than four variable factors, quotients by a variable) follow the
// caller's variables: at xv every auxiliary takes the value its defining row gives it -- the row is linear in it: its values
// with the auxiliary at 0 and at 1 --, level by level (a row reads auxiliaries of lower levels only; its owner is the only
// thread that touches its auxiliary).  Applied to every trial point of the line search: the defining rows then hold exactly at
// every iterate, the merit function is the one of the caller's own problem and no step is cut for what a product moves beyond
// its linearisation (with the rows left to the Newton iteration the l1 penalty on 200 bilinear rows rejected every step
// longer than 2e-3 on the Bicycle class of `vehicles/bicycle.py:53`).  Ends with a barrier when there is anything to do.
template <class C>
OMGX_FN void lift_project(const C& c, const Dims& d, const Tables& T, const Work& w, int m, double* xv) {
  if (d.n_lift == 0) return;
  for (int L = 0; L < d.lift_depth; ++L) {
    const int k0 = T.lift_lev[L], k1 = T.lift_lev[L + 1];
    OMGX_PFOR(kk, k1 - k0) {
      const int32_t* q = T.lift_rec + 2 * (k0 + kk);
      xv[q[1]] = 0.0;
      const double g0 = row_value_ell<C>(d, T, w, q[0], m, xv);
      xv[q[1]] = 1.0;
      const double g1 = row_value_ell<C>(d, T, w, q[0], m, xv);
      // (a quotient row q den - num whose den is zero at this point has no value for its auxiliary: NaN -- the line search rejects
      // such a trial point like any other that leaves the domain, the start of a solve is refused, ipm_solve)
      const double den = g1 - g0;
      xv[q[1]] = den != 0.0 ? -g0 / den : NAN;
    }
    c.sync();
  }
}

// this thread's share of row r (terms strided over the workgroup); the caller sums the shares
template <class C>
OMGX_FN double row_value_share(const C& c, const Tables& T, const Work& w, int r, const double* xv) {
  double g = 0.0;
  for (int t = T.row_ptr[r] + c.tid(); t < T.row_ptr[r + 1]; t += c.nthr()) {
    double v = term_coef(T, w, t);
    const int32_t* tv = T.t_var + OMGX_TV * t;
    for (int k = 0; k < OMGX_TV && tv[k] >= 0; ++k) v *= xv[tv[k]];
    g += v;
  }
  return g;
}

// ---------------------------------------------------------------------------
// block-arrow KKT store
// ---------------------------------------------------------------------------
OMGX_FN int tri(int i, int k) { return i * (i + 1) / 2 + k; }   // packed lower, row-major, i >= k

struct Kkt {
  // (copies of the few table pointers / dimensions it needs, not pointers to the structs: a struct whose address
  // escapes into a helper object cannot be kept in scalar registers -- the compiler copied all of Dims and
  // Tables into every lane's scratch memory at kernel entry)
  const int32_t *d_off, *b_off, *leaf_off, *leaf_bw, *cpl_ptr, *cpl_idx, *blk, *cpl_map, *lf_w, *lf_ldb, *lf_band, *lf_kind, *dl_pos;
  int n_leaf, n_root, root_off;
  double* a;
  OMGX_FN void bind(const Dims& dd, const Tables& TT, double* store) {
    d_off = TT.d_off; b_off = TT.b_off; leaf_off = TT.leaf_off; leaf_bw = TT.leaf_bw; cpl_ptr = TT.cpl_ptr;
    cpl_idx = TT.cpl_idx; blk = TT.blk; cpl_map = TT.cpl_map;
    lf_w = TT.lf_w; lf_ldb = TT.lf_ldb; lf_band = TT.lf_band; lf_kind = TT.lf_kind; dl_pos = TT.dl_pos;
    n_leaf = dd.n_leaf; n_root = dd.n_root; root_off = dd.root_off; a = store;
  }
  // leaf l: panel of (n_l + nc_l + 1) rows with odd leading dimension ld_l; rows [0,n_l) hold the
  // leaf block D_l (lower part), rows [n_l, n_l+nc_l) the coupling rows B_l (one per coupled
  // root position), row n_l+nc_l the right-hand side of the leaf.  Root block R: packed lower,
  // row-major, nr rows + one more for its right-hand side.  The right-hand sides are carried through
  // the factorisation like coupling rows, so the forward substitutions L^{-1} r come out of it.
  OMGX_FN double* P(int l) const { return a + d_off[l]; }
  OMGX_FN int ld(int l) const { return b_off[l]; }
  OMGX_FN double* R() const { return a + d_off[n_leaf]; }
  OMGX_FN int nl(int l) const { return leaf_off[l + 1] - leaf_off[l]; }
  OMGX_FN int nc(int l) const { return cpl_ptr[l + 1] - cpl_ptr[l]; }
};

// ---------------------------------------------------------------------------
// Blocked LDL' (panel width 4 = the K of v_mfma_f64_16x16x4_f64).
//
// A "matrix" here is a set of rows: the first nfact rows/columns form the
// symmetric block to factorise (lower part stored), the remaining rows are
// carried along (leaf panels carry their coupling rows B_l, which end up as
// Wt = B L^{-T}).  Storage is either row-major with leading dimension ld or
// packed lower (ld == 0).  Per block of 4 columns:
//   phase A  every row solves its 4 panel entries against the 4x4 diagonal
//            block (factorised redundantly by each thread: no extra barrier)
//            and publishes U = a L_jj^{-T} and Lp = U Delta^{-1} in a panel buffer;
//   phase B  trailing update  A[r][k] -= sum_q U[r][q] Lp[k][q]: one MFMA per
//            16x16 tile on the device.
// Two workgroup barriers per 4 columns instead of two per column.
// ---------------------------------------------------------------------------
// offsets (not pointers) so that every access stays a provable LDS access (ds_* instead of flat_*)
struct BMat { int a, ld, nfact, rows, npos, dinv, pan, cpl, bw, pad_; };   // cpl: offset of the leaf's coupling index list (cpl_ptr[l])   // a: offset in kkt; dinv: offset in w.dinv (-1: none); pan: offset in w.col
static_assert(sizeof(BMat) <= OMGX_BMAT_DOUBLES * sizeof(double), "BMat larger than its LDS slot");
#define OMGX_NB 4
#define OMGX_PAN_SMALL(n) (4 * (n) + 16)      // per leaf in the spill modes: ldl_left4 keeps n inverse pivots + 10 doubles per 4 columns there

// (branch-free on purpose: with `ld ? row-major : packed` as a conditional the compiler sinks the
// LDS load that uses the address into the two branches, and a sequence of such loads -- the ten
// entries of a diagonal block -- becomes ten serialised round trips)
// Leaf panels are stored by rows in the LDS modes (odd leading dimension: one thread per row is conflict-free) and
// by columns in the spill modes (one thread per row then reads consecutive addresses of the HBM slab; the plan
// writes its precomputed addresses the same way, omgx_plan.h `col_major`): entry (r, k) = a + r * sr + k * sc.
#define OMGX_PANEL_STRIDES(C, M, sr, sc) const int sr = C::hbm ? 1 : (M).ld, sc = C::hbm ? (M).ld : 1
OMGX_FN int baddr(const BMat& M, int r, int k) { return M.a + r * M.ld + (M.ld == 0 ? 1 : 0) * ((r * (r + 1)) >> 1) + k; }
// storage kind known at compile time (one multiply instead of two and a select per address):
// KIND 0 generic, 1 row-major (leaf panels), 2 packed lower (root)
template <int KIND>
OMGX_FN int baddr_k(const BMat& M, int r, int k) {
  return KIND == 1 ? M.a + r * M.ld + k : (KIND == 2 ? M.a + ((r * (r + 1)) >> 1) + k : baddr(M, r, k));
}

// reciprocal of a pivot: v_rcp_f64 (about 2^-29 relative) + two Newton steps instead of the
// ~15-instruction IEEE division sequence; exact division on the host port
#ifdef OMGX_HOST_PORT
OMGX_FN double rcp_pivot(double d) { return 1.0 / d; }
#else
OMGX_FN double rcp_pivot(double d) {
  double y = __builtin_amdgcn_rcp(d);
  y = fma(fma(-d, y, 1.0), y, y);
  y = fma(fma(-d, y, 1.0), y, y);
  return y;
}
#endif

}  // namespace omgx
#ifndef OMGX_HOST_PORT
#include "omgx_wave.h"      // register-resident wave-level LDL' (device only)
#endif
namespace omgx {

// 4x4 (or smaller) diagonal block LDL' from the stored lower entries
struct Blk4 { double l10, l20, l21, l30, l31, l32, d0, d1, d2, d3, i0, i1, i2, i3; };
OMGX_FN Blk4 blk4_from(double g00, double g10, double g11, double g20, double g21, double g22,
                       double g30, double g31, double g32, double g33) {
  Blk4 b;
  b.d0 = g00; b.i0 = rcp_pivot(b.d0);
  b.l10 = g10 * b.i0; b.l20 = g20 * b.i0; b.l30 = g30 * b.i0;
  b.d1 = g11 - b.l10 * g10; b.i1 = rcp_pivot(b.d1);
  const double w21 = g21 - g20 * b.l10, w31 = g31 - g30 * b.l10;
  b.l21 = w21 * b.i1; b.l31 = w31 * b.i1;
  b.d2 = g22 - b.l20 * g20 - b.l21 * w21; b.i2 = rcp_pivot(b.d2);
  const double w32 = g32 - g30 * b.l20 - w31 * b.l21;
  b.l32 = w32 * b.i2;
  b.d3 = g33 - b.l30 * g30 - b.l31 * w31 - b.l32 * w32; b.i3 = rcp_pivot(b.d3);
  return b;
}

template <int KIND>
OMGX_FN Blk4 blk4_factor(const BMat& M, const double* A, int jb, int nb) {
  // all loads first (independent), then the short dependent chain
  // (a partial last block reads rows past the block: still inside the workspace, then masked;
  // unconditional loads issue back to back instead of one LDS round trip per branch)
  double g[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int k = 0; k <= a; ++k) {
      const double v = A[baddr_k<KIND>(M, jb + (a < nb ? a : 0), jb + (a < nb ? k : 0))];
      g[a][k] = (a < nb) ? v : (a == k ? 1.0 : 0.0);
    }
  return blk4_from(g[0][0], g[1][0], g[1][1], g[2][0], g[2][1], g[2][2], g[3][0], g[3][1], g[3][2], g[3][3]);
}

// ---------------------------------------------------------------------------
// Left-looking blocked LDL' for small matrices (order <~ 64), one thread per row, ONE barrier per
// block of 4 columns.  Every thread forms its own 4 entries of the block column AND (redundantly,
// it reads those rows anyway) the 4x4 diagonal block,
//     v_rc = a_rc - sum_{k<jb} U_rk U_ck / d_k ,
// factorises the block in registers and solves its row against it.  During the sweep all rows
// hold U = L D (for the carried rows that is the final Wt = B L^{-T}); the original diagonal blocks
// are parked in a side buffer first, because their matrix slots receive U while other threads
// still need the originals.  Output: factorised rows hold U = L D (pivots on the diagonal, inverse
// pivots in the side array), carried rows U = a L^{-T} as in ldl_blocked.  Latency: 2-3 dependent LDS round trips and one barrier per block instead of ~6 and
// two; the dot products are independent loads the compiler can keep in flight.
// ---------------------------------------------------------------------------
template <class C>
OMGX_FN void ldl_left4(const C& c, const BMat* Ms, int nm, double* A, double* dinvb, double* colb, int* bad) {
  int nmax = 0, total_rows = 0, total_blocks = 0;
  for (int i = 0; i < nm; ++i) {
    if (Ms[i].nfact > nmax) nmax = Ms[i].nfact;
    total_rows += Ms[i].rows; total_blocks += (Ms[i].nfact + 3) >> 2;
  }
  OMGX_TIC();
  // park the original diagonal blocks: 10 doubles per block behind the inverse pivots of the matrix
  OMGX_PFOR(it, total_blocks * 10) {
    int mi = 0, blk = it / 10; const int e = it - 10 * blk;
    while (blk >= ((Ms[mi].nfact + 3) >> 2)) { blk -= (Ms[mi].nfact + 3) >> 2; ++mi; }
    const BMat M = Ms[mi];
    const int a = e < 1 ? 0 : (e < 3 ? 1 : (e < 6 ? 2 : 3)), k = e - ((a * (a + 1)) >> 1);
    const int ra = 4 * blk + a, ck = 4 * blk + k;
    OMGX_PANEL_STRIDES(C, M, sr, sc);
    colb[M.pan + M.nfact + 10 * blk + e] = (ra < M.nfact) ? A[M.a + ra * sr + ck * sc] : (a == k ? 1.0 : 0.0);
  }
  c.sync();
  OMGX_TOC(PH_F_PARK);
  int badl = 0;
  for (int jb = 0; jb < nmax; jb += OMGX_NB) {
    OMGX_PFOR(it, total_rows) {
      int mi = 0, r = it;
      while (r >= Ms[mi].rows) { r -= Ms[mi].rows; ++mi; }
      const BMat M = Ms[mi];
      if (jb >= M.nfact || r < jb) continue;
      const int nb = (M.nfact - jb) < OMGX_NB ? (M.nfact - jb) : OMGX_NB;
      double* iv = (M.dinv >= 0) ? dinvb + M.dinv : colb + M.pan;       // inverse pivots of this matrix
      const double* db = colb + M.pan + M.nfact + 10 * (jb >> 2);
      const bool diag_row = r < jb + nb;
      // block rows (clamped for a partial last block: their products are masked below)
      const int q1 = nb > 1 ? 1 : 0, q2 = nb > 2 ? 2 : 0, q3 = nb > 3 ? 3 : 0;
      OMGX_PANEL_STRIDES(C, M, sr, sc);
      const int b0 = M.a + jb * sr, b1 = M.a + (jb + q1) * sr, b2 = M.a + (jb + q2) * sr, b3 = M.a + (jb + q3) * sr;
      const int br = M.a + r * sr;
      double g00 = db[0], g10 = db[1], g11 = db[2], g20 = db[3], g21 = db[4], g22 = db[5],
             g30 = db[6], g31 = db[7], g32 = db[8], g33 = db[9];
      double v0 = A[br + jb * sc], v1 = A[br + (jb + q1) * sc], v2 = A[br + (jb + q2) * sc], v3 = A[br + (jb + q3) * sc];
      const double m1 = nb > 1 ? 1.0 : 0.0, m2 = nb > 2 ? 1.0 : 0.0, m3 = nb > 3 ? 1.0 : 0.0;
      // banded block (leaf in its Cuthill-McKee order, no fill outside the band): the block rows jb .. jb + 3 are
      // zero left of column jb - bw, those columns contribute exact zeros
      const int k0 = jb > M.bw ? jb - M.bw : 0;
#pragma unroll 4
      for (int k = k0; k < jb; ++k) {
        const double tk = iv[k];
        const double u0 = A[b0 + k * sc], u1 = m1 * A[b1 + k * sc], u2 = m2 * A[b2 + k * sc], u3 = m3 * A[b3 + k * sc], ur = A[br + k * sc];
        const double t0 = u0 * tk, t1 = u1 * tk, t2 = u2 * tk, t3 = u3 * tk;
        g00 -= u0 * t0;
        g10 -= u1 * t0; g11 -= u1 * t1;
        g20 -= u2 * t0; g21 -= u2 * t1; g22 -= u2 * t2;
        g30 -= u3 * t0; g31 -= u3 * t1; g32 -= u3 * t2; g33 -= u3 * t3;
        v0 -= ur * t0; v1 -= ur * t1; v2 -= ur * t2; v3 -= ur * t3;
      }
      const Blk4 B = blk4_from(g00, g10, g11, g20, g21, g22, g30, g31, g32, g33);
      if (diag_row) {
        const int q = r - jb;
        const double dq = q == 0 ? B.d0 : (q == 1 ? B.d1 : (q == 2 ? B.d2 : B.d3));
        const double iq = q == 0 ? B.i0 : (q == 1 ? B.i1 : (q == 2 ? B.i2 : B.i3));
        const bool pos_ok = (jb + q < M.npos) ? (dq > 0.0) : (dq < 0.0);
        if (!pos_ok) badl = 1;
        iv[jb + q] = iq;
        // U = L D inside the block, the pivot itself on the diagonal
        if (q == 1) { A[br + jb * sc] = B.l10 * B.d0; }
        else if (q == 2) { A[br + jb * sc] = B.l20 * B.d0; A[br + (jb + 1) * sc] = B.l21 * B.d1; }
        else if (q == 3) { A[br + jb * sc] = B.l30 * B.d0; A[br + (jb + 1) * sc] = B.l31 * B.d1; A[br + (jb + 2) * sc] = B.l32 * B.d2; }
        A[br + (jb + q) * sc] = dq;
      } else {
        const double u0 = v0;
        const double u1 = v1 - u0 * B.l10;
        const double u2 = v2 - u0 * B.l20 - u1 * B.l21;
        const double u3 = v3 - u0 * B.l30 - u1 * B.l31 - u2 * B.l32;
        A[br + jb * sc] = u0;
        if (nb > 1) A[br + (jb + 1) * sc] = u1;
        if (nb > 2) A[br + (jb + 2) * sc] = u2;
        if (nb > 3) A[br + (jb + 3) * sc] = u3;
      }
    }
    c.sync();
  }
  OMGX_TOC(PH_F_SWEEP);
  // (the factorised rows stay in the form U = L D: the only reader, the leaves' backward
  // substitution, applies the inverse pivots on the fly -- no scaling pass over the panels)
  *bad = c.rmax(badl ? 1.0 : 0.0) > 0.0 ? 1 : 0;
  OMGX_TOC(PH_F_SCALE);
}

#ifndef OMGX_HOST_PORT
// ---------------------------------------------------------------------------
// Cooperative form of ldl_left4 for the case that the workgroup has a spare wave per matrix (config 2:
// 4 leaves, 256 rows = 4 row waves + 4 block waves).  In ldl_left4 every row recomputes the 4x4
// diagonal block (14 of the 21 fp64 operations per row and finished column) because handing it over
// would cost a barrier -- but a barrier costs 15 cycles here, the redundant arithmetic thousands.  So:
//   block wave of matrix l:  S = U_blk diag(1/d) U_blk' over the finished columns on the matrix pipe
//                            (one v_mfma_f64_16x16x4 per 4 columns), G = A_blk - S, 4x4 LDL' of G,
//                            result (14 doubles) and the new inverse pivots to LDS;
//   row waves, meanwhile:    v_rq = a_rq - sum_k U_rk (U_{jb+q,k} / d_k), 8 operations per column;
//   barrier; every row reads the block factor and finishes its four entries; barrier.
// Same storage convention and (up to summation order) the same numbers as ldl_left4.
// ---------------------------------------------------------------------------
template <int KIND, class C>
OMGX_FN void ldl_left4_coop(const C& c, const BMat* Ms, int nm, double* A, double* dinvb, double* colb, int* bad,
                            int total_rows, int nmax) {
  const int lane = c.lane(), wave = c.wave();
  const int row_waves = (total_rows + 63) >> 6;
  const bool is_row = c.tid() < total_rows;
  const bool is_blk = wave >= row_waves && wave < row_waves + nm;
  int r = c.tid();
  BMat M = Ms[0];
  if (is_row) { int mi = 0; while (r >= Ms[mi].rows) { r -= Ms[mi].rows; ++mi; } M = Ms[mi]; }
  else if (is_blk) { M = Ms[wave - row_waves]; r = -1; }
  double* iv = (M.dinv >= 0) ? dinvb + M.dinv : colb + M.pan;          // inverse pivots of this matrix
  double* gb = colb + M.pan + M.nfact;                                // [16] G staging, [16..30) block factor
  double* bb = gb + 16;
  const int br = baddr_k<KIND>(M, r > 0 ? r : 0, 0);
  int badl = 0;
  for (int jb = 0; jb < nmax; jb += OMGX_NB) {
    const bool live = (is_row || is_blk) && jb < M.nfact;
    const int nb = (M.nfact - jb) < OMGX_NB ? (M.nfact - jb) : OMGX_NB;
    const int q1 = nb > 1 ? 1 : 0, q2 = nb > 2 ? 2 : 0, q3 = nb > 3 ? 3 : 0;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    if (live && is_blk) {
      // ---- block wave: G = A_blk - U_blk diag(1/d) U_blk' ---------------------------------
      const int a = lane & 15, kq = lane >> 4;
      const bool in_blk = a < nb;
      const int ra = baddr_k<KIND>(M, jb + (in_blk ? a : 0), 0);
      typedef double v4d __attribute__((ext_vector_type(4)));
      v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
      int k0 = 0;
      for (; k0 + 8 <= jb; k0 += 8) {
        const double u_a = A[ra + k0 + kq], t_a = iv[k0 + kq], u_b = A[ra + k0 + 4 + kq], t_b = iv[k0 + 4 + kq];
        const double ua = in_blk ? u_a : 0.0, ub = in_blk ? u_b : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ua * t_a, ua, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ub * t_b, ub, acc2, 0, 0, 0);
      }
      for (; k0 < jb; k0 += 4) {
        const double u_a = A[ra + k0 + kq], t_a = iv[k0 + kq];
        const double ua = in_blk ? u_a : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ua * t_a, ua, acc, 0, 0, 0);
      }
      acc += acc2;
      // acc[0] of lane (row = lane >> 4, col = lane & 15) is S[row][col], rows and columns 0..3
      {
        const int gr = lane >> 4, gc = lane & 15;
        if (gc <= gr) {                                           // lower part (gr < 4 always)
          const bool ok = gr < nb && gc < nb;
          const double orig = A[baddr_k<KIND>(M, jb + (ok ? gr : 0), jb + (ok ? gc : 0))];
          gb[gr * 4 + gc] = ok ? orig - acc[0] : (gr == gc ? 1.0 : 0.0);
        }
      }
      c.wave_sync();
      const Blk4 B = blk4_from(gb[0], gb[4], gb[5], gb[8], gb[9], gb[10], gb[12], gb[13], gb[14], gb[15]);
      if (lane == 0) {
        bb[0] = B.l10; bb[1] = B.l20; bb[2] = B.l21; bb[3] = B.l30; bb[4] = B.l31; bb[5] = B.l32;
        bb[6] = B.d0; bb[7] = B.d1; bb[8] = B.d2; bb[9] = B.d3;
      }
      if (lane < nb) {
        const double dq = lane == 0 ? B.d0 : (lane == 1 ? B.d1 : (lane == 2 ? B.d2 : B.d3));
        const double iq = lane == 0 ? B.i0 : (lane == 1 ? B.i1 : (lane == 2 ? B.i2 : B.i3));
        const bool pos_ok = (jb + lane < M.npos) ? (dq > 0.0) : (dq < 0.0);
        if (!pos_ok) badl = 1;
        bb[10 + lane] = iq;                                       // published to iv after the barrier
      }
    } else if (live && r >= jb + nb) {
      // ---- row waves: the four entries of this row against the finished columns ------------
      const int b0 = baddr_k<KIND>(M, jb, 0), b1 = baddr_k<KIND>(M, jb + q1, 0), b2 = baddr_k<KIND>(M, jb + q2, 0), b3 = baddr_k<KIND>(M, jb + q3, 0);
      v0 = A[br + jb]; v1 = A[br + jb + q1]; v2 = A[br + jb + q2]; v3 = A[br + jb + q3];
#pragma unroll 4
      for (int k = 0; k < jb; ++k) {
        const double tk = iv[k], ur = A[br + k];
        const double w = ur * tk;
        v0 -= w * A[b0 + k]; v1 -= w * A[b1 + k]; v2 -= w * A[b2 + k]; v3 -= w * A[b3 + k];
      }
    }
    c.sync();
    if (live && is_row && r >= jb) {
      const double l10 = bb[0], l20 = bb[1], l21 = bb[2], l30 = bb[3], l31 = bb[4], l32 = bb[5];
      const double d0 = bb[6], d1 = bb[7], d2 = bb[8], d3 = bb[9];
      if (r < jb + nb) {
        const int q = r - jb;
        iv[jb + q] = bb[10 + q];
        // U = L D inside the block, the pivot itself on the diagonal
        if (q == 1) { A[br + jb] = l10 * d0; }
        else if (q == 2) { A[br + jb] = l20 * d0; A[br + jb + 1] = l21 * d1; }
        else if (q == 3) { A[br + jb] = l30 * d0; A[br + jb + 1] = l31 * d1; A[br + jb + 2] = l32 * d2; }
        A[br + jb + q] = q == 0 ? d0 : (q == 1 ? d1 : (q == 2 ? d2 : d3));
      } else {
        const double u0 = v0;
        const double u1 = v1 - u0 * l10;
        const double u2 = v2 - u0 * l20 - u1 * l21;
        const double u3 = v3 - u0 * l30 - u1 * l31 - u2 * l32;
        A[br + jb] = u0;
        if (nb > 1) A[br + jb + 1] = u1;
        if (nb > 2) A[br + jb + 2] = u2;
        if (nb > 3) A[br + jb + 3] = u3;
      }
    }
    c.sync();
  }
  *bad = c.rmax(badl ? 1.0 : 0.0) > 0.0 ? 1 : 0;
}
#endif

// Factorise `nm` matrices together (same block index for all of them).
// Returns through *bad whether a pivot had the wrong sign.
template <int KIND, class C>
OMGX_FN void ldl_blocked(const C& c, const BMat* Ms, int nm, double* A, double* dinvb, double* colb, double* stage, int* bad) {
  int nmax = 0, total_rows = 0;
  for (int i = 0; i < nm; ++i) { if (Ms[i].nfact > nmax) nmax = Ms[i].nfact; total_rows += Ms[i].rows; }
  int badl = 0;
  // row -> (matrix, local row) of this thread's first row: static over the panel loop, so the
  // descriptor walk (dependent LDS reads) is done once, not once per panel
  int mi_own = 0, r_own = c.tid();
  BMat M_own = Ms[0];
  if (c.tid() < total_rows) {
    while (r_own >= Ms[mi_own].rows) { r_own -= Ms[mi_own].rows; ++mi_own; }
    M_own = Ms[mi_own];
  }
  OMGX_TIC();
  for (int jb = 0; jb < nmax; jb += OMGX_NB) {
    // ---- phase A: panel ---------------------------------------------------------------
    OMGX_PFOR(it, total_rows) {
      int mi = mi_own, r = r_own;
      BMat M = M_own;                              // by value: keep the descriptor in registers
      if (it != c.tid()) {                         // more rows than threads: later passes
        mi = 0; r = it;
        while (r >= Ms[mi].rows) { r -= Ms[mi].rows; ++mi; }
        M = Ms[mi];
      }
      if (jb >= M.nfact || r < jb) continue;
      const int nb = (M.nfact - jb) < OMGX_NB ? (M.nfact - jb) : OMGX_NB;
      const Blk4 B = blk4_factor<KIND>(M, A, jb, nb);
      if (r < jb + nb) {
        // a row of the diagonal block: final values go through the staging area
        // (other threads are still reading the original block).  No dynamically
        // indexed local arrays here: they would live in scratch memory.
        const int q = r - jb;
        double* st = stage + mi * OMGX_STAGE_LD + q * 4;
        const double dq = q == 0 ? B.d0 : (q == 1 ? B.d1 : (q == 2 ? B.d2 : B.d3));
        const double iq = q == 0 ? B.i0 : (q == 1 ? B.i1 : (q == 2 ? B.i2 : B.i3));
        if (q == 1) { st[0] = B.l10; }
        else if (q == 2) { st[0] = B.l20; st[1] = B.l21; }
        else if (q == 3) { st[0] = B.l30; st[1] = B.l31; st[2] = B.l32; }
        st[q] = dq;
        const bool pos_ok = (jb + q < M.npos) ? (dq > 0.0) : (dq < 0.0);
        if (!pos_ok) badl = 1;
        if (M.dinv >= 0) dinvb[M.dinv + jb + q] = iq;
        stage[mi * OMGX_STAGE_LD + 16 + q] = iq;
        if (q == 0) for (int qq = nb; qq < OMGX_NB; ++qq) stage[mi * OMGX_STAGE_LD + 16 + qq] = 0.0;
        double* pr = colb + M.pan + r * OMGX_PAN_LD;
        pr[0] = 0.0; pr[1] = 0.0; pr[2] = 0.0; pr[3] = 0.0;
      } else {
        const int base = baddr_k<KIND>(M, r, jb);
        const double a0 = A[base];
        const double l1 = A[base + (nb > 1 ? 1 : 0)], l2 = A[base + (nb > 2 ? 2 : 0)], l3 = A[base + (nb > 3 ? 3 : 0)];
        const double a1 = nb > 1 ? l1 : 0.0;
        const double a2 = nb > 2 ? l2 : 0.0;
        const double a3 = nb > 3 ? l3 : 0.0;
        const double u0 = a0;
        const double u1 = a1 - u0 * B.l10;
        const double u2 = a2 - u0 * B.l20 - u1 * B.l21;
        const double u3 = a3 - u0 * B.l30 - u1 * B.l31 - u2 * B.l32;
        const double p0 = u0 * B.i0, p1 = nb > 1 ? u1 * B.i1 : 0.0, p2 = nb > 2 ? u2 * B.i2 : 0.0,
                     p3 = nb > 3 ? u3 * B.i3 : 0.0;
        double* pr = colb + M.pan + r * OMGX_PAN_LD;
        pr[0] = u0; pr[1] = nb > 1 ? u1 : 0.0; pr[2] = nb > 2 ? u2 : 0.0; pr[3] = nb > 3 ? u3 : 0.0;
        const bool fact_row = r < M.nfact;
        A[base] = fact_row ? p0 : u0;
        if (nb > 1) A[base + 1] = fact_row ? p1 : u1;
        if (nb > 2) A[base + 2] = fact_row ? p2 : u2;
        if (nb > 3) A[base + 3] = fact_row ? p3 : u3;
      }
    }
    c.sync();
    OMGX_TOC(PH_LA);
    // block rows: staging -> matrix
    OMGX_PFOR(it, nm * 16) {
      const int mi = it >> 4, q = (it >> 2) & 3, k = it & 3;
      const BMat M = Ms[mi];
      if (jb < M.nfact && jb + q < M.nfact && k <= q) A[baddr_k<KIND>(M, jb + q, jb + k)] = stage[mi * OMGX_STAGE_LD + (it & 15)];
    }
    OMGX_TOC(PH_LS);
    // ---- phase B: trailing update, 16x16 tiles -------------------------------------------
    int tile0 = 0;
    for (int mi = 0; mi < nm; ++mi) {
      const BMat M = Ms[mi];
      const int s0 = jb + OMGX_NB;
      if (s0 >= M.nfact) continue;
      const int tr = (M.rows - s0 + 15) >> 4, tc = (M.nfact - s0 + 15) >> 4;
#ifdef OMGX_HOST_PORT
      (void)tile0; (void)tr; (void)tc;
      for (int r = s0; r < M.rows; ++r) {
        const int kmax = r < M.nfact ? r : M.nfact - 1;
        for (int k = s0; k <= kmax; ++k) {
          double acc = 0.0;
          for (int q = 0; q < OMGX_NB; ++q)
            acc += colb[M.pan + r * OMGX_PAN_LD + q] * (colb[M.pan + k * OMGX_PAN_LD + q] * stage[mi * OMGX_STAGE_LD + 16 + q]);
          A[baddr_k<KIND>(M, r, k)] -= acc;
        }
      }
#else
      // tile rows are dealt round-robin to the waves across all matrices (tile0 = running count);
      // the wave count is a power of two, so no integer division anywhere in this loop
      const int lane = c.lane(), nwm = c.nwaves() - 1;
      for (int ti = (c.wave() - tile0) & nwm; ti < tr; ti += nwm + 1) {
        const int R0 = s0 + 16 * ti;
        for (int tj = 0; tj < tc; ++tj) {
          const int K0 = s0 + 16 * tj;
          if (K0 > R0 + 15 && R0 + 15 < M.nfact) break;             // rest of the row is above the diagonal
          const int ra = R0 + (lane & 15), kb = K0 + (lane & 15), q = lane >> 4;
          // all loads unconditional on clamped addresses, masked afterwards: one LDS round trip
          const double av_l = colb[M.pan + (ra < M.rows ? ra : M.rows - 1) * OMGX_PAN_LD + q];
          const double bv_l = colb[M.pan + (kb < M.nfact ? kb : M.nfact - 1) * OMGX_PAN_LD + q];
          const double iv_l = stage[mi * OMGX_STAGE_LD + 16 + q];
          typedef double v4d __attribute__((ext_vector_type(4)));
          v4d acc;
          const int col = K0 + (lane & 15);
          int ad[4]; bool ok[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = R0 + (lane >> 4) + 4 * i;
            ok[i] = row < M.rows && col < M.nfact && (row >= M.nfact || col <= row);
            ad[i] = ok[i] ? baddr_k<KIND>(M, row, col) : M.a;
          }
          double al[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) al[i] = A[ad[i]];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = ok[i] ? al[i] : 0.0;
          const double av = (ra < M.rows) ? -av_l : 0.0;
          const double bv = (kb < M.nfact) ? bv_l * iv_l : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) if (ok[i]) A[ad[i]] = acc[i];
        }
      }
      tile0 += tr;
#endif
    }
    c.sync();
    OMGX_TOC(PH_LB);
  }
  *bad = c.rmax(badl ? 1.0 : 0.0) > 0.0 ? 1 : 0;
}

// Matrix descriptors of the block-arrow store: written once per solve at the head of w.col (shared
// by the workgroup; the plan tables they come from are global memory, i.e. a chain of dependent
// loads per look-up), followed by the staging area and the panel buffers.
template <class C>
OMGX_FN void kkt_describe(const C& c, const Dims& d, const Kkt& K, Work& w, bool sync_after = true) {
  // (one thread, a chain of table loads: the first thread of the SECOND wave, so that a caller whose next stage is serial work
  // of thread 0 -- the parameter program -- can let the two chains run side by side: sync_after = false, the barriers of
  // that stage make the descriptors visible)
  const int who = c.nthr() > 64 ? 64 : 0;
  BMat* Ms = (BMat*)w.col;
  double* stage = w.col + OMGX_BMAT_DOUBLES * (OMGX_MAX_LEAF + 1);
  const int pan0 = (int)(stage + OMGX_STAGE_LD * (OMGX_MAX_LEAF + 1) - w.col);
  if (c.tid() == who) {
    int pan = pan0;
    for (int l = 0; l < d.n_leaf; ++l) {
      BMat& M = Ms[l];
      if constexpr (C::wave_only) {
        // compact store: a = band of the leaf block (sparse diagonal leaf: its arrays), pan = its carried rows
        // (sparse: number of coupling slots), pad_ = row stride of the band, bw = stored band (-1: sparse leaf)
        M.a = K.d_off[l]; M.ld = K.ld(l); M.nfact = K.nl(l); M.rows = K.nl(l) + K.nc(l) + 1; M.npos = M.nfact;
        M.dinv = K.leaf_off[l]; M.pan = K.lf_w[l]; M.cpl = K.cpl_ptr[l]; M.bw = K.lf_kind[l] ? -1 : K.lf_band[l]; M.pad_ = K.lf_ldb[l];
        continue;
      }
      M.a = K.d_off[l]; M.ld = K.ld(l); M.nfact = K.nl(l); M.rows = K.nl(l) + K.nc(l) + 1; M.npos = M.nfact;   // + the rhs row
      M.dinv = K.leaf_off[l]; M.pan = pan; pan += C::hbm ? OMGX_PAN_SMALL(M.nfact) : OMGX_PAN_LD * M.rows; M.cpl = K.cpl_ptr[l]; M.bw = K.leaf_bw[l];
    }
    BMat& Mr = Ms[d.n_leaf];
    Mr.a = C::root_lds ? 0 : K.d_off[d.n_leaf]; Mr.ld = 0; Mr.nfact = d.nr; Mr.rows = d.nr + 1; Mr.npos = d.n_root; Mr.dinv = -1; Mr.pan = pan0; Mr.cpl = 0; Mr.bw = d.nr;
    Mr.pad_ = K.d_off[d.n_leaf];      // where the assembly writes the root block (Mr.a: where the factorisation reads it)
  }
  if (sync_after) c.sync();
}

#ifndef OMGX_HOST_PORT
// ---------------------------------------------------------------------------
// Wave path (every panel of the store fits one wave, Dims::wave_ok): register-resident factorisation
// (omgx_wave.h).  Leaf l is factorised by wave l % nwaves -- all leaves at once, no workgroup barrier inside
// --, its Schur complement S_l = Wt Delta^{-1} Wt' is formed by the same wave on the matrix pipe
// (16x16x4 fp64 MFMA tiles) and subtracted from the root in leaf order (one leaf per barrier-separated
// round: fixed order of the sums); the root is factorised by wave 0.
// ---------------------------------------------------------------------------
OMGX_FN WPanel wpanel_leaf(const BMat& M) {
  WPanel P; P.base = M.a; P.ld = M.ld; P.n = M.nfact; P.nreg = M.rows - 2; P.nvec = 2; P.npos = M.nfact; P.bw = M.bw;
  P.vrow = M.rows - 2; P.band = M.bw; P.ldb = M.pad_; P.wbase = M.pan;
  return P;
}
OMGX_FN WPanel wpanel_root(const BMat& M, int n_root) {
  WPanel P; P.base = M.a; P.ld = 0; P.n = M.nfact; P.nreg = M.nfact; P.nvec = 1; P.npos = n_root; P.bw = M.nfact;
  P.vrow = M.nfact; P.band = -1; P.ldb = 0; P.wbase = 0;
  return P;
}

// Sparse diagonal leaf (the terminal slacks g*: every variable alone on its diagonal, coupled to one trajectory
// coefficient and to t): arrays of n entries each -- diagonal | S coupling slots | t coupling | right-hand side.
// Nothing to factorise; its Schur complement has one owner per target (the plan checked that no root position is
// coupled to two of its variables), the two entries every variable shares -- (t, t) and (rhs, t) -- are wave sums.
struct DiagLeaf { int base, n, S, dinv, dl; };
OMGX_FN DiagLeaf diag_leaf(const BMat& M) {
  DiagLeaf L;
  L.base = __builtin_amdgcn_readfirstlane(M.a); L.n = __builtin_amdgcn_readfirstlane(M.nfact);
  L.S = __builtin_amdgcn_readfirstlane(M.pan); L.dinv = __builtin_amdgcn_readfirstlane(M.dinv); L.dl = 4 * L.dinv;
  return L;
}

template <class C>
OMGX_FN int kkt_factor_wave(const C& c, const Dims& d, const Kkt& K, Work& w) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  const BMat* Ms = (const BMat*)w.col;
  const int lane = c.lane(), wave = c.wave(), nw = c.nwaves();
  const int koff = (int)(w.kkt - omgx_lds);            // the out-of-line wave routines address the LDS by offset
  OMGX_TIC();
  int badl = 0;
#ifdef OMGX_PROFILE
  const long long tw0_ = clock64();
#endif
  for (int l = wave; l < d.n_leaf; l += nw) {
    const int kind_bw = __builtin_amdgcn_readfirstlane(Ms[l].bw);
    if (kind_bw < 0) {
      const DiagLeaf L = diag_leaf(Ms[l]);
      const double dj = w.kkt[L.base + (lane < L.n ? lane : 0)];
      if (lane < L.n) { w.dinv[L.dinv + lane] = rcp_pivot(dj); if (!(dj > 0.0)) badl = 1; }
      continue;
    }
    const WPanel P = wpanel_leaf(Ms[l]);
    // hyperplane leaves are banded (half bandwidth 5 in reverse Cuthill-McKee order): compile-time band of 8
    badl |= (kind_bw <= 8) ? wave_ldl<OMGX_WAVE_COLS, 8, true>(koff, P) : wave_ldl<OMGX_WAVE_COLS, OMGX_WAVE_COLS, true>(koff, P);
    wave_fence();
    const double dl = wave_dinv<true>(w.kkt, P);
    if (lane < P.n) w.dinv[Ms[l].dinv + lane] = dl;
  }
#ifdef OMGX_PROFILE
  if (c.tid() == 0) { c.prof[PH_F_SCALE] += clock64() - tw0_; c.prof[PH_F_PARK] += 1; }      // raw leaf time of wave 0, number of factorisations
#endif
  if (c.rmax(badl ? 1.0 : 0.0) > 0.0) return 1;       // (two barriers: the panels and inverse pivots are visible)
  OMGX_TOC(PH_F_LEAF);
  double* R = K.R();
  for (int base = 0; base < d.n_leaf; base += nw) {
    const int l = base + wave;
    v4d acc00 = {0.0, 0.0, 0.0, 0.0}, acc10 = acc00, acc11 = acc00;
    int nc1 = 0, ci_b0 = 0, ci_b1 = 0, ci_a[8];
    bool sparse = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) ci_a[i] = 0;
    if (l < d.n_leaf) {
      const BMat M = Ms[l];
      sparse = __builtin_amdgcn_readfirstlane(M.bw) < 0;
      if (!sparse) {
      const int n = __builtin_amdgcn_readfirstlane(M.nfact), ld = __builtin_amdgcn_readfirstlane(M.ld);
      nc1 = __builtin_amdgcn_readfirstlane(M.rows) - n;            // coupling rows + the right-hand-side row (last)
      const double* Wt = w.kkt + __builtin_amdgcn_readfirstlane(M.pan);
      const double* di = w.dinv + __builtin_amdgcn_readfirstlane(M.dinv);
      const int32_t* ci = K.cpl_idx + __builtin_amdgcn_readfirstlane(M.cpl);
      // root positions of this lane's rows / columns (global table: requested before the MFMA loop)
      const int cb0 = lane & 15, cb1 = 16 + (lane & 15);
      ci_b0 = ci[cb0 < nc1 - 1 ? cb0 : 0]; ci_b1 = ci[cb1 < nc1 - 1 ? cb1 : 0];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ca = 16 * (i >> 2) + (lane >> 4) + 4 * (i & 3);
        ci_a[i] = ca < nc1 - 1 ? ci[ca] : d.nr;
      }
      const int r0 = lane & 15, r1 = 16 + (lane & 15), q = lane >> 4;
      const int r0c = r0 < nc1 ? r0 : nc1 - 1, r1c = r1 < nc1 ? r1 : nc1 - 1;
      const bool two = nc1 > 16;
      for (int j0 = 0; j0 < n; j0 += 4) {
        const int j = j0 + q, jc = j < n ? j : n - 1;
        const double w0 = Wt[r0c * ld + jc], w1 = Wt[r1c * ld + jc], dj = di[jc];       // unconditional loads
        const double b0 = (r0 < nc1 && j < n) ? w0 : 0.0, b1 = (r1 < nc1 && j < n) ? w1 : 0.0;
        acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(b0 * dj, b0, acc00, 0, 0, 0);
        if (two) {
          acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1 * dj, b0, acc10, 0, 0, 0);
          acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1 * dj, b1, acc11, 0, 0, 0);
        }
      }
      }
    }
    // subtract from the root, one leaf per round
    const int lend = base + nw < d.n_leaf ? base + nw : d.n_leaf;
    for (int lr = base; lr < lend; ++lr) {
      if (l == lr && !sparse) {
        const int cb0 = lane & 15, cb1 = 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ca0 = (lane >> 4) + 4 * i, ca1 = 16 + ca0;
          if (ca0 < nc1 && cb0 < nc1 - 1 && cb0 <= ca0) R[tri(ci_a[i], ci_b0)] -= acc00[i];
          if (ca1 < nc1 && cb0 < nc1 - 1) R[tri(ci_a[4 + i], ci_b0)] -= acc10[i];
          if (ca1 < nc1 && cb1 < nc1 - 1 && cb1 <= ca1) R[tri(ci_a[4 + i], ci_b1)] -= acc11[i];
        }
      } else if (l == lr) {
        // sparse diagonal leaf: lane j owns variable j.  v_0 .. v_{S-1} at root positions p_0 .. p_{S-1}, v_t at
        // the position of t (n_root - 1), the right-hand side r_j at row nr of the root
        const DiagLeaf L = diag_leaf(Ms[l]);
        const int j = lane < L.n ? lane : 0;
        const bool on = lane < L.n;
        const double* A = w.kkt + L.base + j;
        const double di = w.dinv[L.dinv + j];
        const double vt = A[(1 + L.S) * L.n], rj = A[(2 + L.S) * L.n];
        const int pt = d.n_root - 1;
        double vs[4]; int ps[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bool has = s < L.S;
          ps[s] = has ? K.dl_pos[L.dl + s * L.n + j] : -1;
          const double v = A[(1 + (has ? s : 0)) * L.n];
          vs[s] = (has && on && ps[s] >= 0) ? v : 0.0;
        }
        if (on) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (ps[s] < 0 || s >= L.S) continue;
            const double u = vs[s] * di;
#pragma unroll
            for (int s2 = 0; s2 <= s; ++s2) {
              if (ps[s2] < 0) continue;
              const int pa = ps[s] > ps[s2] ? ps[s] : ps[s2], pb = ps[s] > ps[s2] ? ps[s2] : ps[s];
              R[tri(pa, pb)] -= u * vs[s2];
            }
            R[tri(pt, ps[s])] -= u * vt;                  // (t is the last root variable: pt > every p_s)
            R[tri(d.nr, ps[s])] -= u * rj;
          }
        }
        // shared targets: fixed-order wave sums
        const double stt = c.wave_sum(on ? vt * di * vt : 0.0), srt = c.wave_sum(on ? rj * di * vt : 0.0);
        if (lane == 0) { R[tri(pt, pt)] -= stt; R[tri(d.nr, pt)] -= srt; }
      }
      c.sync();
    }
  }
  OMGX_TOC(PH_F_SCHUR);
  // (Round 6 measured the root on the four waves of the workgroup -- root_ldl_4w, omgx_wave.h: the same bits, 30.2 k -> 22.1 k cycles alone
  // on a CU -- and did not adopt it: inside the solve kernel f_root drops by 3.4 k only while the phases of the agent that shares the CU
  // lose what the three parked waves used to leave it, 342 k -> 349 k cycles per solve: profiles/r06_micro_root_ldl.txt)
  int badr = 0;
  if (wave == 0) {
    const WPanel P = wpanel_root(Ms[d.n_leaf], d.n_root);
#ifdef OMGX_PROFILE
    const long long tr0_ = clock64();
#endif
    badr = wave_ldl<OMGX_WAVE_COLS, OMGX_WAVE_COLS, false>(koff, P);
#ifdef OMGX_PROFILE
    if (c.tid() == 0) c.prof[PH_F_SWEEP] += clock64() - tr0_;      // raw root time
#endif
  }
  const int bad = c.rmax(badr ? 1.0 : 0.0) > 0.0 ? 2 : 0;       // 2: the leaves are fine, the root has the wrong inertia
  OMGX_TOC(PH_F_ROOT);
  return bad;
}

// Root block alone once more (its diagonal was changed in place): wave_ldl stores nothing when a pivot has the
// wrong sign, so after a failed attempt the store still holds the root as the Schur complements left it and
// the leaf factors stay valid.
template <class C>
OMGX_FN int kkt_refactor_root_wave(const C& c, const Dims& d, const Kkt& K, Work& w) {
  const BMat* Ms = (const BMat*)w.col;
  const int koff = (int)(w.kkt - omgx_lds);
  OMGX_TIC();
  int badr = 0;
  if (c.wave() == 0) badr = wave_ldl<OMGX_WAVE_COLS, OMGX_WAVE_COLS, false>(koff, wpanel_root(Ms[d.n_leaf], d.n_root));
  const int bad = c.rmax(badr ? 1.0 : 0.0) > 0.0 ? 2 : 0;
  OMGX_TOC(PH_F_ROOT);
  return bad;
}

// the solve that goes with kkt_factor_wave: root backward substitution by wave 0, then every leaf wave
// corrects its right-hand side by the root solution and substitutes backwards -- no workgroup barrier
// inside a leaf, no atomics
template <class C>
OMGX_FN void kkt_solve_wave(const C& c, const Dims& d, const Kkt& K, Work& w, double* sol) {
  const BMat* Ms = (const BMat*)w.col;
  const int lane = c.lane(), wave = c.wave(), nw = c.nwaves();
  const int koff = (int)(w.kkt - omgx_lds);
  OMGX_TIC();
  if (wave == 0) {
    const WPanel P = wpanel_root(Ms[d.n_leaf], d.n_root);
    const double dl = wave_dinv<false>(w.kkt, P);
    const double y = w.kkt[wcarried<false>(P, P.n) + (lane < P.n ? lane : 0)];           // L^{-1} r: the carried vector row
    const double x = wave_bwd<OMGX_WAVE_COLS, false>(koff, P, dl, y * dl);
    if (lane < P.n) sol[d.root_off + lane] = x;
  }
  c.sync();
  OMGX_TOC(PH_K_ROOT);
  double* xg = w.col + OMGX_BMAT_DOUBLES * (OMGX_MAX_LEAF + 1) + wave * 64;      // gathered root solution of this wave's leaf
  for (int l = wave; l < d.n_leaf; l += nw) {
    const BMat M = Ms[l];
    if (__builtin_amdgcn_readfirstlane(M.bw) < 0) {
      // sparse diagonal leaf: x_j = (r_j - sum_s v_s x_r[p_s] - v_t x_t) / d_j
      const DiagLeaf L = diag_leaf(M);
      const int j = lane < L.n ? lane : 0;
      const double* A = w.kkt + L.base + j;
      double acc = A[(2 + L.S) * L.n] - A[(1 + L.S) * L.n] * sol[d.root_off + d.n_root - 1];
      for (int s = 0; s < L.S; ++s) {
        const int ps = K.dl_pos[L.dl + s * L.n + j];
        acc = fma(-A[(1 + s) * L.n], sol[d.root_off + (ps < 0 ? 0 : ps)] * (ps < 0 ? 0.0 : 1.0), acc);
      }
      if (lane < L.n) sol[L.dinv + lane] = acc * w.dinv[L.dinv + j];
      continue;
    }
    const WPanel P = wpanel_uniform(wpanel_leaf(M));
    const int n = P.n, nc = P.nreg + 1 - P.n, ld = P.ld;          // (wave-uniform values: scalar loop bounds)
    const int32_t* ci = K.cpl_idx + __builtin_amdgcn_readfirstlane(M.cpl);
    if (lane < nc) xg[lane] = sol[d.root_off + ci[lane]];
    wave_fence();
    const int j = lane < n ? lane : 0;
    const double* Wt = w.kkt + P.wbase + j;
    double acc = Wt[nc * ld];                          // L^{-1} r_l
#pragma unroll 4
    for (int a = 0; a < nc; ++a) acc = fma(-Wt[a * ld], xg[a], acc);
    const int dv = __builtin_amdgcn_readfirstlane(M.dinv);
    const double dl = w.dinv[dv + j];
    const double x = wave_bwd<OMGX_WAVE_COLS, true>(koff, P, dl, acc * dl);
    if (lane < n) sol[dv + lane] = x;
    wave_fence();
  }
  c.sync();
  OMGX_TOC(PH_K_BWD);
}

// One more solve with the factors kkt_factor_wave left in the store (the factorisation carries the right-hand side of the
// Newton system along; a second-order correction solves once more with the same factors).  In: the raw right-hand side
// where kkt_rhs writes it (last carried row of every leaf, row nr of the root).  Leaf waves substitute forwards in
// registers (wave_fwd_r) and form their share W Delta^{-1} y of the root's right-hand side; wave 0 subtracts the shares in
// leaf order (fixed order of the sums), substitutes the root forwards and backwards; the leaves finish as in
// kkt_solve_wave.  out [N]: the solution in position order (the equality multipliers of this solve are dropped).
template <class C>
OMGX_FN void kkt_solve2_wave(const C& c, const Dims& d, const Kkt& K, Work& w, double* out, bool with_y = false) {
  const BMat* Ms = (const BMat*)w.col;
  const int lane = c.lane(), wave = c.wave(), nw = c.nwaves();
  const int koff = (int)(w.kkt - omgx_lds);
  double* xg0 = w.col + OMGX_BMAT_DOUBLES * (OMGX_MAX_LEAF + 1);
  double* R = K.R();
  const int rrow = tri(d.nr, 0);
  for (int base = 0; base < d.n_leaf; base += nw) {
    const int l = base + wave;
    double* xg = xg0 + wave * 64;
    if (l < d.n_leaf && __builtin_amdgcn_readfirstlane(Ms[l].bw) >= 0) {
      const BMat M = Ms[l];
      const WPanel P = wpanel_uniform(wpanel_leaf(M));
      const int n = P.n, nc = P.nreg + 1 - P.n, ld = P.ld;
      const int j = lane < n ? lane : 0;
      double* Wt = w.kkt + P.wbase;
      const int dv = __builtin_amdgcn_readfirstlane(M.dinv);
      const double dl = w.dinv[dv + j];
      const double y = wave_fwd_r<true>(koff, P, dl, Wt[nc * ld + j]);
      if (lane < n) { Wt[nc * ld + lane] = y; xg[lane] = y * dl; }
      wave_fence();
      // share of coupling row a: sum_j W_aj y_j / d_j
      const int a = lane < nc ? lane : 0;
      double acc = 0.0;
#pragma unroll 4
      for (int q = 0; q < n; ++q) acc = fma(Wt[a * ld + q], xg[q], acc);
      wave_fence();
      if (lane < nc) xg[lane] = acc;
    }
    c.sync();
    if (wave == 0) {
      const int lend = base + nw < d.n_leaf ? base + nw : d.n_leaf;
      for (int lr = base; lr < lend; ++lr) {
        const BMat M = Ms[lr];
        if (__builtin_amdgcn_readfirstlane(M.bw) >= 0) {
          const int nc = __builtin_amdgcn_readfirstlane(M.rows) - 1 - __builtin_amdgcn_readfirstlane(M.nfact);
          const int32_t* ci = K.cpl_idx + __builtin_amdgcn_readfirstlane(M.cpl);
          const double* xs = xg0 + (lr - base) * 64;
          if (lane < nc) R[rrow + ci[lane]] -= xs[lane];
        } else {
          // sparse diagonal leaf: y = r; lane j owns variable j and the root positions it is coupled to
          const DiagLeaf L = diag_leaf(M);
          const int j = lane < L.n ? lane : 0;
          const bool on = lane < L.n;
          const double* A = w.kkt + L.base + j;
          const double u = A[(2 + L.S) * L.n] * w.dinv[L.dinv + j];
          for (int s = 0; s < L.S; ++s) {
            const int ps = K.dl_pos[L.dl + s * L.n + j];
            if (on && ps >= 0) R[rrow + ps] -= A[(1 + s) * L.n] * u;
          }
          const double st = c.wave_sum(on ? A[(1 + L.S) * L.n] * u : 0.0);
          if (lane == 0) R[rrow + d.n_root - 1] -= st;
        }
        wave_fence();
      }
    }
    c.sync();
  }
  if (wave == 0) {
    const WPanel P = wpanel_root(Ms[d.n_leaf], d.n_root);
    const double dl = wave_dinv<false>(w.kkt, P);
    const int j = lane < P.n ? lane : 0;
    const double y = wave_fwd_r<false>(koff, P, dl, w.kkt[wcarried<false>(P, P.n) + j]);
    const double x = wave_bwd_r<false>(koff, P, dl, y * dl);
    if (lane < (with_y ? P.n : d.n_root)) out[d.root_off + lane] = x;      // (with_y: the equality multipliers behind the N positions)
  }
  c.sync();
  double* xg = xg0 + wave * 64;
  for (int l = wave; l < d.n_leaf; l += nw) {
    const BMat M = Ms[l];
    if (__builtin_amdgcn_readfirstlane(M.bw) < 0) {
      const DiagLeaf L = diag_leaf(M);
      const int j = lane < L.n ? lane : 0;
      const double* A = w.kkt + L.base + j;
      double acc = A[(2 + L.S) * L.n] - A[(1 + L.S) * L.n] * out[d.root_off + d.n_root - 1];
      for (int s = 0; s < L.S; ++s) {
        const int ps = K.dl_pos[L.dl + s * L.n + j];
        acc = fma(-A[(1 + s) * L.n], out[d.root_off + (ps < 0 ? 0 : ps)] * (ps < 0 ? 0.0 : 1.0), acc);
      }
      if (lane < L.n) out[L.dinv + lane] = acc * w.dinv[L.dinv + j];
      continue;
    }
    const WPanel P = wpanel_uniform(wpanel_leaf(M));
    const int n = P.n, nc = P.nreg + 1 - P.n, ld = P.ld;
    const int32_t* ci = K.cpl_idx + __builtin_amdgcn_readfirstlane(M.cpl);
    if (lane < nc) xg[lane] = out[d.root_off + ci[lane]];
    wave_fence();
    const int j = lane < n ? lane : 0;
    const double* Wt = w.kkt + P.wbase + j;
    double acc = Wt[nc * ld];
#pragma unroll 4
    for (int a = 0; a < nc; ++a) acc = fma(-Wt[a * ld], xg[a], acc);
    const int dv = __builtin_amdgcn_readfirstlane(M.dinv);
    const double dl = w.dinv[dv + j];
    const double x = wave_bwd_r<true>(koff, P, dl, acc * dl);
    if (lane < n) out[dv + lane] = x;
    wave_fence();
  }
  c.sync();
}
#endif

// Factorise the assembled block-arrow matrix in place.  Returns 0 if the
// inertia is (N positive, n_eq negative), 1 otherwise.
template <class C>
OMGX_FN int kkt_factor(const C& c, const Dims& d, const Kkt& K, Work& w) {
#ifndef OMGX_HOST_PORT
  if constexpr (C::wave_only) return kkt_factor_wave(c, d, K, w);      // (compact store, omgx_plan.h)
#endif
  if constexpr (C::wave_only) return 1; else {
  int bad = 0;
  OMGX_TIC();
  BMat* Ms = (BMat*)w.col;
  double* stage = w.col + OMGX_BMAT_DOUBLES * (OMGX_MAX_LEAF + 1);
  if (d.n_leaf > 0) {
#ifdef OMGX_HOST_PORT
    ldl_left4(c, Ms, d.n_leaf, w.kkt, w.dinv, w.col, &bad);
#else
    {
      int total_rows = 0, nmax = 0; bool room = true;
      for (int l = 0; l < d.n_leaf; ++l) {
        total_rows += Ms[l].rows; if (Ms[l].nfact > nmax) nmax = Ms[l].nfact;
        if (Ms[l].nfact + 30 > OMGX_PAN_LD * Ms[l].rows) room = false;     // staging behind the inverse-pivot slot
      }
      // a spare wave per leaf next to the row waves
      const bool coop = room && ((total_rows + 63) >> 6) + d.n_leaf <= c.nwaves();       // (leaf panels are row-major: baddr_k<1>)
      if constexpr (C::hbm) ldl_left4(c, Ms, d.n_leaf, w.kkt, w.dinv, w.col, &bad);     // (the small panel scratch of the spill modes has no room for the cooperative form)
      else if (coop) ldl_left4_coop<1>(c, Ms, d.n_leaf, w.kkt, w.dinv, w.col, &bad, total_rows, nmax);
      else ldl_left4(c, Ms, d.n_leaf, w.kkt, w.dinv, w.col, &bad);
    }
#endif
#ifdef OMGX_COUNT_FACT
    if (bad) ++omgx_dbg_cnt[0];
#endif
    if (bad) return 1;
  }
  OMGX_TOC(PH_F_LEAF);
  // Schur complement onto the root:  R[ci[a]][ci[b]] -= sum_j Wt[a][j] Wt[b][j] / d_j
  // (spill modes: from here on the root block lives in LDS -- the Schur updates, its factorisation and its substitutions
  // are read-modify-write chains on a few thousand doubles; nothing reads the store's copy again)
  if constexpr (C::root_lds) {
    const double* Rg = K.R();
    const int nrd = ((d.nr + 1) * (d.nr + 2)) / 2;
    OMGX_PFOR(i, nrd) w.root[i] = Rg[i];
    c.sync();
  }
  double* R = C::root_lds ? w.root : K.R();
#ifdef OMGX_HOST_PORT
  for (int l = 0; l < d.n_leaf; ++l) {
    const int n = K.nl(l), nc = K.nc(l), ld = K.ld(l);
    const double* Wt = K.P(l) + n * ld;
    const double* di = w.dinv + K.leaf_off[l];
    const int32_t* ci = K.cpl_idx + K.cpl_ptr[l];
    // carried rows 0..nc-1 are coupling rows (root position ci[a]), row nc the leaf's right-hand side,
    // which lands in the root's right-hand-side row (index nr):  r_r -= Wt D^{-1} (L^{-1} r_l)
    for (int ai = 0; ai <= nc; ++ai) for (int ak = 0; ak <= ai && ak < nc; ++ak) {
      double acc = 0.0;
      for (int j = 0; j < n; ++j) acc += Wt[ai * ld + j] * Wt[ak * ld + j] * di[j];
      R[tri(ai < nc ? ci[ai] : d.nr, ci[ak])] -= acc;
    }
  }
#else
  {
    // S = Wt Delta^{-1} Wt' per leaf as 16x16 MFMA tiles (K swept in steps of 4), the tiles of one leaf
    // dealt round-robin to the waves.  Different leaves add into the same root entries, so the leaves
    // take turns (one barrier per leaf): plain read-modify-write, fixed order of the sums, no atomics.
    const int lane = c.lane();
    for (int l = 0; l < d.n_leaf; ++l) {
      const BMat M = Ms[l];                       // dimensions from LDS, not from the global plan tables
      // carried rows: nc - 1 coupling rows + the right-hand-side row (last), which maps to row nr of the root
      const int n = M.nfact, nc = M.rows - M.nfact;
      OMGX_PANEL_STRIDES(C, M, sr, sc);
      const double* Wt = w.kkt + M.a + n * sr;
      const double* di = w.dinv + M.dinv;
      const int32_t* ci = K.cpl_idx + M.cpl;
      const int tn = (nc + 15) >> 4, nwm = c.nwaves() - 1;
      int tile = 0;
      for (int ti = 0; ti < tn; ++ti) for (int tj = 0; tj <= ti; ++tj, ++tile) {
        if ((tile & nwm) != c.wave()) continue;
        typedef double v4d __attribute__((ext_vector_type(4)));
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        const int ra = 16 * ti + (lane & 15), rb = 16 * tj + (lane & 15), q = lane >> 4;
        const int rac = ra < nc ? ra : nc - 1, rbc = rb < nc ? rb : nc - 1;
        const double* Wa = Wt + rac * sr;
        const double* Wb = Wt + rbc * sr;
        // eight K steps per round: their 24 loads are in flight together (in the spill modes every one of them is
        // an L2 round trip; one step at a time the sweep over a 64-column leaf was 16 of them in a row)
        for (int j0 = 0; j0 < n; j0 += 32) {
          double wa[8], wb[8], wd[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int j = j0 + 4 * u + q;
            const int jc = j < n ? j : n - 1;
            wa[u] = Wa[jc * sc]; wd[u] = di[jc]; wb[u] = Wb[jc * sc];     // unconditional loads
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int j = j0 + 4 * u + q;
            const double av = (ra < nc && j < n) ? wa[u] * wd[u] : 0.0;
            const double bv = (rb < nc && j < n) ? wb[u] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
          }
        }
        const int cb = 16 * tj + (lane & 15);
        for (int i = 0; i < 4; ++i) {
          const int ca = 16 * ti + (lane >> 4) + 4 * i;
          if (ca < nc && cb < nc - 1 && cb <= ca) R[tri(ca < nc - 1 ? ci[ca] : d.nr, ci[cb])] -= acc[i];
        }
      }
      c.sync();
    }
  }
#endif
  c.sync();
  OMGX_TOC(PH_F_SCHUR);
  // the root is one matrix of a few dozen rows (a single wave): the two-phase MFMA routine spreads its
  // trailing tiles over all waves and wins there; the leaves (several matrices, many rows) are faster
  // with the one-barrier left-looking sweep
  // (the cooperative routine was tried for the root as well -- one row wave + one block wave: 31 k against
  // 43 k cycles standalone, but 1.5 % slower inside the fused kernel in an A/B of three bench runs each)
#ifdef OMGX_HOST_PORT
  // the host twin of "a failed root factorisation leaves the store untouched" (kkt_refactor_root): keep a copy
  if (d.wave_ok) omgx_rbak.assign(R, R + ((size_t)(d.nr + 1) * (d.nr + 2)) / 2);
#endif
  ldl_blocked<2>(c, Ms + d.n_leaf, 1, C::root_lds ? w.root : w.kkt, w.dinv, w.col, stage, &bad);
  OMGX_TOC(PH_F_ROOT);
#ifdef OMGX_COUNT_FACT
  if (bad) ++omgx_dbg_cnt[1];
#endif
  return bad ? 2 : 0;
  }
}

// The root block once more after its diagonal was changed (inertia correction of the root variables only: the
// leaf factors and the Schur complements stay).  Device: wave path only.  Host: from the copy taken before the
// failed attempt.
template <class C>
OMGX_FN void kkt_root_before_retry(const C& c, const Dims& d, const Kkt& K, Work& w) {
#ifdef OMGX_HOST_PORT
  double* R = K.R();
  for (size_t i = 0; i < omgx_rbak.size(); ++i) R[i] = omgx_rbak[i];
#endif
}
template <class C>
OMGX_FN int kkt_refactor_root(const C& c, const Dims& d, const Kkt& K, Work& w) {
#ifndef OMGX_HOST_PORT
  if constexpr (!C::wave_only) return 1;      // (never reached: the blocked instances run with Dims::wave_ok = 0)
  else return kkt_refactor_root_wave(c, d, K, w);
#else
  int bad = 0;
  BMat* Ms = (BMat*)w.col;
  double* stage = w.col + OMGX_BMAT_DOUBLES * (OMGX_MAX_LEAF + 1);
  double* R = K.R();
  omgx_rbak.assign(R, R + ((size_t)(d.nr + 1) * (d.nr + 2)) / 2);
  ldl_blocked<2>(c, Ms + d.n_leaf, 1, w.kkt, w.dinv, w.col, stage, &bad);
  return bad ? 2 : 0;
#endif
}

// x <- L^{-T} y for the same storage (entry (i, j) of L, i > j, multiplies x_i into row j).  With
// `iv` (inverse pivots) the storage holds U = L D instead of L (the leaf panels after ldl_left4):
// L_ij = U_ij iv_j, applied on the fly.
template <class C, class Addr>
OMGX_FN void trsv_bwd4(const C& c, const double* A, Addr L, int n, double* y, const double* iv = nullptr) {
  const int lane = c.lane(), nln = c.nlanes();
  const int nblk = (n + 3) >> 2;
  for (int bk = nblk - 1; bk >= 0; --bk) {
    const int jb = 4 * bk;
    const int nb = (n - jb) < 4 ? (n - jb) : 4;
    const int q1 = nb > 1 ? 1 : 0, q2 = nb > 2 ? 2 : 0, q3 = nb > 3 ? 3 : 0;
    const double m1 = nb > 1 ? 1.0 : 0.0, m2 = nb > 2 ? 1.0 : 0.0, m3 = nb > 3 ? 1.0 : 0.0;
    const double r0 = y[jb], r1 = m1 * y[jb + q1], r2 = m2 * y[jb + q2], r3 = m3 * y[jb + q3];
    const double s0 = iv ? iv[jb] : 1.0, s1 = iv ? iv[jb + q1] : 1.0, s2 = iv ? iv[jb + q2] : 1.0;
    const double l10 = m1 * A[L(jb + q1, jb)] * s0;
    const double l20 = m2 * A[L(jb + q2, jb)] * s0, l21 = m2 * A[L(jb + q2, jb + (q2 ? 1 : 0))] * s1;
    const double l30 = m3 * A[L(jb + q3, jb)] * s0, l31 = m3 * A[L(jb + q3, jb + (q3 ? 1 : 0))] * s1,
                 l32 = m3 * A[L(jb + q3, jb + (q3 ? 2 : 0))] * s2;
    const double x3 = r3, x2 = r2 - l32 * x3, x1 = r1 - l21 * x2 - l31 * x3, x0 = r0 - l10 * x1 - l20 * x2 - l30 * x3;
    for (int i = lane; i < jb + nb; i += nln) {
      if (i < jb) {
        const double si = iv ? iv[i] : 1.0;
        y[i] -= si * (A[L(jb, i)] * x0 + m1 * A[L(jb + q1, i)] * x1 + m2 * A[L(jb + q2, i)] * x2 + m3 * A[L(jb + q3, i)] * x3);
      } else {
        const int q = i - jb;
        y[i] = q == 0 ? x0 : (q == 1 ? x1 : (q == 2 ? x2 : x3));
      }
    }
    c.wave_sync();
  }
}

// y <- L^{-1} y for the same storage, by one wave (the forward twin of trsv_bwd4: blocks of four columns, the 4 x 4
// diagonal block solved redundantly in every lane, the rows below updated one per lane).  With `iv` the storage holds
// U = L D (leaf panels), L_ij = U_ij iv_j; without, L itself with the pivots on the diagonal (root).  Used by the
// second solve of an iteration (kkt_solve2_blocked): the first right-hand side goes through the factorisation instead.
template <class C, class Addr>
OMGX_FN void trsv_fwd4(const C& c, const double* A, Addr L, int n, double* y, const double* iv = nullptr) {
  const int lane = c.lane(), nln = c.nlanes();
  const int nblk = (n + 3) >> 2;
  for (int bk = 0; bk < nblk; ++bk) {
    const int jb = 4 * bk;
    const int nb = (n - jb) < 4 ? (n - jb) : 4;
    const int q1 = nb > 1 ? 1 : 0, q2 = nb > 2 ? 2 : 0, q3 = nb > 3 ? 3 : 0;
    const double m1 = nb > 1 ? 1.0 : 0.0, m2 = nb > 2 ? 1.0 : 0.0, m3 = nb > 3 ? 1.0 : 0.0;
    const double s0 = iv ? iv[jb] : 1.0, s1 = iv ? iv[jb + q1] : 1.0, s2 = iv ? iv[jb + q2] : 1.0, s3 = iv ? iv[jb + q3] : 1.0;
    const double l10 = m1 * A[L(jb + q1, jb)] * s0;
    const double l20 = m2 * A[L(jb + q2, jb)] * s0, l21 = m2 * A[L(jb + q2, jb + (q2 ? 1 : 0))] * s1;
    const double l30 = m3 * A[L(jb + q3, jb)] * s0, l31 = m3 * A[L(jb + q3, jb + (q3 ? 1 : 0))] * s1,
                 l32 = m3 * A[L(jb + q3, jb + (q3 ? 2 : 0))] * s2;
    const double x0 = y[jb], x1 = m1 * (y[jb + q1] - l10 * x0), x2 = m2 * (y[jb + q2] - l20 * x0 - l21 * x1),
                 x3 = m3 * (y[jb + q3] - l30 * x0 - l31 * x1 - l32 * x2);
    c.wave_sync();                                  // (every lane has read the block before anybody writes it)
    for (int i = jb + lane; i < n; i += nln) {
      if (i >= jb + nb) {
        y[i] -= A[L(i, jb)] * s0 * x0 + m1 * A[L(i, jb + q1)] * s1 * x1 + m2 * A[L(i, jb + q2)] * s2 * x2 + m3 * A[L(i, jb + q3)] * s3 * x3;
      } else {
        const int q = i - jb;
        y[i] = q == 0 ? x0 : (q == 1 ? x1 : (q == 2 ? x2 : x3));
      }
    }
    c.wave_sync();
  }
}

// Finish the solve of K sol = rhs.  The right-hand side went through the factorisation as the carried
// last row of every leaf panel and of the root (kkt_rhs wrote it there), so the forward substitutions
// are done: the rows hold L^{-1} r.  What is left: scale by the inverse pivots, the root's backward
// substitution, the leaves' correction by the root solution and their backward substitutions.
// sol: position order + equality multipliers.
template <class C>
OMGX_FN void kkt_solve(const C& c, const Dims& d, const Kkt& K, Work& w, double* sol) {
#ifndef OMGX_HOST_PORT
  if constexpr (C::wave_only) { kkt_solve_wave(c, d, K, w, sol); return; }
#endif
  if constexpr (!C::wave_only) {
  double* yr = sol + d.root_off;
  // leaf dimensions from the matrix descriptors kkt_factor left in LDS (not from the global plan
  // tables: every look-up there is a dependent global load)
  const BMat* Ms = (const BMat*)w.col;
  OMGX_TIC();
  // leaves: y_l <- Delta^{-1} (L^{-1} r_l), root: y_r <- D^{-1} (L^{-1} r_r)
  OMGX_PFOR(q, d.root_off) {
    int l = 0;
    while (q >= Ms[l].dinv + Ms[l].nfact) ++l;
    const BMat M = Ms[l];
    OMGX_PANEL_STRIDES(C, M, sr, sc);
    sol[q] = w.kkt[M.a + (M.rows - 1) * sr + (q - M.dinv) * sc] * w.dinv[q];
  }
  {
    const int rbase = Ms[d.n_leaf].a, nr = d.nr;
    const double* rootp = C::root_lds ? w.root : w.kkt;
    OMGX_PFOR(k, nr) yr[k] = rootp[rbase + tri(nr, k)] / rootp[rbase + tri(k, k)];
  }
  c.sync();
  OMGX_TOC(PH_K_FWD);
  // root backward substitution by wave 0 (packed L)
  if (c.wave() == 0) {
    const int n = d.nr, rbase = Ms[d.n_leaf].a;
    auto Lr = [=](int i, int j) { return rbase + tri(i, j); };
    trsv_bwd4(c, C::root_lds ? w.root : w.kkt, Lr, n, yr);
  }
  c.sync();
  OMGX_TOC(PH_K_ROOT);
  // leaves: y_l <- L^{-T} (y_l - Delta^{-1} Wt' x_r); the correction term by one thread per leaf column
  // (coupling rows in order: a fixed-order sum)
  OMGX_PFOR(q, d.root_off) {
    int l = 0;
    while (q >= Ms[l].dinv + Ms[l].nfact) ++l;
    const BMat M = Ms[l];
    const int n = M.nfact, nc = M.rows - 1 - M.nfact, j = q - M.dinv;
    OMGX_PANEL_STRIDES(C, M, sr, sc);
    const double* Pn = w.kkt + M.a + n * sr + j * sc;
    const int32_t* ci = K.cpl_idx + M.cpl;
    double acc = 0.0;
#pragma unroll 4
    for (int a = 0; a < nc; ++a) acc += Pn[a * sr] * yr[ci[a]];
    sol[q] -= acc * w.dinv[q];
  }
  c.sync();
  OMGX_TOC(PH_K_LEAFRHS);
  for (int l = c.wave(); l < d.n_leaf; l += c.nwaves()) {
    const BMat M = Ms[l];
    const int n = M.nfact, base = M.a;
    OMGX_PANEL_STRIDES(C, M, sr, sc);
    double* yl = sol + M.dinv;
    trsv_bwd4(c, w.kkt, [=](int i, int j) { return base + i * sr + j * sc; }, n, yl, w.dinv + M.dinv);   // panels hold U = L D
  }
  c.sync();
  OMGX_TOC(PH_K_BWD);
  }
}

// Right-hand side of the Newton system into the carried rows of the block-arrow store (after the
// store was zeroed, before the factorisation): leaf l gets -gbar of its variables, the root
// -gbar of the root variables and the equality residuals.
template <class C>
OMGX_FN void kkt_rhs(const C& c, const Dims& d, const Tables& T, Work& w, double t) {
  const BMat* Ms = (const BMat*)w.col;
  OMGX_PFOR(q, d.root_off) {
    int l = 0;
    while (q >= Ms[l].dinv + Ms[l].nfact) ++l;
    const BMat M = Ms[l];
    if constexpr (C::wave_only) {       // compact store: the right-hand side is the last carried row / the last array of a sparse leaf
      const int ad = M.bw < 0 ? M.a + (2 + M.pan) * M.nfact + (q - M.dinv) : M.pan + (M.rows - 1 - M.nfact) * M.ld + (q - M.dinv);
      w.kkt[ad] = -w.gbar[q];
    } else {
      OMGX_PANEL_STRIDES(C, M, sr, sc);
      w.kkt[M.a + (M.rows - 1) * sr + (q - M.dinv) * sc] = -w.gbar[q];
    }
  }
  const int rbase = Ms[d.n_leaf].pad_, nr = d.nr;
  OMGX_PFOR(k, nr) {
    double v;
    if (k < d.n_root) v = -w.gbar[d.root_off + k];
    else { const int r = T.eq_rows[k - d.n_root]; v = (w.rtype[r] == ROW_EQ) ? -(w.hv[r] - t * w.vv[r]) : 0.0; }
    w.kkt[rbase + tri(nr, k)] = v;
  }
}

// where kkt_rhs puts the right-hand side entry of position q (< N) in the store
template <class C>
OMGX_FN int kkt_rhs_slot(const Dims& d, const BMat* Ms, int q) {
  if (q >= d.root_off) return Ms[d.n_leaf].pad_ + tri(d.nr, q - d.root_off);
  int l = 0;
  while (q >= Ms[l].dinv + Ms[l].nfact) ++l;
  const BMat M = Ms[l];
  if constexpr (C::wave_only) return M.bw < 0 ? M.a + (2 + M.pan) * M.nfact + (q - M.dinv) : M.pan + (M.rows - 1 - M.nfact) * M.ld + (q - M.dinv);
  else { OMGX_PANEL_STRIDES(C, M, sr, sc); return M.a + (M.rows - 1) * sr + (q - M.dinv) * sc; }
}

#ifdef OMGX_HOST_PORT
static thread_local std::vector<double> omgx_sol2;       // output of a second solve (position order + equality multipliers)
#endif
// One more solve with the factors of the iteration (second-order correction): the caller wrote the right-hand side into
// the slots of kkt_rhs; out [N] receives the solution in position order (the equality multipliers of this solve are
// dropped).  Device: the wave routines on the wave path (Dims::wave_ok), the blocked form below otherwise; host port: plain loops
// over the storage its blocked routines leave (leaf panels U = L D with inverse pivots aside, root L with the pivots
// on the diagonal).
template <class C>
OMGX_FN void kkt_solve2(const C& c, const Dims& d, const Kkt& K, Work& w, double* out, bool with_y = false) {
#ifndef OMGX_HOST_PORT
  if constexpr (C::wave_only) {
    kkt_solve2_wave(c, d, K, w, out, with_y);
  } else {
    // Blocked storage (templates off the wave path, spill modes included): leaf right-hand sides from their carried rows into
    // `out` (LDS), a wave per leaf substitutes forwards, the leaves subtract W Delta^-1 y from the root's right-hand side one
    // after the other (fixed order), wave 0 substitutes the root forwards and backwards in its right-hand-side row, the leaves
    // finish as in kkt_solve.
    const BMat* Ms = (const BMat*)w.col;
    const int nr = d.nr, rbase = Ms[d.n_leaf].a;
    double* rootp = C::root_lds ? w.root : w.kkt;
    double* rr = rootp + rbase + tri(nr, 0);
    if constexpr (C::root_lds) { OMGX_PFOR(k, nr) rr[k] = w.kkt[Ms[d.n_leaf].pad_ + tri(nr, k)]; }
    OMGX_PFOR(q, d.root_off) {
      int l = 0;
      while (q >= Ms[l].dinv + Ms[l].nfact) ++l;
      const BMat M = Ms[l];
      OMGX_PANEL_STRIDES(C, M, sr, sc);
      out[q] = w.kkt[M.a + (M.rows - 1) * sr + (q - M.dinv) * sc];
    }
    c.sync();
    for (int l = c.wave(); l < d.n_leaf; l += c.nwaves()) {
      const BMat M = Ms[l];
      const int base = M.a;
      OMGX_PANEL_STRIDES(C, M, sr, sc);
      trsv_fwd4(c, w.kkt, [=](int i, int j) { return base + i * sr + j * sc; }, M.nfact, out + M.dinv, w.dinv + M.dinv);
    }
    c.sync();
    for (int l = 0; l < d.n_leaf; ++l) {
      const BMat M = Ms[l];
      const int n = M.nfact, nc = M.rows - 1 - n;
      OMGX_PANEL_STRIDES(C, M, sr, sc);
      const int32_t* ci = K.cpl_idx + M.cpl;
      const double* yl = out + M.dinv;
      const double* iv = w.dinv + M.dinv;
      OMGX_PFOR(a, nc) {
        const double* Pa = w.kkt + M.a + (n + a) * sr;
        double acc = 0.0;
#pragma unroll 4
        for (int j = 0; j < n; ++j) acc += Pa[j * sc] * (yl[j] * iv[j]);
        rr[ci[a]] -= acc;
      }
      c.sync();
    }
    if (c.wave() == 0) trsv_fwd4(c, rootp, [=](int i, int j) { return rbase + tri(i, j); }, nr, rr);
    c.sync();
    OMGX_PFOR(q, d.root_off) out[q] *= w.dinv[q];
    OMGX_PFOR(k, nr) rr[k] = rr[k] / rootp[rbase + tri(k, k)];
    c.sync();
    if (c.wave() == 0) trsv_bwd4(c, rootp, [=](int i, int j) { return rbase + tri(i, j); }, nr, rr);
    c.sync();
    OMGX_PFOR(q, d.root_off) {
      int l = 0;
      while (q >= Ms[l].dinv + Ms[l].nfact) ++l;
      const BMat M = Ms[l];
      const int n = M.nfact, nc = M.rows - 1 - M.nfact, j = q - M.dinv;
      OMGX_PANEL_STRIDES(C, M, sr, sc);
      const double* Pn = w.kkt + M.a + n * sr + j * sc;
      const int32_t* ci = K.cpl_idx + M.cpl;
      double acc = 0.0;
#pragma unroll 4
      for (int a = 0; a < nc; ++a) acc += Pn[a * sr] * rr[ci[a]];
      out[q] -= acc * w.dinv[q];
    }
    c.sync();
    for (int l = c.wave(); l < d.n_leaf; l += c.nwaves()) {
      const BMat M = Ms[l];
      const int base = M.a;
      OMGX_PANEL_STRIDES(C, M, sr, sc);
      trsv_bwd4(c, w.kkt, [=](int i, int j) { return base + i * sr + j * sc; }, M.nfact, out + M.dinv, w.dinv + M.dinv);
    }
    OMGX_PFOR(k, d.n_root) out[d.root_off + k] = rr[k];
    c.sync();
  }
#else
  const BMat* Ms = (const BMat*)w.col;
  double* R = w.kkt + Ms[d.n_leaf].a;
  const int nr = d.nr;
  for (int l = 0; l < d.n_leaf; ++l) {
    const BMat M = Ms[l];
    const int n = M.nfact, nc = M.rows - 1 - n, ld = M.ld;
    double* P = w.kkt + M.a;
    double* y = P + (M.rows - 1) * ld;
    const double* iv = w.dinv + M.dinv;
    for (int i = 0; i < n; ++i) {
      double acc = y[i];
      for (int j = 0; j < i; ++j) acc -= P[i * ld + j] * iv[j] * y[j];
      y[i] = acc;
    }
    const int32_t* ci = K.cpl_idx + M.cpl;
    for (int a = 0; a < nc; ++a) {
      double acc = 0.0;
      for (int j = 0; j < n; ++j) acc += P[(n + a) * ld + j] * (y[j] * iv[j]);
      R[tri(nr, ci[a])] -= acc;
    }
  }
  for (int i = 0; i < nr; ++i) {
    double acc = R[tri(nr, i)];
    for (int j = 0; j < i; ++j) acc -= R[tri(i, j)] * R[tri(nr, j)];
    R[tri(nr, i)] = acc;
  }
  omgx_sol2.resize((size_t)d.N + d.n_eq);
  kkt_solve(c, d, K, w, omgx_sol2.data());
  for (int q = 0; q < d.N + (with_y ? d.n_eq : 0); ++q) out[q] = omgx_sol2[q];
#endif
}

// Lagrangian Hessian, the share of owner bin `bin`: the items of the terms with >= 2 variables, weight w.ht[row] =
// multiplier x signed scale (row m = objective: 1), summed per KKT address in table order and added to the store
// (omgx_plan.h (4): ELL records, the target in the last record of its run, everything else to the dump slot).
// Used by the assembly of the solve and by the verification entry (ipm_eval).
template <bool Q4>
OMGX_FN void hess_bin_t(const Dims& d, const Tables& T, Work& w, int m, int bin, int dump) {
  double acc = 0.0;
  for (int e0 = 0; e0 < d.kh_len; e0 += OMGX_REC_BATCH) {
    HItem q[OMGX_REC_BATCH];
#pragma unroll
    for (int i = 0; i < OMGX_REC_BATCH; ++i) q[i] = T.kh_rec[(e0 + i) * OMGX_NBIN + bin];
    double h[OMGX_REC_BATCH];
#pragma unroll
    for (int i = 0; i < OMGX_REC_BATCH; ++i) {
      const int r = q[i].row < m ? q[i].row : 0;
      const double lam = (q[i].row < m) ? w.ht[r] : 1.0;          // (row multiplier x signed scale, set below the residuals)
      const double xs = w.slots[q[i].slot < 0 ? 0 : q[i].slot], x3 = w.x[q[i].vthird < 0 ? 0 : q[i].vthird];
      h[i] = (q[i].kind ? 2.0 : 1.0) * lam * q[i].coef * (q[i].slot < 0 ? 1.0 : xs) * (q[i].vthird < 0 ? 1.0 : x3);
      if (Q4) { const double x4 = w.x[q[i].vfourth < 0 ? 0 : q[i].vfourth]; h[i] *= (q[i].vfourth < 0 ? 1.0 : x4); }
    }
    // (the old values are read first, together: a thread's targets are distinct, only the dump
    // slot repeats, and what ends up there does not matter)
    double old[OMGX_REC_BATCH];
#pragma unroll
    for (int i = 0; i < OMGX_REC_BATCH; ++i) old[i] = w.kkt[q[i].target >= 0 ? q[i].target : dump];
#pragma unroll
    for (int i = 0; i < OMGX_REC_BATCH; ++i) {
      acc += h[i];
      w.kkt[q[i].target >= 0 ? q[i].target : dump] = old[i] + acc;
      acc = q[i].target >= 0 ? 0.0 : acc;
    }
  }
}

template <class C>
OMGX_FN void hess_bin(const Dims& d, const Tables& T, Work& w, int m, int bin, int dump) {
  hess_bin_t<C::general>(d, T, w, m, bin, dump);
}

// cut runs of Hessian items: the side slots are added to their entry, in order (after the owners' pass and a barrier;
// ends with a barrier when there is anything to do)
template <class C>
OMGX_FN void hess_fix(const C& c, const Dims& d, const Tables& T, Work& w) {
  if (d.n_khfix == 0) return;
  OMGX_PFOR(i, d.n_khfix) {
    const int32_t* f = T.kh_fix + 3 * i;
    double a = w.kkt[f[0]];
    for (int k = 0; k < f[2]; ++k) a += w.kkt[d.side_off + f[1] + k];
    w.kkt[f[0]] = a;
  }
  c.sync();
}

// ---------------------------------------------------------------------------
// the solve
// ---------------------------------------------------------------------------
struct Result { int status, iters; double f, mu, t, dw; };

// The reference's stop criterion of the point-mass classes on an agent's parameter vector (`problems/point2point.py:98-102` ->
// `vehicles/holonomic.py:145-151`, `holonomic3d.py`: |state0 - poseT| <= tol and |input0| <= tol, Euclidean norms): what ends a
// vehicle's loop in `execution/simulator.py:39-62`.  One statement for the solve kernel's stop rule and the host build's.
OMGX_HD bool stop_criterium(const double* p, int o_state, int o_input, int o_pose, int n_dim, double tol) {
  double e2 = 0.0, u2 = 0.0;
  for (int k = 0; k < n_dim; ++k) {
    const double e = p[o_state + k] - p[o_pose + k], u = p[o_input + k];
    e2 += e * e; u2 += u * u;
  }
  return sqrt(e2) <= tol && sqrt(u2) <= tol;
}

// The inertia correction at which every nonlinear variable gets at least its Gershgorin row sum g_q (w.xt, position order) under
// the cap min(dw f_q, g_q + 0.03 dw): max_q g_q / f_q, with a margin of 1 % and the smallest correction on top
template <class C>
OMGX_FN double gersh_cap(const C& c, int N, const Tables& T, const Work& w, double reg_root, double reg_leaf) {
  double g = 0.0;
  OMGX_PFOR(q, N) {
    const double wq = T.reg_w[q];
    if (wq == 1.0) g = fmax(g, w.xt[q] / reg_root);
    else if (wq == -1.0) g = fmax(g, w.xt[q] / reg_leaf);
  }
  return c.uni(1.01 * c.rmax(g) + OMGX_DW_FIRST);
}

// What the setup of a solve hands to its iteration (beside the work arrays): status 1 = go on, 3 = invalid rows / start point
struct Start { int status, warm, use_t; double mu, zt, f; };

// Setup of one solve: parameter stage, Jacobian and row values at x0, row classification and gradient-based scaling, start
// values of the multipliers and of the barrier parameter.  Runs either at the head of the solve kernel (ipm_solve) or -- for
// a whole batch at once, many workgroups per CU -- as a kernel of its own ahead of it (ipm_prepare_kernel, omgx.hip; C::prep:
// the row arrays and the Jacobian values then live in the agent's record in global memory, no KKT store, no descriptors):
// the same statements in the same order either way, the same bits.
template <class C>
OMGX_FN Start ipm_setup(const C& c, const Dims& d, const Tables& T, const Opts& o, Work& w,
                        const double* p, const double* x0, const double* lb, const double* ub,
                        const double* lam0, int prev_status, int kkt_doubles, bool describe = true) {
  const int n = d.n_var, m = d.n_con;
  Start res; res.status = 1; res.warm = 0; res.use_t = 0; res.mu = o.mu_init; res.zt = 0.0; res.f = 0.0;
  OMGX_TIC();
  if constexpr (!C::prep) {
    // (describe = false: the caller wrote the matrix descriptors -- the same for every agent -- once for its workgroup)
    if (describe) {
      Kkt K; K.bind(d, T, w.kkt);
      kkt_describe(c, d, K, w, false);      // (beside the parameter program of eval_params, whose barriers publish it)
    }
  }

  // the per-agent inputs come from HBM (~1 us each if loaded where they are first needed): all of
  // them are requested here, so that their latencies overlap with each other and with the parameter
  // stage (p is loaded by eval_params; the multipliers wait in w.ds, which is free until the assembly)
  const bool warm_in = o.warm_start && prev_status == 0;
  const double tol_c = (o.compl_tol > 0.0 && o.compl_tol < o.tol) ? o.compl_tol : o.tol;      // what the complementarity has to reach
  OMGX_PFOR(i, n) w.x[i] = x0[i];
  if constexpr (!C::prep) { if (warm_in) { OMGX_PFOR(r, m) w.ds[r] = lam0[r]; } }
  if (c.tid() == 0) w.x[n] = 1.0;
  eval_params(c, d, T, w, p);
  OMGX_TOC(PH_S_PARAMS);
  // lifted auxiliaries: whatever the caller handed in, the solve starts on their defining rows (eval_params ends with a
  // barrier: the slots are there)
  if constexpr (C::general) {
    lift_project(c, d, T, w, m, w.x);
    if (d.n_lift > 0) {      // an auxiliary without a value at the caller's point (division by a variable that is zero there): Invalid
      double bad_aux = 0.0;
      OMGX_PFOR(k, d.n_lift) if (!isfinite(w.x[n - d.n_lift + k])) bad_aux = 1.0;
      if (c.rmax(bad_aux) > 0.0) { res.status = 3; return res; }
    }
  }

  // ---- row classification, gradient-based scaling, phase-I weights -----------
  // warm start only from a converged previous solve; otherwise a cold start from x0
  const bool warm = o.warm_start && prev_status == 0;     // callers pass lam0 whenever warm_start is set
  const double kpush = warm ? o.kappa_warm : o.kappa_push;
  // unscaled Jacobian entries (one thread per entry) and row values (one thread per row) at x0 ...
  // (the unscaled entries go to the KKT store, which is idle during the setup, when that is LDS: the row classification
  // below reads every entry of its row, and with two agents per CU the Jacobian values themselves live in a slab)
  double* jtmp = (C::hbm || kkt_doubles < d.nnz_j + 1) ? w.jval : w.kkt;
  jac_entries4(c, T.ja_ell, T.ja_own, T.ja_glen, d.n_ja4, w, w.x, [&](int e, int, double v) { jtmp[e] = v; });
  if (c.tid() == 0) { jtmp[d.nnz_j] = 0.0; w.jval[d.nnz_j] = 0.0; }      // the slot padding records point at
  c.sync();
  OMGX_TOC(PH_S_JAC0);
  // ... then one thread per row: classification, gradient-based scale, phase-I weight
  // (rows in the order of the ELL table of their Jacobian entries, eight entries in flight at a time: with the Jacobian
  // values in a slab -- two agents per CU -- a row that walks its entries one by one pays a memory round trip each)
  double bad_local = 0.0, any_local = 0.0;
  OMGX_PFOR(ir, m) {
    const int r = T.row_perm[ir];
    const int Lr = T.jp_glen[ir >> 6];
    const double l = lb[r], u = ub[r];
    const double g = row_value_ell<C>(d, T, w, ir, m, w.x);      // (the row's value in the same pass: its term loads fly with the bounds and the entry list)
    const bool fl = isfinite(l), fu = isfinite(u);
    int ty = ROW_FREE;
    if (fl && fu) ty = (l == u) ? ROW_EQ : ROW_BAD;
    else if (fu) ty = ROW_UPPER;
    else if (fl) ty = ROW_LOWER;
    if (T.eq_index[r] >= 0) { if (ty != ROW_EQ && ty != ROW_FREE) ty = ROW_BAD; }
    else if (ty == ROW_EQ) ty = ROW_BAD;
    if (ty == ROW_BAD) bad_local = 1.0;
    w.rtype[r] = ty;
    double gm = 0.0;
    for (int s0 = 0; s0 < Lr; s0 += 8) {
      int32_t e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = T.jp_ell[2 * ((s0 + k) * m + ir)];
      double jv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) jv[k] = jtmp[e[k]];            // (padding: the slot that holds 0.0)
#pragma unroll
      for (int k = 0; k < 8; ++k) gm = fmax(gm, fabs(jv[k]));
    }
    double rho = (o.scale_gmax > 0.0 && gm > o.scale_gmax) ? o.scale_gmax / gm : 1.0;
    const double sg = (ty == ROW_LOWER) ? -1.0 : 1.0;
    w.rho[r] = sg * rho;                                  // signed scale: h = rho*(g - bound)
    const double bnd = (ty == ROW_LOWER || ty == ROW_EQ) ? l : (ty == ROW_UPPER ? u : 0.0);
    const double h = (ty == ROW_FREE) ? 0.0 : w.rho[r] * (g - bnd);
    w.hv[r] = h;
    double v = 0.0;
    if (ty == ROW_UPPER || ty == ROW_LOWER) {
      v = fmax(h + kpush, 0.0);
    }
    else if (ty == ROW_EQ) v = h;
    w.vv[r] = v;
    if (v != 0.0) any_local = 1.0;
  }
  bool use_t;
  {
    double rv[2] = {bad_local, any_local};
    c.template reduce_ops<1, 1>(rv);
    if (rv[0] > 0.0) { res.status = 3; return res; }
    use_t = rv[1] > 0.0;
  }
  // the Jacobian the first iteration needs is this one with the rows scaled (the objective row as it is): one thread per
  // entry, no second pass over the constraint terms at x0.  (Round 6 measured the row owners writing their scaled entries inside the
  // pass above instead: the scattered stores into the slab cost 3.4 k cycles more than this coalesced pass saves.)
  OMGX_PFOR_U4(e, d.nnz_j) {
    const int r = d.rp_packed ? (int)((uint32_t)T.je_rp[e] >> 16) : T.je_row[e];
    const double sc = (r < m) ? ((w.rtype[r] == ROW_FREE) ? 0.0 : w.rho[r]) : 1.0;
    w.jval[e] = jtmp[e] * sc;
  }
  OMGX_TOC(PH_S_CLASS);
  double mu = o.mu_init, nu = o.nu_init;
  const double t = use_t ? 1.0 : 0.0;
  double zt = 0.0, f;
  if (c.tid() == 0) w.x[n] = t;
  // start values of the multipliers -- cold: on the central path of mu_init; warm: the multipliers of the previous solve (unscaled
  // lam_g), the barrier parameter from their average complementarity -- and, in the same pass, the sums the start scalars need
  // (round 6: one pass and one reduction where there were three of each)
  {
    double sz = 0.0, cnt0 = 0.0, vz = 0.0;
    OMGX_PFOR(r, m) {
      const int ty = w.rtype[r];
      double zr = 0.0;
      if (ty == ROW_UPPER || ty == ROW_LOWER) {
        const double s_r = row_slack(w, r, t);
        if (warm) {
          // (a row with a small slack AND a vanishing multiplier is invisible to the Newton system -- Sigma = z / s -- until the
          // step runs into it: hundreds of iterations with step lengths of 1e-2 on a knot-crossing x-update; the floor keeps
          // such rows in the picture)
          zr = fmax((C::prep ? lam0[r] : w.ds[r]) / w.rho[r], fmax(OMGX_WARM_ZMIN, fmin(o.warm_z_floor * tol_c, o.warm_z_cap > 0.0 ? o.warm_z_cap * tol_c / s_r : 1e300)));
          sz += s_r * zr; cnt0 += 1.0;
        } else zr = mu / s_r;
      } else if (ty == ROW_EQ && warm) {
        zr = (C::prep ? lam0[r] : w.ds[r]) / w.rho[r];
      }
      w.z[r] = zr;
      if (ty != ROW_FREE) vz += w.vv[r] * zr;
    }
    double rv[4] = {sz, cnt0, vz, row_value_share(c, T, w, m, w.x)};
    c.template reduce_ops<0, 0, 0, 0>(rv);
    if (warm) mu = c.uni(fmin(o.mu_init, fmax(tol_c / 10.0, o.warm_mu_factor * rv[0] / fmax(1.0, rv[1]))));
    // multiplier of t >= 0: dual feasible in t (nu - v'z - zt = 0) rather than on the central path,
    // so that the first Newton step in t is O(t) instead of O(nu t^2 / mu)
    zt = c.uni(use_t ? fmax(mu / t, nu - rv[2]) : 0.0);
    f = rv[3];
  }
  OMGX_TOC(PH_S_INIT);
#if defined(OMGX_PROFILE) && !defined(OMGX_HOST_PORT)
  if (c.tid() == 0 && c.prof) c.prof[PH_SETUP] += c.prof[PH_S_DESC] + c.prof[PH_S_PARAMS] + c.prof[PH_S_JAC0] + c.prof[PH_S_CLASS] + c.prof[PH_S_INIT];
#endif
  res.warm = warm ? 1 : 0; res.use_t = use_t ? 1 : 0; res.mu = mu; res.zt = zt; res.f = f;
  return res;
}

// Record of a prepared solve (one per agent, global memory; written by ipm_prepare_kernel, read by the solve kernel): offsets
// in doubles.  sc: {status, warm, use_t, mu, zt, f, -, -}; x: [N] start point (lifted auxiliaries on their rows, x[n_var] = t);
// the row arrays; rtype bytes; the scaled Jacobian values (+ the padding slot).
struct PrepLayout { int sc, x, slots, hv, rho, vv, z, rtype, jval, total; };
OMGX_HD PrepLayout prep_layout(const Dims& d) {
  PrepLayout L; int o = 0;
  L.sc = o; o += 8;
  L.x = o; o += d.N;
  L.slots = o; o += d.n_slots;
  L.hv = o; o += d.n_con; L.rho = o; o += d.n_con; L.vv = o; o += d.n_con; L.z = o; o += d.n_con;
  L.rtype = o; o += (d.n_con + 7) / 8;
  o = (o + 1) & ~1;                       // (the Jacobian values 16-byte aligned)
  L.jval = o; o += d.nnz_j + 1;
  L.total = (o + 1) & ~1;
  return L;
}

#ifndef OMGX_HOST_PORT
// The solve kernel's side of a prepared solve: the record into the work arrays (every load independent of the others, all in
// flight together -- one memory round trip where the in-kernel setup spends ~40 dependent ones).  JAC_GLOBAL: the Jacobian
// values stay where the setup kernel wrote them (the caller points w.jval at the record); otherwise they are copied to LDS.
template <bool JAC_GLOBAL, class C>
OMGX_FN Start ipm_load_start(const C& c, const Dims& d, Work& w, const double* rec) {
  const PrepLayout L = prep_layout(d);
  OMGX_TIC();
  Start st;
  st.status = (int)rec[L.sc]; st.warm = (int)rec[L.sc + 1]; st.use_t = (int)rec[L.sc + 2];
  st.mu = rec[L.sc + 3]; st.zt = rec[L.sc + 4]; st.f = rec[L.sc + 5];
  OMGX_PFOR(i, d.N) w.x[i] = rec[L.x + i];
  OMGX_PFOR(i, d.n_slots) w.slots[i] = rec[L.slots + i];
  OMGX_PFOR(r, d.n_con) {
    const double a = rec[L.hv + r], b = rec[L.rho + r], v = rec[L.vv + r], z = rec[L.z + r];
    w.hv[r] = a; w.rho[r] = b; w.vv[r] = v; w.z[r] = z;
  }
  { double* rt = (double*)w.rtype; OMGX_PFOR(i, (d.n_con + 7) / 8) rt[i] = rec[L.rtype + i]; }
  if constexpr (!JAC_GLOBAL) { OMGX_PFOR_U4(e, d.nnz_j + 1) w.jval[e] = rec[L.jval + e]; }
  st.status = __builtin_amdgcn_readfirstlane(st.status); st.warm = __builtin_amdgcn_readfirstlane(st.warm);
  st.use_t = __builtin_amdgcn_readfirstlane(st.use_t);
  st.mu = c.uni(st.mu); st.zt = c.uni(st.zt); st.f = c.uni(st.f);
  c.sync();
  OMGX_TOC(PH_S_INIT);
#if defined(OMGX_PROFILE)
  if (c.tid() == 0 && c.prof) c.prof[PH_SETUP] += c.prof[PH_S_INIT];
#endif
  return st;
}
#endif

// The iteration of a solve whose setup is done (the work arrays hold x, the slots, the scaled Jacobian, the row arrays and the
// multipliers; `st` the scalars).
template <class C>
OMGX_FN Result ipm_iterate(const C& c, const Dims& d, const Tables& T, const Opts& o, Work& w,
                           const double* lb, const double* ub, const Start& st, int kkt_doubles, double dw_prev = 0.0) {
  const int n = d.n_var, m = d.n_con, N = d.N;
  Result res; res.status = st.status == 3 ? 3 : 1; res.iters = 0; res.f = 0; res.mu = o.mu_init; res.t = 0; res.dw = 0;
  if (st.status == 3) return res;
  Kkt K; K.bind(d, T, w.kkt);
  const bool warm = st.warm != 0, use_t = st.use_t != 0;
  const double tol_c = (o.compl_tol > 0.0 && o.compl_tol < o.tol) ? o.compl_tol : o.tol;
  const double mu_floor = tol_c / 10.0;
  // (rows are tested scaled, `viol <= tol`; with viol_tol set also unscaled, folded into the same maximum: value / |rho| <= viol_tol)
  const double viol_fold = o.viol_tol > 0.0 ? o.tol / o.viol_tol : 0.0;
  double mu = st.mu, nu = o.nu_init, zt = st.zt, f = st.f;
  double t = use_t ? 1.0 : 0.0;
  // a warm start also inherits the inertia correction the previous solve of this agent ended with
  // (first factorisation at that value instead of climbing 0, 1e-4, 1e-3, ... again)
  double dw_last = c.uni((warm && dw_prev > 0.0) ? dw_prev : 0.0), t_check = t;
  // Round 5: `hess_approx` (the reference's examples of the nonholonomic classes ask IPOPT for a limited-memory Hessian,
  // `examples/p2p_dubins.py:42`, `p2p_agv.py:43`: their authors did not trust the exact one there).  The analogue here: the
  // Lagrangian Hessian WITHOUT the curvature of the rows (objective Hessian + J' Sigma J: positive semidefinite, no inertia
  // trouble however large the multipliers of phase I), damped by `lm` -- a Levenberg-Marquardt weight driven by the accepted
  // step length: x 4 after a step cut below a quarter, / 2 after a full one.  Linear convergence (hundreds of iterations), but
  // phase I no longer drowns in an inertia correction of 1e6.  Compiled into the general instances only.
  const bool gn = C::general ? (o.hess_approx != 0) : false;
  double lm = 1.0;
  int dw_hold = dw_last > 0.0 ? 1 : 0, dw_backoff = 1;   // inertia-correction tracking (see the factorisation loop)
  // cold starts may damp the leaf (hyperplane) variables less and the root (trajectory) variables more
  // than dw (same product: the same bilinear negative curvature is covered); warm starts use dw on both
  const double reg_leaf = warm ? 1.0 : o.dw_leaf_ratio_cold, reg_root = warm ? 1.0 : 1.0 / o.dw_leaf_ratio_cold;
  int it = 0, status = 1, ls_fail = 0, full_steps = 0;
  double alpha_prev = 1.0;      // step length the previous iteration accepted
  const double nu_stall_max = warm ? OMGX_NU_MAX : 0.0;     // see the stall test in the loop

#if defined(OMGX_PROFILE) && !defined(OMGX_HOST_PORT)
  long long resid_seen_ = 0;
#endif
  for (it = 0; it <= o.max_iter; ++it) {
#ifndef OMGX_HOST_PORT
    // a solve that is still running after this many iterations is one the rest of the batch will wait for: its waves win
    // the issue arbitration against the agent that shares the CU from here on (reset by the kernel after the solve)
    if (o.prio_iter > 0 && it == o.prio_iter) __builtin_amdgcn_s_setprio(2);
#endif
    OMGX_TIC();
    // ---- Jacobian (scaled): one thread per entry; only the entries that depend on x ------------
    if (it > 0) {        // (iteration 0: left by the setup)
      jac_entries4(c, T.jv_ell, T.jv_own, T.jv_glen, d.n_jv4, w, w.x, [&](int e, int r, double sj) {
        const double sc = (r < m) ? ((w.rtype[r] == ROW_FREE) ? 0.0 : w.rho[r]) : 1.0;
        w.jval[e] = sc * sj;
      });
    }
    // per row: 1/s (-> ht) and Sigma = z/s (-> ds; stays there for the assembly)
    OMGX_PFOR(r, m) {
      const int ty = w.rtype[r];
      const bool ineq = (ty == ROW_UPPER || ty == ROW_LOWER);
      const double is = ineq ? 1.0 / row_slack(w, r, t) : 0.0;
      w.ht[r] = is; w.ds[r] = ineq ? w.z[r] * is : 0.0;
    }
    c.sync();
    OMGX_TOC(PH_JAC);
    // ---- dual residual, barrier gradient (position order), error measures -------
    // Column sums over the Jacobian: a column's entries in row order, in chunks of at most cs_cap records, one owner
    // thread per chunk (cs_own[j] .. cs_own[j + 1]: the owners of the column in slot j); the chunks are combined in
    // order (fixed-order sums):
    //   dinv <- grad f + J'z (the dual residual; w.dinv is free until the factorisation),  gbar <- grad f,
    //   xt <- J'(1/s),  sol <- J'(Sigma v) (the phase-I column of the KKT matrix; w.sol is free until the
    //   Newton system is solved, also across the retries of the inertia correction)
    // the barrier gradient grad f + mu J'(1/s) is formed once mu is settled below
    {
      const int no = d.n_cs_own;
      double* part = w.kkt;                               // [n_cs_own][3] staging (the KKT store is idle here)
      OMGX_PFOR(o, no) {
        const int L = T.cs_glen[o >> 6];
        double a_z = 0.0, a_s = 0.0, a_t = 0.0;
        for (int s0 = 0; s0 < L; s0 += 8) {               // eight {entry, row} records at a time
          int32_t e[8], r[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) { const int32_t* q = T.cs_ell + 2 * ((s0 + k) * no + o); e[k] = q[0]; r[k] = q[1]; }
          double jv[8], vz[8], vs[8], vt[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) { jv[k] = w.jval[e[k]]; vz[k] = w.z[r[k]]; vs[k] = w.ht[r[k]]; vt[k] = w.ds[r[k]] * w.vv[r[k]]; }
#pragma unroll
          for (int k = 0; k < 8; ++k) { a_z += jv[k] * vz[k]; a_s += jv[k] * vs[k]; a_t += jv[k] * vt[k]; }
        }
        part[3 * o] = a_z; part[3 * o + 1] = a_s; part[3 * o + 2] = a_t;
      }
      c.sync();
      OMGX_PFOR(j, n) {
        double a_z = 0.0, a_s = 0.0, a_t = 0.0;
        for (int k = T.cs_own[j]; k < T.cs_own[j + 1]; ++k) { const double* pp = part + 3 * k; a_z += pp[0]; a_s += pp[1]; a_t += pp[2]; }
        const int q = T.cs_col[j];
        const int eo = T.obj_ent[q];
        const double gf = eo >= 0 ? w.jval[eo] : 0.0;
        w.dinv[q] = gf + a_z; w.gbar[q] = gf; w.xt[q] = a_s; w.sol[q] = a_t;
      }
      if (c.tid() == 0) { w.gbar[N - 1] = 0.0; w.xt[N - 1] = 0.0; }
    }
    c.sync();
    OMGX_TOC(PH_LA);                                      // (profiling build: column sums of the residual phase)
    double rd_max = 0.0;
    OMGX_PFOR(q, n) rd_max = fmax(rd_max, fabs(w.dinv[q]));
    double viol = 0.0, zh = 0.0, rE_max = 0.0, rE_sum = 0.0, vz = 0.0, lam_sum = 0.0, cnt = 0.0;
    OMGX_PFOR(r, m) {
      const int ty = w.rtype[r];
      const double vmul = viol_fold > 0.0 ? fmax(1.0, viol_fold / fabs(w.rho[r])) : 1.0;
      if (ty == ROW_UPPER || ty == ROW_LOWER) {
        cnt += 1.0; lam_sum += fabs(w.z[r]); vz += w.vv[r] * w.z[r];
        viol = fmax(viol, w.hv[r] * vmul); zh = fmax(zh, fabs(w.z[r] * w.hv[r]));
      } else if (ty == ROW_EQ) {
        cnt += 1.0; lam_sum += fabs(w.z[r]); vz += w.vv[r] * w.z[r];
        viol = fmax(viol, fabs(w.hv[r]) * vmul);
        const double re = w.hv[r] - t * w.vv[r];
        rE_max = fmax(rE_max, fabs(re)); rE_sum += fabs(re);
      }
    }
    {
      double rv[8] = {rd_max, lam_sum, vz, cnt, viol, zh, rE_max, rE_sum};
      c.template reduce_ops<1, 0, 0, 0, 1, 1, 1, 0>(rv);
      rd_max = rv[0]; lam_sum = rv[1]; vz = rv[2]; cnt = rv[3]; viol = rv[4]; zh = rv[5]; rE_max = rv[6]; rE_sum = rv[7];
    }
    OMGX_TOC(PH_LS);                                      // (error measures and their reduction)
    const double sd = c.uni(fmax(OMGX_S_MAX, lam_sum / fmax(1.0, cnt)) / OMGX_S_MAX);
    const double err0 = fmax(rd_max / sd, fmax(viol, zh / sd));
    res.f = f; res.mu = mu; res.t = t; res.iters = it;
    if (err0 <= o.tol && (o.compl_tol <= 0.0 || zh <= o.compl_tol)) { status = 0; break; }
    if (it == o.max_iter) break;
    // barrier-problem error at a given mu.  Under a heavy inertia correction (concave rows with
    // multipliers ~ mu/s: the negative curvature itself scales with mu) the damped Newton method
    // crawls on the barrier subproblem, so it is solved less accurately before mu is reduced.
    const double keps = (dw_last > OMGX_DW_HEAVY) ? OMGX_KAPPA_EPS_HEAVY : OMGX_KAPPA_EPS;
    int infeasible = 0;
    for (;;) {
      double comp = 0.0;
      OMGX_PFOR(r, m) {
        const int ty = w.rtype[r];
        if (ty == ROW_UPPER || ty == ROW_LOWER) comp = fmax(comp, fabs(row_slack(w, r, t) * w.z[r] - mu));
      }
      comp = c.rmax(comp);
      if (use_t) comp = fmax(comp, fabs(t * zt - mu));
      const double rd_t = use_t ? (nu - vz - zt) : 0.0;
      const double emu = fmax(fmax(rd_max, fabs(rd_t)) / sd, fmax(rE_max, comp / sd));
      if (mu > mu_floor && emu <= keps * mu) {
        mu = c.uni(fmax(mu_floor, fmin(OMGX_KAPPA_MU * mu, pow(mu, OMGX_THETA_MU))));
        continue;
      }
      if (use_t && zt < 0.1 * nu && t > o.tol && emu <= 100.0 * OMGX_KAPPA_EPS * mu) {
        if (nu >= OMGX_NU_MAX) { infeasible = 1; break; }   // phase I stalls at t > 0: local infeasibility
        nu = c.uni(nu * 10.0); zt = c.uni(zt + 0.9 * nu);
        continue;
      }
      break;
    }
    // (version 8, constr_viol_tol: everything but the violation of the rows is converged and the barrier parameter sits on its
    // floor -- what is left is t v_i, the shift phase I still holds: the penalty weight takes t down, a factor ten at a time)
    if (use_t && viol_fold > 0.0 && !infeasible && mu <= mu_floor && nu < OMGX_NU_MAX && viol > o.tol &&
        fmax(rd_max, zh) / sd <= o.tol && (o.compl_tol <= 0.0 || zh <= o.compl_tol)) {
      nu = c.uni(nu * 10.0); zt = c.uni(zt + 0.9 * nu);
    }
    // stall test: phase I must shrink t by at least 10 % over OMGX_STALL_ITERS (20) iterations.
    // A warm-started solve (the previous solve of this agent converged, so a stall is likely
    // transient: e.g. an ADMM x-update next to a moving obstacle) first raises the penalty weight
    // and is declared locally infeasible only at nu_max; a cold solve gives up at once (an agent
    // that is locally infeasible would otherwise burn ~150 iterations in every receding-horizon step).
    if (use_t && it > 0 && it % OMGX_STALL_ITERS == 0) {
      // (t sits at mu / z_t ~ mu / nu on the central path of a feasible problem: only a t well above
      // that counts as phase I not finishing)
      if (t > fmax(o.tol, 10.0 * mu / nu) && t > 0.9 * t_check) {
        if (nu >= nu_stall_max) infeasible = 1;
        else { nu = c.uni(nu * 10.0); zt = c.uni(zt + 0.9 * nu); }
      }
      t_check = t;
    }
    if (infeasible) { status = 2; break; }
    // barrier gradient with the current mu
    OMGX_PFOR(q, n) w.gbar[q] += mu * w.xt[q];
    double vms = 0.0;
    OMGX_PFOR(r, m) if (w.rtype[r] == ROW_UPPER || w.rtype[r] == ROW_LOWER) vms += w.vv[r] * (mu / row_slack(w, r, t));
    vms = c.rsum(vms);
    const double gbar_t = c.uni(use_t ? (nu - vms - mu / t) : 0.0);
    if (c.tid() == 0) w.gbar[N - 1] = gbar_t;
    c.sync();

    OMGX_TOC(PH_LB);                                      // (barrier update loop, barrier gradient)
#if defined(OMGX_PROFILE) && !defined(OMGX_HOST_PORT)
    if (c.tid() == 0) { c.prof[PH_RESID] += c.prof[PH_LA] + c.prof[PH_LS] + c.prof[PH_LB] - resid_seen_; resid_seen_ = c.prof[PH_LA] + c.prof[PH_LS] + c.prof[PH_LB]; }
#endif
    // The equality block of the quasi-definite system carries -delta_c I.  A step leaves delta_c |y| of equality residual
    // behind, so a fixed 1e-8 is a floor under the feasibility a solve can reach: with multipliers of ~100 (an ADMM
    // x-update whose consensus weight is large) 1.3e-6, and a solve at 1e-6 never ends.  Like IPOPT the regularisation
    // follows the barrier parameter: 1e-8 mu^(1/4) (1.8e-10 at mu = 1e-7).
    const double delta_c = c.uni(OMGX_DELTA_C * sqrt(sqrt(mu)));
    // ---- assemble + factorise with inertia correction --------------------------------
    // Tracking of the inertia correction dw.  When the previous iteration needed dw > 0 the
    // (doomed) dw = 0 attempt is skipped.  A decrease dw_last/3 is attempted only every
    // dw_backoff-th iteration (doubling, up to OMGX_DW_BACKOFF_MAX, each time the decrease
    // fails; reset by a success); a failed decrease falls back to the value that worked last
    // time before escalating by OMGX_DW_INC.  Near a solution whose Lagrangian Hessian is
    // indefinite this keeps dw within [dw*, 3 dw*] instead of cycling through [dw*/3, 10 dw*/3].
    double dw; int decreasing = 0;
    if (dw_last < OMGX_DW_ZERO) dw = 0.0;
    else if (dw_hold > 0) { dw = dw_last; --dw_hold; }
    else { dw = c.uni(dw_last * OMGX_DW_DEC); decreasing = 1; }
    int failed = 0;
    // Gershgorin row sums g_q of the Lagrangian Hessian (term by term, position order; w.xt is free
    // until the line search): H + diag(g) is diagonally dominant, so no variable ever needs more
    // damping than g_q -- the inertia correction of variable q is min(dw * weight, g_q + 0.03 dw).  A
    // trajectory coefficient whose bilinear rows are all inactive (multipliers ~ mu / s) is then
    // practically undamped even while dw covers the active hyperplane rows elsewhere.
    int first_trial = 1;
    double dw_leaf_used = 0.0, dw_root_used = 0.0;      // the inertia correction the factors of this iteration carry (leaf / root variables)
    double gcap = -1.0;      // the Gershgorin guarantee of this iteration (computed when first needed)
    // (side sums of cut runs: w.dinv -- the dual residual has been read, the factorisation has not started -- or the pairs' side slots)
    double* const gside = d.kg_side_dinv ? w.dinv : w.kkt + d.side_off;
    OMGX_PFOR(q, N) w.xt[q] = 0.0;
    // weight of every row in the Lagrangian Hessian (multiplier x signed scale): one read per Hessian item instead of
    // three (w.ht -- 1/s during the residual phase -- is free until the line search)
    OMGX_PFOR(r, m) w.ht[r] = (w.rtype[r] != ROW_FREE && !gn) ? w.z[r] * w.rho[r] : 0.0;
    if (gn && dw < lm) dw = lm;
    for (;;) {
      OMGX_PFOR(i, kkt_doubles) w.kkt[i] = 0.0;
      c.sync();
      OMGX_TOC(PH_A_ZERO);
      // J' Sigma J + Lagrangian Hessian, owner-computes: every KKT address belongs to one of OMGX_NBIN
      // bins; a bin's pair records {entry a, entry b, address, row} and Hessian items are sorted by
      // address, the owner sums each run in table order and stores it (Sigma = w.ds, set with the
      // residuals).  No atomics: the same bits in every run, for every batch composition.
      // (branch-free loop bodies -- the store of a record that does not end a segment goes to a dump slot
      // behind the store -- and the records of a batch loaded first, all in flight together: left to itself
      // the compiler issues every load right before its use, behind the previous LDS store, ~1000 cycles each)
      for (int bin = c.tid(); bin < d.n_owner; bin += c.nthr()) {
        const int dump = d.dump_off + (bin & 63);       // (a slot per lane behind the store: nothing lives there)
        struct PairRec { int32_t a, b, ad, r; };
        const PairRec* recs = (const PairRec*)T.ka_rec;
        double acc = 0.0;
        for (int r0 = 0; r0 < d.ka_len; r0 += OMGX_REC_BATCH) {
          PairRec q[OMGX_REC_BATCH];
#pragma unroll
          for (int i = 0; i < OMGX_REC_BATCH; ++i) q[i] = recs[(r0 + i) * OMGX_NBIN + bin];
          double v[OMGX_REC_BATCH];
#pragma unroll
          for (int i = 0; i < OMGX_REC_BATCH; ++i) v[i] = w.ds[q[i].r] * w.jval[q[i].a] * w.jval[q[i].b];
#pragma unroll
          for (int i = 0; i < OMGX_REC_BATCH; ++i) {
            acc += v[i];
            w.kkt[q[i].ad >= 0 ? q[i].ad : dump] = acc;
            acc = q[i].ad >= 0 ? 0.0 : acc;
          }
        }
      }
      c.sync();
      OMGX_TOC(PH_A_PAIRS);
      // cut runs: the side slots are added to their address, in order
      OMGX_PFOR(i, d.n_kafix) {
        const int32_t* f = T.ka_fix + 3 * i;
        double a = w.kkt[f[0]];
        for (int k = 0; k < f[2]; ++k) a += w.kkt[d.side_off + f[1] + k];
        w.kkt[f[0]] = a;
      }
      // phase-I column: -J'(Sigma v) from the column sums, its diagonal from the rows
      double tt_acc = 0.0;
      if (use_t) {
        OMGX_PFOR(q, n) w.kkt[T.tq_addr[q]] = -w.sol[q];
        OMGX_PFOR(r, m) tt_acc += w.ds[r] * w.vv[r] * w.vv[r];
      }
      // equality rows straight into the root block
      OMGX_PFOR(i, d.n_eqe) {                                   // one thread per equality-row entry
        const int32_t* q = T.eqe3 + 3 * i;                     // {Jacobian entry, KKT address, row}
        if (w.rtype[q[2]] == ROW_EQ) w.kkt[q[1]] = w.jval[q[0]];
      }
      OMGX_PFOR(k, d.n_eq) {
        const int r = T.eq_rows[k];
        double* Rr = K.R();
        if (use_t && w.rtype[r] == ROW_EQ) Rr[tri(d.n_root + k, d.n_root - 1)] = -w.vv[r];
        Rr[tri(d.n_root + k, d.n_root + k)] = -delta_c;
      }
      c.sync();
      OMGX_TOC(PH_A_TCOL);
      for (int bin = c.tid(); bin < d.n_owner; bin += c.nthr()) {
        const int dump = d.dump_off + (bin & 63);
        hess_bin<C>(d, T, w, m, bin, dump);
        if (first_trial) {
          // Gershgorin row sums of the Hessian, term by term (no cancellation), owner = position
          double acc = 0.0;
          for (int e0 = 0; e0 < d.kg_len; e0 += OMGX_REC_BATCH) {
            HItem q[OMGX_REC_BATCH];
#pragma unroll
            for (int i = 0; i < OMGX_REC_BATCH; ++i) q[i] = T.kg_rec[(e0 + i) * OMGX_NBIN + bin];
            double g[OMGX_REC_BATCH];
#pragma unroll
            for (int i = 0; i < OMGX_REC_BATCH; ++i) {
              const int r = q[i].row < m ? q[i].row : 0;
              const double lam = (q[i].row < m) ? w.ht[r] : 1.0;
              const double xs = w.slots[q[i].slot < 0 ? 0 : q[i].slot], x3 = w.x[q[i].vthird < 0 ? 0 : q[i].vthird];
              double h = lam * q[i].coef * (q[i].slot < 0 ? 1.0 : xs) * (q[i].vthird < 0 ? 1.0 : x3);
              if constexpr (C::general) { const double x4 = w.x[q[i].vfourth < 0 ? 0 : q[i].vfourth]; h *= (q[i].vfourth < 0 ? 1.0 : x4); }
              g[i] = q[i].kind ? (h < 0.0 ? -2.0 * h : 0.0) : fabs(h);
            }
#pragma unroll
            for (int i = 0; i < OMGX_REC_BATCH; ++i) {
              acc += g[i];
              const int tg = q[i].target;
              double* dst = tg >= N ? gside + (tg - N) : (tg >= 0 ? w.xt + tg : w.kkt + dump);
              *dst = acc;
              acc = tg >= 0 ? 0.0 : acc;
            }
          }
        }
      }
      OMGX_TOC(PH_A_HESS);
      kkt_rhs(c, d, T, w, t);
      tt_acc = use_t ? c.rsum(tt_acc) : 0.0;
      c.sync();
      hess_fix(c, d, T, w);
      if (first_trial) {
        OMGX_PFOR(i, d.n_kgfix) {
          const int32_t* f = T.kg_fix + 3 * i;
          double a = w.xt[f[0]];
          for (int k = 0; k < f[2]; ++k) a += gside[f[1] + k];
          w.xt[f[0]] = a;
        }
        c.sync();
      }
      OMGX_TOC(PH_A_REST);
      // Round 5: the Gershgorin guarantee.  With dw >= g_q / f_q for every nonlinear variable (f_q: its weight) the capped
      // correction min(dw f_q, g_q + 0.03 dw) is at least g_q everywhere, H + D is diagonally dominant and the primal
      // block positive definite: no iteration needs more than gcap = max_q g_q / f_q.  The escalation ladder (x10 per failed
      // attempt) is cut there, and a correction carried over from an earlier iteration that exceeds it (> OMGX_DW_CLAMP_FROM:
      // normal operation never pays the reduction) is taken back to it -- a knot-crossing step whose multipliers blew up for
      // one iteration climbed to 6e4 and then walked down by thirds for ten iterations with g_max = 0.3 all along.  Not after
      // a failed line search: that retry WANTS the heavier, steepest-descent-like direction.
      if (first_trial && ls_fail == 0 && dw > OMGX_DW_CLAMP_FROM && !gn) {
        gcap = gersh_cap(c, N, T, w, reg_root, reg_leaf);
        if (dw > gcap) dw = gcap;
      }
      OMGX_PFOR(q, N) {
        // variables without a nonlinear term have zero rows in the Lagrangian Hessian: negative
        // curvature cannot come from them, and damping them would stall LP-like directions
        const double wq = T.reg_w[q];
        double add = dw * (wq == 1.0 ? reg_root : (wq == -1.0 ? reg_leaf : wq));
        // never (much) more than diagonal dominance needs; the small share of dw that stays keeps the
        // step of a practically free variable inside the range of the quadratic model
        if (wq == 1.0 || wq == -1.0) add = fmin(add, w.xt[q] + OMGX_DW_CAP_FLOOR * dw);
        if (q == N - 1) add += (use_t ? zt / t : 1.0) + tt_acc;
        w.kkt[T.diag_addr[q]] += add;
      }
      c.sync();
      OMGX_TOC(PH_A_DIAG);
#if defined(OMGX_PROFILE) && !defined(OMGX_HOST_PORT)
      if (c.tid() == 0) c.prof[PH_ASSEMBLE] = c.prof[PH_A_ZERO] + c.prof[PH_A_PAIRS] + c.prof[PH_A_TCOL] + c.prof[PH_A_HESS] + c.prof[PH_A_REST] + c.prof[PH_A_DIAG];
#endif
      int bad = kkt_factor(c, d, K, w);
#ifdef OMGX_COUNT_FACT
      ++omgx_dbg_nfact;
      if (first_trial) ++omgx_dbg_cnt[4];
      if (first_trial && dw == 0.0) ++omgx_dbg_cnt[5];
      if (!bad && dw == 0.0) ++omgx_dbg_cnt[6];
      if (bad && dw == 0.0) ++omgx_dbg_cnt[2];
      if (bad && decreasing) ++omgx_dbg_cnt[3];
#endif
      OMGX_TOC(PH_FACTOR);
#if defined(OMGX_HOST_PORT) && defined(OMGX_TRACE)
      { double gmx = 0.0; int qg = -1; for (int q = 0; q < N; ++q) if (w.xt[q] > gmx) { gmx = w.xt[q]; qg = q; }
        fprintf(stderr, "      factorisation: dw %.3e decreasing %d -> bad %d   (largest Gershgorin sum %.3e at x[%d])\n", dw, decreasing, bad, gmx, qg >= 0 ? T.order[qg] : -1); }
#endif
      first_trial = 0;
      if (!bad) { if (decreasing) dw_backoff = 1; dw_leaf_used = dw_root_used = dw; break; }
      if (bad == 2 && d.wave_ok && warm) {
        // The leaves are positive definite at this dw and their Schur complements are in the root, which alone
        // has the wrong inertia: raise the inertia correction of the root variables only (with positive definite
        // leaves a large enough one always works) and factorise the root again -- no reassembly, no leaf
        // factorisation (the failed attempt left the store untouched).  The leaves keep the smaller value for
        // this iteration; the next one starts from the root's value on both.  (Tried and dropped: separate
        // tracking of a leaf and a root value -- the leaf value then decays to zero, fails there and pays full
        // retries: 1.14 instead of 1.00 full factorisations per cold iteration, and four instead of two of
        // 1024 cold solves fail.)  Warm starts only: on cold solves the same rule saves 13 % of the cycles per
        // iteration but sends 2 of 1024 agents of config 2 into a phase-I stall that the plain rule avoids, and
        // their restart costs more than was saved (cold solve of the batch 69 ms instead of 15.7 ms).
        double dwr = dw;
        int leave_root = 0;
        for (;;) {
          const double dw_prev_try = dwr;
          if (decreasing) {          // back to the last value that worked, try less often
            decreasing = 0; dwr = dw_last;
            dw_backoff = dw_backoff < OMGX_DW_BACKOFF_MAX ? 2 * dw_backoff : OMGX_DW_BACKOFF_MAX;
            dw_hold = dw_backoff;
          } else {
            // (the root at the Gershgorin guarantee and still the wrong inertia: it is the LEAVES that lack damping -- positive
            // definite but nearly singular, their Schur complements swamp the root; back to the full factorisation at gcap)
            if (gcap < 0.0) gcap = gersh_cap(c, N, T, w, reg_root, reg_leaf);
            if (ls_fail == 0 && dwr >= gcap && dw < gcap) { leave_root = 1; break; }
            const double nxt = (dwr == 0.0) ? OMGX_DW_FIRST : dwr * OMGX_DW_INC;
            dwr = c.uni((ls_fail == 0 && dwr < gcap && nxt > gcap) ? gcap : nxt);
          }
          if (dwr > OMGX_DW_MAX) { failed = 1; break; }
          c.sync();
          kkt_root_before_retry(c, d, K, w);
          OMGX_PFOR(k, d.n_root) {
            const int q = d.root_off + k;
            const double wq = T.reg_w[q];
            const double f = (wq == 1.0 ? reg_root : (wq == -1.0 ? reg_leaf : wq));
            double a_new = dwr * f, a_old = dw_prev_try * f;
            if (wq == 1.0 || wq == -1.0) {
              a_new = fmin(a_new, w.xt[q] + OMGX_DW_CAP_FLOOR * dwr);
              a_old = fmin(a_old, w.xt[q] + OMGX_DW_CAP_FLOOR * dw_prev_try);
            }
            w.kkt[T.diag_addr[q]] += a_new - a_old;
          }
          c.sync();
          bad = kkt_refactor_root(c, d, K, w);
#ifdef OMGX_COUNT_FACT
          ++omgx_dbg_cnt[7];
#endif
          OMGX_TOC(PH_FACTOR);
          if (!bad) break;
        }
        if (failed) break;
        if (leave_root) { dw = c.uni(gcap); decreasing = 0; c.sync(); continue; }      // (full reassembly at the guarantee)
        dw_leaf_used = dw; dw_root_used = dwr;
        dw = dwr;                    // what the next iteration starts from
        break;
      }
      if (decreasing) {            // back to the last value that worked, try less often
        decreasing = 0; dw = dw_last;
        dw_backoff = dw_backoff < OMGX_DW_BACKOFF_MAX ? 2 * dw_backoff : OMGX_DW_BACKOFF_MAX;
        dw_hold = dw_backoff;
      } else {
        if (gcap < 0.0) gcap = gersh_cap(c, N, T, w, reg_root, reg_leaf);
        const double nxt = (dw == 0.0) ? OMGX_DW_FIRST : dw * OMGX_DW_INC;
        dw = c.uni((ls_fail == 0 && dw < gcap && nxt > gcap) ? gcap : nxt);
      }
      if (dw > OMGX_DW_MAX) { failed = 1; break; }
      c.sync();
    }
    if (failed) { status = 4; break; }
    dw_last = dw;

    // ---- Newton step -------------------------------------------------------------
    kkt_solve(c, d, K, w, w.sol);
    bool refined = false;      // w.sol carries refinement terms (their sum parked in the store's right-hand-side slots)
    // Iterative refinement of a regularised step (o.refine, off by default; round 6).  With an inertia correction D the factors are
    // those of K + D and the step s1 = -(K + D)^-1 r is a proximal step: in the directions the Lagrangian Hessian leaves nearly flat
    // -- hyperplanes sliding along an inactive face, the LP-like directions of the L1 objective -- it covers lambda / (lambda + delta)
    // of the way, and the dual residual it leaves, D s1, is the regularisation term itself (1.1-1.6e-3 against tol 1e-3 on a typical
    // warm step: why the slowest agent of a step needs three iterations).  Since K s1 = -r - D s1, the step of the unregularised
    // system is s1 + s2 + ... with (K + D) s_{k+1} = D s_k; ONE term is added -- one more solve with the factors of the iteration
    // (kkt_solve2 with the equality multipliers), no Hessian product; both s1 and s1 + s2 are descent directions of the regularised
    // model (s2' g = -g' M^-1 D M^-1 g <= 0).  More terms were measured: stragglers move around, nothing gained.  Rules, each from a
    // measured failure:  the term is taken only if it is no longer than the step (where K is indefinite along the step the series
    // grows);  from the second iteration of a solve on (nine of ten warm solves end after one iteration and pay nothing);  not after
    // an iteration that accepted less than a tenth of its step (there the line search holds the solve back and a longer direction is
    // cut further: a formation x-update in a phase-I crawl, 212 iterations, ran into the cap with refined steps);  in cold solves only
    // once phase I is over (`examples/revolving_door.py` from the reference's guess: Infeasible_Detected after 20 iterations with
    // refined phase-I steps, 46 iterations without);  no lifted auxiliaries (trial points projected onto defining rows: a toy
    // problem ended in Numerical_Failure);  wave-path templates with the exact Hessian (a second solve of the spill classes runs out
    // of their slab);  and the refined step stands only if the FIRST trial of the line search accepts it as it is -- else the plain
    // step takes over with its own boundary rule and line search (the pass loop below; the term waits in the store's right-hand-side
    // slots).  Host build, three fresh seeds x (cold + 110 updates of 1024 agents): cold 21.6 -> 20.8 iterations, warm 1.208 -> 1.192,
    // slowest agent per update summed 562 -> 447, solves with more than five iterations 501 -> 308, nothing unsolved either way; tight
    // tolerances gain most (tol 1e-6: 9.5 -> 6.9 iterations per warm solve, unsolved steps 24 -> 7 of 46 k).  On the device the second
    // solve and the repeated step phase cost more than the saved iterations give back where solves are short: headline 2.26 -> 2.19 M
    // solves/s at tol 1e-3, +22 % / +52 % at 1e-4 / 1e-6 (profiles/r06_refine_ab.txt).  Hence an option.  The term lives in w.ht
    // (free between the factorisation and the line search).
#ifndef OMGX_REFINE_ALPHA
#define OMGX_REFINE_ALPHA 0.1      // the refinement is skipped after an iteration whose accepted step length was below this
#endif
#ifndef OMGX_REFINE_PHASE2_ONLY
#define OMGX_REFINE_PHASE2_ONLY 0
#endif
#ifndef OMGX_REFINE_TMAX
#define OMGX_REFINE_TMAX 1e-4     // cold solves: no refinement while phase I is under way (t above this)
#endif
    if constexpr (C::refine)
    if (o.refine > 0 && it >= 1 && alpha_prev >= OMGX_REFINE_ALPHA && ((warm && !OMGX_REFINE_PHASE2_ONLY) || !use_t || t <= OMGX_REFINE_TMAX) && d.wave_ok && d.n_lift == 0 && !gn && dw_leaf_used > 0.0) {
      const BMat* Ms2 = (const BMat*)w.col;
      const int rb2 = Ms2[d.n_leaf].pad_;
      double n1 = 0.0;
      OMGX_PFOR(q, N) {
        const double wq = T.reg_w[q];
        const double dq = q >= d.root_off ? dw_root_used : dw_leaf_used;
        double add = dq * (wq == 1.0 ? reg_root : (wq == -1.0 ? reg_leaf : wq));
        if (wq == 1.0 || wq == -1.0) add = fmin(add, w.xt[q] + OMGX_DW_CAP_FLOOR * dq);      // (the assembly's statement)
        w.kkt[kkt_rhs_slot<C>(d, Ms2, q)] = add * w.sol[q];
        n1 = fmax(n1, fabs(w.sol[q]));
      }
      OMGX_PFOR(k, d.n_eq) w.kkt[rb2 + tri(d.nr, d.n_root + k)] = 0.0;
      c.sync();
      kkt_solve2(c, d, K, w, w.ht, true);
      double n2 = 0.0;
      OMGX_PFOR(q, N) n2 = fmax(n2, fabs(w.ht[q]));
      double rv2[2] = {n1, n2};
      c.template reduce_ops<1, 1>(rv2);
      if (rv2[1] <= rv2[0]) {
        // (the term also waits in the right-hand-side slots of the store -- free until a second-order correction solves again --
        // in case the first trial of the line search sends the iteration back to the plain step, below)
        OMGX_PFOR(q, N) { w.sol[q] += w.ht[q]; w.kkt[kkt_rhs_slot<C>(d, Ms2, q)] = w.ht[q]; }
        OMGX_PFOR(k, d.n_eq) { w.sol[N + k] += w.ht[N + k]; w.kkt[rb2 + tri(d.nr, d.n_root + k)] = w.ht[N + k]; }
        refined = true;
      }
#if defined(OMGX_HOST_PORT) && defined(OMGX_TRACE)
      fprintf(stderr, "      refinement: |s1| %.3e |s2| %.3e -> %s\n", rv2[0], rv2[1], refined ? "taken" : "not taken");
#endif
      c.sync();
    }
    OMGX_TOC(PH_SOLVE);
    if (!use_t && c.tid() == 0) w.sol[N - 1] = 0.0;
    c.sync();
    // (what the update below and the traces read of the step and its line search; two passes at most: a refined step whose first
    // trial is not accepted gives way to the plain step of the same factors, and the pass runs again as if there had been no refinement)
    double dt = 0.0, gdx = 0.0, a_p = 0.0, a_d = 0.0, dzt = 0.0, phi0 = 0.0, dphi = 0.0, alpha = 0.0, ft = f, tt = t;
    int ok = 0;
    const double tau = c.uni(fmax(OMGX_TAU_MIN, 1.0 - mu));
    for (;;) {
    bool back_to_plain = false;
    dt = c.uni(w.sol[N - 1]);
    // (the primal boundary step is collected up to OMGX_EXPAND_MAX: a heavily regularised step may be lengthened, below)
    double ap_l = OMGX_EXPAND_MAX, ad_l = 1.0, ymax = 0.0, ysum = 0.0;
    gdx = 0.0;
    OMGX_PFOR(ir, m) {
      const int r = T.row_perm[ir];
      const int ty = w.rtype[r];
      if (ty == ROW_UPPER || ty == ROW_LOWER) {
        double jd = 0.0;
        {
          const int L = T.jp_glen[ir >> 6];
          for (int s0 = 0; s0 < L; s0 += 8) {
            int32_t e[8], ps[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int32_t* q = T.jp_ell + 2 * ((s0 + k) * m + ir); e[k] = q[0]; ps[k] = q[1]; }
            double jv[8], dx[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { jv[k] = w.jval[e[k]]; dx[k] = w.sol[ps[k]]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) jd += jv[k] * dx[k];
          }
        }
        const double dsr = -(jd - w.vv[r] * dt);
        w.ds[r] = dsr;
        const double sr = row_slack(w, r, t);
        const double dzr = mu / sr - w.z[r] - (w.z[r] / sr) * dsr;
        if (dsr < 0.0) ap_l = fmin(ap_l, -tau * sr / dsr);
        if (dzr < 0.0) ad_l = fmin(ad_l, -tau * w.z[r] / dzr);
      } else if (ty == ROW_EQ) {
        ymax = fmax(ymax, fabs(w.sol[N + T.eq_index[r]]));
        ysum += fabs(w.sol[N + T.eq_index[r]]);
      }
    }
    OMGX_PFOR(q, N) gdx += w.gbar[q] * w.sol[q];
    double lns = 0.0;
    OMGX_PFOR(r, m) if (w.rtype[r] == ROW_UPPER || w.rtype[r] == ROW_LOWER) lns += log(row_slack(w, r, t));
    {
      double rv[6] = {ap_l, ad_l, ymax, gdx, lns, ysum};
      c.template reduce_ops<2, 2, 1, 0, 0, 0>(rv);
      a_p = rv[0]; a_d = rv[1]; ymax = rv[2]; gdx = rv[3]; lns = rv[4]; ysum = rv[5];
    }
    dzt = 0.0;
    if (use_t) {
      dzt = c.uni(mu / t - zt - (zt / t) * dt);
      if (dt < 0.0) a_p = fmin(a_p, -tau * t / dt);
      if (dzt < 0.0) a_d = fmin(a_d, -tau * zt / dzt);
      a_p = c.uni(a_p); a_d = c.uni(a_d);
    }
    const double nuE = c.uni(2.0 * fmax(1.0, ymax));
    // The equality block carries -delta_c I (quasi-definite system): the linearised equality residual after the full
    // step is delta_c * y_new, not zero -- the penalty term can only promise the difference.  (Without this the
    // Armijo test asks, at the end of a tight solve, for a decrease of nuE * 2e-8 that no step can deliver, the
    // step length collapses and the solve stalls a factor 1.2 above a tolerance of 1e-6.)  Round 4: the merit function
    // itself counts the equality residual only above that floor.  Once the rows are satisfied to 1e-13 every step of the
    // regularised system puts delta_c * |y| ~ 1e-8 of residual back: the penalty term then GROWS by nuE * 4e-8 per unit step
    // against a predicted decrease of 1e-11, every trial is an ascent step and the solve ends in Numerical_Failure (seen on
    // the quartic free-end-time problem and on 1e-6 formation x-updates).
    const double floorE = c.uni(delta_c * ysum);
    phi0 = c.uni(f + nu * t - mu * lns - (use_t ? mu * log(t) : 0.0) + nuE * fmax(0.0, rE_sum - floorE));
    dphi = c.uni(gdx - nuE * fmax(0.0, rE_sum - floorE));

    OMGX_TOC(PH_STEP);
    // ---- Armijo backtracking on the barrier function, iterate stays strictly feasible ----
    // A step of the regularised system (inertia correction dw > 0) is a proximal step: in the directions the Lagrangian
    // Hessian leaves flat -- hyperplanes that may slide along an inactive face, the released leading coefficient -- it
    // moves by (gradient / dw) only, and a solve near its solution crawls towards the tolerance at 1 % per iteration (observed:
    // 112 iterations at dw = 29 on a knot-crossing step, err 3e-3 -> 1e-3).  Such a step is offered LONGER first: the
    // largest power of two times the Newton step that the boundary allows, up to OMGX_EXPAND_MAX, then the usual halving;
    // the Armijo test on the barrier function decides.  Only in that regime: after OMGX_EXPAND_FROM iterations in a row
    // whose full step was accepted at once; undamped Newton steps (dw = 0) are never lengthened.
    const double a_bnd = a_p;
    a_p = c.uni(fmin(a_bnd, 1.0));
    const bool phi_noise = fabs(a_p * dphi) <= OMGX_PHI_NOISE * (1.0 + fabs(phi0));
    if (!phi_noise && dw_last > OMGX_EXPAND_DW && full_steps >= OMGX_EXPAND_FROM) {
      double ex = 1.0;
      while (2.0 * ex <= a_bnd && 2.0 * ex <= OMGX_EXPAND_MAX) ex *= 2.0;
      a_p = c.uni(fmin(a_bnd, ex));
    }
    alpha = a_p; ft = f; tt = t; ok = 0;
    // Second-order correction (o.max_soc; templates on the wave path): when the first trial is rejected, the rows have
    // moved by e = (s + alpha ds) - s(x + alpha dx) more than their linearisation said (the bilinear hyperplane rows: a term
    // of second order in the step, which is what cuts a step along an active face to a few per cent).  One more solve
    // with the factors of this iteration, K d_c = -[J' Sigma e; e_E], gives the step alpha d + d_c that accounts for it;
    // it is offered once, before the halving starts, and accepted by the same Armijo test (IPOPT does the same inside
    // its filter line search).  The correction lives in w.gbar (free once the directional derivative is formed).
#ifndef OMGX_SOC_COMPILED
#define OMGX_SOC_COMPILED 1
#endif
    // (wave path: kkt_solve2_wave; every other template: kkt_solve2's blocked form -- round 4: the spill classes gain most,
    // 39 -> 33 cold iterations on the 3-D class, and a tube of quartic range rows 717 -> 149)
    int soc = (OMGX_SOC_COMPILED && o.max_soc > 0) ? 0 : 2;      // 0: not tried yet, 1: the trial under way is the corrected one, 2: done
    int soc_rounds = 0;                                          // corrections computed for this step (at most o.max_soc)
    const bool soc_levels = OMGX_SOC_COMPILED && o.max_soc > 0 && d.wave_ok && !gn;      // the correction at every step length a row rejects (below)
    for (int bt = 0; bt < OMGX_MAX_BACKTRACK; ++bt) {
      if (soc == 1 && soc_rounds > 1) { OMGX_PFOR(q, N) { const int v = T.order[q]; w.xt[v] += w.gbar[q]; } }      // (a further correction on top of the corrected trial)
      else if (soc == 1) { OMGX_PFOR(q, N) { const int v = T.order[q]; w.xt[v] = w.x[v] + (alpha * w.sol[q] + w.gbar[q]); } }
      else { OMGX_PFOR(q, N) { const int v = T.order[q]; w.xt[v] = w.x[v] + alpha * w.sol[q]; } }
      c.sync();
      if constexpr (C::general) lift_project(c, d, T, w, m, w.xt);
      tt = use_t ? c.uni(w.xt[n]) : 0.0;
      // row values at the trial point: one thread per row (long rows first), terms in table order
      double smin = 1e300, lnst = 0.0, rEt = 0.0;
      OMGX_PFOR(i, m) {
        const int r = T.row_perm[i];
        const int ty = w.rtype[r];
        const double gv = row_value_ell<C>(d, T, w, i, m, w.xt);      // (also for free rows: the loop bound is per wave)
        if (ty == ROW_FREE) { w.ht[r] = 0.0; continue; }
        const double h = w.rho[r] * (gv - ((ty == ROW_LOWER || ty == ROW_EQ) ? lb[r] : ub[r]));
        w.ht[r] = h;
        if (ty == ROW_EQ) rEt += fabs(h - tt * w.vv[r]);
        else {
          // (the fraction-to-boundary rule on the slack the row really has at the trial point, not only on its linear
          // prediction: along a curved row a trial may keep a billionth of the slack, and with a small barrier parameter
          // the merit function does not mind -- the next steps then start on the boundary)
          const double st = tt * w.vv[r] - h; smin = fmin(smin, st - OMGX_FTB_ACTUAL * (1.0 - tau) * row_slack(w, r, t)); if (st > 0.0) lnst += log(st);
        }
      }
      OMGX_TOC(PH_L_TERMS);
      {
        double rv[4] = {smin, lnst, rEt, row_value_share(c, T, w, m, w.xt)};
        c.template reduce_ops<2, 0, 0, 0>(rv);
        smin = rv[0]; lnst = rv[1]; rEt = rv[2]; ft = rv[3];
      }
#ifdef OMGX_COUNT_FACT
      ++omgx_dbg_cnt[8];
#endif
#if defined(OMGX_HOST_PORT) && defined(OMGX_TRACE_LS)
      {
        int rmin = -1; double sm = 1e300;
        for (int r = 0; r < m; ++r) if (w.rtype[r] == ROW_UPPER || w.rtype[r] == ROW_LOWER) { const double st = tt * w.vv[r] - w.ht[r]; if (st < sm) { sm = st; rmin = r; } }
        const double phit_ = ft + nu * tt - mu * lnst - (use_t ? mu * log(tt) : 0.0) + nuE * fmax(0.0, rEt - floorE);
        fprintf(stderr, "        bt %d soc %d alpha %.3e smin %.3e (row %d: s0 %.3e ds %.3e z %.3e) tt %.3e dphi_pred %.3e dphi_act %.3e\n", bt, soc, alpha, sm, rmin,
                rmin >= 0 ? row_slack(w, rmin, t) : 0.0, rmin >= 0 ? w.ds[rmin] : 0.0, rmin >= 0 ? w.z[rmin] : 0.0, tt, OMGX_ETA * alpha * dphi, phit_ - phi0);
      }
#endif
      if (smin > 0.0 && (!use_t || tt > 0.0)) {
        const double phit = ft + nu * tt - mu * lnst - (use_t ? mu * log(tt) : 0.0) + nuE * fmax(0.0, rEt - floorE);
        // (near the solution the decrease a Newton step predicts, ~ error^2, drops below what the merit function can
        // resolve -- its value is a sum of ~n_con terms of size 1 -- and the Armijo test then compares rounding
        // noise: such a step is taken as it is, like IPOPT's tiny-step rule; the error test decides about the rest)
        if (phi_noise || phit <= phi0 + OMGX_ETA * alpha * dphi || phit - phi0 <= 10.0 * 2.220446049250313e-16 * fabs(phi0)) { ok = 1; break; }
      }
      if constexpr (C::refine)
      if (refined && bt == 0) {
        // the refined step is taken only where its first trial is accepted as it stands; else the plain step of the same factors,
        // with its own boundary rule, directional derivative and line search (the term comes back out of the right-hand-side slots)
        const BMat* Ms2 = (const BMat*)w.col;
        const int rb2 = Ms2[d.n_leaf].pad_;
        c.sync();
        OMGX_PFOR(q, N) w.sol[q] -= w.kkt[kkt_rhs_slot<C>(d, Ms2, q)];
        OMGX_PFOR(k, d.n_eq) w.sol[N + k] -= w.kkt[rb2 + tri(d.nr, d.n_root + k)];
        if (!use_t && c.tid() == 0) w.sol[N - 1] = 0.0;
        c.sync();
        refined = false; back_to_plain = true;
#if defined(OMGX_HOST_PORT) && defined(OMGX_TRACE)
        fprintf(stderr, "      refined step rejected at its first trial (alpha %.3e smin %.3e): back to the plain step\n", alpha, smin);
#endif
        break;
      }
      // (round 6, max_soc > 1 -- IPOPT's max_soc is a count, default 4: the corrected trial failed too: one more correction from where
      // it landed while its rows still violate -- a step along a curved row whose slack is tiny needs the correction to be exact to
      // that slack --, else plain backtracking from here.  Holonomic3D cold solves: 36.3 -> 33.2 (2 rounds) -> 32.0 (4) iterations)
      if (soc == 1 && soc_rounds < o.max_soc && !(smin > 0.0)) soc = 0;
      if (soc == 1) {
        soc = 2;
        // Round 6: the correction is offered again at the shorter step when it was a ROW, not the merit function, that rejected the
        // trial.  A hyperplane normal glued to its norm constraint (slack 6e-8) has to move 2e-2 along the sphere: the linear model
        // loses 3.4e-4 at the full step, one correction leaves 5e-5 -- but at a quarter of the step the loss is 2e-5 and ITS correction
        // leaves less than the slack.  With the correction at the first trial only, that solve took ninety iterations at alpha = 2^-8
        // (knot-crossing step at IPOPT's tolerances: slowest agent 112 -> 16 iterations, tol 1e-4: 83 -> 20; nothing got slower)
        // (templates on the wave path -- the second solve is a substitution in registers there; the spill classes pay a blocked solve out of
        // their slab per correction and keep the first-trial rule, as does the damped-Hessian mode whose weight follows the accepted step)
        if (soc_levels && !(smin > 0.0)) { soc = 0; soc_rounds = 0; }
        alpha = c.uni(alpha * 0.5);
        c.sync(); continue;
      }
      if (soc == 0) {
        soc = 1; ++soc_rounds;
#ifdef OMGX_COUNT_FACT
        ++omgx_dbg_cnt[9];
#endif
        // per row: Sigma_r e_r (inequality rows) -> w.ht; the equality rows' part of the right-hand side goes straight to the store
        const BMat* Ms = (const BMat*)w.col;
        const int rbase = Ms[d.n_leaf].pad_, nr = d.nr;
        double vte = 0.0;
        OMGX_PFOR(r, m) {
          const int ty = w.rtype[r];
          double v = 0.0;
          if (ty == ROW_UPPER || ty == ROW_LOWER) {
            const double s0 = row_slack(w, r, t);
            const double e = (s0 + alpha * w.ds[r]) - (tt * w.vv[r] - w.ht[r]);
            v = (w.z[r] / s0) * e;
            vte += w.vv[r] * v;
          } else if (ty == ROW_EQ) {
            // what the row is off its linear prediction (1 - alpha) r_E (zero for the linear initial / terminal conditions)
            w.kkt[rbase + tri(nr, d.n_root + T.eq_index[r])] = -((w.ht[r] - tt * w.vv[r]) - (1.0 - alpha) * (w.hv[r] - t * w.vv[r]));
          }
          w.ht[r] = v;
        }
        OMGX_PFOR(k, d.n_eq) { const int r = T.eq_rows[k]; if (w.rtype[r] != ROW_EQ) w.kkt[rbase + tri(nr, d.n_root + k)] = 0.0; }
        vte = c.rsum(vte);
        // -J' (Sigma e): one thread per column over its chunks in order (the records of the column sums; the KKT store
        // holds the factors: no staging area for partial sums here)
        {
          const int no = d.n_cs_own;
          OMGX_PFOR(j, n) {
            double acc = 0.0;
            for (int ow = T.cs_own[j]; ow < T.cs_own[j + 1]; ++ow) {
              const int L = T.cs_glen[ow >> 6];
              for (int s0 = 0; s0 < L; s0 += 8) {
                int32_t e[8], r[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { const int32_t* q = T.cs_ell + 2 * ((s0 + i) * no + ow); e[i] = q[0]; r[i] = q[1]; }
                double jv[8], ve[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { jv[i] = w.jval[e[i]]; ve[i] = w.ht[r[i]]; }
#pragma unroll
                for (int i = 0; i < 8; ++i) acc += jv[i] * ve[i];
              }
            }
            w.kkt[kkt_rhs_slot<C>(d, Ms, T.cs_col[j])] = -acc;
          }
          if (c.tid() == 0) w.kkt[kkt_rhs_slot<C>(d, Ms, N - 1)] = use_t ? vte : 0.0;
        }
        c.sync();
        kkt_solve2(c, d, K, w, w.gbar);
        if (!use_t && c.tid() == 0) w.gbar[N - 1] = 0.0;
        c.sync();
        continue;
      }
      if (soc == 2 && soc_levels && !(smin > 0.0)) { soc = 0; soc_rounds = 0; }      // (see above: a row rejected this step length too)
      alpha = c.uni(alpha * 0.5);
      c.sync();
    }
    if (!back_to_plain) break;
    }
    OMGX_TOC(PH_L_ROWS);
#if defined(OMGX_PROFILE) && !defined(OMGX_HOST_PORT)
    if (c.tid() == 0) c.prof[PH_LINESEARCH] = c.prof[PH_L_TERMS] + c.prof[PH_L_ROWS];
#endif
#if !defined(OMGX_HOST_PORT) && defined(OMGX_TRACE_DEV)
    if (blockIdx.x == 0 && threadIdx.x == 0)
      printf("it %3d mu %.2e t %.3e nu %.1e zt %.2e err %.3e rd %.2e viol %.2e zh %.2e sd %.2e dw %.2e alpha %.2e ok %d f %.10e dphi %.3e phi0 %.6e a_p %.2e gdx %.3e\n", it, mu, t, nu, zt, err0, rd_max, viol, zh, sd, dw_last, alpha, ok, f, dphi, phi0, a_p, gdx);
#endif
#if defined(OMGX_HOST_PORT) && defined(OMGX_TRACE)
    {
      int rb = -1; double best = 1e300;
      for (int r = 0; r < m; ++r) if ((w.rtype[r] == ROW_UPPER || w.rtype[r] == ROW_LOWER) && w.ds[r] < 0.0) { const double q_ = -row_slack(w, r, t) / w.ds[r]; if (q_ < best) { best = q_; rb = r; } }
      if (rb >= 0) fprintf(stderr, "      blocking row %d ratio %.3e s %.3e ds %.3e z %.3e vv %.3e | dt %.3e t %.3e\n", rb, best, row_slack(w, rb, t), w.ds[rb], w.z[rb], w.vv[rb], dt, t);
    }
    {
      int q1 = -1, q2 = -1; double b1 = 0.0, b2 = 0.0;      // the two largest components of the Newton step (variable index, value)
      for (int q = 0; q < N; ++q) { const double a_ = fabs(w.sol[q]); if (a_ > b1) { b2 = b1; q2 = q1; b1 = a_; q1 = q; } else if (a_ > b2) { b2 = a_; q2 = q; } }
      fprintf(stderr, "      largest step components: x[%d] %.3e, x[%d] %.3e\n", q1 >= 0 ? T.order[q1] : -1, q1 >= 0 ? w.sol[q1] : 0.0, q2 >= 0 ? T.order[q2] : -1, q2 >= 0 ? w.sol[q2] : 0.0);
    }
    fprintf(stderr, "it %3d mu %.2e t %.3e nu %.1e zt %.2e err %.2e (rd %.2e viol %.2e zh %.2e sd %.1e) dw %.2e alpha %.2e rE %.2e f %.4e\n", it, mu, t, nu, zt, err0, rd_max, viol, zh, sd, dw_last, alpha, rE_sum, f);
#endif
    if (!ok) {
      // No acceptable step along this direction: at tight tolerances the decrease the Armijo test asks for sinks
      // into the rounding noise of the merit function (which lanes add what first decides which agent it hits).
      // Before giving up, take the iteration again with a heavier inertia correction -- a direction closer to
      // steepest descent of the barrier function -- up to OMGX_LS_RETRY times in a row.
      if (ls_fail < OMGX_LS_RETRY) {
        ++ls_fail;
        dw_last = c.uni(fmax(dw_last, OMGX_DW_FIRST) * OMGX_LS_RETRY_DW);
        dw_hold = 2; dw_backoff = OMGX_DW_BACKOFF_MAX;
        c.sync();
        continue;
      }
      status = 4; break;
    }
    if (gn) { if (alpha < 0.25) lm = fmin(lm * 4.0, 1e8); else if (alpha >= 1.0) lm = fmax(lm * 0.5, 1e-6); }
    ls_fail = 0;
    full_steps = (alpha >= 1.0 && alpha == a_p) ? full_steps + 1 : 0;      // (accepted at the first trial, not cut by the boundary)
    alpha_prev = alpha;
    // ---- accept --------------------------------------------------------------------
    c.sync();
    OMGX_PFOR(q, N) w.x[q] = w.xt[q];
    const double t_old = t;
    t = tt; f = ft;
    OMGX_PFOR(r, m) {
      const int ty = w.rtype[r];
      const double s_old = row_slack(w, r, t_old);
      w.hv[r] = w.ht[r];
      if (ty == ROW_UPPER || ty == ROW_LOWER) {
        const double dzr = mu / s_old - w.z[r] - (w.z[r] / s_old) * w.ds[r];
        const double sn = t * w.vv[r] - w.ht[r];
        // warm starts take the dual step component-wise (full Newton step, each multiplier clipped by
        // its own fraction-to-boundary rule): one multiplier on its way to zero -- a row the moving
        // horizon releases -- does not scale down the step of all the others
        double zn = warm ? fmax(w.z[r] + dzr, (1.0 - tau) * w.z[r]) : w.z[r] + a_d * dzr;
        zn = fmin(fmax(zn, mu / (OMGX_KAPPA_SIGMA * sn)), OMGX_KAPPA_SIGMA * mu / sn);
        w.z[r] = zn;
      } else if (ty == ROW_EQ) {
        const double yn = w.sol[N + T.eq_index[r]];
        w.z[r] = w.z[r] + fmin(alpha, 1.0) * (yn - w.z[r]);      // (a lengthened primal step: the multipliers take the Newton step)
      }
    }
    if (use_t) {
      zt = warm ? fmax(zt + dzt, (1.0 - tau) * zt) : zt + a_d * dzt;
      zt = c.uni(fmin(fmax(zt, mu / (OMGX_KAPPA_SIGMA * t)), OMGX_KAPPA_SIGMA * mu / t));
    }
    c.sync();
    OMGX_TOC(PH_UPDATE);
  }
  res.status = status; res.iters = it > o.max_iter ? o.max_iter : it; res.f = f; res.mu = mu; res.t = t;
  res.dw = dw_last * reg_root;      // handed to the next (warm, symmetric) solve: the damping the root block had
  return res;
}

// One solve: setup, then the iteration (the head of the solve kernel when no prepared record is handed in; the host build).
template <class C>
OMGX_FN Result ipm_solve(const C& c, const Dims& d, const Tables& T, const Opts& o, Work& w,
                         const double* p, const double* x0, const double* lb, const double* ub,
                         const double* lam0, int prev_status, int kkt_doubles, double dw_prev = 0.0) {
  const Start st = ipm_setup(c, d, T, o, w, p, x0, lb, ub, lam0, prev_status, kkt_doubles);
  return ipm_iterate(c, d, T, o, w, lb, ub, st, kkt_doubles, dw_prev);
}

// Verification entry: what the tables of the solve evaluate at a given point, by the device code of
// the solve itself -- parameter stage, Jacobian items, row terms, the Hessian items of the assembly pass -- without any
// scaling: w.hv <- g(x, p) (unscaled row values), *f <- objective, w.jval <- every Jacobian entry (objective row
// included), the KKT store <- the Hessian of f + lam' g at its addresses (nothing else in it).
template <class C>
OMGX_FN void ipm_eval(const C& c, const Dims& d, const Tables& T, Work& w, const double* p, const double* x,
                      const double* lam, int kkt_doubles, double* f_out) {
  const int n = d.n_var, m = d.n_con;
  Kkt K; K.bind(d, T, w.kkt);
  kkt_describe(c, d, K, w);
  OMGX_PFOR(i, n) w.x[i] = x[i];
  if (c.tid() == 0) w.x[n] = 0.0;
  eval_params(c, d, T, w, p);
  jac_entries4(c, T.ja_ell, T.ja_own, T.ja_glen, d.n_ja4, w, w.x, [&](int e, int, double v) { w.jval[e] = v; });
  OMGX_PFOR(i, m) { const int r = T.row_perm[i]; w.hv[r] = row_value_ell<C>(d, T, w, i, m, w.x); }
  OMGX_PFOR(r, m) w.ht[r] = lam[r];
  OMGX_PFOR(i, kkt_doubles) w.kkt[i] = 0.0;
  const double f = c.rsum(row_value_share(c, T, w, m, w.x));
  if (c.tid() == 0) *f_out = f;
  c.sync();
  for (int bin = c.tid(); bin < d.n_owner; bin += c.nthr()) hess_bin<C>(d, T, w, m, bin, d.dump_off + (bin & 63));
  c.sync();
  hess_fix(c, d, T, w);
}

}  // namespace omgx
