// omgx.hip -- libomgx.so: C ABI (include/omgx.h) + gfx950 kernels.
//
// Kernels (all hand-written HIP for CDNA4, wave64):
//   ipm_solve_kernel   one 512-thread workgroup (8 waves) per agent, whole interior-point
//                      solve with every per-agent array resident in LDS
//                      (<= 160 KiB / CU); only p, x0, bounds are read from HBM and
//                      x, lam_g, status written back.
//   sample_kernel      post-solve trajectory sampling (reference
//                      `vehicles/vehicle.py:250-300`, `spline_extra.py:406-410`,
//                      C++ `Vehicle::sampleSplines` Vehicle.cpp:112-129): the
//                      HBM-write-bound stage, coalesced along the sample index.
//   shift_kernel       warm-start shift T*coeffs (`spline_extra.py:165-191`).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/omgx.h"
#include "omgx_core.h"
#include "omgx_plan.h"

namespace {

thread_local std::string g_err;

#define HIPCHK(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      g_err = std::string(#call) + ": " + hipGetErrorString(e_);                  \
      return OMGX_E_HIP;                                                          \
    }                                                                             \
  } while (0)

constexpr int kThreads = 512;
constexpr int kLdsLimit = 160 * 1024;
constexpr int kLdsHalf = kLdsLimit / 2 - 512;      // two workgroups per CU (their static LDS and the allocation granule taken off)

}  // namespace

// out[b, o, k, i] = d^o/dt^o spline_k(t0[b] + i*dt); one block = (agent, 1024-sample chunk).
//
// Bound: HBM writes (8 n_der n_spl bytes per sample).  The de Boor recursion per sample (about 20
// fp64 divisions for 3 derivative orders of 2 cubic splines) kept the first version at 20 % of the HBM
// roofline, so the recursion is run once per knot span instead of once per sample: every block turns
// its agent's splines into local power series  p_{o,k,j}(h) = sum_m S_k^{(o+m)}(k_j+) / m!  h^m  around
// the left knot of each span j (a few dozen small de Boor evaluations, spread over the block) and a
// sample is a span look-up plus one Horner evaluation per output value.
struct KnotArg { double k[40]; };
#define OMGX_SAMPLE_CHUNK 1024      // samples per block: the per-block set-up is amortised over 4 samples per thread

// What `Vehicle.store` extracts from a solution (reference `vehicles/vehicle.py:250-300` ->
// `splines2signals`, e.g. `vehicles/holonomic.py:116-124`): derivative orders 0 .. n_der-1 of the n_spl
// splines on a time grid, order o scaled by inv_T^o (time derivatives), and optionally the speed
// v_tot = |first derivative|.  Passed by value to the kernels; out == nullptr: nothing to do.
// `omgx_admm_center_ex` fused behind the solve (omgx_batch_set_center): x_i[b] = shared coefficients of the solution + the
// agent's relative position, and its published copy
struct CenterArgs {
  int x_spl, p_rel, n_dim, L;
  double* x_i;
  const int32_t* pub_inv;     // [B] slot of the agent's row in x_send (-1: not published); nullptr: nothing published
  double* x_send;
};
// The reference's stop criterion inside the solve launch (omgx_batch_set_stop): parameter offsets of the state, the input and
// the target of the vehicle, its dimension, the tolerance; under_way [B] (device, owned by the caller): 1 while the agent's loop runs
struct StopArgs {
  int o_state, o_input, o_pose, n_dim;
  double tol;
  int32_t* under_way;
};
struct StoreArgs {
  double* out;            // [B, n_der, n_spl, n_samp]
  double* v_tot;          // [B, n_samp] or nullptr (needs n_der >= 2)
  const double* t0;       // [B] first sample, spline domain
  int coeff_off, n_spl, degree, n_knots, n_der, n_samp;
  double dt, inv_T;
  KnotArg knots;
};

__host__ __device__ inline size_t sample_scratch_doubles(int n_spl, int degree, int n_knots, int n_der) {
  const int L = n_knots - degree - 1, n_span = n_knots - 2 * degree - 1, D1 = degree + 1;
  return (size_t)n_knots + (size_t)D1 * n_spl * L + (size_t)D1 * n_spl * n_span + (size_t)n_der * n_spl * n_span * D1;
}

// value at u of the spline with coefficients c on the knot vector kk (degree dg <= 5), inside span jo.
// Fully unrolled triangle with compile-time indices: a dynamically indexed local array would live in
// scratch (global) memory.
__device__ __forceinline__ double deboor_at(const double* c, const double* kk, int dg, int jo, double u) {
  double dbo[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) dbo[r] = (r <= dg) ? c[jo - dg + r] : 0.0;
#pragma unroll
  for (int lev = 1; lev <= 5; ++lev) {
#pragma unroll
    for (int r = 5; r >= 1; --r) {
      if (lev <= dg && r >= lev && r <= dg) {
        const int idx = jo - dg + r;
        const double den = kk[idx + dg - lev + 1] - kk[idx];
        const double a = den != 0.0 ? (u - kk[idx]) / den : 0.0;
        dbo[r] = (1.0 - a) * dbo[r - 1] + a * dbo[r];
      }
    }
  }
  return dg == 0 ? dbo[0] : (dg == 1 ? dbo[1] : (dg == 2 ? dbo[2] : (dg == 3 ? dbo[3] : (dg == 4 ? dbo[4] : dbo[5]))));
}

// Samples [i_begin, i_end) of one agent by the whole workgroup.  coeffs: [n_spl][L] (global memory or
// LDS), scratch: sample_scratch_doubles() doubles the workgroup may overwrite.  Used by sample_kernel and
// by the epilogue of the solve kernel (the solution is still in LDS there).
template <typename OutT>
__device__ void sample_agent(const double* coeffs, double* scratch, int n_spl, int degree, const KnotArg& knots,
                             int n_knots, int n_der, double tb, double dt, double inv_T, int n_samp, int i_begin,
                             int i_end, OutT* out_b, OutT* vtot_b) {
  const int L = n_knots - degree - 1;
  const int n_span = n_knots - 2 * degree - 1;    // spans j = degree .. degree + n_span - 1
  const int D1 = degree + 1;
  double* kn = scratch;                           // [n_knots]
  double* cf = kn + n_knots;                      // [D1][n_spl][L]      coefficients of every derivative order
  double* val = cf + D1 * n_spl * L;              // [D1][n_spl][n_span] S^{(q)}(k_j+)
  double* pw = val + D1 * n_spl * n_span;         // [n_der][n_spl][n_span][D1] local power series
  for (int i = threadIdx.x; i < n_knots; i += blockDim.x) kn[i] = knots.k[i];
  for (int i = threadIdx.x; i < n_spl * L; i += blockDim.x) cf[i] = coeffs[i];
  __syncthreads();
  for (int o = 1; o <= degree; ++o) {             // c^(o)_i = (d-o+1) (c^(o-1)_{i+1}-c^(o-1)_i)/(k_{i+d+1}-k_{i+o})
    const int Lo = L - o, dd = degree - o + 1;
    for (int e = threadIdx.x; e < n_spl * Lo; e += blockDim.x) {
      const int k = e / Lo, i = e - k * Lo;
      const double* src = cf + ((o - 1) * n_spl + k) * L;
      const double den = kn[i + degree + 1] - kn[i + o];
      cf[(o * n_spl + k) * L + i] = den != 0.0 ? dd * (src[i + 1] - src[i]) / den : 0.0;
    }
    __syncthreads();
  }
  // right-hand limits of every derivative order at the left knot of every span
  for (int e = threadIdx.x; e < D1 * n_spl * n_span; e += blockDim.x) {
    const int q = e / (n_spl * n_span), r = e - q * n_spl * n_span, k = r / n_span, sp = r - k * n_span;
    const int j = degree + sp;
    // the q-th derivative lives on the knot vector kn[q .. n_knots-q), its span index there is j - q
    val[e] = deboor_at(cf + (q * n_spl + k) * L, kn + q, degree - q, j - q, kn[j]);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n_der * n_spl * n_span * D1; e += blockDim.x) {
    const int m = e % D1, r = e / D1;             // r = (o, k, sp)
    const int o = r / (n_spl * n_span), ks = r - o * n_spl * n_span;
    double f = 1.0;
    for (int q = 2; q <= m; ++q) f *= q;
    for (int q = 0; q < o; ++q) f /= inv_T;       // time derivative of order o: spline-domain derivative * inv_T^o
    pw[e] = (o + m <= degree) ? val[(o + m) * n_spl * n_span + ks] / f : 0.0;
  }
  __syncthreads();
  for (int i = i_begin + threadIdx.x; i < i_end; i += blockDim.x) {
    const double u = tb + i * dt;
    // span j: k_j < u <= k_{j+1} (reference convention, `basics/spline.py:131-136`)
    int j = degree;
    for (int q = degree + 1; q < n_knots - degree - 1; ++q) if (kn[q] < u) j = q;
    const double h = u - kn[j];
    const int sp = j - degree;
    double v2 = 0.0;
    for (int o = 0; o < n_der; ++o) {
      const int dg = degree - o;
      for (int k = 0; k < n_spl; ++k) {
        const double* c = pw + (((o * n_spl + k) * n_span) + sp) * D1;
        double v = c[dg];
        for (int m = dg - 1; m >= 0; --m) v = fma(v, h, c[m]);
        out_b[((size_t)o * n_spl + k) * n_samp + i] = (OutT)v;
        if (o == 1) v2 = fma(v, v, v2);
      }
    }
    if (vtot_b) vtot_b[i] = (OutT)sqrt(v2);
  }
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
template <int MODE, bool WAVE_ONLY, bool GEN, bool REFINE = false>
__global__ void __launch_bounds__(512)
ipm_solve_kernel(omgx::Dims d, omgx::Tables T, omgx::Opts o, int kkt_doubles,
                 const double* __restrict__ p, const double* __restrict__ x0,
                 const double* __restrict__ lb, const double* __restrict__ ub, int bounds_shared,
                 double* __restrict__ x, double* __restrict__ lam, int32_t* __restrict__ status,
                 int32_t* __restrict__ iters, int n_agents, long long* __restrict__ prof,
                 double* __restrict__ slabs, size_t slab_doubles, double* __restrict__ dw_state,
                 const int32_t* __restrict__ order, const StoreArgs* __restrict__ stp, int only_failed,
                 int* __restrict__ next_slot, const double* __restrict__ x0_alt, int n_alt, int32_t* __restrict__ attempts,
                 unsigned long long* __restrict__ stats, int stagger, const CenterArgs* __restrict__ ctr,
                 double* __restrict__ prep, size_t prep_doubles, const StopArgs* __restrict__ stop) {
  extern __shared__ __align__(16) double lds[];
  omgx::Work w;
    omgx::work_carve_split<MODE>(w, lds, MODE == omgx::WS_LDS ? nullptr : slabs + (size_t)blockIdx.x * slab_doubles,
                               d, kkt_doubles);
  omgx::CtxT<omgx::ws_kkt_hbm(MODE), WAVE_ONLY, omgx::ws_root_lds(MODE), GEN, false, REFINE> c; c.red = w.red;
#ifdef OMGX_PROFILE
  __shared__ long long prof_lds[omgx::PH_COUNT];
  c.prof = prof_lds;
#else
  c.prof = nullptr;
#endif
  // `prep` (round 6): the setup of every agent's solve -- parameter stage, Jacobian and rows at x0, classification, scaling, start
  // values -- was done for the whole batch by ipm_prepare_kernel ahead of this launch; a solve then starts by loading its record.
  // The matrix descriptors (the same for every agent) are written once per workgroup.
  double* const jval_own = w.jval;
  // (the descriptors are rewritten per solve only where the fused trajectory store may use the space behind a small KKT store as scratch)
  const bool describe_once = prep != nullptr || stp == nullptr;
  if (describe_once) { omgx::Kkt K0; K0.bind(d, T, w.kkt); omgx::kkt_describe(c, d, K0, w, true); }
  // mode 0: one workgroup per agent.  Spill modes: the grid is capped at the number of HBM
  // slabs and every workgroup walks over its agents.
  // `order` (optional) maps launch slots to agents: the host can put expected stragglers first so
  // that their long solves overlap the rest of the batch instead of trailing it
  // Spill modes hand the launch slots out dynamically (next_slot[0]: a counter that is zero at every launch -- the
  // workgroup that finishes last resets it, next_slot[1] counts the finished ones):
  // solves differ by a factor of several in their iteration counts, and a fixed share of agents per workgroup would
  // leave most of the chip waiting for the unluckiest one.  Which workgroup solves an agent does not change its result.
  __shared__ int slot_lds;
  // Two workgroups per CU start together and would run their (equally long) solves in lockstep -- both in a one-wave
  // phase, then both in an all-waves phase.  The workgroup in the second wave slot starts `stagger` x 8 k cycles late.
  if (stagger > 0 && (__builtin_amdgcn_s_getreg(6148) & 1))      // HW_ID[3:0] = wave slot within the SIMD
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  for (int slot = blockIdx.x; slot < n_agents;) {
    const int b = order ? order[slot] : slot;
    if (!next_slot) slot += gridDim.x;
    else {
      if (threadIdx.x == 0) slot_lds = gridDim.x + atomicAdd(next_slot, 1);
      __syncthreads();
      slot = slot_lds;
      __syncthreads();
    }
    // restart pass (OMGX_ONLY_FAILED): agents that are solved already keep x, lam_g, status, iters
    if (only_failed && status[b] == 0) continue;
    // stop rule (omgx_batch_set_stop): a vehicle whose loop has ended -- the criterion held at this or an earlier update -- is not
    // solved again: it keeps its plan (x <- x0), its multipliers and its status; iters = 0.  Every thread evaluates the same
    // numbers from the same loads (a thread that reads the flag after thread 0 cleared it takes the same branch).
    if (stop) {
      const StopArgs sa = *stop;
      bool go = sa.under_way[b] != 0;
      if (go && omgx::stop_criterium(p + (size_t)b * d.n_par, sa.o_state, sa.o_input, sa.o_pose, sa.n_dim, sa.tol)) go = false;
      if (!go) {
        for (int i = threadIdx.x; i < d.n_var; i += blockDim.x) x[(size_t)b * d.n_var + i] = x0[(size_t)b * d.n_var + i];
        if (threadIdx.x == 0) { sa.under_way[b] = 0; iters[b] = 0; }
        continue;
      }
    }
#ifdef OMGX_PROFILE
    if (threadIdx.x < omgx::PH_COUNT) prof_lds[threadIdx.x] = 0;
    __syncthreads();
    const long long t_begin = clock64();
#endif
    const double* lbb = lb + (bounds_shared ? 0 : (size_t)b * d.n_con);
    const double* ubb = ub + (bounds_shared ? 0 : (size_t)b * d.n_con);
    // Restart guesses (omgx_batch_set_restarts; cold solves only): an agent that does not converge from x0 is solved
    // again from x0_alt[0], x0_alt[1], ... by the same workgroup right away -- a separate pass over the failed agents
    // would leave the chip to a handful of them for as long as their slowest solve takes.
    omgx::Result r;
    int attempt = 0;
    for (;;) {
      const double* xs = attempt == 0 ? x0 + (size_t)b * d.n_var : x0_alt + ((size_t)(attempt - 1) * n_agents + b) * d.n_var;
      omgx::Start st;
      if (prep && attempt == 0) {
        double* rec = prep + (size_t)b * prep_doubles;
        if (omgx::ws_jac_hbm(MODE)) w.jval = rec + omgx::prep_layout(d).jval;      // (the scaled Jacobian stays where the setup kernel left it)
        st = omgx::ipm_load_start<omgx::ws_jac_hbm(MODE)>(c, d, w, rec);
      } else {
        w.jval = jval_own;
        st = omgx::ipm_setup(c, d, T, o, w, p + (size_t)b * d.n_par, xs, lbb, ubb, o.warm_start ? lam + (size_t)b * d.n_con : nullptr,
                             o.warm_start ? status[b] : 0, kkt_doubles, !describe_once);
      }
      r = omgx::ipm_iterate(c, d, T, o, w, lbb, ubb, st, kkt_doubles, o.warm_start ? dw_state[b] : 0.0);
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      if (r.status == 0 || o.warm_start || attempt >= n_alt) break;
      ++attempt;
    }
    if (attempts && threadIdx.x == 0) attempts[b] = attempt;
    for (int i = threadIdx.x; i < d.n_var; i += blockDim.x) x[(size_t)b * d.n_var + i] = w.x[i];
    for (int q = threadIdx.x; q < d.n_con; q += blockDim.x)
      lam[(size_t)b * d.n_con + q] =
          (r.status == 3 || w.rtype[q] == omgx::ROW_FREE) ? 0.0 : w.rho[q] * w.z[q];
    if (threadIdx.x == 0) {
      status[b] = r.status; iters[b] = r.iters; dw_state[b] = r.dw;
      if (stats) {      // launch statistics (omgx_batch_set_stats): integer atomics, the same totals in any order
        atomicAdd(stats + 0, r.status == 0 ? 1ull : 0ull);
        atomicAdd(stats + 1, (unsigned long long)r.iters);
        atomicMax(stats + 2, (unsigned long long)r.iters);
        atomicAdd(stats + 3, 1ull);
      }
    }
    if (ctr) {
      // the ADMM x-update's centre (`omgx_admm_center_ex`) from the solution in LDS: no launch of its own
      const CenterArgs ca = *ctr;
      const int ns = ca.n_dim * ca.L;
      const int slot = ca.pub_inv ? ca.pub_inv[b] : -1;
      for (int q = threadIdx.x; q < ns; q += blockDim.x) {
        const double v = w.x[ca.x_spl + q] + p[(size_t)b * d.n_par + ca.p_rel + q / ca.L];
        ca.x_i[(size_t)b * ns + q] = v;
        if (slot >= 0) ca.x_send[(size_t)slot * ns + q] = v;
      }
    }
    if (stp) {
      // `Vehicle.store` fused behind the solve (reference `vehicles/vehicle.py:250-300`): the trajectories of
      // this agent straight from the solution in LDS; the KKT store is free now and serves as scratch
      // (the specification sits in device memory: as a by-value kernel argument its 90 dwords would be kept in
      // scalar registers across the whole solve)
      const StoreArgs st = *stp;
      __syncthreads();
      sample_agent<double>(w.x + st.coeff_off, w.kkt, st.n_spl, st.degree, st.knots, st.n_knots, st.n_der, st.t0[b],
                           st.dt, st.inv_T, st.n_samp, 0, st.n_samp, st.out + (size_t)b * st.n_der * st.n_spl * st.n_samp,
                           st.v_tot ? st.v_tot + (size_t)b * st.n_samp : nullptr);
      if (prep) { __syncthreads(); omgx::Kkt K0; K0.bind(d, T, w.kkt); omgx::kkt_describe(c, d, K0, w, true); }      // (the scratch may have reached the descriptors)
    }
#ifdef OMGX_PROFILE
    __syncthreads();
    if (threadIdx.x == 0) prof_lds[omgx::PH_TOTAL] = clock64() - t_begin;
    __syncthreads();
    if (prof && threadIdx.x < omgx::PH_COUNT) prof[(size_t)b * omgx::PH_COUNT + threadIdx.x] = prof_lds[threadIdx.x];
#endif
    __syncthreads();
  }
  if (next_slot && threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(next_slot + 1, 1) == (int)gridDim.x - 1) { next_slot[0] = 0; next_slot[1] = 0; __threadfence(); }
  }
}

// Round 6: the setup of a batch of solves as a kernel of its own -- north_star's "basis evaluation on the sample grid and the
// constraint Jacobian assembled with coalesced loads across a batch of agents".  One workgroup per agent, a few KB of LDS
// (atoms, knots, slots, x) and half the registers of the solve kernel: four to eight workgroups per CU hide the table-load
// latencies that the same statements pay in full at the head of the solve kernel, where two agents fill a CU (78 k of the
// 339 k cycles of a warm-started solve, profiles/r05_phase_cycles_mpc.json).  Same device function (omgx::ipm_setup), same
// thread count as the solve kernel (the fixed-order reductions depend on it): the same bits as the in-kernel setup.
// Output: the agent's record (omgx::prep_layout) -- start point, slots, row arrays, scaled Jacobian, start scalars.
__host__ __device__ inline size_t prepare_lds_doubles(const omgx::Dims& d) {
  return (size_t)d.n_slots + d.n_atoms + d.n_knots + d.N + 64;
}
template <bool GEN>
__global__ void __launch_bounds__(512, 4)      // (second argument: waves per SIMD the register allocation must leave room for -- 128 VGPRs)
ipm_prepare_kernel(omgx::Dims d, omgx::Tables T, omgx::Opts o, const double* __restrict__ p, const double* __restrict__ x0,
                   const double* __restrict__ lb, const double* __restrict__ ub, int bounds_shared,
                   const double* __restrict__ lam, const int32_t* __restrict__ status, int n_agents,
                   double* __restrict__ prep, size_t prep_doubles, int only_failed) {
  extern __shared__ __align__(16) double lds[];
  const int b = blockIdx.x;
  if (b >= n_agents) return;
  if (only_failed && status[b] == 0) return;
  const omgx::PrepLayout L = omgx::prep_layout(d);
  double* rec = prep + (size_t)b * prep_doubles;
  omgx::Work w;
  {
    double* q = lds;
    w.slots = q; q += d.n_slots; w.atoms = q; q += d.n_atoms; w.knots = q; q += d.n_knots;
    w.x = q; q += d.N; w.red = q; q += 64;
    w.xt = nullptr; w.gbar = nullptr; w.sol = nullptr; w.dinv = nullptr; w.kkt = nullptr; w.col = nullptr; w.root = nullptr;
    w.ht = nullptr; w.ds = nullptr;
    w.hv = rec + L.hv; w.rho = rec + L.rho; w.vv = rec + L.vv; w.z = rec + L.z;
    w.rtype = (int8_t*)(rec + L.rtype); w.jval = rec + L.jval;
  }
  omgx::CtxT<false, false, false, GEN, true> c; c.red = w.red; c.prof = nullptr;
  const double* lbb = lb + (bounds_shared ? 0 : (size_t)b * d.n_con);
  const double* ubb = ub + (bounds_shared ? 0 : (size_t)b * d.n_con);
  const omgx::Start st = omgx::ipm_setup(c, d, T, o, w, p + (size_t)b * d.n_par, x0 + (size_t)b * d.n_var, lbb, ubb,
                                         o.warm_start ? lam + (size_t)b * d.n_con : nullptr, o.warm_start ? status[b] : 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < d.N; i += blockDim.x) rec[L.x + i] = w.x[i];
  for (int i = threadIdx.x; i < d.n_slots; i += blockDim.x) rec[L.slots + i] = w.slots[i];
  if (threadIdx.x == 0) {
    rec[L.sc] = (double)st.status; rec[L.sc + 1] = (double)st.warm; rec[L.sc + 2] = (double)st.use_t;
    rec[L.sc + 3] = st.mu; rec[L.sc + 4] = st.zt; rec[L.sc + 5] = st.f;
  }
}

// Verification entry (omgx_batch_eval): one workgroup evaluates the tables of the solve at a caller's point and dumps the
// raw arrays -- row values, objective, Jacobian entries, the KKT store holding the Lagrangian Hessian -- to `out`
// [agent][n_con + 1 + nnz_j + kkt_doubles]; the host scatters them into dense matrices.
template <int MODE, bool WAVE_ONLY, bool GEN>
__global__ void __launch_bounds__(512)
ipm_eval_kernel(omgx::Dims d, omgx::Tables T, int kkt_doubles, const double* __restrict__ p, const double* __restrict__ x,
                const double* __restrict__ lam, int n_agents, double* __restrict__ slabs, size_t slab_doubles,
                double* __restrict__ out) {
  extern __shared__ __align__(16) double lds[];
  omgx::Work w;
  omgx::work_carve_split<MODE>(w, lds, MODE == omgx::WS_LDS ? nullptr : slabs + (size_t)blockIdx.x * slab_doubles,
                               d, kkt_doubles);
  omgx::CtxT<omgx::ws_kkt_hbm(MODE), WAVE_ONLY, omgx::ws_root_lds(MODE), GEN> c; c.red = w.red; c.prof = nullptr;
  const size_t stride = (size_t)d.n_con + 1 + d.nnz_j + kkt_doubles;
  for (int b = blockIdx.x; b < n_agents; b += gridDim.x) {
    double* o = out + (size_t)b * stride;
    omgx::ipm_eval(c, d, T, w, p + (size_t)b * d.n_par, x + (size_t)b * d.n_var, lam + (size_t)b * d.n_con, kkt_doubles,
                   o + d.n_con);
    for (int i = threadIdx.x; i < d.n_con; i += blockDim.x) o[i] = w.hv[i];
    for (int i = threadIdx.x; i < d.nnz_j; i += blockDim.x) o[d.n_con + 1 + i] = w.jval[i];
    for (int i = threadIdx.x; i < kkt_doubles; i += blockDim.x) o[d.n_con + 1 + d.nnz_j + i] = w.kkt[i];
    __syncthreads();
  }
}

typedef void (*ipm_eval_kernel_t)(omgx::Dims, omgx::Tables, int, const double*, const double*, const double*, int, double*,
                                  size_t, double*);
template <bool GEN>
static ipm_eval_kernel_t ipm_eval_kernel_gen(int mode, int wave_ok) {
#ifdef OMGX_ONLY_HEADLINE
  return ipm_eval_kernel<omgx::WS_JAC_HV, true, false>;
#else
  switch (mode) {
    case omgx::WS_LDS: return wave_ok ? ipm_eval_kernel<omgx::WS_LDS, true, GEN> : ipm_eval_kernel<omgx::WS_LDS, false, GEN>;
    case omgx::WS_KKT_HBM: return ipm_eval_kernel<omgx::WS_KKT_HBM, false, GEN>;
    case omgx::WS_JAC_HBM: return ipm_eval_kernel<omgx::WS_JAC_HBM, false, GEN>;
    case omgx::WS_JAC_ONLY: return ipm_eval_kernel<omgx::WS_JAC_ONLY, true, GEN>;
    case omgx::WS_JAC_HV: return ipm_eval_kernel<omgx::WS_JAC_HV, true, GEN>;
    case omgx::WS_ROOT_HBM: return ipm_eval_kernel<omgx::WS_ROOT_HBM, false, true>;      // (one instance: pick_mode hands mode 6 to general templates only)
    default: return ipm_eval_kernel<omgx::WS_ROWS_HBM, false, GEN>;
  }
#endif
}
static ipm_eval_kernel_t ipm_eval_kernel_for(int mode, int wave_ok, int general) {
  return general ? ipm_eval_kernel_gen<true>(mode, wave_ok) : ipm_eval_kernel_gen<false>(mode, wave_ok);
}

typedef void (*ipm_kernel_t)(omgx::Dims, omgx::Tables, omgx::Opts, int, const double*, const double*, const double*,
                             const double*, int, double*, double*, int32_t*, int32_t*, int, long long*, double*, size_t, double*,
                             const int32_t*, const StoreArgs*, int, int*, const double*, int, int32_t*, unsigned long long*, int, const CenterArgs*,
                             double*, size_t, const StopArgs*);
// (GEN: the instance that carries the terms with four factors, the cos / sin atoms and the basis rows of any degree --
// Dims::general; the other one is the kernel of the benchmark classes, free of that code)
template <bool GEN>
static ipm_kernel_t ipm_kernel_gen(int mode, int wave_ok, int refine = 0) {
  // (the refinement of regularised steps -- omgx_options.refine -- has instances of its own: templates on the wave path, not general)
  if (refine && wave_ok && !GEN) {
    switch (mode) {
      case omgx::WS_LDS: return ipm_solve_kernel<omgx::WS_LDS, true, false, true>;
      case omgx::WS_JAC_ONLY: return ipm_solve_kernel<omgx::WS_JAC_ONLY, true, false, true>;
      case omgx::WS_JAC_HV: return ipm_solve_kernel<omgx::WS_JAC_HV, true, false, true>;
      default: break;
    }
  }
#ifdef OMGX_ONLY_HEADLINE      // developer builds (register counts of one instance in a third of the compile time): only the kernel of the benchmark class
  return ipm_solve_kernel<omgx::WS_JAC_HV, true, false>;
#else
  switch (mode) {
    case omgx::WS_LDS: return wave_ok ? ipm_solve_kernel<omgx::WS_LDS, true, GEN> : ipm_solve_kernel<omgx::WS_LDS, false, GEN>;
    case omgx::WS_KKT_HBM: return ipm_solve_kernel<omgx::WS_KKT_HBM, false, GEN>;
    case omgx::WS_JAC_HBM: return ipm_solve_kernel<omgx::WS_JAC_HBM, false, GEN>;
    case omgx::WS_JAC_ONLY: return ipm_solve_kernel<omgx::WS_JAC_ONLY, true, GEN>;
    case omgx::WS_JAC_HV: return ipm_solve_kernel<omgx::WS_JAC_HV, true, GEN>;
    case omgx::WS_ROOT_HBM: return ipm_solve_kernel<omgx::WS_ROOT_HBM, false, true>;
    default: return ipm_solve_kernel<omgx::WS_ROWS_HBM, false, GEN>;
  }
#endif
}
static ipm_kernel_t ipm_kernel_for(int mode, int wave_ok, int general, int refine = 0) {
  return general ? ipm_kernel_gen<true>(mode, wave_ok) : ipm_kernel_gen<false>(mode, wave_ok, refine);
}

template <typename OutT>
__global__ void __launch_bounds__(256)
sample_kernel(const double* __restrict__ x, int x_stride, int coeff_off, int n_spl, int degree,
              KnotArg knots, int n_knots, int n_der,
              const double* __restrict__ t0, double dt, double inv_T, int n_samp, OutT* __restrict__ out,
              OutT* __restrict__ v_tot) {
  extern __shared__ __align__(16) double lds[];
  const int b = blockIdx.y;
  const int i_end = min(n_samp, (int)(blockIdx.x + 1) * OMGX_SAMPLE_CHUNK);
  sample_agent<OutT>(x + (size_t)b * x_stride + coeff_off, lds, n_spl, degree, knots, n_knots, n_der, t0[b], dt, inv_T,
                     n_samp, blockIdx.x * OMGX_SAMPLE_CHUNK, i_end, out + (size_t)b * n_der * n_spl * n_samp,
                     v_tot ? v_tot + (size_t)b * n_samp : nullptr);
}

// Prediction of one receding-horizon step: thread (agent b, spline k) evaluates the plan and its time
// derivatives at tau by de Boor on the active span and writes them into the parameter vector (state0 /
// input0 / ...), thread k == 0 also the time since the last knot crossing.  RK4 mode: the state is the
// caller's current state integrated over the n_sub sample intervals before tau with the inputs the plan
// holds there, by the statements of the reference's `Vehicle::integrate` (export/vehicles/Vehicle.cpp:82-110)
// for the integrator models (`ode` = input: Holonomic, Holonomic3D).
struct PredictArgs {
  KnotArg kn;
  int coeff_off, n_spl, degree, n_knots, n_out, p_off[4], p_t, mode, n_sub;
  double tau, inv_T, t_value, dtau;
  const double* state_in;
};

// d-th derivative (spline-domain units) at u of the spline whose coefficients on span j are c[j-degree .. j]
__device__ __forceinline__ double spline_der_at(const double* c, const double* kk, int degree, int j, double u, int dord) {
  double v[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) v[r] = (r <= degree) ? c[j - degree + r] : 0.0;
  // after o differences v[r] (r = o .. degree) holds c^(o)_{j-degree+r}
#pragma unroll
  for (int o = 1; o <= 3; ++o) {
    if (o <= dord) {
#pragma unroll
      for (int r = 5; r >= 1; --r) {
        if (r >= o && r <= degree) {
          const int i = j - degree + r;                       // c^(o)_i = (degree-o+1) (c^(o-1)_i - c^(o-1)_{i-1}) / (k_{i+degree-o+1} - k_i)
          const double den = kk[i + degree - o + 1] - kk[i];
          v[r] = den != 0.0 ? (degree - o + 1) * (v[r] - v[r - 1]) / den : 0.0;
        }
      }
    }
  }
  const int dg = degree - dord;
#pragma unroll
  for (int lev = 1; lev <= 5; ++lev) {
#pragma unroll
    for (int r = 5; r >= 1; --r) {
      if (lev <= dg && r >= dord + lev && r <= degree) {
        const int i = j - degree + r;
        const double den = kk[i + dg - lev + 1] - kk[i];
        const double a = den != 0.0 ? (u - kk[i]) / den : 0.0;
        v[r] = (1.0 - a) * v[r - 1] + a * v[r];
      }
    }
  }
  double out = 0.0;
#pragma unroll
  for (int r = 0; r < 6; ++r) if (r == degree) out = v[r];
  return out;
}

__device__ __forceinline__ int span_of(const double* kk, int degree, int n_knots, double u) {
  int j = degree;                                  // span: k_j < u <= k_{j+1} (`basics/spline.py:131-136`)
  for (int q = degree + 1; q < n_knots - degree - 1; ++q) if (kk[q] < u) j = q;
  return j;
}

// Launch order for the next solve: agents bucketed by the iteration count of their previous solve,
// largest first (64 buckets, counting sort in LDS by one workgroup; the order inside a bucket is
// arbitrary).  Replaces a device-wide sort: this is ~10 us.
// Round 4: among the agents with the same previous count (most have 1) those that carry a heavy inertia correction go
// first -- a large dw is the signature of the slow ones (multipliers of 5-20 on bilinear rows: the regularised Newton
// iteration converges linearly).  On the host model of the step (512 slots, greedy queue) the total time of 20 headline
// steps drops by 4 % against the count alone; the perfect order would gain 6.6 %.
__device__ __forceinline__ int order_bucket(int it, double dw) {
  const int a = it < 0 ? 0 : (it > 15 ? 15 : it);
  const int q = dw > 10.0 ? 3 : (dw > 1.0 ? 2 : (dw > 0.1 ? 1 : 0));
  return 63 - (4 * a + q);
}
__device__ __forceinline__ void order_block(const int32_t* __restrict__ iters, const double* __restrict__ dw, int32_t* __restrict__ order, int B) {
  __shared__ int cnt[64], off[64];
  if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) atomicAdd(&cnt[order_bucket(iters[b], dw ? dw[b] : 0.0)], 1);
  __syncthreads();
  if (threadIdx.x == 0) { int a = 0; for (int k = 0; k < 64; ++k) { off[k] = a; a += cnt[k]; } }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) order[atomicAdd(&off[order_bucket(iters[b], dw ? dw[b] : 0.0)], 1)] = b;
}

// (ord_iters != nullptr: the launch carries one more workgroup, which computes the launch order of the next solve --
// the receding-horizon step then has one launch less in front of its solve kernel)
__global__ void __launch_bounds__(256)
predict_kernel(const double* __restrict__ x, int n_var, double* __restrict__ p, int n_par, int B, PredictArgs a,
               const int32_t* __restrict__ ord_iters, int32_t* __restrict__ ord_out, const double* __restrict__ ord_dw) {
  if (ord_iters && blockIdx.x == gridDim.x - 1) { order_block(ord_iters, ord_dw, ord_out, B); return; }
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * a.n_spl) return;
  const int b = id / a.n_spl, k = id - b * a.n_spl;
  const int L = a.n_knots - a.degree - 1;
  const double* c = x + (size_t)b * n_var + a.coeff_off + k * L;
  double* pb = p + (size_t)b * n_par;
  const int j = span_of(a.kn.k, a.degree, a.n_knots, a.tau);
  double sc = 1.0;
  for (int o = 0; o < a.n_out; ++o) {
    if (a.p_off[o] >= 0 && !(o == 0 && a.mode == OMGX_PREDICT_RK4))
      pb[a.p_off[o] + k] = spline_der_at(c, a.kn.k, a.degree, j, a.tau, o) * sc;
    sc *= a.inv_T;
  }
  if (a.mode == OMGX_PREDICT_RK4 && a.p_off[0] >= 0) {
    // k1 = k2 = k3 = u_i, k4 = u_{i+1} for an integrator; h = sample time
    const double h = a.dtau / a.inv_T;
    double st = a.state_in[(size_t)b * a.n_spl + k];
    double u0 = a.tau - a.n_sub * a.dtau;
    double ui = spline_der_at(c, a.kn.k, a.degree, span_of(a.kn.k, a.degree, a.n_knots, u0), u0, 1) * a.inv_T;
    for (int i = 0; i < a.n_sub; ++i) {
      const double u1 = a.tau - (a.n_sub - 1 - i) * a.dtau;
      const double un = spline_der_at(c, a.kn.k, a.degree, span_of(a.kn.k, a.degree, a.n_knots, u1), u1, 1) * a.inv_T;
      st += (h / 6.0) * (ui + 2.0 * ui + 2.0 * ui + un);
      ui = un;
    }
    pb[a.p_off[0] + k] = st;
  }
  if (k == 0 && a.p_t >= 0) pb[a.p_t] = a.t_value;
}

// Non-ideal prediction of the Quadrotor (`vehicles/vehicle.py:323-337` with the model's own `ode`,
// `vehicles/quadrotor.py:149-152`: state (x, y, dx, dy, theta), inputs (u1, u2) = thrust and pitch rate, which the plan
// holds as functions of its second and third derivatives, `quadrotor.py:121-140`).  One thread per agent: the inputs at
// the n_sub + 1 sample points that end at tau, classical Runge-Kutta over the sample intervals with the input taken
// linearly between the samples (the reference integrates with odeint on a linear interpolation of the sampled inputs,
// `vehicle.py:412-423`), the position of the integrated state into spl0, the plan's own derivatives at tau into
// dspl0 / ddspl0 (`quadrotor.py:110-114`: only state[:2] of the prediction enters the parameters).
__device__ __forceinline__ void quad_inputs(const double* cx, const double* cy, const PredictArgs& a, double u, double g, double* u1, double* u2) {
  const int j = span_of(a.kn.k, a.degree, a.n_knots, u);
  const double s2 = a.inv_T * a.inv_T, s3 = s2 * a.inv_T;
  const double ddx = spline_der_at(cx, a.kn.k, a.degree, j, u, 2) * s2, ddy = spline_der_at(cy, a.kn.k, a.degree, j, u, 2) * s2;
  const double dddx = spline_der_at(cx, a.kn.k, a.degree, j, u, 3) * s3, dddy = spline_der_at(cy, a.kn.k, a.degree, j, u, 3) * s3;
  const double n2 = (ddy + g) * (ddy + g) + ddx * ddx;
  *u1 = sqrt(n2);
  *u2 = (dddx * (ddy + g) - ddx * dddy) / n2;
}

__device__ __forceinline__ void quad_ode(const double* s, double u1, double u2, double g, double* k) {
  k[0] = s[2]; k[1] = s[3]; k[2] = u1 * sin(s[4]); k[3] = u1 * cos(s[4]) - g; k[4] = u2;
}

__global__ void __launch_bounds__(256)
predict_quadrotor_kernel(const double* __restrict__ x, int n_var, double* __restrict__ p, int n_par, int B, PredictArgs a,
                         double g, double* __restrict__ state_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int L = a.n_knots - a.degree - 1;
  const double* cx = x + (size_t)b * n_var + a.coeff_off;
  const double* cy = cx + L;
  double* pb = p + (size_t)b * n_par;
  const double h = a.dtau / a.inv_T;
  double s[5];
  for (int q = 0; q < 5; ++q) s[q] = a.state_in[(size_t)b * 5 + q];
  double u1a, u2a;
  quad_inputs(cx, cy, a, a.tau - a.n_sub * a.dtau, g, &u1a, &u2a);
  for (int i = 0; i < a.n_sub; ++i) {
    double u1b, u2b, k1[5], k2[5], k3[5], k4[5], st[5];
    quad_inputs(cx, cy, a, a.tau - (a.n_sub - 1 - i) * a.dtau, g, &u1b, &u2b);
    const double u1m = 0.5 * (u1a + u1b), u2m = 0.5 * (u2a + u2b);
    quad_ode(s, u1a, u2a, g, k1);
    for (int q = 0; q < 5; ++q) st[q] = s[q] + 0.5 * h * k1[q];
    quad_ode(st, u1m, u2m, g, k2);
    for (int q = 0; q < 5; ++q) st[q] = s[q] + 0.5 * h * k2[q];
    quad_ode(st, u1m, u2m, g, k3);
    for (int q = 0; q < 5; ++q) st[q] = s[q] + h * k3[q];
    quad_ode(st, u1b, u2b, g, k4);
    for (int q = 0; q < 5; ++q) s[q] += (h / 6.0) * (k1[q] + 2.0 * k2[q] + 2.0 * k3[q] + k4[q]);
    u1a = u1b; u2a = u2b;
  }
  if (state_out) for (int q = 0; q < 5; ++q) state_out[(size_t)b * 5 + q] = s[q];
  const int j = span_of(a.kn.k, a.degree, a.n_knots, a.tau);
  if (a.p_off[0] >= 0) { pb[a.p_off[0]] = s[0]; pb[a.p_off[0] + 1] = s[1]; }
  double sc = a.inv_T;
  for (int o = 1; o < a.n_out; ++o) {
    if (a.p_off[o] >= 0) {
      pb[a.p_off[o]] = spline_der_at(cx, a.kn.k, a.degree, j, a.tau, o) * sc;
      pb[a.p_off[o] + 1] = spline_der_at(cy, a.kn.k, a.degree, j, a.tau, o) * sc;
    }
    sc *= a.inv_T;
  }
  if (a.p_t >= 0) pb[a.p_t] = a.t_value;
}

__global__ void __launch_bounds__(1024)
order_kernel(const int32_t* __restrict__ iters, const double* __restrict__ dw, int32_t* __restrict__ order, int B) { order_block(iters, dw, order, B); }

// warm-start shift of one row: every entry block <- T * block (`spline_extra.py:165-191`); scratch: LDS doubles for the
// largest block; all threads of the workgroup take part (barriers inside)
__device__ __forceinline__ void shift_row(double* __restrict__ xrow, const int32_t* __restrict__ entries, int n_ent,
                                          const double* __restrict__ Tm, double* scratch) {
  for (int e = 0; e < n_ent; ++e) {
    const int off = entries[4 * e], rows = entries[4 * e + 1], cols = entries[4 * e + 2];
    const double* Tmat = Tm + entries[4 * e + 3];
    double* xe = xrow + off;
    for (int i = threadIdx.x; i < rows * cols; i += blockDim.x) scratch[i] = xe[i];
    __syncthreads();
    for (int i = threadIdx.x; i < rows * cols; i += blockDim.x) {
      const int k = i / rows, r = i - k * rows;
      double acc = 0.0;
      for (int q = 0; q < rows; ++q) acc += Tmat[r * rows + q] * scratch[k * rows + q];
      xe[i] = acc;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(64)
shift_kernel(double* __restrict__ x, int x_stride, const uint8_t* __restrict__ mask,
             const int32_t* __restrict__ entries, int n_ent, const double* __restrict__ Tm) {
  extern __shared__ __align__(16) double lds[];
  const int b = blockIdx.x;
  if (mask && !mask[b]) return;
  shift_row(x + (size_t)b * x_stride, entries, n_ent, Tm, lds);
}

// ---------------------------------------------------------------------------
// Rollout: K receding-horizon steps of every agent in ONE launch (omgx_batch_rollout).  Agents of a point-to-point batch
// are independent, so nothing in the protocol asks for a barrier between the steps of different agents: a persistent
// workgroup takes an agent and runs its whole loop -- ideal prediction from the current plan, obstacles advanced, the
// knot-crossing shift of the plan and of the multipliers, warm-started solve -- K times, statement for statement what
// `BatchP2P.step` issues as separate launches (predict_kernel, tensor updates, shift_kernel, index_select, solve): the same
// bits per agent (tests/test_gpu_rollout.py).  What it removes is the step barrier: with 1024 agents on 512 resident
// workgroups a step launched on its own is two rounds plus a third for whoever a straggler displaced.
// ---------------------------------------------------------------------------
struct RolloutStep { double tau, t_rel; int32_t crossed, pad; };
struct RolloutArgs {
  KnotArg kn;
  int coeff_off, n_spl, degree, n_knots, n_out, p_off[4], p_t;
  double inv_T, dt;
  int n_obst, obst[8][4];
  const int32_t* sh_ent; int n_ent; const double* sh_T;
  const int32_t* lam_perm;
  const RolloutStep* steps; int K;
  omgx::Opts o_cross;
  unsigned long long* stats;            // [K][4] {solved, sum of iterations, max, agents} or nullptr
  int32_t* iters_log; int32_t* status_log;     // [K][n_agents] or nullptr
  StopArgs stop; int stop_on;           // omgx_batch_set_stop: an agent's loop ends at the step its state meets the criterion
};

template <int MODE, bool WAVE_ONLY, bool GEN>
__global__ void __launch_bounds__(512)
ipm_rollout_kernel(omgx::Dims d, omgx::Tables T, omgx::Opts o, int kkt_doubles, double* __restrict__ p, double* __restrict__ x,
                   const double* __restrict__ lb, const double* __restrict__ ub, int bounds_shared, double* __restrict__ lam,
                   int32_t* __restrict__ status, int32_t* __restrict__ iters, int n_agents, double* __restrict__ slabs,
                   size_t slab_doubles, double* __restrict__ dw_state, int* __restrict__ next_slot,
                   const RolloutArgs* __restrict__ rop, int stagger, const int32_t* __restrict__ order,
                   const StoreArgs* __restrict__ stp) {
  extern __shared__ __align__(16) double lds[];
  omgx::Work w;
  omgx::work_carve_split<MODE>(w, lds, MODE == omgx::WS_LDS ? nullptr : slabs + (size_t)blockIdx.x * slab_doubles, d, kkt_doubles);
  omgx::CtxT<omgx::ws_kkt_hbm(MODE), WAVE_ONLY, omgx::ws_root_lds(MODE), GEN> c; c.red = w.red;
  c.prof = nullptr;
  __shared__ int slot_lds;
  if (stagger > 0 && (__builtin_amdgcn_s_getreg(6148) & 1))
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  const int K = rop->K;
  // (agents in the caller's order -- the ones whose last solve was slow first: a workgroup that gets a long loop early is
  // given a short one by the queue afterwards)
  for (int slot = blockIdx.x; slot < n_agents;) {
    const int b = order ? order[slot] : slot;
    if (threadIdx.x == 0) slot_lds = gridDim.x + atomicAdd(next_slot, 1);
    __syncthreads();
    slot = slot_lds;
    __syncthreads();
    double* pb = p + (size_t)b * d.n_par;
    double* xb = x + (size_t)b * d.n_var;
    double* lamb = lam + (size_t)b * d.n_con;
    const double* lbb = lb + (bounds_shared ? 0 : (size_t)b * d.n_con);
    const double* ubb = ub + (bounds_shared ? 0 : (size_t)b * d.n_con);
    for (int k = 0; k < K; ++k) {
      const RolloutStep st = rop->steps[k];
      // (1) ideal prediction (predict_kernel): the plan and its time derivatives at tau, the new t
      {
        const int n_spl = rop->n_spl, degree = rop->degree, n_knots = rop->n_knots, n_out = rop->n_out;
        const int L = n_knots - degree - 1;
        if ((int)threadIdx.x < n_spl) {
          const int ks = threadIdx.x;
          const double* cc = xb + rop->coeff_off + ks * L;
          const int j = span_of(rop->kn.k, degree, n_knots, st.tau);
          double sc = 1.0;
          for (int q = 0; q < n_out; ++q) {
            if (rop->p_off[q] >= 0) pb[rop->p_off[q] + ks] = spline_der_at(cc, rop->kn.k, degree, j, st.tau, q) * sc;
            sc *= rop->inv_T;
          }
          if (ks == 0 && rop->p_t >= 0) pb[rop->p_t] = st.t_rel;
        }
      }
      // (2) obstacles move on: x <- x + (dt v + dt^2 / 2 a), v <- v + dt a (each product and sum rounded on its own, as the
      //     tensor statements of BatchP2P.step are)
      for (int q = 0; q < rop->n_obst; ++q) {
        const int ox = rop->obst[q][0], ov = rop->obst[q][1], oa = rop->obst[q][2], nd = rop->obst[q][3];
        if ((int)threadIdx.x < nd) {
#pragma clang fp contract(off)      // (no fused multiply-add here: the tensor statements round every product)
          const int i = threadIdx.x;
          const double dt = rop->dt, c2 = 0.5 * dt * dt;
          const double pv = pb[ov + i], pa = pb[oa + i];
          const double m1 = dt * pv, m2 = c2 * pa, m3 = dt * pa;
          const double s1 = m1 + m2;
          pb[ox + i] = pb[ox + i] + s1;
          pb[ov + i] = pv + m3;
        }
      }
      __syncthreads();
      // (3) knot crossing: plan <- T plan, multipliers by index (the KKT store is idle between two solves: scratch)
      if (st.crossed) {
        shift_row(xb, rop->sh_ent, rop->n_ent, rop->sh_T, w.kkt);
        for (int i = threadIdx.x; i < d.n_con; i += blockDim.x) w.kkt[i] = lamb[i];
        __syncthreads();
        for (int i = threadIdx.x; i < d.n_con; i += blockDim.x) { const int s = rop->lam_perm[i]; lamb[i] = s >= 0 ? w.kkt[s] : 0.0; }
        __syncthreads();
      }
      // stop rule (omgx_batch_set_stop), where the solve kernel of the per-step path tests it -- after the glue of the step: the
      // vehicle has arrived, its loop ends here (`execution/simulator.py:39-62`): plan, multipliers and status stay as they are,
      // the remaining steps of the call log iters 0
      if (rop->stop_on) {
        const StopArgs sa = rop->stop;
        bool go = sa.under_way[b] != 0;
        if (go && omgx::stop_criterium(pb, sa.o_state, sa.o_input, sa.o_pose, sa.n_dim, sa.tol)) go = false;
        __syncthreads();
        if (!go) {
          if (threadIdx.x == 0) {
            sa.under_way[b] = 0; iters[b] = 0;
            for (int k2 = k; k2 < K; ++k2) {
              if (rop->iters_log) rop->iters_log[(size_t)k2 * n_agents + b] = 0;
              if (rop->status_log) rop->status_log[(size_t)k2 * n_agents + b] = status[b];
            }
          }
          break;
        }
      }
      // (4) warm-started solve, results back to the agent's rows
      const omgx::Result r = omgx::ipm_solve(c, d, T, st.crossed ? rop->o_cross : o, w, pb, xb, lbb, ubb, o.warm_start ? lamb : nullptr,
                                             o.warm_start ? status[b] : 0, kkt_doubles, o.warm_start ? dw_state[b] : 0.0);
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      for (int i = threadIdx.x; i < d.n_var; i += blockDim.x) xb[i] = w.x[i];
      for (int q = threadIdx.x; q < d.n_con; q += blockDim.x)
        lamb[q] = (r.status == 3 || w.rtype[q] == omgx::ROW_FREE) ? 0.0 : w.rho[q] * w.z[q];
      if (threadIdx.x == 0) {
        status[b] = r.status; iters[b] = r.iters; dw_state[b] = r.dw;
        if (rop->stats) {
          unsigned long long* sk = rop->stats + 4 * (size_t)k;
          atomicAdd(sk + 0, r.status == 0 ? 1ull : 0ull);
          atomicAdd(sk + 1, (unsigned long long)r.iters);
          atomicMax(sk + 2, (unsigned long long)r.iters);
          atomicAdd(sk + 3, 1ull);
        }
        if (rop->iters_log) rop->iters_log[(size_t)k * n_agents + b] = r.iters;
        if (rop->status_log) rop->status_log[(size_t)k * n_agents + b] = r.status;
      }
      if (stp) {      // `Vehicle.store` of this step (omgx_batch_set_store), as in the solve kernel's epilogue
        const StoreArgs st2 = *stp;
        __syncthreads();
        sample_agent<double>(w.x + st2.coeff_off, w.kkt, st2.n_spl, st2.degree, st2.knots, st2.n_knots, st2.n_der, st2.t0[b],
                             st2.dt, st2.inv_T, st2.n_samp, 0, st2.n_samp, st2.out + (size_t)b * st2.n_der * st2.n_spl * st2.n_samp,
                             st2.v_tot ? st2.v_tot + (size_t)b * st2.n_samp : nullptr);
      }
      __syncthreads();
    }
  }
  // (the last workgroup out resets the queue counter for the next launch, as in the solve kernel)
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(next_slot + 1, 1) == (int)gridDim.x - 1) { next_slot[0] = 0; next_slot[1] = 0; }
  }
}

// ---------------------------------------------------------------------------
// Formation ADMM kernels (all pointers are device pointers; tiny, memory-bound)
// ---------------------------------------------------------------------------
// (elements [B ns, (B + n_pub) ns): the rows other ranks need, written to the send buffer of the exchange by the same
// launch -- x_send[i] = row pub_rows[i])
__global__ void admm_center_kernel(omgx_admm_layout lay, const double* __restrict__ x, int n_var,
                                   const double* __restrict__ p, int n_par, double* __restrict__ x_i, int B,
                                   const int32_t* __restrict__ pub_rows, int n_pub, double* __restrict__ x_send) {
  const int ns = lay.n_dim * lay.L;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (B + n_pub) * ns) return;
  const int row = i / ns, q = i - row * ns, k = q / lay.L;
  const int b = row < B ? row : pub_rows[row - B];
  const double v = x[(size_t)b * n_var + lay.x_spl + q] + p[(size_t)b * n_par + lay.p_rel + k];
  if (row < B) x_i[i] = v; else x_send[(size_t)(row - B) * ns + q] = v;
}

// one block per agent; thread r owns row r of the stacked vectors (n_all <= blockDim)
__global__ void __launch_bounds__(256)
admm_update_kernel(omgx_admm_layout lay, const double* __restrict__ x_ext, const int32_t* __restrict__ nbr,
                   const double* __restrict__ M, const double* __restrict__ F, double rho,
                   double* __restrict__ p, int n_par, double* __restrict__ z_ij, double* __restrict__ l_ij, int zl_stride,
                   double* __restrict__ res, double* __restrict__ sums, int* __restrict__ done,
                   const int32_t* __restrict__ pub_slot, double* __restrict__ zl_send, int send_stride) {
  extern __shared__ __align__(16) double lds[];
  const int ns = lay.n_dim * lay.L, nn = lay.n_nghb, na = (1 + nn) * ns;
  double* xa = lds; double* la = lds + na; double* zp = lds + 2 * na; double* va = lds + 3 * na;
  double* d1 = lds + 4 * na; double* d2 = lds + 5 * na; double* red = lds + 6 * na;
  const int b = blockIdx.x, r = threadIdx.x;
  double* pb = p + (size_t)b * n_par;
  if (r < na) {
    const int blk = r / ns, q = r - blk * ns;
    if (blk == 0) { xa[r] = x_ext[(size_t)b * ns + q]; la[r] = pb[lay.p_li + q]; zp[r] = pb[lay.p_zi + q]; }
    else {
      const int j = nbr[b * nn + blk - 1];
      xa[r] = x_ext[(size_t)j * ns + q];
      la[r] = l_ij[(size_t)b * zl_stride + (blk - 1) * ns + q];
      zp[r] = z_ij[(size_t)b * zl_stride + (blk - 1) * ns + q];
    }
    va[r] = xa[r] + la[r] / rho;
  }
  __syncthreads();
  double zr = 0.0, lr = 0.0;
  if (r < na) {
    const double* Mr = M + (size_t)r * na;
    for (int c = 0; c < na; ++c) zr += Mr[c] * va[c];
    lr = la[r] + rho * (xa[r] - zr);
    d1[r] = xa[r] - zr; d2[r] = zr - zp[r];
    const int blk = r / ns, q = r - blk * ns;
    if (blk == 0) { pb[lay.p_zi + q] = zr; pb[lay.p_li + q] = lr; }
    else {
      z_ij[(size_t)b * zl_stride + (blk - 1) * ns + q] = zr; l_ij[(size_t)b * zl_stride + (blk - 1) * ns + q] = lr;
      // a row another rank needs goes to the send buffer of the second exchange as well: [z_ij | l_ij]
      const int ps = pub_slot ? pub_slot[b] : -1;
      if (ps >= 0) {
        zl_send[(size_t)ps * send_stride + (blk - 1) * ns + q] = zr;
        zl_send[(size_t)ps * send_stride + nn * ns + (blk - 1) * ns + q] = lr;
      }
    }
  }
  __syncthreads();
  double pr = 0.0, dr = 0.0;
  if (r < na) {
    const double* Fr = F + (size_t)r * na;
    double a1 = 0.0, a2 = 0.0;
    for (int c = 0; c < na; ++c) { a1 += Fr[c] * d1[c]; a2 += Fr[c] * d2[c]; }
    pr = a1 * a1; dr = a2 * a2;
  }
  // block sum of (pr, dr)
  for (int off = 32; off > 0; off >>= 1) { pr += __shfl_down(pr, off, 64); dr += __shfl_down(dr, off, 64); }
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[2 * wave] = pr; red[2 * wave + 1] = dr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double P = 0.0, D = 0.0;
    for (int w = 0; w < nw; ++w) { P += red[2 * w]; D += red[2 * w + 1]; }
    D *= rho;
    res[3 * b] = P; res[3 * b + 1] = D; res[3 * b + 2] = rho * P + D;
  }
  if (!sums) return;
  // Fleet sums of the three residuals by the workgroup that finishes last (no second launch): thread t adds the
  // agents t, t + 256, ... in that order, then a fixed tree over the 256 partial sums -- the same bits whichever
  // workgroup happens to be the last one.
  __shared__ int last;
  if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); last = atomicAdd(done, 1) == (int)gridDim.x - 1 ? 1 : 0; }
  __syncthreads();
  if (!last) return;
  // (acquire at device scope: the other workgroups' res rows -- written before their release + counter increment --
  // are visible to plain loads from here on, which the compiler can keep in flight together)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const int B = gridDim.x;
  const double* rv = res;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll 4
  for (int a = threadIdx.x; a < B; a += blockDim.x) { s0 += rv[3 * a]; s1 += rv[3 * a + 1]; s2 += rv[3 * a + 2]; }
  double* t3 = lds;                       // (6 na + 16 doubles are there; 3 x 256 are needed: see the launch)
  t3[threadIdx.x] = s0; t3[256 + threadIdx.x] = s1; t3[512 + threadIdx.x] = s2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      t3[threadIdx.x] += t3[threadIdx.x + off]; t3[256 + threadIdx.x] += t3[256 + threadIdx.x + off];
      t3[512 + threadIdx.x] += t3[512 + threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { sums[0] = t3[0]; sums[1] = t3[256]; sums[2] = t3[512]; *done = 0; }
}

// (sum_rows: the residual sums every rank sent along with its rows, [n_sum_rows] rows of sum_stride doubles with the
// three sums in front; their total over the ranks goes to sums_out -- rank order, the same bits on every rank)
__global__ void admm_comm_kernel(omgx_admm_layout lay, const int32_t* __restrict__ nbr,
                                 const int32_t* __restrict__ slot, const double* __restrict__ z_ext,
                                 const double* __restrict__ l_ext, int zl_stride, double* __restrict__ p, int n_par, int B,
                                 const double* __restrict__ sum_rows, int n_sum_rows, int sum_stride, double* __restrict__ sums_out) {
  const int ns = lay.n_dim * lay.L, nn = lay.n_nghb;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 && sums_out) {
    double acc = 0.0;
    for (int r = 0; r < n_sum_rows; ++r) acc += sum_rows[(size_t)r * sum_stride + i];
    sums_out[i] = acc;
  }
  if (i >= B * nn * ns) return;
  const int b = i / (nn * ns), rem = i - b * nn * ns, k = rem / ns, q = rem - k * ns;
  const size_t src = (size_t)nbr[b * nn + k] * zl_stride + (size_t)slot[b * nn + k] * ns + q;
  p[(size_t)b * n_par + lay.p_zji + k * ns + q] = z_ext[src];
  p[(size_t)b * n_par + lay.p_lji + k * ns + q] = l_ext[src];
}

// ---------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------
struct omgx_batch {
  int device = 0, n_agents = 0;
  omgx::Dims dims;
  omgx::Tables dev;               // device pointers
  omgx::Opts opts;
  int kkt_doubles = 0;
  size_t lds_bytes = 0;
  int ws_mode = 0, n_slabs = 0;        // workspace placement (omgx::WS_*), HBM slabs (= grid cap)
  int threads = kThreads;              // workgroup size of the solve kernel
  int per_cu = 1;                      // workgroups (agents in flight) per CU the workspace allows
  int prio_iter = 0, stagger = 0; // straggler priority / start offset of the second workgroup of a CU (two per CU only)
  size_t slab_doubles = 0;
  double* d_slabs = nullptr;
  double* d_dw = nullptr;          // per-agent inertia correction carried between warm-started solves
  // round 6: the setup of every solve as a kernel of its own ahead of the solve kernel (ipm_prepare_kernel): one record per agent
  double* d_prep = nullptr; size_t prep_doubles = 0; size_t prep_lds = 0; bool prepare_on = false;
  int order_dw = 1;                // omgx_batch_order_by_iters: ties of the iteration count broken by the carried inertia correction (OMGX_ORDER_DW=0: developer switch)
  const int32_t* d_order = nullptr; // optional launch order (device pointer owned by the caller)
  const int32_t* pend_iters = nullptr; int32_t* pend_order = nullptr;   // omgx_batch_order_by_iters not launched yet
  int* d_next = nullptr;            // spill modes: counter of the dynamic slot hand-out
  int* d_admm_done = nullptr;       // admm_update_kernel: finished workgroups (fleet sums by the last one)
  const double* d_x0_alt = nullptr; // restart guesses [n_alt][n_agents][n_var] (device, owned by the caller)
  int n_alt = 0;
  int32_t* d_attempts = nullptr;    // optional [n_agents] (device, owned by the caller): restarts each agent used
  int64_t* d_stats = nullptr;       // optional [stats_slots][4] launch statistics (device, owned by the caller)
  int stats_slots = 0; long long stats_launch = 0;
  StoreArgs store = {};             // trajectories written by the solve kernel (omgx_batch_set_store); out == nullptr: off
  StoreArgs* d_store = nullptr;     // its copy in device memory (what the kernel reads)
  RolloutArgs* d_rollout = nullptr; RolloutStep* d_ro_steps = nullptr; int ro_steps_cap = 0; int32_t* d_ro_perm = nullptr;      // omgx_batch_rollout
  std::vector<int32_t> ro_perm_host;
  StopArgs* d_stop = nullptr;       // omgx_batch_set_stop: device copy of the arguments
  StopArgs stop_host = {};          // (and the host copy: omgx_batch_rollout hands it to its kernel inside RolloutArgs)
  bool stop_on = false;
  CenterArgs* d_center = nullptr;   // omgx_batch_set_center: device copy of the arguments (nullptr: off); the slot map behind it
  int32_t* d_pub_inv = nullptr;
  bool center_on = false;
  std::vector<void*> allocs;
  char* arena = nullptr; size_t arena_cap = 0, arena_used = 0;      // bump allocator of the plan's tables (arena_alloc)
  hipStream_t own_stream = nullptr, stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ext_ev0 = nullptr, ext_ev1 = nullptr;   // caller's events for the next solve launch (one shot)
  bool timed = false;
  bool timing = true;              // bracket every solve kernel with events (omgx_batch_set_timing)
  // staging buffers for host-pointer calls
  double *d_p = nullptr, *d_x0 = nullptr, *d_lb = nullptr, *d_ub = nullptr, *d_x = nullptr, *d_lam = nullptr;
  // Two-sided rows lb < g < ub (`basics/optilayer.py:634-666`): the solve kernel knows one-sided rows, so the library hands it
  // the template with every such row twice -- row r as g <= ub, a copy behind the last row as g >= lb -- and maps bounds and
  // multipliers between the caller's n_con_user rows and the kernel's dims.n_con rows around every solve (range_* kernels).
  int n_con_user = 0, n_range = 0;
  int32_t *d_range_src = nullptr, *d_range_dup = nullptr;      // [n_range] source row of copy k; [n_con_user] copy of row r (-1: none)
  double *d_lam_user = nullptr, *d_lb_user = nullptr, *d_ub_user = nullptr;   // staging of the caller's arrays (host-pointer calls)
  int32_t *d_status = nullptr, *d_iters = nullptr;
  long long* d_prof = nullptr;
  // shift tables (entries + T matrices) live in the handle: uploaded when they change (a receding-horizon loop
  // passes the same ones at every knot crossing), so that a shift is one stream-ordered launch
  // (a receding-horizon loop passes the same few sets at every knot crossing -- an ADMM fleet four of them: x, p, z_ij,
  // l_ij --: each set is uploaded once and found again by content, so that a shift is stream-ordered launches only)
  struct ShiftSet { std::vector<int32_t> ent; std::vector<double> T; int32_t* d_ent; double* d_T; unsigned long long used; };
  std::vector<ShiftSet> shift_sets; unsigned long long shift_clock = 0;
  int32_t* d_shift_ent = nullptr; double* d_shift_T = nullptr;      // the set the last staging call selected
  uint8_t* d_mask = nullptr;
  // omgx_batch_eval: Jacobian entry -> (row, variable), stored Hessian entry -> (address, variable a, variable b)
  std::vector<int32_t> ev_jrow, ev_jvar, ev_hess;
};

namespace {

// The ~60 tables of a plan live in one arena (chunks of 4 MiB, 256-byte aligned pieces) instead of sixty separately
// placed hipMalloc'ed buffers: every thread of every solve walks them.
int arena_alloc(omgx_batch* b, size_t bytes, void** out) {
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (!b->arena || b->arena_used + need > b->arena_cap) {
    const size_t cap = need > ((size_t)4 << 20) ? need : ((size_t)4 << 20);
    void* ptr = nullptr;
    HIPCHK(hipMalloc(&ptr, cap));
    b->allocs.push_back(ptr);
    b->arena = (char*)ptr; b->arena_cap = cap; b->arena_used = 0;
  }
  *out = b->arena + b->arena_used;
  b->arena_used += need;
  return OMGX_OK;
}

template <typename T>
int upload(omgx_batch* b, const T* src, size_t n, const T** dst) {
  void* ptr = nullptr;
  const size_t bytes = (n > 0 ? n : 1) * sizeof(T);
  { const int rc = arena_alloc(b, bytes, &ptr); if (rc != OMGX_OK) return rc; }
  if (n > 0) HIPCHK(hipMemcpy(ptr, src, n * sizeof(T), hipMemcpyHostToDevice));
  *dst = (const T*)ptr;
  return OMGX_OK;
}

template <typename T>
int dalloc(omgx_batch* b, size_t n, T** dst) {
  void* ptr = nullptr;
  HIPCHK(hipMalloc(&ptr, (n > 0 ? n : 1) * sizeof(T)));
  b->allocs.push_back(ptr);
  *dst = (T*)ptr;
  return OMGX_OK;
}

#define UP(field, count)                                             \
  do { int rc_ = upload(b, H.field, (size_t)(count), &b->dev.field); \
       if (rc_ != OMGX_OK) return rc_; } while (0)

// smallest spill mode whose LDS part fits one CU (WS_MODES: none does)
int pick_mode(const omgx::Dims& d, int kkt_doubles, size_t* lds_doubles, size_t* hbm_doubles) {
  int mode = 0;
  for (; mode <= omgx::WS_ROWS_HBM; ++mode) {
    omgx::work_split(d, kkt_doubles, mode, lds_doubles, hbm_doubles);
    if (*lds_doubles * sizeof(double) <= (size_t)kLdsLimit) break;
  }
  if (mode <= omgx::WS_ROWS_HBM) return mode;
  // the root block alone is too large for LDS: it stays in the slab (mode 6; compiled for the general instance only -- the
  // templates that get here are the ones with lifted auxiliaries, hundreds of equality rows in the root)
  omgx::work_split(d, kkt_doubles, omgx::WS_ROOT_HBM, lds_doubles, hbm_doubles);
  return (d.general && *lds_doubles * sizeof(double) <= (size_t)kLdsLimit) ? (int)omgx::WS_ROOT_HBM : (int)omgx::WS_MODES;
}

// The plan of a template for the workspace mode it gets: the spill modes store the leaf panels by columns
// (omgx_plan.h `col_major`), so their plan is built a second time once the mode is known.
// Templates on the register-resident wave path get the compact store; when their workspace then fits half a CU -- all
// of it, or with the Jacobian values (WS_JAC_HV) and the row values hv (WS_JAC_ONLY) in a slab -- two agents share a CU: *per_cu = 2, workgroups
// of 256 threads (a solve is latency bound: four waves are as fast as eight, and the second agent fills the gaps).
bool plan_for_mode(omgx::HostPlan& plan, const omgx_template& t, int* mode, size_t* lds_doubles, size_t* hbm_doubles, int* per_cu) {
  *per_cu = 1;
  if (!plan.build(t)) return false;
  if (plan.dims.wave_ok && !getenv("OMGX_NO_COMPACT")) {
    // (built in place: HostPlan::tables points into the plan's own vectors)
    plan = omgx::HostPlan();
    plan.owners = 256;        // (the assembly records dealt to the 256 threads of a two-per-CU workgroup)
    if (!plan.build(t, false, true)) return false;
    const int cand[3] = {omgx::WS_LDS, omgx::WS_JAC_HV, omgx::WS_JAC_ONLY};
    // (two per CU = workgroups of four waves: the substitutions gather 64 doubles per wave in the scratch behind the
    // matrix descriptors, four of the eight blocks suffice)
    plan.dims.col_doubles -= 4 * 64;
    for (int k = 0; k < 3; ++k) {
      omgx::work_split(plan.dims, plan.kkt_doubles, cand[k], lds_doubles, hbm_doubles);
      if (*lds_doubles * sizeof(double) <= (size_t)kLdsHalf && !getenv("OMGX_ONE_PER_CU")) { *mode = cand[k]; *per_cu = 2; return true; }
    }
    plan = omgx::HostPlan();
    if (!plan.build(t, false, true)) return false;
    // one agent per CU: everything in LDS, or the Jacobian values in a slab (the register-resident factorisation either way)
    for (int k = 0; k < 3; ++k) {
      omgx::work_split(plan.dims, plan.kkt_doubles, cand[k], lds_doubles, hbm_doubles);
      if (*lds_doubles * sizeof(double) <= (size_t)kLdsLimit && !(k >= 1 && getenv("OMGX_NO_JAC_ONLY_FULL"))) { *mode = cand[k]; return true; }
    }
    plan = omgx::HostPlan();
    if (!plan.build(t)) return false;
  }
  plan.dims.wave_ok = 0;      // the blocked routines (dense panels)
  *mode = pick_mode(plan.dims, plan.kkt_doubles, lds_doubles, hbm_doubles);
  if (const char* e = getenv("OMGX_FORCE_MODE")) {      // (developer knob: a deeper spill mode than the class needs)
    const int fm = atoi(e);
    if (*mode != omgx::WS_MODES && fm > *mode && fm <= omgx::WS_ROWS_HBM) *mode = fm;
  }
  if (*mode != omgx::WS_LDS && *mode != omgx::WS_MODES) {
    plan = omgx::HostPlan();
    if (!plan.build(t, true)) return false;
    omgx::work_split(plan.dims, plan.kkt_doubles, *mode, lds_doubles, hbm_doubles);
    // a spill class is bound by dependent accesses to its slab: when the part that stays in LDS fits half a CU, two
    // agents share the CU and hide each other's round trips
    if (*lds_doubles * sizeof(double) <= (size_t)kLdsHalf && getenv("OMGX_SPILL_PER_CU") && atoi(getenv("OMGX_SPILL_PER_CU")) == 2) *per_cu = 2;
  }
  return true;
}

typedef void (*ipm_rollout_t)(omgx::Dims, omgx::Tables, omgx::Opts, int, double*, double*, const double*, const double*, int, double*,
                              int32_t*, int32_t*, int, double*, size_t, double*, int*, const RolloutArgs*, int, const int32_t*, const StoreArgs*);
// (the classes of the wave path without quartic terms / cos / sin atoms: the receding-horizon classes that fit LDS)
static ipm_rollout_t rollout_kernel_for(int mode, int wave_ok, int general) {
  if (!wave_ok || general) return nullptr;
  switch (mode) {
    case omgx::WS_LDS: return ipm_rollout_kernel<omgx::WS_LDS, true, false>;
    case omgx::WS_JAC_ONLY: return ipm_rollout_kernel<omgx::WS_JAC_ONLY, true, false>;
    case omgx::WS_JAC_HV: return ipm_rollout_kernel<omgx::WS_JAC_HV, true, false>;
    default: return nullptr;
  }
}

int check_template(const omgx_template* t) {
  if (!t || t->n_var <= 0 || t->n_par < 0 || t->n_con < 0 || t->n_terms < 0 || t->n_eq < 0 || t->n_root_vars < 0 ||
      !t->row_ptr || (t->n_terms > 0 && (!t->t_coef || !t->t_slot || !t->t_var)) || (t->n_eq > 0 && !t->eq_rows) ||
      (t->n_root_vars > 0 && !t->root_vars)) { g_err = "bad template"; return OMGX_E_INVALID; }
  if (t->n_lift < 0 || (t->n_lift > 0 && (t->n_lift > t->n_var || t->lift_row0 < 0 || t->lift_row0 + t->n_lift > t->n_con))) { g_err = "bad template: lifted rows outside the template"; return OMGX_E_INVALID; }
  // the block table (optional): every entry inside its flat vector -- callers fill p / x0 and read x through these offsets
  // (`Point2Point::fillParameterDict / extractData`, export/point2point/Point2Point.cpp:263-294)
  if (t->n_blocks > 0) {
    if (!t->block_kind || !t->block_off || !t->block_rows || !t->block_cols) { g_err = "bad template: block table without its arrays"; return OMGX_E_INVALID; }
    for (int i = 0; i < t->n_blocks; ++i) {
      const int k = t->block_kind[i];
      const long long n = k == OMGX_BLOCK_VAR ? t->n_var : (k == OMGX_BLOCK_PAR ? t->n_par : (k == OMGX_BLOCK_CON ? t->n_con : -1));
      const long long off = t->block_off[i], sz = (long long)t->block_rows[i] * t->block_cols[i];
      if (n < 0 || off < 0 || t->block_rows[i] < 0 || t->block_cols[i] < 0 || off + sz > n) {
        char buf[160];
        snprintf(buf, sizeof buf, "bad template: block %d (kind %d, offset %lld, %d x %d) reaches outside its vector of %lld", i, k, off,
                 t->block_rows[i], t->block_cols[i], n);
        g_err = buf; return OMGX_E_INVALID;
      }
    }
  }
  return OMGX_OK;
}

// ---- two-sided rows: caller's rows <-> the kernel's rows ----------------------------------------------------------
__global__ void range_expand_bounds(const double* __restrict__ lb_u, const double* __restrict__ ub_u, double* __restrict__ lb_i,
                                    double* __restrict__ ub_i, int sets, int nu, int ni, const int32_t* __restrict__ src,
                                    const int32_t* __restrict__ dup) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= sets * ni) return;
  const int s = id / ni, r = id - s * ni;
  if (r < nu) {                                  // a two-sided row keeps its upper bound here
    lb_i[id] = dup[r] >= 0 ? -INFINITY : lb_u[(size_t)s * nu + r];
    ub_i[id] = ub_u[(size_t)s * nu + r];
  } else {                                       // its copy carries the lower bound
    lb_i[id] = lb_u[(size_t)s * nu + src[r - nu]];
    ub_i[id] = INFINITY;
  }
}
// multipliers in: lam_g of a two-sided row is positive when its upper bound is active, negative for the lower one
__global__ void range_expand_lam(const double* __restrict__ lam_u, double* __restrict__ lam_i, int B, int nu, int ni,
                                 const int32_t* __restrict__ src, const int32_t* __restrict__ dup) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * ni) return;
  const int b = id / ni, r = id - b * ni;
  if (r < nu) { const double v = lam_u[(size_t)b * nu + r]; lam_i[id] = dup[r] >= 0 ? fmax(v, 0.0) : v; }
  else lam_i[id] = fmin(lam_u[(size_t)b * nu + src[r - nu]], 0.0);
}
__global__ void range_contract_lam(const double* __restrict__ lam_i, double* __restrict__ lam_u, int B, int nu, int ni,
                                   const int32_t* __restrict__ dup) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * nu) return;
  const int b = id / nu, r = id - b * nu;
  lam_u[id] = lam_i[(size_t)b * ni + r] + (dup[r] >= 0 ? lam_i[(size_t)b * ni + dup[r]] : 0.0);
}

// The template with every two-sided row of its default bounds twice (see omgx_batch::n_range).  Host arrays owned by `own`.
struct ExpandedTemplate {
  omgx_template t;
  std::vector<int32_t> row_ptr, t_slot, t_var, src, dup;
  std::vector<double> t_coef, lb, ub;
};
static bool expand_range_rows(const omgx_template* in, ExpandedTemplate& e) {
  e.src.clear(); e.dup.assign(in->n_con, -1);
  if (!in->has_bounds || !in->lbg_def || !in->ubg_def) return false;
  std::vector<char> is_eq(in->n_con, 0);
  for (int k = 0; k < in->n_eq; ++k) if (in->eq_rows[k] >= 0 && in->eq_rows[k] < in->n_con) is_eq[in->eq_rows[k]] = 1;
  for (int r = 0; r < in->n_con; ++r)
    if (!is_eq[r] && std::isfinite(in->lbg_def[r]) && std::isfinite(in->ubg_def[r]) && in->lbg_def[r] < in->ubg_def[r]) {
      e.dup[r] = in->n_con + (int)e.src.size(); e.src.push_back(r);
    }
  if (e.src.empty()) return false;
  const int nu = in->n_con, ni = nu + (int)e.src.size(), W = OMGX_TERM_VARS;
  e.t = *in;
  e.row_ptr.assign(ni + 2, 0);
  auto put_row = [&](int from) {
    for (int i = in->row_ptr[from]; i < in->row_ptr[from + 1]; ++i) {
      e.t_coef.push_back(in->t_coef[i]); e.t_slot.push_back(in->t_slot[i]);
      for (int q = 0; q < W; ++q) e.t_var.push_back(in->t_var[(size_t)W * i + q]);
    }
  };
  int out = 0;
  for (int r = 0; r < nu; ++r) { put_row(r); e.row_ptr[++out] = (int)e.t_coef.size(); }
  for (int k = 0; k < (int)e.src.size(); ++k) { put_row(e.src[k]); e.row_ptr[++out] = (int)e.t_coef.size(); }
  put_row(nu); e.row_ptr[++out] = (int)e.t_coef.size();                 // the objective row stays last
  e.lb.resize(ni); e.ub.resize(ni);
  for (int r = 0; r < nu; ++r) { e.lb[r] = e.dup[r] >= 0 ? -INFINITY : in->lbg_def[r]; e.ub[r] = in->ubg_def[r]; }
  for (int k = 0; k < (int)e.src.size(); ++k) { e.lb[nu + k] = in->lbg_def[e.src[k]]; e.ub[nu + k] = INFINITY; }
  e.t_coef.push_back(0.0); e.t_slot.push_back(0); e.t_var.push_back(0);
  e.t.n_con = ni; e.t.n_terms = (int)e.t_coef.size() - 1;
  e.t.row_ptr = e.row_ptr.data(); e.t.t_coef = e.t_coef.data(); e.t.t_slot = e.t_slot.data(); e.t.t_var = e.t_var.data();
  e.t.lbg_def = e.lb.data(); e.t.ubg_def = e.ub.data();
  e.t.n_blocks = 0; e.t.block_names_len = 0;                            // (the block table describes the caller's rows)
  return true;
}

int build_batch(omgx_batch* b, const omgx_template* t) {
  omgx::HostPlan plan;
  size_t nl = 0, ng = 0;
  int mode = 0;
  if (!plan_for_mode(plan, *t, &mode, &nl, &ng, &b->per_cu)) { g_err = "inconsistent template: " + plan.error; return OMGX_E_INVALID; }
  b->dims = plan.dims;
  b->kkt_doubles = plan.kkt_doubles;
  if (mode == omgx::WS_MODES) {
    char buf[160];
    snprintf(buf, sizeof buf, "per-agent O(n_var) vectors (%zu B) exceed the %d B LDS of one CU", nl * sizeof(double), kLdsLimit);
    g_err = buf;
    return OMGX_E_TOOLARGE;
  }
  b->ws_mode = mode; b->lds_bytes = nl * sizeof(double); b->slab_doubles = ng;
  b->threads = b->per_cu >= 2 ? 256 : kThreads;
  if (const char* e = getenv("OMGX_THREADS")) { const int t2 = atoi(e); if (t2 == 256 || (t2 == 512 && b->per_cu < 2)) b->threads = t2; }      // (developer knob; the workspace of two per CU is sized for four waves)
  if (b->per_cu >= 2) { b->prio_iter = 2; b->stagger = 0; }
  if (const char* e = getenv("OMGX_PRIO_ITER")) b->prio_iter = atoi(e);      // (developer knobs)
  if (const char* e = getenv("OMGX_STAGGER")) b->stagger = atoi(e);
  if (const char* e = getenv("OMGX_ORDER_DW")) b->order_dw = atoi(e);
  const omgx::Tables& H = plan.tables;
  const omgx::Dims& d = plan.dims;
  {
    b->ev_jrow.assign(plan.je_row.begin(), plan.je_row.begin() + d.nnz_j);
    b->ev_jvar.resize(d.nnz_j);
    for (int e = 0; e < d.nnz_j; ++e) b->ev_jvar[e] = plan.order[plan.jr_pos[e]];
    for (int q1 = 0; q1 + 1 < d.N; ++q1)          // (positions but the phase-I variable t, the last one)
      for (int q2 = 0; q2 <= q1; ++q2) {
        const int32_t ad = plan.kkt_addr(q1, q2);
        if (ad >= 0) { b->ev_hess.push_back(ad); b->ev_hess.push_back(plan.order[q1]); b->ev_hess.push_back(plan.order[q2]); }
      }
  }
  UP(prog, 6 * d.n_prog); UP(knots, t->n_knots); UP(pp_ptr, t->n_pp + 1); UP(pm_coef, t->n_mono);
  UP(pm_ptr, t->n_mono + 1); UP(pm_atom, t->n_matom); UP(slot_pp, d.n_slots);
  // (packed monomial records: 16-byte MonoRec or, with 5..8 atoms per monomial, 24-byte MonoRec8 behind the same pointers)
  if (d.mono_packed == 2) {
    const omgx::MonoRec8* dev8 = nullptr;
    int rc8 = upload(b, plan.pm_rec8.data(), plan.pm_rec8.size(), &dev8); if (rc8 != OMGX_OK) return rc8;
    b->dev.pm_rec = (const omgx::MonoRec*)dev8;
    rc8 = upload(b, plan.sl_ell8.data(), plan.sl_ell8.size(), &dev8); if (rc8 != OMGX_OK) return rc8;
    b->dev.sl_ell = (const omgx::MonoRec*)dev8;
  } else { UP(pm_rec, plan.pm_rec.size()); UP(sl_ell, plan.sl_ell.size()); }
  UP(row_ptr, d.n_con + 2); UP(t_coef, d.n_terms); UP(t_slot, d.n_terms); UP(t_var, OMGX_TERM_VARS * d.n_terms);
  UP(order, d.N); UP(leaf_off, d.n_leaf + 1); UP(leaf_bw, plan.leaf_bw.size()); UP(blk, d.N);
  UP(eq_rows, d.n_eq); UP(eq_index, d.n_con);
  UP(cpl_ptr, d.n_leaf + 1); UP(cpl_idx, plan.cpl_idx.size()); UP(cpl_map, plan.cpl_map.size());
  UP(d_off, d.n_leaf + 1); UP(b_off, plan.b_off.size());
  UP(lf_w, plan.lf_w.size()); UP(lf_ldb, plan.lf_ldb.size()); UP(lf_band, plan.lf_band.size()); UP(lf_kind, plan.lf_kind.size()); UP(dl_pos, plan.dl_pos.size());
  UP(pair4, plan.pair4.size()); UP(eqe3, plan.eqe3.size());
  UP(je_row, plan.je_row.size()); UP(diag_addr, d.N); UP(tq_addr, plan.tq_addr.size());
  UP(reg_w, d.N); UP(je_rp, plan.je_rp.size());
  UP(slot_rng, plan.slot_rng.size());
  UP(row_perm, plan.row_perm.size()); UP(cs_ptr, plan.cs_ptr.size()); UP(cs_rec, plan.cs_rec.size());
  UP(obj_ent, plan.obj_ent.size());
  UP(ka_rec, plan.ka_rec.size()); UP(kh_rec, plan.kh_rec.size()); UP(kg_rec, plan.kg_rec.size());
  UP(ka_fix, plan.ka_fix.size()); UP(kg_fix, plan.kg_fix.size()); UP(kh_fix, plan.kh_fix.size());
  UP(rt_ell, plan.rt_ell.size()); UP(rt_glen, plan.rt_glen.size()); UP(jp_ell, plan.jp_ell.size()); UP(jp_glen, plan.jp_glen.size());
  UP(cs_ell, plan.cs_ell.size()); UP(cs_glen, plan.cs_glen.size()); UP(cs_col, plan.cs_col.size()); UP(cs_own, plan.cs_own.size());
  UP(jv_ell, plan.jv_ell.size()); UP(jv_own, plan.jv_own.size()); UP(jv_glen, plan.jv_glen.size());
  UP(ja_ell, plan.ja_ell.size()); UP(ja_own, plan.ja_own.size()); UP(ja_glen, plan.ja_glen.size());
  UP(sl_list, plan.sl_list.size()); UP(sl_glen, plan.sl_glen.size());
  UP(lift_rec, plan.lift_rec.size()); UP(lift_lev, plan.lift_lev.size());
  return OMGX_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int omgx_version(void) { return OMGX_VERSION; }
const char* omgx_last_error(void) { return g_err.c_str(); }

const char* omgx_status_string(int32_t s) {
  switch (s) {
    case OMGX_SOLVE_SUCCEEDED: return "Solve_Succeeded";
    case OMGX_MAX_ITER_EXCEEDED: return "Maximum_Iterations_Exceeded";
    case OMGX_INFEASIBLE_DETECTED: return "Infeasible_Problem_Detected";
    case OMGX_UNSUPPORTED_BOUNDS: return "Unsupported_Bounds";
    case OMGX_NUMERICAL_FAILURE: return "Numerical_Failure";
    default: return "Unknown";
  }
}

int omgx_plan_describe(const omgx_template* tpl, omgx_plan_info* info, int32_t* order) {
  if (!info) { g_err = "null argument"; return OMGX_E_INVALID; }
  int rc = check_template(tpl);
  if (rc != OMGX_OK) return rc;
  omgx::HostPlan plan;
  size_t nl = 0, ng = 0;
  int mode = 0, per_cu = 1;
  if (!plan_for_mode(plan, *tpl, &mode, &nl, &ng, &per_cu)) { g_err = "inconsistent template: " + plan.error; return OMGX_E_INVALID; }
  const omgx::Dims& d = plan.dims;
  memset(info, 0, sizeof *info);
  info->n_leaf = d.n_leaf; info->n_root = d.n_root; info->n_eq = d.n_eq; info->nnz_j = d.nnz_j;
  info->kkt_doubles = plan.kkt_doubles; info->wave_path = d.wave_ok;
  info->ws_mode = mode;
  info->lds_bytes = (int64_t)(nl * sizeof(double));
  for (int l = 0; l < d.n_leaf && l < OMGX_PLAN_MAX_LEAF; ++l) {
    info->leaf_size[l] = plan.leaf_off[l + 1] - plan.leaf_off[l];
    info->leaf_bw[l] = plan.leaf_bw[l];
    info->leaf_cpl[l] = plan.cpl_ptr[l + 1] - plan.cpl_ptr[l];
  }
  info->n_pairs = d.n_pairs; info->ka_len = d.ka_len; info->kh_len = d.kh_len; info->kg_len = d.kg_len;
  if (order) for (int q = 0; q < d.N; ++q) order[q] = plan.order[q];
  return OMGX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Template files: the counts and arrays of omgx_template, written by the Python front end
// (omgtools.backend.save_template) once per problem class and read by C/C++ callers -- the role the
// generated nlp.so plays for the reference's C++ export (`export/export.py:236-262`, loaded in
// `Point2Point.cpp:80-91`).  Layout: "OMGXTPL4", 16 int32 counts (the last three: block-table entries, the length of
// their names, whether default bounds follow; "OMGXTPL5", written for templates with lifted auxiliaries: 18 counts, + n_lift and
// lift_row0), then the arrays in struct order, the block table last ("OMGXTPL3" files -- three variables per term -- and "OMGXTPL2" files -- 13 counts, no table -- are still read).
namespace {
struct TplField { int kind; size_t count; const void* const* src; void** dst; };     // kind 0 int32, 1 double, 2 char

size_t tpl_fields(const omgx_template& t, omgx_template* m, TplField* f, int term_vars = OMGX_TERM_VARS) {
  const omgx_template& s = t;
  size_t k = 0;
#define OMGX_F(KD, NAME, CNT) f[k].kind = KD; f[k].count = (size_t)(CNT); f[k].src = (const void* const*)&s.NAME; f[k].dst = m ? (void**)&m->NAME : nullptr; ++k;
  OMGX_F(0, prog, 6 * (size_t)t.n_prog)      OMGX_F(1, knots, t.n_knots)       OMGX_F(0, pp_ptr, t.n_pp + 1)
  OMGX_F(1, pm_coef, t.n_mono)               OMGX_F(0, pm_ptr, t.n_mono + 1)   OMGX_F(0, pm_atom, t.n_matom)
  OMGX_F(0, slot_pp, t.n_slots)              OMGX_F(0, row_ptr, t.n_con + 2)   OMGX_F(1, t_coef, t.n_terms)
  OMGX_F(0, t_slot, t.n_terms)               OMGX_F(0, t_var, (size_t)term_vars * t.n_terms)
  OMGX_F(0, eq_rows, t.n_eq)                 OMGX_F(0, root_vars, t.n_root_vars)
  OMGX_F(2, block_names, t.block_names_len)  OMGX_F(0, block_kind, t.n_blocks)        OMGX_F(0, block_off, t.n_blocks)
  OMGX_F(0, block_rows, t.n_blocks)          OMGX_F(0, block_cols, t.n_blocks)
  OMGX_F(1, lbg_def, t.has_bounds ? t.n_con : 0)  OMGX_F(1, ubg_def, t.has_bounds ? t.n_con : 0)
#undef OMGX_F
  return k;
}
}  // namespace

int omgx_template_write(const omgx_template* tpl, const char* path) {
  int rc = check_template(tpl);
  if (rc != OMGX_OK) return rc;
  if (!path) { g_err = "null path"; return OMGX_E_INVALID; }
  FILE* fp = fopen(path, "wb");
  if (!fp) { g_err = std::string("cannot write ") + path; return OMGX_E_INVALID; }
  const int32_t counts[18] = {tpl->n_var, tpl->n_par, tpl->n_con, tpl->n_atoms, tpl->n_slots, tpl->n_terms, tpl->n_prog,
                              tpl->n_knots, tpl->n_pp, tpl->n_mono, tpl->n_matom, tpl->n_eq, tpl->n_root_vars,
                              tpl->n_blocks, tpl->block_names_len, tpl->has_bounds ? 1 : 0, tpl->n_lift, tpl->lift_row0};
  // (a template without lifted auxiliaries is written as before: "OMGXTPL4", 16 counts)
  const bool v5 = tpl->n_lift > 0;
  bool ok = fwrite(v5 ? "OMGXTPL5" : "OMGXTPL4", 1, 8, fp) == 8 && fwrite(counts, sizeof(int32_t), v5 ? 18 : 16, fp) == (size_t)(v5 ? 18 : 16);
  TplField f[24];
  const size_t nf = tpl_fields(*tpl, nullptr, f);
  for (size_t i = 0; i < nf && ok; ++i) {
    const size_t sz = f[i].kind == 1 ? sizeof(double) : (f[i].kind == 2 ? 1 : sizeof(int32_t));
    if (f[i].count && fwrite(*f[i].src, sz, f[i].count, fp) != f[i].count) ok = false;
  }
  if (fclose(fp) != 0) ok = false;
  if (!ok) { g_err = std::string("short write to ") + path; return OMGX_E_INVALID; }
  return OMGX_OK;
}

void omgx_template_free(omgx_template* t) {
  if (!t) return;
  TplField f[24];
  const size_t nf = tpl_fields(*t, t, f);
  for (size_t i = 0; i < nf; ++i) free(*f[i].dst);
  free(t);
}

int omgx_template_read(const char* path, omgx_template** out) {
  if (!path || !out) { g_err = "null argument"; return OMGX_E_INVALID; }
  *out = nullptr;
  FILE* fp = fopen(path, "rb");
  if (!fp) { g_err = std::string("cannot read ") + path; return OMGX_E_INVALID; }
  char magic[8];
  int32_t c[18] = {0};
  const bool head = fread(magic, 1, 8, fp) == 8;
  const int file_version = !head ? 0 : (memcmp(magic, "OMGXTPL5", 8) == 0 ? 5 : (memcmp(magic, "OMGXTPL4", 8) == 0 ? 4 : (memcmp(magic, "OMGXTPL3", 8) == 0 ? 3 : (memcmp(magic, "OMGXTPL2", 8) == 0 ? 2 : 0))));
  const int n_counts = file_version >= 5 ? 18 : (file_version >= 3 ? 16 : (file_version == 2 ? 13 : 0));
  const int file_tv = file_version >= 4 ? OMGX_TERM_VARS : 3;
  if (n_counts == 0 || fread(c, sizeof(int32_t), n_counts, fp) != (size_t)n_counts) {
    fclose(fp); g_err = std::string(path) + " is not an omgx template file"; return OMGX_E_INVALID;
  }
  for (int i = 0; i < 18; ++i) if (c[i] < 0 || c[i] > (1 << 26)) { fclose(fp); g_err = "template file: bad counts"; return OMGX_E_INVALID; }
  omgx_template* t = (omgx_template*)calloc(1, sizeof(omgx_template));
  if (!t) { fclose(fp); g_err = "out of memory"; return OMGX_E_INVALID; }
  t->n_var = c[0]; t->n_par = c[1]; t->n_con = c[2]; t->n_atoms = c[3]; t->n_slots = c[4]; t->n_terms = c[5]; t->n_prog = c[6];
  t->n_knots = c[7]; t->n_pp = c[8]; t->n_mono = c[9]; t->n_matom = c[10]; t->n_eq = c[11]; t->n_root_vars = c[12];
  t->n_blocks = c[13]; t->block_names_len = c[14]; t->has_bounds = c[15];
  t->n_lift = c[16]; t->lift_row0 = c[17];
  TplField f[24];
  const size_t nf = tpl_fields(*t, t, f, file_tv);
  bool ok = true;
  for (size_t i = 0; i < nf; ++i) {
    const size_t sz = f[i].kind == 1 ? sizeof(double) : (f[i].kind == 2 ? 1 : sizeof(int32_t));
    *f[i].dst = calloc(f[i].count + 1, sz);                  // (+1: an empty array still gets an address)
    if (!*f[i].dst) { ok = false; continue; }
    if (ok && f[i].count && fread(*f[i].dst, sz, f[i].count, fp) != f[i].count) ok = false;
  }
  fclose(fp);
  if (ok && file_tv != OMGX_TERM_VARS) {
    // an older file: three variables per term
    int32_t* wide = (int32_t*)calloc((size_t)OMGX_TERM_VARS * t->n_terms + 1, sizeof(int32_t));
    if (!wide) ok = false;
    else {
      for (int i = 0; i < t->n_terms; ++i)
        for (int k = 0; k < OMGX_TERM_VARS; ++k) wide[OMGX_TERM_VARS * i + k] = k < file_tv ? t->t_var[file_tv * i + k] : -1;
      free((void*)t->t_var);
      t->t_var = wide;
    }
  }
  if (!ok) { omgx_template_free(t); g_err = std::string("truncated template file ") + path; return OMGX_E_INVALID; }
  const int rc = check_template(t);
  if (rc != OMGX_OK) { omgx_template_free(t); return rc; }
  *out = t;
  return OMGX_OK;
}

// ---- block table -----------------------------------------------------------------------------------------
namespace {
// name of block i (its offset inside block_names), or nullptr when the table is malformed
const char* block_name(const omgx_template* t, int i) {
  if (!t->block_names || t->block_names_len <= 0 || t->block_names[t->block_names_len - 1] != 0) return nullptr;
  const char* p = t->block_names;
  const char* end = t->block_names + t->block_names_len;
  for (int k = 0; k < i; ++k) { p += strlen(p) + 1; if (p >= end) return nullptr; }
  return p < end ? p : nullptr;
}
}  // namespace

int omgx_template_n_blocks(const omgx_template* tpl, int32_t kind) {
  if (!tpl) { g_err = "null template"; return OMGX_E_INVALID; }
  int n = 0;
  for (int i = 0; i < tpl->n_blocks; ++i) if (tpl->block_kind[i] == kind) ++n;
  return n;
}

int omgx_template_block_at(const omgx_template* tpl, int32_t kind, int32_t i, const char** name, int32_t* off,
                           int32_t* rows, int32_t* cols) {
  if (!tpl || i < 0) { g_err = "bad argument"; return OMGX_E_INVALID; }
  int n = 0;
  for (int k = 0; k < tpl->n_blocks; ++k) {
    if (tpl->block_kind[k] != kind) continue;
    if (n++ == i) {
      const char* nm = block_name(tpl, k);
      if (!nm) { g_err = "malformed block table"; return OMGX_E_INVALID; }
      if (name) *name = nm;
      if (off) *off = tpl->block_off[k];
      if (rows) *rows = tpl->block_rows[k];
      if (cols) *cols = tpl->block_cols[k];
      return OMGX_OK;
    }
  }
  g_err = "block index out of range"; return OMGX_E_INVALID;
}

int omgx_template_block(const omgx_template* tpl, int32_t kind, const char* name, int32_t* off, int32_t* rows,
                        int32_t* cols) {
  if (!tpl || !name) { g_err = "null argument"; return OMGX_E_INVALID; }
  for (int k = 0; k < tpl->n_blocks; ++k) {
    const char* nm = tpl->block_kind[k] == kind ? block_name(tpl, k) : nullptr;
    if (nm && strcmp(nm, name) == 0) {
      if (off) *off = tpl->block_off[k];
      if (rows) *rows = tpl->block_rows[k];
      if (cols) *cols = tpl->block_cols[k];
      return OMGX_OK;
    }
  }
  g_err = std::string("the template has no entry named ") + name; return OMGX_E_INVALID;
}

void omgx_default_options(omgx_options* o) {
  o->tol = 1e-3; o->max_iter = 300; o->mu_init = 0.1; o->kappa_push = 1.0;
  o->nu_init = 100.0; o->scale_gmax = 100.0; o->warm_start = 0; o->kappa_warm = 1e-3;
  o->dw_leaf_ratio_cold = 1.0; o->warm_mu_factor = 1.0; o->warm_z_floor = 0.1; o->warm_z_cap = 0.01; o->max_soc = 1; o->hess_approx = 0; o->compl_inf_tol = 0.0; o->constr_viol_tol = 0.0; o->refine = 0;
}

int omgx_batch_create(const omgx_template* tpl, int32_t n_agents, int32_t device, omgx_batch** out) {
  if (!tpl || !out || n_agents <= 0 || device < 0) { g_err = "bad argument"; return OMGX_E_INVALID; }
  { const int rc0 = check_template(tpl); if (rc0 != OMGX_OK) return rc0; }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device >= count) {
    g_err = "no usable HIP device (the solve path has no CPU fallback)";
    return OMGX_E_NODEVICE;
  }
  HIPCHK(hipSetDevice(device));
  omgx_batch* b = new omgx_batch();
  b->device = device; b->n_agents = n_agents;
  omgx_options o; omgx_default_options(&o);
  b->opts = {o.tol, o.max_iter, o.mu_init, o.kappa_push, o.nu_init, o.scale_gmax, o.warm_start, o.kappa_warm, o.dw_leaf_ratio_cold, 0, o.warm_mu_factor, o.warm_z_floor, o.warm_z_cap, o.max_soc, o.hess_approx, o.compl_inf_tol, o.constr_viol_tol, o.refine};
  ExpandedTemplate ex;
  const bool ranged = expand_range_rows(tpl, ex);
  b->n_con_user = tpl->n_con; b->n_range = ranged ? (int)ex.src.size() : 0;
  int rc = build_batch(b, ranged ? &ex.t : tpl);
  if (rc != OMGX_OK) { omgx_batch_destroy(b); return rc; }
  const omgx::Dims& d = b->dims;
  if (ranged) {
    if (hipMalloc((void**)&b->d_range_src, ex.src.size() * sizeof(int32_t)) != hipSuccess ||
        hipMalloc((void**)&b->d_range_dup, ex.dup.size() * sizeof(int32_t)) != hipSuccess ||
        hipMemcpy(b->d_range_src, ex.src.data(), ex.src.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b->d_range_dup, ex.dup.data(), ex.dup.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
      g_err = "hipMalloc failed"; omgx_batch_destroy(b); return OMGX_E_HIP;
    }
    if ((rc = dalloc(b, (size_t)n_agents * b->n_con_user, &b->d_lam_user)) || (rc = dalloc(b, (size_t)n_agents * b->n_con_user, &b->d_lb_user)) ||
        (rc = dalloc(b, (size_t)n_agents * b->n_con_user, &b->d_ub_user))) { omgx_batch_destroy(b); return rc; }
  }
  if ((rc = dalloc(b, (size_t)n_agents * d.n_par, &b->d_p)) || (rc = dalloc(b, (size_t)n_agents * d.n_var, &b->d_x0)) ||
      (rc = dalloc(b, (size_t)n_agents * d.n_con, &b->d_lb)) || (rc = dalloc(b, (size_t)n_agents * d.n_con, &b->d_ub)) ||
      (rc = dalloc(b, (size_t)n_agents * d.n_var, &b->d_x)) || (rc = dalloc(b, (size_t)n_agents * d.n_con, &b->d_lam)) ||
      (rc = dalloc(b, (size_t)n_agents, &b->d_status)) || (rc = dalloc(b, (size_t)n_agents, &b->d_iters)) ||
      (rc = dalloc(b, (size_t)n_agents * omgx::PH_COUNT, &b->d_prof)) || (rc = dalloc(b, (size_t)n_agents, &b->d_dw))) {
    omgx_batch_destroy(b); return rc;
  }
  if (hipStreamCreate(&b->own_stream) != hipSuccess || hipEventCreate(&b->ev0) != hipSuccess ||
      hipEventCreate(&b->ev1) != hipSuccess) { g_err = "stream/event creation failed"; omgx_batch_destroy(b); return OMGX_E_HIP; }
  b->stream = b->own_stream;
  if (hipMemset(b->d_dw, 0, (size_t)n_agents * sizeof(double)) != hipSuccess) { g_err = "hipMemset failed"; omgx_batch_destroy(b); return OMGX_E_HIP; }
  // (the attribute belongs to the kernel, not to the handle: keep the largest request of the process)
  static int lds_reserved[4 * omgx::WS_MODES] = {0};
  int& reserved = lds_reserved[4 * b->ws_mode + (b->dims.wave_ok ? 1 : 0) + (b->dims.general ? 2 : 0)];
  if ((int)b->lds_bytes > reserved) reserved = (int)b->lds_bytes;
  if (hipFuncSetAttribute((const void*)ipm_kernel_for(b->ws_mode, b->dims.wave_ok, b->dims.general), hipFuncAttributeMaxDynamicSharedMemorySize,
                          reserved) != hipSuccess) {
    g_err = "cannot reserve dynamic LDS for ipm_solve_kernel"; omgx_batch_destroy(b); return OMGX_E_HIP;
  }
  if (ipm_kernel_for(b->ws_mode, b->dims.wave_ok, b->dims.general, 1) != ipm_kernel_for(b->ws_mode, b->dims.wave_ok, b->dims.general) &&
      hipFuncSetAttribute((const void*)ipm_kernel_for(b->ws_mode, b->dims.wave_ok, b->dims.general, 1), hipFuncAttributeMaxDynamicSharedMemorySize,
                          reserved) != hipSuccess) {      // (the instance with the refinement of regularised steps, omgx_options.refine)
    g_err = "cannot reserve dynamic LDS for ipm_solve_kernel"; omgx_batch_destroy(b); return OMGX_E_HIP;
  }
  if (ipm_rollout_t rk = rollout_kernel_for(b->ws_mode, b->dims.wave_ok, b->dims.general)) {
    if (hipFuncSetAttribute((const void*)rk, hipFuncAttributeMaxDynamicSharedMemorySize, reserved) != hipSuccess) {
      g_err = "cannot reserve dynamic LDS for ipm_rollout_kernel"; omgx_batch_destroy(b); return OMGX_E_HIP;
    }
  }
  {
    // Persistent workgroups, as many as fit the chip at once (one or two per CU), that take their agents from an atomic
    // counter -- in every mode: solves differ by a factor of several in their iteration counts.  The workgroups of the
    // spill modes own a slab each.
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_err = "hipGetDeviceProperties failed"; omgx_batch_destroy(b); return OMGX_E_HIP; }
    int per_cu = (int)((size_t)kLdsLimit / (b->lds_bytes > 0 ? b->lds_bytes : 1));
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    if (b->ws_mode == omgx::WS_LDS || b->ws_mode == omgx::WS_JAC_ONLY || b->ws_mode == omgx::WS_JAC_HV) per_cu = b->per_cu;      // (bound by the registers of 512-thread workgroups otherwise)
    int slabs = prop.multiProcessorCount * per_cu;
    if (slabs > n_agents) slabs = n_agents;
    b->n_slabs = slabs;
    if (b->slab_doubles > 0 && (rc = dalloc(b, (size_t)slabs * b->slab_doubles, &b->d_slabs))) { omgx_batch_destroy(b); return rc; }
    if ((rc = dalloc(b, (size_t)2, &b->d_next))) { omgx_batch_destroy(b); return rc; }
    if (hipMemset(b->d_next, 0, 2 * sizeof(int)) != hipSuccess) { g_err = "hipMemset failed"; omgx_batch_destroy(b); return OMGX_E_HIP; }
  }
  {
    // the setup kernel's records (omgx::prep_layout: ~43 KB per agent for config 2) and its LDS (atoms, knots, slots, x)
    b->prep_doubles = (size_t)omgx::prep_layout(d).total;
    b->prep_lds = prepare_lds_doubles(d) * sizeof(double);
    // OFF by default (omgx_batch_set_prepare / OMGX_PREPARE=1 switch it on).  Measured on the 1024-agent benchmark batch (round 6,
    // profiles/r06_prepare_ab.txt): the solve kernel drops from 341 k to 277 k cycles per warm-started solve (its setup phase 77 k ->
    // 6 k), but the setup kernel takes 64 us for the 1024 agents -- every workgroup walks the same chain of ~80 dependent table
    // loads whatever the occupancy, and at the head of the solve kernel that chain already overlaps with the agent that shares
    // the CU -- against 43 us saved: 1.73-1.74 M against 1.74-1.79 M solves/s with one launch per step.
    const char* env = getenv("OMGX_PREPARE");
    if (env && env[0] == '1') { const int rcp = omgx_batch_set_prepare(b, 1); if (rcp != OMGX_OK) { omgx_batch_destroy(b); return rcp; } }
  }
  *out = b;
  return OMGX_OK;
}

int omgx_batch_set_prepare(omgx_batch* b, int32_t on) {
  if (!b) return OMGX_E_INVALID;
  if (!on) { b->prepare_on = false; return OMGX_OK; }
  if (b->prep_lds > (size_t)kLdsLimit) { g_err = "the setup kernel's LDS (atoms, knots, slots, x) does not fit one CU for this template"; return OMGX_E_INVALID; }
  if (!b->d_prep) {      // the records: allocated when the kernel is first asked for
    HIPCHK(hipSetDevice(b->device));
    const int rc = dalloc(b, (size_t)b->n_agents * b->prep_doubles, &b->d_prep);
    if (rc != OMGX_OK) return rc;
    const void* pk = b->dims.general ? (const void*)ipm_prepare_kernel<true> : (const void*)ipm_prepare_kernel<false>;
    static int prep_reserved[2] = {0, 0};
    int& res = prep_reserved[b->dims.general ? 1 : 0];
    if ((int)b->prep_lds > res) res = (int)b->prep_lds;
    if (hipFuncSetAttribute(pk, hipFuncAttributeMaxDynamicSharedMemorySize, res) != hipSuccess) { g_err = "cannot reserve dynamic LDS for ipm_prepare_kernel"; return OMGX_E_HIP; }
  }
  b->prepare_on = true;
  return OMGX_OK;
}

int omgx_batch_set_stop(omgx_batch* b, int32_t o_state0, int32_t o_input0, int32_t o_poseT, int32_t n_dim, double stop_tol, int32_t* under_way) {
  if (!b) { g_err = "bad argument"; return OMGX_E_INVALID; }
  if (!under_way) { b->stop_on = false; return OMGX_OK; }
  b->stop_on = false;      // (a failed registration leaves the rule OFF)
  const int np = b->dims.n_par;
  if (n_dim <= 0 || o_state0 < 0 || o_input0 < 0 || o_poseT < 0 || o_state0 + n_dim > np || o_input0 + n_dim > np || o_poseT + n_dim > np ||
      !(stop_tol >= 0.0)) { g_err = "stop rule: parameter offsets out of range or a negative tolerance"; return OMGX_E_INVALID; }
  HIPCHK(hipSetDevice(b->device));
  if (!b->d_stop) HIPCHK(hipMalloc((void**)&b->d_stop, sizeof(StopArgs)));
  StopArgs sa;
  sa.o_state = o_state0; sa.o_input = o_input0; sa.o_pose = o_poseT; sa.n_dim = n_dim; sa.tol = stop_tol; sa.under_way = under_way;
  HIPCHK(hipMemcpyAsync(b->d_stop, &sa, sizeof(StopArgs), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  b->stop_host = sa;
  b->stop_on = true;
  return OMGX_OK;
}

void omgx_batch_destroy(omgx_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  for (void* p : b->allocs) (void)hipFree(p);
  if (b->d_store) (void)hipFree(b->d_store);
  if (b->d_center) (void)hipFree(b->d_center);
  if (b->d_stop) (void)hipFree(b->d_stop);
  if (b->d_rollout) (void)hipFree(b->d_rollout);
  if (b->d_ro_steps) (void)hipFree(b->d_ro_steps);
  if (b->d_ro_perm) (void)hipFree(b->d_ro_perm);
  if (b->d_pub_inv) (void)hipFree(b->d_pub_inv);
  if (b->d_range_src) (void)hipFree(b->d_range_src);
  if (b->d_range_dup) (void)hipFree(b->d_range_dup);
  for (auto& ss : b->shift_sets) { if (ss.d_ent) (void)hipFree(ss.d_ent); if (ss.d_T) (void)hipFree(ss.d_T); }
  if (b->d_mask) (void)hipFree(b->d_mask);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
  delete b;
}

int omgx_batch_set_options(omgx_batch* b, const omgx_options* o) {
  if (!b || !o || !(o->tol > 0) || o->max_iter < 0) { g_err = "bad options"; return OMGX_E_INVALID; }
  b->opts = {o->tol, o->max_iter, o->mu_init, o->kappa_push, o->nu_init, o->scale_gmax, o->warm_start, o->kappa_warm,
             o->dw_leaf_ratio_cold > 0 ? o->dw_leaf_ratio_cold : 1.0, 0, o->warm_mu_factor >= 0 ? o->warm_mu_factor : 0.0,
             o->warm_z_floor >= 0 ? o->warm_z_floor : 0.0, o->warm_z_cap >= 0 ? o->warm_z_cap : 0.0, o->max_soc > 0 ? (o->max_soc > 8 ? 8 : o->max_soc) : 0, o->hess_approx > 0 ? 1 : 0,
             o->compl_inf_tol > 0 ? o->compl_inf_tol : 0.0, o->constr_viol_tol > 0 ? o->constr_viol_tol : 0.0, o->refine > 0 ? 1 : 0};
  return OMGX_OK;
}

int omgx_batch_set_stream(omgx_batch* b, void* s) {
  if (!b) return OMGX_E_INVALID;
  b->stream = s ? (hipStream_t)s : b->own_stream;
  return OMGX_OK;
}

static int flush_order(omgx_batch* b) {
  if (!b->pend_iters) return OMGX_OK;
  hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, b->stream, b->pend_iters, (const double*)(b->order_dw ? b->d_dw : nullptr), b->pend_order, b->n_agents);
  b->pend_iters = nullptr; b->pend_order = nullptr;
  HIPCHK(hipGetLastError());
  return OMGX_OK;
}

int omgx_batch_set_order(omgx_batch* b, const int32_t* order_device) {
  if (!b) return OMGX_E_INVALID;
  b->d_order = order_device;
  b->pend_iters = nullptr; b->pend_order = nullptr;
  return OMGX_OK;
}

int omgx_batch_set_restarts(omgx_batch* b, const double* x0_alt_device, int32_t n_alt, int32_t* attempts_device) {
  if (!b || n_alt < 0 || (n_alt > 0 && !x0_alt_device)) { g_err = "bad argument"; return OMGX_E_INVALID; }
  b->d_x0_alt = n_alt > 0 ? x0_alt_device : nullptr;
  b->n_alt = n_alt;
  b->d_attempts = attempts_device;
  return OMGX_OK;
}

int omgx_batch_order_by_iters(omgx_batch* b, const int32_t* iters_device, int32_t* order_device) {
  if (!b || !iters_device || !order_device) { g_err = "null argument"; return OMGX_E_INVALID; }
  // deferred: the next omgx_batch_predict(_ex) launch carries the ordering as one more workgroup; a solve that comes
  // first launches order_kernel itself (flush_order)
  b->pend_iters = iters_device; b->pend_order = order_device;
  b->d_order = order_device;
  return OMGX_OK;
}

int omgx_batch_lds_bytes(const omgx_batch* b) { return b ? (int)b->lds_bytes : OMGX_E_INVALID; }

int omgx_batch_workspace(const omgx_batch* b, int32_t* mode, int64_t* lds_bytes, int64_t* hbm_bytes_per_slab, int32_t* n_slabs) {
  if (!b) return OMGX_E_INVALID;
  if (mode) *mode = b->ws_mode;
  if (lds_bytes) *lds_bytes = (int64_t)b->lds_bytes;
  if (hbm_bytes_per_slab) *hbm_bytes_per_slab = (int64_t)(b->slab_doubles * sizeof(double));
  if (n_slabs) *n_slabs = b->n_slabs;
  return OMGX_OK;
}

int omgx_batch_solve(omgx_batch* b, const double* p, const double* x0, const double* lbg, const double* ubg,
                     double* x, double* lam_g, int32_t* status, int32_t* iters, int32_t flags) {
  if (!b || !p || !x0 || !lbg || !ubg || !x || !lam_g || !status || !iters) { g_err = "null argument"; return OMGX_E_INVALID; }
  HIPCHK(hipSetDevice(b->device));
  const omgx::Dims& d = b->dims;
  const int B = b->n_agents;
  const bool dev = flags & OMGX_PTR_DEVICE, shared = flags & OMGX_BOUNDS_SHARED;
  const bool bdev = (flags & OMGX_BOUNDS_DEVICE) != 0;
  // (nu: rows of the caller's arrays; d.n_con: rows of the kernel's -- more when two-sided rows were doubled, omgx_batch::n_range)
  const int nu = b->n_con_user, ni = d.n_con;
  const bool ranged = b->n_range > 0;
  const size_t nb = (shared ? 1 : (size_t)B) * nu;
  const double *kp = p, *kx0 = x0, *klb = lbg, *kub = ubg;
  double *kx = x, *klam = lam_g; int32_t *kst = status, *kit = iters;
  const bool lam_in = b->opts.warm_start || (flags & OMGX_ONLY_FAILED);
  double* lam_user_dev = ranged ? (dev ? lam_g : b->d_lam_user) : nullptr;       // the caller's multipliers on the device
  if (!dev) {
    HIPCHK(hipMemcpyAsync(b->d_p, p, (size_t)B * d.n_par * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_x0, x0, (size_t)B * d.n_var * sizeof(double), hipMemcpyHostToDevice, b->stream));
    double* lam_to = ranged ? b->d_lam_user : b->d_lam;
    if (b->opts.warm_start) {
      HIPCHK(hipMemcpyAsync(lam_to, lam_g, (size_t)B * nu * sizeof(double), hipMemcpyHostToDevice, b->stream));
      HIPCHK(hipMemcpyAsync(b->d_status, status, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
    }
    if (flags & OMGX_ONLY_FAILED) {       // the skipped agents keep what the caller's buffers hold
      HIPCHK(hipMemcpyAsync(b->d_x, x, (size_t)B * d.n_var * sizeof(double), hipMemcpyHostToDevice, b->stream));
      HIPCHK(hipMemcpyAsync(lam_to, lam_g, (size_t)B * nu * sizeof(double), hipMemcpyHostToDevice, b->stream));
      HIPCHK(hipMemcpyAsync(b->d_status, status, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
      HIPCHK(hipMemcpyAsync(b->d_iters, iters, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
    }
    kp = b->d_p; kx0 = b->d_x0; kx = b->d_x; klam = b->d_lam; kst = b->d_status; kit = b->d_iters;
  }
  if (ranged) {
    klam = b->d_lam;
    if (lam_in)
      hipLaunchKernelGGL(range_expand_lam, dim3((B * ni + 255) / 256), dim3(256), 0, b->stream, (const double*)lam_user_dev, b->d_lam, B, nu, ni,
                         (const int32_t*)b->d_range_src, (const int32_t*)b->d_range_dup);
  }
  if (!bdev) {
    double* lb_to = ranged ? b->d_lb_user : b->d_lb;
    double* ub_to = ranged ? b->d_ub_user : b->d_ub;
    HIPCHK(hipMemcpyAsync(lb_to, lbg, nb * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(ub_to, ubg, nb * sizeof(double), hipMemcpyHostToDevice, b->stream));
    klb = lb_to; kub = ub_to;
  }
  if (ranged) {
    const int sets = shared ? 1 : B;
    hipLaunchKernelGGL(range_expand_bounds, dim3((sets * ni + 255) / 256), dim3(256), 0, b->stream, klb, kub, b->d_lb, b->d_ub, sets, nu, ni,
                       (const int32_t*)b->d_range_src, (const int32_t*)b->d_range_dup);
    klb = b->d_lb; kub = b->d_ub;
  }
  { const int rc_o = flush_order(b); if (rc_o != OMGX_OK) return rc_o; }
  // Timing events ride on the dispatch packet of the solve kernel (hipExtLaunchKernelGGL: the packet's own begin / end
  // stamps) -- separate hipEventRecord calls around it cost two more packets, ~25 us of stream time per solve.
  // The caller's pair (omgx_batch_set_launch_events, one launch) goes first, else the handle's own when timing is on.
  hipEvent_t e0 = b->ext_ev0 ? b->ext_ev0 : (b->timing ? b->ev0 : nullptr);
  hipEvent_t e1 = b->ext_ev0 ? b->ext_ev1 : (b->timing ? b->ev1 : nullptr);
  b->timed = b->timing && !b->ext_ev0;
  b->ext_ev0 = b->ext_ev1 = nullptr;
  b->opts.prio_iter = b->prio_iter;
  const bool prepared = b->prepare_on && b->d_prep && !b->stop_on;      // (the stop rule is the solve kernel's: it does its own setup then)
  if (prepared) {
    // the setup of all B solves, many workgroups per CU; the begin stamp of the caller's / the handle's event pair rides on this
    // launch, the end stamp on the solve kernel's: the pair brackets both
    if (b->dims.general)
      hipExtLaunchKernelGGL(ipm_prepare_kernel<true>, dim3(B), dim3(b->threads), (uint32_t)b->prep_lds, b->stream, e0, nullptr, 0u, d, b->dev, b->opts,
                            kp, kx0, klb, kub, shared ? 1 : 0, (const double*)klam, (const int32_t*)kst, B, b->d_prep, b->prep_doubles, (flags & OMGX_ONLY_FAILED) ? 1 : 0);
    else
      hipExtLaunchKernelGGL(ipm_prepare_kernel<false>, dim3(B), dim3(b->threads), (uint32_t)b->prep_lds, b->stream, e0, nullptr, 0u, d, b->dev, b->opts,
                            kp, kx0, klb, kub, shared ? 1 : 0, (const double*)klam, (const int32_t*)kst, B, b->d_prep, b->prep_doubles, (flags & OMGX_ONLY_FAILED) ? 1 : 0);
    HIPCHK(hipGetLastError());
    e0 = nullptr;
  }
  hipExtLaunchKernelGGL(ipm_kernel_for(b->ws_mode, b->dims.wave_ok, b->dims.general, b->opts.refine), dim3(b->n_slabs), dim3(b->threads), (uint32_t)b->lds_bytes, b->stream,
                        e0, e1, 0u, d, b->dev,
                        b->opts, b->kkt_doubles, kp, kx0, klb, kub, shared ? 1 : 0, kx, klam, kst, kit, B, b->d_prof,
                        b->d_slabs, b->slab_doubles, b->d_dw, b->d_order, (const StoreArgs*)(b->store.out ? b->d_store : nullptr), (flags & OMGX_ONLY_FAILED) ? 1 : 0,
                        b->d_next, b->d_x0_alt, b->d_x0_alt ? b->n_alt : 0, b->d_attempts,
                        (unsigned long long*)(b->d_stats ? b->d_stats + 4 * (size_t)(b->stats_launch++ % b->stats_slots) : nullptr),
                        b->stagger, (const CenterArgs*)(b->center_on ? b->d_center : nullptr),
                        prepared ? b->d_prep : (double*)nullptr, b->prep_doubles, (const StopArgs*)(b->stop_on ? b->d_stop : nullptr));
  HIPCHK(hipGetLastError());
  if (ranged)
    hipLaunchKernelGGL(range_contract_lam, dim3((B * nu + 255) / 256), dim3(256), 0, b->stream, (const double*)b->d_lam, lam_user_dev, B, nu, ni,
                       (const int32_t*)b->d_range_dup);
  if (!dev) {
    HIPCHK(hipMemcpyAsync(x, b->d_x, (size_t)B * d.n_var * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(lam_g, ranged ? b->d_lam_user : b->d_lam, (size_t)B * nu * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(status, b->d_status, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(iters, b->d_iters, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
  }
  return OMGX_OK;
}

#ifdef OMGX_PROFILE
int omgx_batch_phase_cycles(omgx_batch* b, long long* out) {   // profiling build only
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(out, b->d_prof, (size_t)b->n_agents * omgx::PH_COUNT * sizeof(long long), hipMemcpyDeviceToHost));
  return OMGX_OK;
}
#endif

// omgx_batch_transfer: up to OMGX_TRANSFER_MAX segments moved by one launch (device <-> pinned host memory over the host link, or
// device <-> device): a workgroup range per segment in proportion to its size, 16-byte words, coalesced
struct TransferArgs { const double* src[OMGX_TRANSFER_MAX]; double* dst[OMGX_TRANSFER_MAX]; long long n8[OMGX_TRANSFER_MAX]; int blk0[OMGX_TRANSFER_MAX + 1]; int n_seg; };
__global__ void __launch_bounds__(256)
transfer_kernel(TransferArgs a) {
  int sg = 0;
  while (sg + 1 < a.n_seg && (int)blockIdx.x >= a.blk0[sg + 1]) ++sg;
  const int nb = a.blk0[sg + 1] - a.blk0[sg], lb = blockIdx.x - a.blk0[sg];
  const long long n8 = a.n8[sg], n16 = n8 >> 1;
  const bool wide = ((((size_t)a.src[sg]) | ((size_t)a.dst[sg])) & 15) == 0;
  if (wide) {
    const double2* s = (const double2*)a.src[sg];
    double2* d = (double2*)a.dst[sg];
    for (long long i = (long long)lb * 256 + threadIdx.x; i < n16; i += (long long)nb * 256) d[i] = s[i];
    if ((n8 & 1) && lb == 0 && threadIdx.x == 0) a.dst[sg][n8 - 1] = a.src[sg][n8 - 1];
  } else {
    for (long long i = (long long)lb * 256 + threadIdx.x; i < n8; i += (long long)nb * 256) a.dst[sg][i] = a.src[sg][i];
  }
}

int omgx_batch_transfer(omgx_batch* b, int32_t n_seg, const void* const* src, void* const* dst, const int64_t* bytes) {
  if (!b || n_seg < 0 || n_seg > OMGX_TRANSFER_MAX || (n_seg > 0 && (!src || !dst || !bytes))) { g_err = "bad argument"; return OMGX_E_INVALID; }
  if (n_seg == 0) return OMGX_OK;
  TransferArgs a;
  memset(&a, 0, sizeof a);
  a.n_seg = n_seg;
  int blocks = 0;
  for (int i = 0; i < n_seg; ++i) {
    if (!src[i] || !dst[i] || bytes[i] < 0 || (bytes[i] & 7) || (((size_t)src[i] | (size_t)dst[i]) & 7)) { g_err = "transfer: segments are 8-byte aligned multiples of 8 bytes"; return OMGX_E_INVALID; }
    a.src[i] = (const double*)src[i]; a.dst[i] = (double*)dst[i]; a.n8[i] = bytes[i] / 8;
    a.blk0[i] = blocks;
    // (a workgroup per 16 KB, at least one, at most 256 per segment: enough requests in flight to fill the host link)
    long long nb = (bytes[i] + 16383) / 16384;
    blocks += (int)(nb < 1 ? 1 : (nb > 256 ? 256 : nb));
  }
  a.blk0[n_seg] = blocks;
  HIPCHK(hipSetDevice(b->device));
  hipLaunchKernelGGL(transfer_kernel, dim3(blocks), dim3(256), 0, b->stream, a);
  HIPCHK(hipGetLastError());
  return OMGX_OK;
}

int omgx_batch_sync(omgx_batch* b) {
  if (!b) return OMGX_E_INVALID;
  { const int rc_o = flush_order(b); if (rc_o != OMGX_OK) return rc_o; }      // (a deferred omgx_batch_order_by_iters)
  HIPCHK(hipStreamSynchronize(b->stream));
  return OMGX_OK;
}

int omgx_batch_eval(omgx_batch* b, const double* p, const double* x, const double* lam_g, double* g, double* f,
                    double* jac, double* hess) {
  if (!b || !p || !x || !lam_g) { g_err = "null argument"; return OMGX_E_INVALID; }
  if (b->n_range > 0) { g_err = "omgx_batch_eval: not available for a template with two-sided rows (the kernel's rows are not the caller's)"; return OMGX_E_INVALID; }
  HIPCHK(hipSetDevice(b->device));
  const omgx::Dims& d = b->dims;
  const int B = b->n_agents;
  const size_t stride = (size_t)d.n_con + 1 + d.nnz_j + b->kkt_doubles;
  double *d_out = nullptr, *d_lam = nullptr;
  HIPCHK(hipMalloc((void**)&d_out, (size_t)B * stride * sizeof(double)));
  if (hipMalloc((void**)&d_lam, (size_t)B * d.n_con * sizeof(double)) != hipSuccess) { (void)hipFree(d_out); g_err = "hipMalloc failed"; return OMGX_E_HIP; }
  std::vector<double> out((size_t)B * stride);
  hipError_t e = hipMemcpyAsync(b->d_p, p, (size_t)B * d.n_par * sizeof(double), hipMemcpyHostToDevice, b->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(b->d_x0, x, (size_t)B * d.n_var * sizeof(double), hipMemcpyHostToDevice, b->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_lam, lam_g, (size_t)B * d.n_con * sizeof(double), hipMemcpyHostToDevice, b->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(ipm_eval_kernel_for(b->ws_mode, b->dims.wave_ok, b->dims.general), dim3(b->n_slabs), dim3(b->threads), (uint32_t)b->lds_bytes,
                       b->stream, d, b->dev, b->kkt_doubles, (const double*)b->d_p, (const double*)b->d_x0, (const double*)d_lam, B,
                       b->d_slabs, b->slab_doubles, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out.data(), d_out, out.size() * sizeof(double), hipMemcpyDeviceToHost, b->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  (void)hipFree(d_out); (void)hipFree(d_lam);
  if (e != hipSuccess) { g_err = std::string("omgx_batch_eval: ") + hipGetErrorString(e); return OMGX_E_HIP; }
  const size_t nv = d.n_var, nc = d.n_con;
  for (int a = 0; a < B; ++a) {
    const double* o = out.data() + (size_t)a * stride;
    if (g) for (size_t r = 0; r < nc; ++r) g[a * nc + r] = o[r];
    if (f) f[a] = o[nc];
    if (jac) {
      double* J = jac + (size_t)a * (nc + 1) * nv;
      for (size_t i = 0; i < (nc + 1) * nv; ++i) J[i] = 0.0;
      for (int en = 0; en < d.nnz_j; ++en) J[(size_t)b->ev_jrow[en] * nv + b->ev_jvar[en]] = o[nc + 1 + en];
    }
    if (hess) {
      double* Hm = hess + (size_t)a * nv * nv;
      for (size_t i = 0; i < nv * nv; ++i) Hm[i] = 0.0;
      const double* kk = o + nc + 1 + d.nnz_j;
      for (size_t i = 0; i + 2 < b->ev_hess.size(); i += 3) {
        const int va = b->ev_hess[i + 1], vb = b->ev_hess[i + 2];
        Hm[(size_t)va * nv + vb] = kk[b->ev_hess[i]]; Hm[(size_t)vb * nv + va] = kk[b->ev_hess[i]];
      }
    }
  }
  return OMGX_OK;
}

int omgx_batch_set_stats(omgx_batch* b, int64_t* stats_device, int32_t n_slots) {
  if (!b || n_slots < 0 || (n_slots > 0 && !stats_device)) { g_err = "bad argument"; return OMGX_E_INVALID; }
  b->d_stats = n_slots > 0 ? stats_device : nullptr;
  b->stats_slots = n_slots; b->stats_launch = 0;
  return OMGX_OK;
}

int omgx_batch_set_launch_events(omgx_batch* b, void* start_event, void* stop_event) {
  if (!b || (!start_event) != (!stop_event)) { g_err = "bad argument"; return OMGX_E_INVALID; }
  b->ext_ev0 = (hipEvent_t)start_event; b->ext_ev1 = (hipEvent_t)stop_event;
  return OMGX_OK;
}

int omgx_batch_set_timing(omgx_batch* b, int32_t on) {
  if (!b) { g_err = "null handle"; return OMGX_E_INVALID; }
  b->timing = on != 0;
  if (!b->timing) b->timed = false;
  return OMGX_OK;
}

int omgx_batch_last_kernel_ms(omgx_batch* b, double* ms) {
  if (!b || !ms || !b->timed) { g_err = "no timed launch"; return OMGX_E_INVALID; }
  HIPCHK(hipEventSynchronize(b->ev1));
  float f = 0.f;
  HIPCHK(hipEventElapsedTime(&f, b->ev0, b->ev1));
  *ms = f;
  return OMGX_OK;
}

namespace {
// entries / T matrices of a shift into the handle's device buffers (no-op when unchanged); max_elems: LDS doubles
int stage_shift_tables(omgx_batch* b, const int32_t* entries, int32_t n_ent, const double* Tmats, int32_t n_tmat,
                       int limit, int* max_elems) {
  *max_elems = 0;
  for (int e = 0; e < n_ent; ++e) {
    const int32_t* q = entries + 4 * e;
    if (q[0] < 0 || q[1] <= 0 || q[2] <= 0 || q[3] < 0 || q[0] + q[1] * q[2] > limit || q[3] + q[1] * q[1] > n_tmat) {
      g_err = "shift entry outside the array / the matrices"; return OMGX_E_INVALID;
    }
    if (q[1] * q[2] > *max_elems) *max_elems = q[1] * q[2];
  }
  const size_t ne = 4 * (size_t)n_ent, nt = (size_t)n_tmat;
  ++b->shift_clock;
  for (auto& ss : b->shift_sets)
    if (ss.ent.size() == ne && ss.T.size() == nt && memcmp(ss.ent.data(), entries, ne * sizeof(int32_t)) == 0 &&
        memcmp(ss.T.data(), Tmats, nt * sizeof(double)) == 0) {
      ss.used = b->shift_clock; b->d_shift_ent = ss.d_ent; b->d_shift_T = ss.d_T;
      return OMGX_OK;
    }
  const size_t kShiftSets = 16;
  if (b->shift_sets.size() >= kShiftSets) {       // the least recently used set goes (its buffers may still be read by a queued launch: hipFree waits)
    size_t old = 0;
    for (size_t i = 1; i < b->shift_sets.size(); ++i) if (b->shift_sets[i].used < b->shift_sets[old].used) old = i;
    (void)hipFree(b->shift_sets[old].d_ent); (void)hipFree(b->shift_sets[old].d_T);
    b->shift_sets.erase(b->shift_sets.begin() + old);
  }
  omgx_batch::ShiftSet ss;
  ss.ent.assign(entries, entries + ne); ss.T.assign(Tmats, Tmats + nt); ss.d_ent = nullptr; ss.d_T = nullptr; ss.used = b->shift_clock;
  HIPCHK(hipMalloc((void**)&ss.d_ent, ne * sizeof(int32_t)));
  if (hipMalloc((void**)&ss.d_T, nt * sizeof(double)) != hipSuccess) { (void)hipFree(ss.d_ent); g_err = "hipMalloc failed"; return OMGX_E_HIP; }
  // (fresh buffers nothing in flight reads: plain synchronous copies)
  if (hipMemcpy(ss.d_ent, ss.ent.data(), ne * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(ss.d_T, ss.T.data(), nt * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(ss.d_ent); (void)hipFree(ss.d_T); g_err = "hipMemcpy failed"; return OMGX_E_HIP;
  }
  b->d_shift_ent = ss.d_ent; b->d_shift_T = ss.d_T;
  b->shift_sets.push_back(std::move(ss));
  return OMGX_OK;
}
}  // namespace

int omgx_batch_shift(omgx_batch* b, double* x, const uint8_t* mask, const int32_t* entries, int32_t n_ent,
                     const double* Tmats, int32_t n_tmat, int32_t flags) {
  if (!b || !x || !entries || !Tmats || n_ent <= 0 || n_tmat <= 0) { g_err = "bad argument"; return OMGX_E_INVALID; }
  HIPCHK(hipSetDevice(b->device));
  const omgx::Dims& d = b->dims;
  const int B = b->n_agents;
  const bool dev = flags & OMGX_PTR_DEVICE;
  int max_elems = 0;
  int rc = stage_shift_tables(b, entries, n_ent, Tmats, n_tmat, d.n_var, &max_elems);
  if (rc != OMGX_OK) return rc;
  const uint8_t* d_mask = mask; double* d_xx = x;
  if (!dev) {
    HIPCHK(hipMemcpyAsync(b->d_x, x, (size_t)B * d.n_var * sizeof(double), hipMemcpyHostToDevice, b->stream));
    d_xx = b->d_x;
    if (mask) {
      if (!b->d_mask) HIPCHK(hipMalloc((void**)&b->d_mask, B));
      HIPCHK(hipMemcpyAsync(b->d_mask, mask, B, hipMemcpyHostToDevice, b->stream));
      d_mask = b->d_mask;
    }
  }
  hipLaunchKernelGGL(shift_kernel, dim3(B), dim3(64), max_elems * sizeof(double), b->stream, d_xx, d.n_var,
                     d_mask, b->d_shift_ent, n_ent, b->d_shift_T);
  HIPCHK(hipGetLastError());
  if (!dev) {
    HIPCHK(hipMemcpyAsync(x, b->d_x, (size_t)B * d.n_var * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
  }
  return OMGX_OK;      // device pointers: stream-ordered, no host synchronisation
}

int omgx_batch_sample(omgx_batch* b, const double* x, int32_t coeff_off, int32_t n_spl, int32_t degree,
                      const double* knots, int32_t n_knots, int32_t n_der, const double* t0, double dt,
                      int32_t n_samp, void* out, int32_t as_f32, int32_t flags) {
  if (!b || !x || !knots || !t0 || !out || degree < 0 || degree > 5 || n_der < 1 || n_der > degree + 1 || n_samp <= 0 ||
      n_spl <= 0 || n_knots < 2 * (degree + 1) || n_knots > 40 || coeff_off < 0 ||
      coeff_off + n_spl * (n_knots - degree - 1) > b->dims.n_var) {
    g_err = "bad argument"; return OMGX_E_INVALID;
  }
  HIPCHK(hipSetDevice(b->device));
  const omgx::Dims& d = b->dims;
  const int B = b->n_agents, L = n_knots - degree - 1;
  const bool dev = flags & OMGX_PTR_DEVICE;
  const size_t out_elems = (size_t)B * n_der * n_spl * n_samp, esz = as_f32 ? 4 : 8;
  if (n_knots > 40) { g_err = "knot vector longer than 40"; return OMGX_E_INVALID; }
  KnotArg kn;
  for (int i = 0; i < 40; ++i) kn.k[i] = i < n_knots ? knots[i] : 0.0;
  double* d_t0 = nullptr; void* d_out = out; const double* d_xx = x;
  struct DevTmp {                                   // temporaries of the host-pointer path: freed on every way out
    void* p[2] = {nullptr, nullptr};
    ~DevTmp() { for (void* q : p) if (q) (void)hipFree(q); }
  } tmp;
  if (!dev) {
    HIPCHK(hipMemcpyAsync(b->d_x, x, (size_t)B * d.n_var * sizeof(double), hipMemcpyHostToDevice, b->stream));
    d_xx = b->d_x;
    HIPCHK(hipMalloc((void**)&d_t0, B * sizeof(double)));
    tmp.p[0] = d_t0;
    HIPCHK(hipMemcpyAsync(d_t0, t0, B * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMalloc(&d_out, out_elems * esz));
    tmp.p[1] = d_out;
  } else {
    d_t0 = (double*)t0;
  }
  const dim3 grid((n_samp + OMGX_SAMPLE_CHUNK - 1) / OMGX_SAMPLE_CHUNK, B), block(256);
  const int n_span = n_knots - 2 * degree - 1, D1 = degree + 1;
  const size_t lds = sample_scratch_doubles(n_spl, degree, n_knots, n_der) * sizeof(double);
  (void)L; (void)n_span; (void)D1;
  // (the caller's one-shot event pair of omgx_batch_set_launch_events stamps this launch as well: the duration of a
  // 20 us kernel cannot be taken with events recorded around the call)
  hipEvent_t e0 = b->ext_ev0, e1 = b->ext_ev1;
  b->ext_ev0 = b->ext_ev1 = nullptr;
  if (as_f32)
    hipExtLaunchKernelGGL(sample_kernel<float>, grid, block, lds, b->stream, e0, e1, 0, d_xx, d.n_var, coeff_off, n_spl, degree,
                          kn, n_knots, n_der, d_t0, dt, 1.0, n_samp, (float*)d_out, (float*)nullptr);
  else
    hipExtLaunchKernelGGL(sample_kernel<double>, grid, block, lds, b->stream, e0, e1, 0, d_xx, d.n_var, coeff_off, n_spl, degree,
                          kn, n_knots, n_der, d_t0, dt, 1.0, n_samp, (double*)d_out, (double*)nullptr);
  HIPCHK(hipGetLastError());
  if (!dev) {
    HIPCHK(hipMemcpyAsync(out, d_out, out_elems * esz, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
  }
  return OMGX_OK;      // device pointers: stream-ordered, the caller synchronises (omgx_batch_sync)
}

namespace {
int fill_store(omgx_batch* b, const omgx_store_spec* sp, StoreArgs* st) {
  if (!sp->out || !sp->t0 || !sp->knots || sp->degree < 1 || sp->degree > 5 || sp->n_der < 1 || sp->n_der > sp->degree + 1 ||
      sp->n_samp <= 0 || sp->n_spl <= 0 || sp->n_knots > 40 || sp->n_knots < 2 * sp->degree + 2 || !(sp->inv_T > 0.0) ||
      (sp->v_tot && sp->n_der < 2)) {
    g_err = "bad store specification"; return OMGX_E_INVALID;
  }
  const int L = sp->n_knots - sp->degree - 1;
  if (sp->coeff_off < 0 || sp->coeff_off + sp->n_spl * L > b->dims.n_var) { g_err = "store: coefficients outside x"; return OMGX_E_INVALID; }
  st->out = sp->out; st->v_tot = sp->v_tot; st->t0 = sp->t0;
  st->coeff_off = sp->coeff_off; st->n_spl = sp->n_spl; st->degree = sp->degree; st->n_knots = sp->n_knots;
  st->n_der = sp->n_der; st->n_samp = sp->n_samp; st->dt = sp->dt; st->inv_T = sp->inv_T;
  for (int i = 0; i < 40; ++i) st->knots.k[i] = i < sp->n_knots ? sp->knots[i] : 0.0;
  return OMGX_OK;
}
}  // namespace

int omgx_batch_store(omgx_batch* b, const double* x, const omgx_store_spec* sp) {
  if (!b || !x || !sp) { g_err = "null argument"; return OMGX_E_INVALID; }
  StoreArgs st;
  int rc = fill_store(b, sp, &st);
  if (rc != OMGX_OK) return rc;
  HIPCHK(hipSetDevice(b->device));
  const dim3 grid((st.n_samp + OMGX_SAMPLE_CHUNK - 1) / OMGX_SAMPLE_CHUNK, b->n_agents), block(256);
  const size_t lds = sample_scratch_doubles(st.n_spl, st.degree, st.n_knots, st.n_der) * sizeof(double);
  hipLaunchKernelGGL(sample_kernel<double>, grid, block, lds, b->stream, x, b->dims.n_var, st.coeff_off, st.n_spl,
                     st.degree, st.knots, st.n_knots, st.n_der, st.t0, st.dt, st.inv_T, st.n_samp, st.out, st.v_tot);
  HIPCHK(hipGetLastError());
  return OMGX_OK;
}

int omgx_batch_set_store(omgx_batch* b, const omgx_store_spec* sp) {
  if (!b) { g_err = "null handle"; return OMGX_E_INVALID; }
  if (!sp) { b->store = StoreArgs{}; return OMGX_OK; }
  HIPCHK(hipSetDevice(b->device));
  StoreArgs st;
  int rc = fill_store(b, sp, &st);
  if (rc != OMGX_OK) return rc;
  if (sample_scratch_doubles(st.n_spl, st.degree, st.n_knots, st.n_der) > (size_t)b->kkt_doubles) {
    g_err = "store: the per-agent scratch does not fit the KKT store"; return OMGX_E_TOOLARGE;
  }
  if (!b->d_store) HIPCHK(hipMalloc((void**)&b->d_store, sizeof(StoreArgs)));
  b->store = st;
  HIPCHK(hipMemcpyAsync(b->d_store, &b->store, sizeof(StoreArgs), hipMemcpyHostToDevice, b->stream));
  return OMGX_OK;
}

int omgx_batch_predict_quadrotor(omgx_batch* b, const double* x, double* p, int32_t coeff_off, int32_t degree, const double* knots,
                                 int32_t n_knots, double tau, double inv_T, int32_t n_out, const int32_t* p_off, int32_t p_t,
                                 double t_value, const double* state_in, double* state_out, int32_t n_sub, double dtau, double g) {
  if (!b || !x || !p || !knots || !p_off || !state_in || n_knots > 40 || degree > 5 || degree < 3 || n_out < 1 || n_out > 4 ||
      n_out > degree + 1 || n_sub < 1 || !(dtau > 0.0) || !(g > 0.0)) { g_err = "bad argument"; return OMGX_E_INVALID; }
  const omgx::Dims& d = b->dims;
  const int L = n_knots - degree - 1;
  if (coeff_off < 0 || coeff_off + 2 * L > d.n_var || p_t >= d.n_par) { g_err = "predict: offsets outside x / p"; return OMGX_E_INVALID; }
  PredictArgs a;
  for (int o = 0; o < 4; ++o) {
    a.p_off[o] = o < n_out ? p_off[o] : -1;
    if (a.p_off[o] >= 0 && a.p_off[o] + 2 > d.n_par) { g_err = "predict: offsets outside p"; return OMGX_E_INVALID; }
  }
  HIPCHK(hipSetDevice(b->device));
  for (int i = 0; i < 40; ++i) a.kn.k[i] = i < n_knots ? knots[i] : 0.0;
  a.coeff_off = coeff_off; a.n_spl = 2; a.degree = degree; a.n_knots = n_knots; a.n_out = n_out;
  a.tau = tau; a.inv_T = inv_T; a.p_t = p_t; a.t_value = t_value; a.mode = OMGX_PREDICT_RK4; a.state_in = state_in; a.n_sub = n_sub; a.dtau = dtau;
  hipLaunchKernelGGL(predict_quadrotor_kernel, dim3((b->n_agents + 255) / 256), dim3(256), 0, b->stream, x, d.n_var, p, d.n_par, b->n_agents, a, g, state_out);
  HIPCHK(hipGetLastError());
  return OMGX_OK;
}

int omgx_batch_rollout(omgx_batch* b, const omgx_rollout_spec* sp, double* p, double* x, const double* lbg, const double* ubg,
                       double* lam_g, int32_t* status, int32_t* iters, int32_t flags) {
  if (!b || !sp || !p || !x || !lbg || !ubg || !lam_g || !status || !iters) { g_err = "null argument"; return OMGX_E_INVALID; }
  if (!(flags & OMGX_PTR_DEVICE) || !(flags & OMGX_BOUNDS_DEVICE)) { g_err = "rollout: device pointers only (OMGX_PTR_DEVICE | OMGX_BOUNDS_DEVICE)"; return OMGX_E_INVALID; }
  const omgx::Dims& d = b->dims;
  ipm_rollout_t kern = rollout_kernel_for(b->ws_mode, d.wave_ok, d.general);
  if (!kern || b->n_range > 0 || !b->d_next) {
    g_err = "rollout: not available for this template class (spill modes, general instance, two-sided rows): step with omgx_batch_solve";
    return OMGX_E_INVALID;
  }
  if (sp->n_steps <= 0 || !sp->tau || !sp->t_rel || !sp->crossed || !sp->knots || !sp->p_off || sp->n_knots > 40 || sp->degree > 5 ||
      sp->degree < 1 || sp->n_spl <= 0 || sp->n_spl > 64 || sp->n_out < 1 || sp->n_out > 4 || sp->n_out > sp->degree + 1 || sp->n_obst < 0 ||
      sp->n_obst > 8 || (sp->n_obst > 0 && !sp->obst) || sp->n_ent < 0 || (sp->n_ent > 0 && (!sp->shift_entries || !sp->shift_T || !sp->lam_perm))) {
    g_err = "rollout: bad specification"; return OMGX_E_INVALID;
  }
  const int L = sp->n_knots - sp->degree - 1;
  if (sp->coeff_off < 0 || sp->coeff_off + sp->n_spl * L > d.n_var || sp->p_t >= d.n_par) { g_err = "rollout: offsets outside x / p"; return OMGX_E_INVALID; }
  HIPCHK(hipSetDevice(b->device));
  RolloutArgs a;
  memset(&a, 0, sizeof a);
  for (int i = 0; i < 40; ++i) a.kn.k[i] = i < sp->n_knots ? sp->knots[i] : 0.0;
  a.coeff_off = sp->coeff_off; a.n_spl = sp->n_spl; a.degree = sp->degree; a.n_knots = sp->n_knots; a.n_out = sp->n_out; a.p_t = sp->p_t;
  for (int o = 0; o < 4; ++o) {
    a.p_off[o] = o < sp->n_out ? sp->p_off[o] : -1;
    if (a.p_off[o] >= 0 && a.p_off[o] + sp->n_spl > d.n_par) { g_err = "rollout: offsets outside p"; return OMGX_E_INVALID; }
  }
  a.inv_T = sp->inv_T; a.dt = sp->dt; a.n_obst = sp->n_obst;
  for (int q = 0; q < sp->n_obst; ++q) {
    for (int k = 0; k < 4; ++k) a.obst[q][k] = sp->obst[4 * q + k];
    const int nd = a.obst[q][3];
    if (nd <= 0 || nd > 64 || a.obst[q][0] < 0 || a.obst[q][1] < 0 || a.obst[q][2] < 0 || a.obst[q][0] + nd > d.n_par || a.obst[q][1] + nd > d.n_par ||
        a.obst[q][2] + nd > d.n_par) { g_err = "rollout: obstacle entries outside p"; return OMGX_E_INVALID; }
  }
  // knot-crossing tables: the shift set (cached on the device by content) and the multiplier map
  a.n_ent = sp->n_ent;
  bool any_cross = false;
  for (int k = 0; k < sp->n_steps; ++k) any_cross = any_cross || sp->crossed[k] != 0;
  if (any_cross && sp->n_ent <= 0) { g_err = "rollout: a step crosses a knot but no shift tables were given"; return OMGX_E_INVALID; }
  if (sp->n_ent > 0) {
    int max_elems = 0;
    const int rc = stage_shift_tables(b, sp->shift_entries, sp->n_ent, sp->shift_T, sp->n_tmat, d.n_var, &max_elems);
    if (rc != OMGX_OK) return rc;
    if (max_elems > b->kkt_doubles || d.n_con > b->kkt_doubles) { g_err = "rollout: shift scratch exceeds the KKT store"; return OMGX_E_INVALID; }
    a.sh_ent = b->d_shift_ent; a.sh_T = b->d_shift_T;
    for (int i = 0; i < d.n_con; ++i) if (sp->lam_perm[i] >= d.n_con) { g_err = "rollout: multiplier map out of range"; return OMGX_E_INVALID; }
    if (b->ro_perm_host.size() != (size_t)d.n_con || memcmp(b->ro_perm_host.data(), sp->lam_perm, d.n_con * sizeof(int32_t)) != 0) {
      if (!b->d_ro_perm) HIPCHK(hipMalloc((void**)&b->d_ro_perm, sizeof(int32_t) * (size_t)d.n_con));
      b->ro_perm_host.assign(sp->lam_perm, sp->lam_perm + d.n_con);
      // (stream-ordered like the two copies below: a rollout still running on a non-blocking caller stream reads the old map)
      HIPCHK(hipMemcpyAsync(b->d_ro_perm, b->ro_perm_host.data(), sizeof(int32_t) * (size_t)d.n_con, hipMemcpyHostToDevice, b->stream));
    }
    a.lam_perm = b->d_ro_perm;
  }
  if (sp->n_steps > b->ro_steps_cap) {
    if (b->d_ro_steps) (void)hipFree(b->d_ro_steps);
    b->d_ro_steps = nullptr; b->ro_steps_cap = 0;
    HIPCHK(hipMalloc((void**)&b->d_ro_steps, sizeof(RolloutStep) * (size_t)sp->n_steps));
    b->ro_steps_cap = sp->n_steps;
  }
  std::vector<RolloutStep> steps((size_t)sp->n_steps);
  for (int k = 0; k < sp->n_steps; ++k) { steps[k].tau = sp->tau[k]; steps[k].t_rel = sp->t_rel[k]; steps[k].crossed = sp->crossed[k] ? 1 : 0; steps[k].pad = 0; }
  a.steps = b->d_ro_steps; a.K = sp->n_steps;
  b->opts.prio_iter = b->prio_iter;
  a.o_cross = b->opts;
  if (sp->cross_options) {
    const omgx_options& co = *sp->cross_options;
    a.o_cross.kappa_warm = co.kappa_warm; a.o_cross.warm_mu_factor = co.warm_mu_factor; a.o_cross.warm_z_floor = co.warm_z_floor;
    a.o_cross.warm_z_cap = co.warm_z_cap; a.o_cross.max_iter = co.max_iter; a.o_cross.tol = co.tol; a.o_cross.max_soc = co.max_soc; a.o_cross.refine = co.refine > 0 ? 1 : 0;
  }
  // per-step statistics: the slots the next n_steps single launches would have taken (omgx_batch_set_stats)
  a.stop_on = b->stop_on ? 1 : 0; a.stop = b->stop_host;
  a.stats = nullptr;
  if (b->d_stats) {
    if (sp->n_steps > b->stats_slots - (int)(b->stats_launch % b->stats_slots)) { g_err = "rollout: the stats array has fewer free slots than steps"; return OMGX_E_INVALID; }
    a.stats = (unsigned long long*)b->d_stats + 4 * (size_t)(b->stats_launch % b->stats_slots);
    b->stats_launch += sp->n_steps;
  }
  a.iters_log = sp->iters_log; a.status_log = sp->status_log;
  if (!b->d_rollout) HIPCHK(hipMalloc((void**)&b->d_rollout, sizeof(RolloutArgs)));
  // Two small copies per call (one call is n_steps steps of the whole batch), ORDERED ON THE HANDLE'S STREAM: the persistent
  // kernel of a previous rollout reads these tables for its whole run, and on a non-blocking caller stream a null-stream
  // hipMemcpy would overwrite them under it (pageable sources: staged before the calls return, executed in stream order).
  HIPCHK(hipMemcpyAsync(b->d_ro_steps, steps.data(), sizeof(RolloutStep) * steps.size(), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_rollout, &a, sizeof(RolloutArgs), hipMemcpyHostToDevice, b->stream));
  { const int rc_o = flush_order(b); if (rc_o != OMGX_OK) return rc_o; }
  const bool shared = flags & OMGX_BOUNDS_SHARED;
  hipEvent_t e0 = b->ext_ev0, e1 = b->ext_ev0 ? b->ext_ev1 : nullptr;
  b->ext_ev0 = b->ext_ev1 = nullptr;
  b->timed = false;
  hipExtLaunchKernelGGL(kern, dim3(b->n_slabs < b->n_agents ? b->n_slabs : b->n_agents), dim3(b->threads), (uint32_t)b->lds_bytes, b->stream, e0, e1, 0u,
                        d, b->dev, b->opts, b->kkt_doubles, p, x, lbg, ubg, shared ? 1 : 0, lam_g, status, iters, b->n_agents,
                        b->d_slabs, b->slab_doubles, b->d_dw, b->d_next, (const RolloutArgs*)b->d_rollout, b->stagger, b->d_order,
                        (const StoreArgs*)(b->store.out ? b->d_store : nullptr));
  HIPCHK(hipGetLastError());
  return OMGX_OK;
}

int omgx_batch_predict_ex(omgx_batch* b, const double* x, double* p, int32_t coeff_off, int32_t n_spl, int32_t degree,
                          const double* knots, int32_t n_knots, double tau, double inv_T, int32_t n_out,
                          const int32_t* p_off, int32_t p_t, double t_value, int32_t mode, const double* state_in,
                          int32_t n_sub, double dtau) {
  if (!b || !x || !p || !knots || !p_off || n_knots > 40 || degree > 5 || degree < 1 || n_spl <= 0 || n_out < 1 || n_out > 4 ||
      n_out > degree + 1 || (mode != OMGX_PREDICT_IDEAL && mode != OMGX_PREDICT_RK4) ||
      (mode == OMGX_PREDICT_RK4 && (!state_in || n_sub < 1 || !(dtau > 0.0) || n_out < 2))) {
    g_err = "bad argument"; return OMGX_E_INVALID;
  }
  const omgx::Dims& d = b->dims;
  const int L = n_knots - degree - 1;
  if (coeff_off < 0 || coeff_off + n_spl * L > d.n_var || p_t >= d.n_par) { g_err = "predict: offsets outside x / p"; return OMGX_E_INVALID; }
  PredictArgs a;
  for (int o = 0; o < 4; ++o) {
    a.p_off[o] = o < n_out ? p_off[o] : -1;
    if (a.p_off[o] >= 0 && a.p_off[o] + n_spl > d.n_par) { g_err = "predict: offsets outside p"; return OMGX_E_INVALID; }
  }
  HIPCHK(hipSetDevice(b->device));
  for (int i = 0; i < 40; ++i) a.kn.k[i] = i < n_knots ? knots[i] : 0.0;
  a.coeff_off = coeff_off; a.n_spl = n_spl; a.degree = degree; a.n_knots = n_knots; a.n_out = n_out;
  a.tau = tau; a.inv_T = inv_T; a.p_t = p_t; a.t_value = t_value; a.mode = mode; a.state_in = state_in; a.n_sub = n_sub; a.dtau = dtau;
  const int n = b->n_agents * n_spl;
  const int32_t* oi = b->pend_iters; int32_t* oo = b->pend_order;
  b->pend_iters = nullptr; b->pend_order = nullptr;
  hipLaunchKernelGGL(predict_kernel, dim3((n + 255) / 256 + (oi ? 1 : 0)), dim3(256), 0, b->stream, x, d.n_var, p, d.n_par, b->n_agents, a, oi, oo, (const double*)(b->order_dw ? b->d_dw : nullptr));
  HIPCHK(hipGetLastError());
  return OMGX_OK;
}

int omgx_batch_predict(omgx_batch* b, const double* x, double* p, int32_t coeff_off, int32_t n_spl, int32_t degree,
                       const double* knots, int32_t n_knots, double tau, double inv_T, int32_t p_state0,
                       int32_t p_input0, int32_t p_t, double t_value) {
  const int32_t off[2] = {p_state0, p_input0};
  return omgx_batch_predict_ex(b, x, p, coeff_off, n_spl, degree, knots, n_knots, tau, inv_T, 2, off, p_t, t_value,
                               OMGX_PREDICT_IDEAL, nullptr, 0, 0.0);
}

int omgx_admm_center(omgx_batch* b, const omgx_admm_layout* lay, const double* x, const double* p, double* x_i) {
  return omgx_admm_center_ex(b, lay, x, p, x_i, nullptr, 0, nullptr);
}

int omgx_admm_center_ex(omgx_batch* b, const omgx_admm_layout* lay, const double* x, const double* p, double* x_i,
                        const int32_t* pub_rows, int32_t n_pub, double* x_send) {
  if (!b || !lay || !x || !p || !x_i || n_pub < 0 || (n_pub > 0 && (!pub_rows || !x_send))) { g_err = "bad argument"; return OMGX_E_INVALID; }
  HIPCHK(hipSetDevice(b->device));
  const int n = (b->n_agents + n_pub) * lay->n_dim * lay->L;
  hipLaunchKernelGGL(admm_center_kernel, dim3((n + 255) / 256), dim3(256), 0, b->stream, *lay, x, b->dims.n_var,
                     p, b->dims.n_par, x_i, b->n_agents, pub_rows, n_pub, x_send);
  HIPCHK(hipGetLastError());
  return OMGX_OK;
}

int omgx_batch_set_center(omgx_batch* b, const omgx_admm_layout* lay, double* x_i, const int32_t* pub_rows, int32_t n_pub, double* x_send) {
  if (!b) { g_err = "bad argument"; return OMGX_E_INVALID; }
  if (!lay) { b->center_on = false; return OMGX_OK; }
  // (a failed registration leaves the epilogue OFF: the previous x_i may be a buffer the caller has released since)
  b->center_on = false;
  if (!x_i || n_pub < 0 || (n_pub > 0 && (!pub_rows || !x_send)) || lay->n_dim <= 0 || lay->L <= 0 ||
      lay->x_spl < 0 || lay->x_spl + lay->n_dim * lay->L > b->dims.n_var || lay->p_rel < 0 || lay->p_rel + lay->n_dim > b->dims.n_par) {
    g_err = "bad argument"; return OMGX_E_INVALID;
  }
  HIPCHK(hipSetDevice(b->device));
  std::vector<int32_t> inv((size_t)b->n_agents, -1);
  for (int i = 0; i < n_pub; ++i) {
    const int r = pub_rows[i];
    if (r < 0 || r >= b->n_agents) { g_err = "published row out of range"; return OMGX_E_INVALID; }
    if (inv[r] >= 0) { g_err = "a row published twice cannot ride on the solve (use omgx_admm_center_ex)"; return OMGX_E_INVALID; }
    inv[r] = i;
  }
  if (!b->d_center) HIPCHK(hipMalloc((void**)&b->d_center, sizeof(CenterArgs)));
  if (n_pub > 0 && !b->d_pub_inv) HIPCHK(hipMalloc((void**)&b->d_pub_inv, sizeof(int32_t) * (size_t)b->n_agents));
  CenterArgs ca;
  ca.x_spl = lay->x_spl; ca.p_rel = lay->p_rel; ca.n_dim = lay->n_dim; ca.L = lay->L;
  ca.x_i = x_i; ca.pub_inv = n_pub > 0 ? b->d_pub_inv : nullptr; ca.x_send = x_send;
  // (pageable host memory: the copies are staged before the calls return)
  if (n_pub > 0) HIPCHK(hipMemcpyAsync(b->d_pub_inv, inv.data(), sizeof(int32_t) * inv.size(), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_center, &ca, sizeof(CenterArgs), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  b->center_on = true;
  return OMGX_OK;
}

int omgx_admm_update(omgx_batch* b, const omgx_admm_layout* lay, const double* x_ext, const int32_t* nbr,
                     const double* M, const double* F, double rho, double* p, double* z_ij, double* l_ij,
                     double* res) {
  return omgx_admm_update_sums(b, lay, x_ext, nbr, M, F, rho, p, z_ij, l_ij, res, nullptr);
}

int omgx_admm_update_sums(omgx_batch* b, const omgx_admm_layout* lay, const double* x_ext, const int32_t* nbr,
                          const double* M, const double* F, double rho, double* p, double* z_ij, double* l_ij,
                          double* res, double* sums) {
  if (!lay) { g_err = "bad argument"; return OMGX_E_INVALID; }
  return omgx_admm_update_ex(b, lay, x_ext, nbr, M, F, rho, p, z_ij, l_ij, lay->n_nghb * lay->n_dim * lay->L, res, sums,
                             nullptr, nullptr, 0);
}

int omgx_admm_update_ex(omgx_batch* b, const omgx_admm_layout* lay, const double* x_ext, const int32_t* nbr,
                        const double* M, const double* F, double rho, double* p, double* z_ij, double* l_ij,
                        int32_t zl_stride, double* res, double* sums, const int32_t* pub_slot, double* zl_send,
                        int32_t send_stride) {
  if (!b || !lay || !x_ext || !nbr || !M || !F || !p || !z_ij || !l_ij || !res || !(rho > 0) ||
      zl_stride < lay->n_nghb * lay->n_dim * lay->L || (pub_slot && (!zl_send || send_stride < 2 * lay->n_nghb * lay->n_dim * lay->L)) ||
      (pub_slot && sums && send_stride < 3)) {      // (sharded callers keep the three residual sums in a row of zl_send)
    g_err = "bad argument"; return OMGX_E_INVALID;
  }
  const int na = (1 + lay->n_nghb) * lay->n_dim * lay->L;
  if (na > 256) { g_err = "stacked consensus vector longer than 256"; return OMGX_E_TOOLARGE; }
  HIPCHK(hipSetDevice(b->device));
  if (sums && !b->d_admm_done) {
    int rc = dalloc(b, (size_t)1, &b->d_admm_done); if (rc != OMGX_OK) return rc;
    HIPCHK(hipMemset(b->d_admm_done, 0, sizeof(int)));
  }
  const size_t lds_doubles = (size_t)std::max(6 * na + 16, sums ? 3 * 256 : 0);
  hipLaunchKernelGGL(admm_update_kernel, dim3(b->n_agents), dim3(256), lds_doubles * sizeof(double), b->stream,
                     *lay, x_ext, nbr, M, F, rho, p, b->dims.n_par, z_ij, l_ij, (int)zl_stride, res, sums, b->d_admm_done,
                     pub_slot, zl_send, (int)send_stride);
  HIPCHK(hipGetLastError());
  return OMGX_OK;
}

int omgx_admm_communicate(omgx_batch* b, const omgx_admm_layout* lay, const int32_t* nbr, const int32_t* slot,
                          const double* z_ij_ext, const double* l_ij_ext, double* p) {
  if (!lay) { g_err = "null argument"; return OMGX_E_INVALID; }
  return omgx_admm_communicate_ex(b, lay, nbr, slot, z_ij_ext, l_ij_ext, lay->n_nghb * lay->n_dim * lay->L, p, nullptr, 0, 0, nullptr);
}

int omgx_admm_communicate_ex(omgx_batch* b, const omgx_admm_layout* lay, const int32_t* nbr, const int32_t* slot,
                             const double* z_ij_ext, const double* l_ij_ext, int32_t zl_stride, double* p,
                             const double* sum_rows, int32_t n_sum_rows, int32_t sum_stride, double* sums_out) {
  if (!b || !lay || !nbr || !slot || !z_ij_ext || !l_ij_ext || !p || zl_stride < lay->n_nghb * lay->n_dim * lay->L ||
      (sums_out && (!sum_rows || n_sum_rows <= 0 || sum_stride < 3))) { g_err = "bad argument"; return OMGX_E_INVALID; }
  HIPCHK(hipSetDevice(b->device));
  const int n = std::max(3, b->n_agents * lay->n_nghb * lay->n_dim * lay->L);
  hipLaunchKernelGGL(admm_comm_kernel, dim3((n + 255) / 256), dim3(256), 0, b->stream, *lay, nbr, slot, z_ij_ext,
                     l_ij_ext, (int)zl_stride, p, b->dims.n_par, b->n_agents, sum_rows, (int)n_sum_rows, (int)sum_stride, sums_out);
  HIPCHK(hipGetLastError());
  return OMGX_OK;
}

int omgx_shift_rows(omgx_batch* b, double* data, int32_t stride, int32_t n_rows, const uint8_t* mask,
                    const int32_t* entries, int32_t n_ent, const double* Tmats, int32_t n_tmat) {
  // device-pointer variant of omgx_batch_shift for arbitrary row-major arrays (p, z_ij, l_ij ...); stream-ordered
  if (!b || (!data && n_rows > 0) || !entries || !Tmats || n_ent <= 0 || n_rows < 0 || n_tmat <= 0 || stride <= 0) { g_err = "bad argument"; return OMGX_E_INVALID; }
  HIPCHK(hipSetDevice(b->device));
  int max_elems = 0;
  int rc = stage_shift_tables(b, entries, n_ent, Tmats, n_tmat, stride, &max_elems);
  if (rc != OMGX_OK) return rc;
  if (n_rows == 0) return OMGX_OK;            // (tables uploaded ahead of the loop that will use them)
  hipLaunchKernelGGL(shift_kernel, dim3(n_rows), dim3(64), max_elems * sizeof(double), b->stream, data, stride,
                     mask, b->d_shift_ent, n_ent, b->d_shift_T);
  HIPCHK(hipGetLastError());
  return OMGX_OK;
}

}  // extern "C"
