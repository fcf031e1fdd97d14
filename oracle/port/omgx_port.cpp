// ORACLE / CPU BASELINE (test infrastructure; never part of the product library).
//
// Host build of the per-agent interior-point iteration: the statements of
// omg-tools_amd/csrc/omgx_core.h executed by one host thread per agent (agents
// are independent; omgx_port_solve_mt spreads them over n_threads), behind a C
// entry point that mirrors omgx_batch_solve.  Used for
//   * bench.py's `cpu_baseline` leg (kind "port": same algorithm, host cores),
//   * tests that compare the HIP kernel, this port and the independent numpy
//     statement (oracle/ipm_numpy.py) on the same inputs.
// It restates, it does not link, the reference: CasADi/IPOPT (reference
// `problems/problem.py:113`) is absent from /root/reference and not installable.
// Build: oracle/Makefile -> oracle/_build/libomgx_port.so
#define OMGX_HOST_PORT 1
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>
#include "../../include/omgx.h"
#include "../../omg-tools_amd/csrc/omgx_core.h"
#include "../../omg-tools_amd/csrc/omgx_plan.h"

// dw_state [n_agents] (may be null): inertia correction carried between warm-started solves, the
// state the HIP library keeps inside its handle
extern "C" int omgx_port_solve_mt(const omgx_template* tpl, const omgx_options* opt, int32_t n_agents,
                                  const double* p, const double* x0, const double* lbg, const double* ubg,
                                  int32_t bounds_shared, double* x, double* lam_g, int32_t* status,
                                  int32_t* iters, int32_t n_threads, double* dw_state) {
  omgx::HostPlan plan;
  if (!plan.build(*tpl)) return OMGX_E_INVALID;
  omgx::Opts o;
  o.tol = opt->tol; o.max_iter = opt->max_iter; o.mu_init = opt->mu_init;
  o.kappa_push = opt->kappa_push; o.nu_init = opt->nu_init; o.scale_gmax = opt->scale_gmax;
  o.warm_start = opt->warm_start; o.kappa_warm = opt->kappa_warm;
  o.dw_leaf_ratio_cold = opt->dw_leaf_ratio_cold > 0 ? opt->dw_leaf_ratio_cold : 1.0;
  const omgx::Dims& d = plan.dims;
  std::atomic<int> next(0);
  auto worker = [&]() {
    std::vector<double> buf(omgx::work_doubles(d, plan.kkt_doubles) + 8);
    omgx::Work w;
    omgx::work_carve(w, buf.data(), d, plan.kkt_doubles);
    omgx::Ctx c; c.red = w.red;
    for (int b = next.fetch_add(1); b < n_agents; b = next.fetch_add(1)) {
      const double* lb = lbg + (bounds_shared ? 0 : (size_t)b * d.n_con);
      const double* ub = ubg + (bounds_shared ? 0 : (size_t)b * d.n_con);
      omgx::Result r = omgx::ipm_solve(c, d, plan.tables, o, w, p + (size_t)b * d.n_par,
                                       x0 + (size_t)b * d.n_var, lb, ub,
                                       opt->warm_start ? lam_g + (size_t)b * d.n_con : nullptr,
                                       opt->warm_start ? status[b] : 0, plan.kkt_doubles,
                                       (opt->warm_start && dw_state) ? dw_state[b] : 0.0);
      if (dw_state) dw_state[b] = r.dw;
      for (int i = 0; i < d.n_var; ++i) x[(size_t)b * d.n_var + i] = w.x[i];
      for (int r_ = 0; r_ < d.n_con; ++r_)
        lam_g[(size_t)b * d.n_con + r_] = (r.status == 3 || w.rtype[r_] == omgx::ROW_FREE) ? 0.0 : w.rho[r_] * w.z[r_];
      status[b] = r.status; iters[b] = r.iters;
    }
  };
  if (n_threads <= 1) { worker(); return OMGX_OK; }
  std::vector<std::thread> pool;
  for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker);
  for (auto& t : pool) t.join();
  return OMGX_OK;
}

extern "C" int omgx_port_solve(const omgx_template* tpl, const omgx_options* opt, int32_t n_agents,
                               const double* p, const double* x0, const double* lbg, const double* ubg,
                               int32_t bounds_shared, double* x, double* lam_g, int32_t* status,
                               int32_t* iters) {
  return omgx_port_solve_mt(tpl, opt, n_agents, p, x0, lbg, ubg, bounds_shared, x, lam_g, status, iters, 1, nullptr);
}

#ifdef OMGX_COUNT_FACT
extern "C" long omgx_port_nfact(int reset) { long v = omgx_dbg_nfact.load(); if (reset) omgx_dbg_nfact = 0; return v; }
#endif
