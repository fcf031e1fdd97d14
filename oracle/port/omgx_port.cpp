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
#include <pthread.h>
#include <sched.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
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
  o.prio_iter = 0; o.warm_mu_factor = opt->warm_mu_factor >= 0 ? opt->warm_mu_factor : 0.0;
  o.warm_z_floor = opt->warm_z_floor >= 0 ? opt->warm_z_floor : 0.0; o.warm_z_cap = opt->warm_z_cap >= 0 ? opt->warm_z_cap : 0.0; o.max_soc = opt->max_soc > 0 ? (opt->max_soc > 8 ? 8 : opt->max_soc) : 0; o.hess_approx = opt->hess_approx > 0 ? 1 : 0;
  o.compl_tol = opt->compl_inf_tol > 0 ? opt->compl_inf_tol : 0.0; o.viol_tol = opt->constr_viol_tol > 0 ? opt->constr_viol_tol : 0.0;
  o.refine = opt->refine > 0 ? 1 : 0;
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

// ---------------------------------------------------------------------------------------------------------
// CPU baseline of bench.py: the plan is built once, the worker threads live as long as the handle and are
// pinned to the cpus the caller lists (one per physical core), and the glue of a receding-horizon step
// (prediction, obstacle motion, knot shift, multiplier shift: omgtools/batch.py `BatchP2P.step`) runs per
// agent inside the worker that then solves that agent -- no Python between the agents of a step.
struct PortPool {
  omgx::HostPlan plan;
  int n_threads = 0;
  std::vector<std::thread> threads;
  std::vector<std::vector<double>> bufs;
  std::mutex m;
  std::condition_variable cv_go, cv_done;
  long generation = 0;
  int running = 0;
  bool quit = false;
  std::function<void(int)> job;

  void loop(int tid, int cpu) {
    if (cpu >= 0) {
      cpu_set_t set; CPU_ZERO(&set); CPU_SET(cpu, &set);
      pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    }
    long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv_go.wait(lk, [&] { return quit || generation != seen; });
        if (quit) return;
        seen = generation;
      }
      job(tid);
      {
        std::lock_guard<std::mutex> lk(m);
        if (--running == 0) cv_done.notify_all();
      }
    }
  }
  void run(std::function<void(int)> f) {
    std::unique_lock<std::mutex> lk(m);
    job = std::move(f);
    running = n_threads;
    ++generation;
    cv_go.notify_all();
    cv_done.wait(lk, [&] { return running == 0; });
  }
};

struct omgx_port_step_desc {           // constants of one receding-horizon step (mirrors BatchP2P.step)
  int32_t o_spl, n_dim, L, o_state0, o_input0, o_t;
  double t_rel, dt;
  const double* E;                     // [L] basis values at tau
  const double* Ed;                    // [L] d/dt basis values at tau
  int32_t n_obst; const int32_t* obst; // [n_obst, 4] = x offset, v offset, a offset, n_dim
  int32_t crossed, n_shift;
  const int32_t* shift_entries;        // [n_shift, 4] = lo, rows, cols, matrix offset
  const double* shift_mats;
  const int64_t* perm;                 // [n_con] multiplier source row after a knot crossing (-1: none)
};

extern "C" void* omgx_port_pool_create(const omgx_template* tpl, int32_t n_threads, const int32_t* cpus) {
  PortPool* pp = new PortPool;
  if (!pp->plan.build(*tpl)) { delete pp; return nullptr; }
  pp->n_threads = n_threads < 1 ? 1 : n_threads;
  const omgx::Dims& d = pp->plan.dims;
  pp->bufs.resize(pp->n_threads);
  for (auto& b : pp->bufs) b.assign(omgx::work_doubles(d, pp->plan.kkt_doubles) + 8 + d.n_var + d.n_con, 0.0);
  for (int t = 0; t < pp->n_threads; ++t) pp->threads.emplace_back(&PortPool::loop, pp, t, cpus ? cpus[t] : -1);
  return pp;
}

extern "C" void omgx_port_pool_destroy(void* h) {
  PortPool* pp = (PortPool*)h;
  { std::lock_guard<std::mutex> lk(pp->m); pp->quit = true; }
  pp->cv_go.notify_all();
  for (auto& t : pp->threads) t.join();
  delete pp;
}

// step == null: plain solve of every agent (x in/out).  Otherwise the step glue of agent b, then its solve.
extern "C" int omgx_port_pool_solve(void* h, const omgx_options* opt, int32_t n_agents, double* p, double* x,
                                    const double* lbg, const double* ubg, double* lam_g, int32_t* status,
                                    int32_t* iters, double* dw_state, const omgx_port_step_desc* step) {
  PortPool* pp = (PortPool*)h;
  const omgx::Dims& d = pp->plan.dims;
  omgx::Opts o;
  o.tol = opt->tol; o.max_iter = opt->max_iter; o.mu_init = opt->mu_init;
  o.kappa_push = opt->kappa_push; o.nu_init = opt->nu_init; o.scale_gmax = opt->scale_gmax;
  o.warm_start = opt->warm_start; o.kappa_warm = opt->kappa_warm;
  o.dw_leaf_ratio_cold = opt->dw_leaf_ratio_cold > 0 ? opt->dw_leaf_ratio_cold : 1.0;
  o.prio_iter = 0; o.warm_mu_factor = opt->warm_mu_factor >= 0 ? opt->warm_mu_factor : 0.0;
  o.warm_z_floor = opt->warm_z_floor >= 0 ? opt->warm_z_floor : 0.0; o.warm_z_cap = opt->warm_z_cap >= 0 ? opt->warm_z_cap : 0.0; o.max_soc = opt->max_soc > 0 ? (opt->max_soc > 8 ? 8 : opt->max_soc) : 0; o.hess_approx = opt->hess_approx > 0 ? 1 : 0;
  o.compl_tol = opt->compl_inf_tol > 0 ? opt->compl_inf_tol : 0.0; o.viol_tol = opt->constr_viol_tol > 0 ? opt->constr_viol_tol : 0.0;
  o.refine = opt->refine > 0 ? 1 : 0;
  std::atomic<int> next(0);
  pp->run([&](int tid) {
    double* buf = pp->bufs[tid].data();
    size_t wd = omgx::work_doubles(d, pp->plan.kkt_doubles) + 8;
    double* tmp = buf + wd;                                   // [n_var + n_con] scratch of the shifts
    omgx::Work w;
    omgx::work_carve(w, buf, d, pp->plan.kkt_doubles);
    omgx::Ctx c; c.red = w.red;
    for (int b = next.fetch_add(1); b < n_agents; b = next.fetch_add(1)) {
      double* pb = p + (size_t)b * d.n_par;
      double* xb = x + (size_t)b * d.n_var;
      double* lb_ = lam_g + (size_t)b * d.n_con;
      if (step) {
        const omgx_port_step_desc& s = *step;
        for (int k = 0; k < s.n_dim; ++k) {                   // ideal prediction on the current plan
          double v = 0.0, dv = 0.0;
          for (int j = 0; j < s.L; ++j) { double cj = xb[s.o_spl + k * s.L + j]; v += cj * s.E[j]; dv += cj * s.Ed[j]; }
          pb[s.o_state0 + k] = v; pb[s.o_input0 + k] = dv;
        }
        pb[s.o_t] = s.t_rel;
        for (int q = 0; q < s.n_obst; ++q) {                  // obstacles move on
          const int32_t* ob = s.obst + 4 * q;
          for (int k = 0; k < ob[3]; ++k) {
            pb[ob[0] + k] += s.dt * pb[ob[1] + k] + 0.5 * s.dt * s.dt * pb[ob[2] + k];
            pb[ob[1] + k] += s.dt * pb[ob[2] + k];
          }
        }
        if (s.crossed) {                                      // warm-start shift over one knot interval
          for (int e = 0; e < s.n_shift; ++e) {
            const int32_t* en = s.shift_entries + 4 * e;
            const double* T = s.shift_mats + en[3];
            for (int col = 0; col < en[2]; ++col) {
              double* blk = xb + en[0] + col * en[1];
              for (int i = 0; i < en[1]; ++i) {
                double acc = 0.0;
                for (int j = 0; j < en[1]; ++j) acc += T[i * en[1] + j] * blk[j];
                tmp[i] = acc;
              }
              for (int i = 0; i < en[1]; ++i) blk[i] = tmp[i];
            }
          }
          for (int r = 0; r < d.n_con; ++r) tmp[r] = s.perm[r] >= 0 ? lb_[s.perm[r]] : 0.0;
          for (int r = 0; r < d.n_con; ++r) lb_[r] = tmp[r];
        }
      }
      const double* lb = lbg, *ub = ubg;
      omgx::Result r = omgx::ipm_solve(c, d, pp->plan.tables, o, w, pb, xb, lb, ub,
                                       opt->warm_start ? lb_ : nullptr, opt->warm_start ? status[b] : 0,
                                       pp->plan.kkt_doubles, (opt->warm_start && dw_state) ? dw_state[b] : 0.0);
      if (dw_state) dw_state[b] = r.dw;
      for (int i = 0; i < d.n_var; ++i) xb[i] = w.x[i];
      for (int r_ = 0; r_ < d.n_con; ++r_)
        lb_[r_] = (r.status == 3 || w.rtype[r_] == omgx::ROW_FREE) ? 0.0 : w.rho[r_] * w.z[r_];
      status[b] = r.status; iters[b] = r.iters;
    }
  });
  return OMGX_OK;
}

#ifdef OMGX_COUNT_FACT
extern "C" long omgx_port_cnt(int k, int reset) { long v = omgx_dbg_cnt[k].load(); if (reset) omgx_dbg_cnt[k] = 0; return v; }
extern "C" long omgx_port_nfact(int reset) { long v = omgx_dbg_nfact.load(); if (reset) omgx_dbg_nfact = 0; return v; }
#endif
