/* A C caller of libomgx.so shaped like the reference's C++ export:
 *   Point2Point::generateProblem   (export/point2point/Point2Point.cpp:80-91)   -> omgx_template_read + omgx_batch_create
 *   Point2Point::solve             (Point2Point.cpp:207-231: args p, x0, lbg, ubg -> sol x, return_status)
 *                                                                             -> omgx_batch_solve on host buffers
 * Plain C, no Python, no torch: `gcc p2p_solve.c -I include -L csrc -lomgx`.
 *
 *   p2p_solve <template file> plan                 host only: prints what the library derives from the template
 *   p2p_solve <template file> solve <input> <out>  input: int32 n_agents, then p [B,n_par], x0 [B,n_var], lbg, ubg [n_con];
 *                                                  out:   x [B,n_var], lam_g [B,n_con], status [B], iters [B]
 */
#include <stdio.h>
#include <stdlib.h>
#include "omgx.h"

static int fail(const char* what) {
  fprintf(stderr, "%s: %s\n", what, omgx_last_error());
  return 1;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s template plan | solve input output\n", argv[0]); return 2; }
  omgx_template* tpl = NULL;
  if (omgx_template_read(argv[1], &tpl) != OMGX_OK) return fail("omgx_template_read");
  if (argv[2][0] == 'p') {
    omgx_plan_info info;
    if (omgx_plan_describe(tpl, &info, NULL) != OMGX_OK) return fail("omgx_plan_describe");
    printf("version %d n_var %d n_con %d n_par %d n_leaf %d n_root %d n_eq %d nnz_j %d wave_path %d lds_bytes %lld\n",
           omgx_version(), tpl->n_var, tpl->n_con, tpl->n_par, info.n_leaf, info.n_root, info.n_eq, info.nnz_j,
           info.wave_path, (long long)info.lds_bytes);
    omgx_template_free(tpl);
    return 0;
  }
  if (argc < 5) return 2;
  FILE* fp = fopen(argv[3], "rb");
  int32_t B = 0;
  if (!fp || fread(&B, sizeof B, 1, fp) != 1 || B <= 0) { fprintf(stderr, "bad input file\n"); return 2; }
  const size_t nv = (size_t)tpl->n_var, np = (size_t)tpl->n_par, nc = (size_t)tpl->n_con;
  double* p = malloc(B * np * sizeof(double));
  double* x0 = malloc(B * nv * sizeof(double));
  double* lbg = malloc(nc * sizeof(double));
  double* ubg = malloc(nc * sizeof(double));
  double* x = malloc(B * nv * sizeof(double));
  double* lam = malloc(B * nc * sizeof(double));
  int32_t* status = malloc(B * sizeof(int32_t));
  int32_t* iters = malloc(B * sizeof(int32_t));
  if (fread(p, sizeof(double), B * np, fp) != B * np || fread(x0, sizeof(double), B * nv, fp) != B * nv ||
      fread(lbg, sizeof(double), nc, fp) != nc || fread(ubg, sizeof(double), nc, fp) != nc) {
    fprintf(stderr, "short input file\n");
    return 2;
  }
  fclose(fp);

  omgx_batch* batch = NULL;                                          /* generateProblem */
  if (omgx_batch_create(tpl, B, 0, &batch) != OMGX_OK) return fail("omgx_batch_create");
  omgx_options opt;
  omgx_default_options(&opt);
  opt.tol = 1e-6;
  opt.max_iter = 500;
  if (omgx_batch_set_options(batch, &opt) != OMGX_OK) return fail("omgx_batch_set_options");
  /* solve(): args["p"], args["x0"], args["lbg"], args["ubg"] -> sol["x"], return_status */
  if (omgx_batch_solve(batch, p, x0, lbg, ubg, x, lam, status, iters, OMGX_BOUNDS_SHARED) != OMGX_OK)
    return fail("omgx_batch_solve");
  int ok = 0;
  for (int b = 0; b < B; ++b) {
    printf("agent %d: %s after %d iterations\n", b, omgx_status_string(status[b]), iters[b]);
    ok += status[b] == OMGX_SOLVE_SUCCEEDED;
  }
  fp = fopen(argv[4], "wb");
  if (!fp) return 2;
  fwrite(x, sizeof(double), B * nv, fp);
  fwrite(lam, sizeof(double), B * nc, fp);
  fwrite(status, sizeof(int32_t), B, fp);
  fwrite(iters, sizeof(int32_t), B, fp);
  fclose(fp);
  omgx_batch_destroy(batch);
  omgx_template_free(tpl);
  free(p); free(x0); free(lbg); free(ubg); free(x); free(lam); free(status); free(iters);
  return ok == B ? 0 : 1;
}
