/* A C caller of libomgx.so shaped like the reference's C++ export:
 *   Point2Point::generateProblem   (export/point2point/Point2Point.cpp:80-91)   -> omgx_template_read + omgx_batch_create
 *   Point2Point::solve             (Point2Point.cpp:207-231: args p, x0, lbg, ubg -> sol x, return_status)
 *                                                                             -> omgx_batch_solve on host buffers
 * Plain C, no Python, no torch: `gcc p2p_solve.c -I include -L csrc -lomgx`.
 *
 *   Point2Point::fillParameterDict (Point2Point.cpp:263-294: conditions and obstacles into p by entry name, with the
 *   offsets export.py:302-353 generates)                                      -> omgx_template_block* on the block table
 *
 *   p2p_solve <template file> plan                 host only: prints what the library derives from the template
 *   p2p_solve <template file> blocks               host only: prints the block table (kind name offset rows cols)
 *   p2p_solve <template file> fill <cond> <out>    host only: cond: int32 n_agents, int32 n_obst, then per agent
 *                                                  state0 [2], poseT [2], T, n_obst x (x [2], rad); out: p [B,n_par] and
 *                                                  x0 [B,n_var] (straight-line guess) filled BY ENTRY NAME
 *   p2p_solve <template file> solve <input> <out>  input: int32 n_agents, then p [B,n_par], x0 [B,n_var], lbg, ubg [n_con];
 *                                                  out:   x [B,n_var], lam_g [B,n_con], status [B], iters [B]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "omgx.h"

static int fail(const char* what) {
  fprintf(stderr, "%s: %s\n", what, omgx_last_error());
  return 1;
}

/* the n-th entry of a kind whose name ends with `suffix` (labels such as vehicle0 / obstacle3 are numbered by the
 * front end; the generated code of the reference carries them too) */
static int find_suffix(const omgx_template* tpl, int kind, const char* suffix, int nth, int32_t* off, int32_t* rows, int32_t* cols) {
  const int n = omgx_template_n_blocks(tpl, kind);
  const size_t ls = strlen(suffix);
  for (int i = 0; i < n; ++i) {
    const char* name;
    if (omgx_template_block_at(tpl, kind, i, &name, off, rows, cols) != OMGX_OK) return -1;
    const size_t ln = strlen(name);
    if (ln >= ls && strcmp(name + ln - ls, suffix) == 0 && nth-- == 0) return 0;
  }
  return -1;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s template plan | blocks | fill conditions output | solve input output\n", argv[0]); return 2; }
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
  if (argv[2][0] == 'b') {
    const char* kinds[3] = {"var", "par", "con"};
    for (int kind = 0; kind < 3; ++kind)
      for (int i = 0; i < omgx_template_n_blocks(tpl, kind); ++i) {
        const char* name; int32_t off, rows, cols;
        if (omgx_template_block_at(tpl, kind, i, &name, &off, &rows, &cols) != OMGX_OK) return fail("omgx_template_block_at");
        printf("%s %s %d %d %d\n", kinds[kind], name, off, rows, cols);
      }
    omgx_template_free(tpl);
    return 0;
  }
  if (argc < 5) return 2;
  if (argv[2][0] == 'f') {
    /* fillParameterDict: conditions and obstacles into p, the initial guess into x0, by entry name */
    FILE* fc = fopen(argv[3], "rb");
    int32_t hdr[2];
    if (!fc || fread(hdr, sizeof(int32_t), 2, fc) != 2) { fprintf(stderr, "bad condition file\n"); return 2; }
    const int Bc = hdr[0], n_obst = hdr[1];
    double* pp = calloc((size_t)Bc * tpl->n_par, sizeof(double));
    double* xx = calloc((size_t)Bc * tpl->n_var, sizeof(double));
    int32_t o_state, o_pose, o_T, o_spl, L, n_spl, r_, c_;
    if (find_suffix(tpl, OMGX_BLOCK_PAR, ".state0", 0, &o_state, &r_, &c_) || find_suffix(tpl, OMGX_BLOCK_PAR, ".poseT", 0, &o_pose, &r_, &c_) ||
        find_suffix(tpl, OMGX_BLOCK_PAR, ".T", 0, &o_T, &r_, &c_) || find_suffix(tpl, OMGX_BLOCK_VAR, ".splines_seg0", 0, &o_spl, &L, &n_spl))
      { fprintf(stderr, "the template lacks an entry\n"); return 1; }
    for (int b = 0; b < Bc; ++b) {
      double cond[5];
      if (fread(cond, sizeof(double), 5, fc) != 5) return 2;
      double* p = pp + (size_t)b * tpl->n_par;
      double* x = xx + (size_t)b * tpl->n_var;
      p[o_state] = cond[0]; p[o_state + 1] = cond[1]; p[o_pose] = cond[2]; p[o_pose + 1] = cond[3]; p[o_T] = cond[4];
      for (int l = 0; l < n_obst; ++l) {
        double ob[3]; int32_t o_x, o_rad;
        if (fread(ob, sizeof(double), 3, fc) != 3) return 2;
        if (find_suffix(tpl, OMGX_BLOCK_PAR, ".x", l, &o_x, &r_, &c_) || find_suffix(tpl, OMGX_BLOCK_PAR, ".rad", l, &o_rad, &r_, &c_)) return 1;
        p[o_x] = ob[0]; p[o_x + 1] = ob[1]; p[o_rad] = ob[2];
      }
      /* straight line from start to goal: spline k of the (L x n_spl) entry occupies [off + k L, off + (k + 1) L) */
      for (int k = 0; k < n_spl; ++k)
        for (int j = 0; j < L; ++j) x[o_spl + k * L + j] = cond[k] + (cond[2 + k] - cond[k]) * (double)j / (double)(L - 1);
    }
    fclose(fc);
    FILE* fo = fopen(argv[4], "wb");
    if (!fo) return 2;
    fwrite(pp, sizeof(double), (size_t)Bc * tpl->n_par, fo);
    fwrite(xx, sizeof(double), (size_t)Bc * tpl->n_var, fo);
    fclose(fo);
    free(pp); free(xx);
    omgx_template_free(tpl);
    return 0;
  }
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
