// Developer micro-benchmark (not part of the product): the factorisation routines of omgx_core.h
// alone in a kernel of the solve kernel's shape, on config-2 sized blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../omg-tools_amd/csrc/omgx_core.h"
using namespace omgx;

__global__ __launch_bounds__(512) void ldl_kernel(long long* out, double* dout, int variant) {
  extern __shared__ double lds[];
  Ctx c; c.red = lds; c.prof = nullptr;
  double* kkt = lds + 64;
  const int n = 36, nc = 28, ld = 37, nl = 4, nr = 39;      // 64 rows per leaf: 4 row waves + 4 block waves
  const int kkt_doubles = nl * (n + nc) * ld + nr * (nr + 1) / 2;
  double* dinv = kkt + kkt_doubles;
  double* col = dinv + 256;
  BMat* Ms = (BMat*)col;
  double* stage = col + OMGX_BMAT_DOUBLES * (OMGX_MAX_LEAF + 1);
  const int pan0 = (int)(stage + OMGX_STAGE_LD * (OMGX_MAX_LEAF + 1) - col);
  if (threadIdx.x == 0) {
    int pan = pan0;
    for (int l = 0; l < nl; ++l) {
      BMat& M = Ms[l];
      M.a = l * (n + nc) * ld; M.ld = ld; M.nfact = n; M.rows = n + nc; M.npos = n; M.dinv = l * n; M.pan = pan; pan += OMGX_PAN_LD * M.rows; M.cpl = 0;
    }
    BMat& Mr = Ms[nl];
    Mr.a = nl * (n + nc) * ld; Mr.ld = 0; Mr.nfact = nr; Mr.rows = nr; Mr.npos = nr; Mr.dinv = -1; Mr.pan = pan0;
  }
  long long tl = 0, tr = 0;
  int bad = 0;
  for (int rep = 0; rep < 8; ++rep) {
    __syncthreads();
    for (int i = threadIdx.x; i < kkt_doubles; i += 512) kkt[i] = 0.01 * ((i * 7919) % 13) - 0.05;
    __syncthreads();
    for (int l = 0; l < nl; ++l) for (int i = threadIdx.x; i < n; i += 512) kkt[l * (n + nc) * ld + i * ld + i] = 50.0 + i;
    for (int i = threadIdx.x; i < nr; i += 512) kkt[nl * (n + nc) * ld + tri(i, i)] = 50.0 + i;
    __syncthreads();
    long long t0 = clock64();
    if (variant == 0) ldl_left4(c, Ms, nl, kkt, dinv, col, &bad); else if (variant == 1) ldl_blocked<1>(c, Ms, nl, kkt, dinv, col, stage, &bad); else ldl_left4_coop<1>(c, Ms, nl, kkt, dinv, col, &bad, nl * (n + nc), n);
    __syncthreads();
    long long t1 = clock64();
    if (variant == 3) ldl_left4_coop<2>(c, Ms + nl, 1, kkt, dinv, col, &bad, nr, nr); else if (variant != 1) ldl_blocked<2>(c, Ms + nl, 1, kkt, dinv, col, stage, &bad); else ldl_left4(c, Ms + nl, 1, kkt, dinv, col, &bad);
    __syncthreads();
    long long t2 = clock64();
    if (rep >= 4) { tl += t1 - t0; tr += t2 - t1; }
  }
  if (blockIdx.x == 0) { for (int i = threadIdx.x; i < kkt_doubles; i += 512) dout[i] = kkt[i]; for (int i = threadIdx.x; i < 256; i += 512) dout[kkt_doubles + i] = dinv[i]; if (threadIdx.x == 0) dout[kkt_doubles + 256] = bad; }
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = tl / 4; out[blockIdx.x * 2 + 1] = tr / 4; }
}

int main() {
  long long* d_out; double* d_d;
  const int nb = 256, nd = 4 * 64 * 37 + 780 + 257;
  hipMalloc(&d_out, nb * 2 * sizeof(long long));
  hipMalloc(&d_d, nd * sizeof(double));
  hipFuncSetAttribute((const void*)ldl_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  static double res[4][4 * 64 * 37 + 780 + 257];
  for (int variant = 0; variant < 4; ++variant) {
    hipLaunchKernelGGL(ldl_kernel, dim3(nb), dim3(512), 150 * 1024, 0, d_out, d_d, variant);
    hipDeviceSynchronize();
    long long ho[512];
    hipMemcpy(ho, d_out, sizeof(ho), hipMemcpyDeviceToHost);
    hipMemcpy(res[variant], d_d, nd * sizeof(double), hipMemcpyDeviceToHost);
    double a = 0, b = 0; for (int i = 0; i < nb; ++i) { a += ho[2 * i]; b += ho[2 * i + 1]; }
    printf("variant %d (%s): 4 leaves (36 + 28 carried rows) %.0f cycles;  root 39: %.0f cycles  bad=%g [%s]\n", variant,
           variant == 0 ? "leaves ldl_left4 / root ldl_blocked" : (variant == 1 ? "leaves ldl_blocked / root ldl_left4" : (variant == 2 ? "leaves ldl_left4_coop / root ldl_blocked" : "leaves ldl_left4_coop / root ldl_left4_coop")), a / nb, b / nb, res[variant][nd - 1], hipGetErrorString(hipGetLastError()));
  }
  // compare the lower parts / carried rows and the inverse pivots
  double md = 0.0, mx = 0.0;
  for (int l = 0; l < 4; ++l) for (int r = 36; r < 64; ++r) for (int k = 0; k < 36; ++k) {      // carried rows (the factorised rows are L in one routine, U = L D in the other)
    const int i = l * 64 * 37 + r * 37 + k;
    md = fmax(md, fabs(res[0][i] - res[2][i])); mx = fmax(mx, fabs(res[0][i]));
  }
  for (int i = 4 * 64 * 37 + 780; i < 4 * 64 * 37 + 780 + 144; ++i) { md = fmax(md, fabs(res[0][i] - res[2][i])); mx = fmax(mx, fabs(res[0][i])); }
  printf("carried rows and inverse pivots of the leaves, ldl_left4 vs ldl_left4_coop: max |difference| = %.3e (max |entry| %.3e)\n", md, mx);
  return 0;
}
