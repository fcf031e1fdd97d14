// Developer micro-benchmark (not part of the product): the packed root block of the KKT store factorised by one wave --
// wave_ldl<40, 40, false> (column by column in registers) against wave_ldl_packed16<40> (panels of 16, trailing updates on the
// matrix pipe) -- on random quasi-definite matrices of run-time order n <= 40 with a right-hand-side row, checked against a
// plain host LDL' and timed alone and with a second workgroup on the CU (256 threads each, like the solve kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include "../../omg-tools_amd/csrc/omgx_core.h"
#include "root_ldl_variants.h"
using namespace omgx;

__global__ __launch_bounds__(256) void k_root(const double* in, double* out, long long* cyc, int reps, int n, int npos, int blocked) {
  extern __shared__ double lds[];
  const int total = (n + 1) * (n + 2) / 2;
  double* kkt = lds;                       // offset 0 of the dynamic LDS
  const int soff = 1024;                   // scratch behind the block
  const int wave = threadIdx.x >> 6;
  long long t_sum = 0; int bad_any = 0;
  for (int rep = 0; rep < reps; ++rep) {
    __syncthreads();
    for (int i = threadIdx.x; i < total; i += blockDim.x) kkt[i] = in[i];
    __syncthreads();
    const long long t0 = clock64();
    WPanel P; P.base = 0; P.ld = 0; P.n = n; P.nreg = n; P.nvec = 1; P.npos = npos; P.bw = n; P.vrow = n; P.band = -1; P.ldb = 0; P.wbase = 0;
    if (blocked == 2) bad_any |= root_ldl_4w<OMGX_WAVE_COLS>(0, P, soff);      // (round 6: all four waves)
    else if (wave == 0) {
      bad_any |= blocked ? wave_ldl_packed16<OMGX_WAVE_COLS>(0, P, soff) : wave_ldl<OMGX_WAVE_COLS, OMGX_WAVE_COLS, false>(0, P);
      wave_fence();
    }
    const long long t1 = clock64();
    __syncthreads();
    if (rep >= reps / 2) t_sum += t1 - t0;
  }
  if (blockIdx.x == 0) for (int i = threadIdx.x; i < total; i += blockDim.x) out[i] = kkt[i];
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t_sum / (reps - reps / 2); cyc[2 * blockIdx.x + 1] = bad_any; }
}

static void host_ldl(std::vector<double>& M, int n) {   // (n + 1) x n, lower part + the right-hand-side row; U = L D convention
  for (int j = 0; j < n; ++j) {
    const double d = M[j * n + j];
    for (int i = j + 1; i <= n; ++i) {
      const double l = M[i * n + j] / d;
      const int kmax = i < n ? i : n - 1;
      for (int k = j + 1; k <= kmax; ++k) M[i * n + k] -= l * M[k * n + j];
    }
  }
}

int main() {
  int rc = 0;
  const int cases[4][2] = {{40, 30}, {39, 29}, {30, 24}, {17, 12}};
  for (int cs = 0; cs < 4; ++cs) {
    const int n = cases[cs][0], npos = cases[cs][1], total = (n + 1) * (n + 2) / 2;
    std::vector<double> in(total, 0.0);
    unsigned s = 777 + cs;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
    for (int i = 0; i <= n; ++i) for (int k = 0; k <= i && k < n; ++k) {
      double v = 0.4 * rnd();
      if (i == k) v = (i < npos) ? 7.0 + rnd() : -(7.0 + rnd());
      if (i == n) v = rnd();
      in[i * (i + 1) / 2 + k] = v;
    }
    std::vector<double> M((n + 1) * n, 0.0);
    for (int i = 0; i <= n; ++i) for (int k = 0; k < n && k <= i; ++k) M[i * n + k] = in[i * (i + 1) / 2 + k];
    host_ldl(M, n);
    double *d_in, *d_out; long long* d_cyc;
    const int nb_max = 512;
    hipMalloc(&d_in, total * 8); hipMalloc(&d_out, total * 8); hipMalloc(&d_cyc, nb_max * 2 * sizeof(long long));
    hipMemcpy(d_in, in.data(), total * 8, hipMemcpyHostToDevice);
    const size_t lds = (1024 + 256) * 8;
    hipFuncSetAttribute((const void*)k_root, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<double> first;
    for (int blocked = 0; blocked < 3; ++blocked) {
      for (int nb = 256; nb <= 512; nb += 256) {           // one / two workgroups per CU
        hipMemset(d_out, 0, total * 8);
        hipLaunchKernelGGL(k_root, dim3(nb), dim3(256), lds, 0, d_in, d_out, d_cyc, 16, n, npos, blocked);
        hipDeviceSynchronize();
        std::vector<double> out(total); std::vector<long long> cyc(nb * 2);
        hipMemcpy(out.data(), d_out, total * 8, hipMemcpyDeviceToHost);
        hipMemcpy(cyc.data(), d_cyc, nb * 2 * sizeof(long long), hipMemcpyDeviceToHost);
        double c = 0; long long bad = 0;
        for (int i = 0; i < nb; ++i) { c += cyc[2 * i]; bad |= cyc[2 * i + 1]; }
        double err = 0.0, mag = 0.0;
        for (int i = 0; i <= n; ++i) for (int k = 0; k < n && k <= i; ++k) {
          err = fmax(err, fabs(M[i * n + k] - out[i * (i + 1) / 2 + k])); mag = fmax(mag, fabs(M[i * n + k]));
        }
        if (blocked == 0) first = out;
        bool same_bits = true;
        for (int i = 0; i < total; ++i) same_bits = same_bits && out[i] == first[i];
        printf("n %2d (%2d positive pivots)  %-22s %d workgroup(s) per CU  %6.0f cycles  bad %lld  max |device - host| %.2e (max |entry| %.1f) %s, same bits as column by column: %s (%s)\n",
               n, npos, blocked == 2 ? "four waves through LDS" : (blocked ? "panels of 16 + MFMA" : "column by column"), nb / 256, c / nb, bad, err, mag, err < 1e-11 ? "OK" : "MISMATCH", same_bits ? "yes" : "no", hipGetErrorString(hipGetLastError()));
        if (blocked == 2 && !same_bits) rc = 1;
        if (!(err < 1e-11) || bad) rc = 1;
      }
    }
    // a pivot of the wrong sign must be reported and nothing stored
    in[(npos - 1) * npos / 2 + npos - 1] = -3.0;
    hipMemcpy(d_in, in.data(), total * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_root, dim3(1), dim3(256), lds, 0, d_in, d_out, d_cyc, 2, n, npos, 2);
    hipDeviceSynchronize();
    std::vector<double> out(total); long long cyc[2];
    hipMemcpy(out.data(), d_out, total * 8, hipMemcpyDeviceToHost);
    hipMemcpy(cyc, d_cyc, sizeof cyc, hipMemcpyDeviceToHost);
    bool untouched = true;
    for (int i = 0; i < total; ++i) untouched = untouched && out[i] == in[i];
    printf("n %2d wrong-sign pivot: bad %lld, store untouched: %s\n", n, cyc[1], untouched ? "yes" : "NO");
    if (!cyc[1] || !untouched) rc = 1;
    hipFree(d_in); hipFree(d_out); hipFree(d_cyc);
  }
  return rc;
}
