// Developer micro-benchmark (not part of the product): the register-resident wave-level LDL' of
// omgx_wave.h on config-2 sized blocks (four leaf panels 36 + 28 register rows + 2 vector rows, one root
// 39 + right-hand side), one workgroup of 512 threads per CU like the solve kernel.  Checks the result
// against a plain host LDL' and prints cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include "../../omg-tools_amd/csrc/omgx_core.h"
using namespace omgx;

#ifndef NCL
#define NCL 40
#endif
#ifndef BWL
#define BWL 35
#endif
#ifndef NCR
#define NCR 40
#endif

struct Geo { int n, ncar, ld, nl, nr, leaf_doubles, root_off, total; };

__host__ __device__ inline Geo geo() {
  Geo g; g.n = 36; g.ncar = 30; g.ld = 37; g.nl = 4; g.nr = 39;
  g.leaf_doubles = (g.n + g.ncar) * g.ld; g.root_off = g.nl * g.leaf_doubles;
  g.total = g.root_off + (g.nr + 1) * (g.nr + 2) / 2;
  return g;
}

// 128 KB of straight-line code: run between the timed sections to evict the instruction cache (EVICT=1),
// the state the routines find inside the solve kernel, where a whole iteration's code passes between two calls
__device__ __noinline__ double evict_icache(double x) {
  double x0 = x, x1 = x + 1.0, x2 = x + 2.0, x3 = x + 3.0;
#pragma unroll
  for (int i = 0; i < 4096; ++i) { x0 = fma(x0, 0.999, 1e-3); x1 = fma(x1, 0.999, 1e-3); x2 = fma(x2, 0.999, 1e-3); x3 = fma(x3, 0.999, 1e-3); }
  return x0 + x1 + x2 + x3;
}
#ifndef EVICT
#define EVICT 0
#endif

__global__ __launch_bounds__(512) void k_wave(const double* in, double* out, double* dinv_out, double* sol_out, long long* cyc, int reps, int n_rt, int nr_rt) {
  extern __shared__ double lds[];
  Geo g = geo(); g.n = n_rt; g.nr = nr_rt;            // (run-time orders, like the solve kernel sees them)
  double* kkt = lds;
  double* dinv = lds + g.total;            // [4*36 + 40]
  double* solb = dinv + 256;               // backward-substitution results
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long long t_leaf = 0, t_root = 0, t_bwd = 0;
  int bad_any = 0;
  for (int rep = 0; rep < reps; ++rep) {
    __syncthreads();
    for (int i = threadIdx.x; i < g.total; i += blockDim.x) kkt[i] = in[i];
    if (EVICT) solb[300 + (threadIdx.x & 63)] = evict_icache(kkt[threadIdx.x]);
    __syncthreads();
    long long t0 = clock64();
    if (wave < g.nl) {
      WPanel P; P.base = wave * g.leaf_doubles; P.ld = g.ld; P.packed = 0; P.n = g.n; P.nreg = g.n + g.ncar - 2; P.nvec = 2; P.npos = g.n; P.bw = BWL;
      const int bad = (BWL <= 8) ? wave_ldl<NCL, 8>(0, P) : wave_ldl<NCL, NCL>(0, P);
      bad_any |= bad;
      wave_fence();
      const double dl = wave_dinv(kkt, P);
      if (lane < g.n) dinv[wave * g.n + lane] = dl;
    }
    __syncthreads();
    if (EVICT) solb[300 + (threadIdx.x & 63)] = evict_icache(kkt[threadIdx.x]);
    __syncthreads();
    long long t1 = clock64();
    if (wave == 0) {
      WPanel P; P.base = g.root_off; P.ld = 0; P.packed = 1; P.n = g.nr; P.nreg = g.nr; P.nvec = 1; P.npos = 29; P.bw = g.nr;
      const int bad = wave_ldl<NCR, NCR>(0, P);
      bad_any |= bad;
      wave_fence();
      const double dl = wave_dinv(kkt, P);
      const double x = wave_bwd<NCR>(0, P, dl, kkt[wrow(P, P.n) + (lane < P.n ? lane : 0)] * dl);
      if (lane < g.nr) { dinv[g.nl * g.n + lane] = dl; solb[g.nl * g.n + lane] = x; }
    }
    __syncthreads();
    long long t2 = clock64();
    if (wave < g.nl) {          // leaves: backward substitution of the carried right-hand side (no root correction here)
      WPanel P; P.base = wave * g.leaf_doubles; P.ld = g.ld; P.packed = 0; P.n = g.n; P.nreg = g.n + g.ncar - 2; P.nvec = 2; P.npos = g.n; P.bw = BWL;
      const double dl = lane < g.n ? dinv[wave * g.n + lane] : 0.0;
      const double z = lane < g.n ? kkt[wrow(P, P.nreg + 1) + lane] * dl : 0.0;
      const double x = wave_bwd<NCL>(0, P, dl, z);
      if (lane < g.n) solb[wave * g.n + lane] = x;
    }
    __syncthreads();
    long long t3 = clock64();
    if (rep >= reps / 2) { t_leaf += t1 - t0; t_root += t2 - t1; t_bwd += t3 - t2; }
  }
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < g.total; i += blockDim.x) out[i] = kkt[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) { dinv_out[i] = dinv[i]; sol_out[i] = solb[i]; }
  }
  if (threadIdx.x == 0) {
    const int half = reps - reps / 2;
    cyc[blockIdx.x * 4] = t_leaf / half; cyc[blockIdx.x * 4 + 1] = t_root / half; cyc[blockIdx.x * 4 + 2] = t_bwd / half; cyc[blockIdx.x * 4 + 3] = bad_any;
  }
}

// host reference: right-looking LDL' of the first n rows, carried rows follow; U = L D convention
static void host_ldl(std::vector<double>& M, int n, int rows, std::vector<double>& dinv) {   // M: rows x n, row-major full rows (lower part used)
  dinv.assign(n, 0.0);
  for (int j = 0; j < n; ++j) {
    const double d = M[j * n + j];
    dinv[j] = 1.0 / d;
    for (int i = j + 1; i < rows; ++i) {
      const double l = M[i * n + j] / d;
      const int kmax = i < n ? i : n - 1;
      for (int k = j + 1; k <= kmax; ++k) M[i * n + k] -= l * M[k * n + j];
    }
  }
}

int main() {
  const Geo g = geo();
  std::vector<double> in(g.total, 0.0);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
  for (int l = 0; l < g.nl; ++l) {
    double* P = in.data() + l * g.leaf_doubles;
    for (int i = 0; i < g.n + g.ncar; ++i) for (int k = 0; k < g.n; ++k) {
      if (i < g.n) { if (k <= i) P[i * g.ld + k] = (i == k) ? 6.0 + rnd() : (i - k <= BWL ? 0.3 * rnd() : 0.0); }
      else P[i * g.ld + k] = rnd();
    }
  }
  {
    double* R = in.data() + g.root_off;
    for (int i = 0; i <= g.nr; ++i) for (int k = 0; k <= i && k < g.nr; ++k) {
      double v = 0.3 * rnd();
      if (i == k) v = (i < 29) ? 7.0 + rnd() : -(7.0 + rnd());
      if (i == g.nr) v = rnd();
      R[i * (i + 1) / 2 + k] = v;
    }
  }
  double *d_in, *d_out, *d_dinv, *d_sol; long long* d_cyc;
  const int nb = 256;
  hipMalloc(&d_in, g.total * 8); hipMalloc(&d_out, g.total * 8); hipMalloc(&d_dinv, 256 * 8); hipMalloc(&d_sol, 256 * 8);
  hipMalloc(&d_cyc, nb * 4 * sizeof(long long));
  hipMemcpy(d_in, in.data(), g.total * 8, hipMemcpyHostToDevice);
  const size_t lds = (g.total + 1024) * 8;
  hipFuncSetAttribute((const void*)k_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_wave, dim3(nb), dim3(512), lds, 0, d_in, d_out, d_dinv, d_sol, d_cyc, 8, g.n, g.nr);
  hipDeviceSynchronize();
  printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
  std::vector<double> out(g.total), dinv(256), sol(256);
  std::vector<long long> cyc(nb * 4);
  hipMemcpy(out.data(), d_out, g.total * 8, hipMemcpyDeviceToHost);
  hipMemcpy(dinv.data(), d_dinv, 256 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(sol.data(), d_sol, 256 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(cyc.data(), d_cyc, nb * 4 * sizeof(long long), hipMemcpyDeviceToHost);
  double a = 0, b = 0, c = 0; long long bad = 0;
  for (int i = 0; i < nb; ++i) { a += cyc[4 * i]; b += cyc[4 * i + 1]; c += cyc[4 * i + 2]; bad |= cyc[4 * i + 3]; }
  printf("wave_ldl: 4 leaves (36 cols, 64 register rows + 2 vector rows, NC=%d) %.0f cycles; root 39 (NC=%d) incl. backward %.0f cycles; leaf backward %.0f cycles; bad=%lld\n",
         NCL, a / nb, NCR, b / nb, c / nb, bad);
  // ---- check against the host factorisation ---------------------------------------------------
  double err = 0.0, mag = 0.0;
  for (int l = 0; l < g.nl; ++l) {
    const int rows = g.n + g.ncar;
    std::vector<double> M(rows * g.n, 0.0), di;
    for (int i = 0; i < rows; ++i) for (int k = 0; k < g.n; ++k) if (i >= g.n || k <= i) M[i * g.n + k] = in[l * g.leaf_doubles + i * g.ld + k];
    host_ldl(M, g.n, rows, di);
    for (int i = 0; i < rows; ++i) for (int k = 0; k < g.n; ++k) if (i >= g.n || k <= i) {
      err = fmax(err, fabs(M[i * g.n + k] - out[l * g.leaf_doubles + i * g.ld + k])); mag = fmax(mag, fabs(M[i * g.n + k]));
    }
    for (int k = 0; k < g.n; ++k) err = fmax(err, fabs(di[k] - dinv[l * g.n + k]));
    // backward substitution of the carried right-hand side
    std::vector<double> x(g.n);
    for (int k = 0; k < g.n; ++k) x[k] = M[(rows - 1) * g.n + k] * di[k];
    for (int i = g.n - 1; i >= 1; --i) for (int j = 0; j < i; ++j) x[j] -= M[i * g.n + j] * di[j] * x[i];
    for (int k = 0; k < g.n; ++k) err = fmax(err, fabs(x[k] - sol[l * g.n + k]));
  }
  {
    const int n = g.nr, rows = n + 1;
    std::vector<double> M(rows * n, 0.0), di;
    for (int i = 0; i < rows; ++i) for (int k = 0; k < n && k <= i; ++k) M[i * n + k] = in[g.root_off + i * (i + 1) / 2 + k];
    host_ldl(M, n, rows, di);
    for (int i = 0; i < rows; ++i) for (int k = 0; k < n && k <= i; ++k) {
      err = fmax(err, fabs(M[i * n + k] - out[g.root_off + i * (i + 1) / 2 + k])); mag = fmax(mag, fabs(M[i * n + k]));
    }
    std::vector<double> x(n);
    for (int k = 0; k < n; ++k) x[k] = M[n * n + k] * di[k];
    for (int i = n - 1; i >= 1; --i) for (int j = 0; j < i; ++j) x[j] -= M[i * n + j] * di[j] * x[i];
    for (int k = 0; k < n; ++k) err = fmax(err, fabs(x[k] - sol[g.nl * g.n + k]));
  }
  printf("max |device - host| over factors, inverse pivots and solutions = %.3e (max |entry| %.3e) -> %s\n", err, mag, err < 1e-10 ? "OK" : "MISMATCH");
  return err < 1e-10 ? 0 : 1;
}
