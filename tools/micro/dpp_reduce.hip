// Developer check: the DPP wave reduction of omgx_core.h (CtxT::wave_reduce) against a sequential host reduction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#define OMGX_MICRO 1
template <int OP> static __device__ __forceinline__ double comb(double a, double b) { return OP == 0 ? a + b : (OP == 1 ? fmax(a, b) : fmin(a, b)); }
template <int OP, int CTRL> static __device__ __forceinline__ double dpp_comb(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return comb<OP>(v, __hiloint2double(hi2, lo2));
}
static __device__ __forceinline__ double rl(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <int OP> static __device__ __forceinline__ double wave_reduce(double v) {
  v = dpp_comb<OP, 0x128>(v); v = dpp_comb<OP, 0x124>(v); v = dpp_comb<OP, 0x122>(v); v = dpp_comb<OP, 0x121>(v);
  const double r0 = rl(v, 0), r1 = rl(v, 16), r2 = rl(v, 32), r3 = rl(v, 48);
  return comb<OP>(comb<OP>(r0, r1), comb<OP>(r2, r3));
}
__global__ void k(const double* in, double* out) {
  const double v = in[blockIdx.x * 64 + threadIdx.x];
  const double s = wave_reduce<0>(v), mx = wave_reduce<1>(v), mn = wave_reduce<2>(v);
  out[(blockIdx.x * 64 + threadIdx.x) * 3 + 0] = s; out[(blockIdx.x * 64 + threadIdx.x) * 3 + 1] = mx; out[(blockIdx.x * 64 + threadIdx.x) * 3 + 2] = mn;
}
int main() {
  const int W = 256; std::vector<double> h(W * 64), o(W * 64 * 3);
  srand(3); for (auto& x : h) x = (rand() / (double)RAND_MAX - 0.5) * pow(10.0, rand() % 9 - 4);
  double *d, *e; hipMalloc(&d, h.size() * 8); hipMalloc(&e, o.size() * 8);
  hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, d, e); hipMemcpy(o.data(), e, o.size() * 8, hipMemcpyDeviceToHost);
  double worst = 0; int bad = 0;
  for (int w = 0; w < W; ++w) {
    double s = 0, mx = -1e300, mn = 1e300, sa = 0;
    for (int i = 0; i < 64; ++i) { double x = h[w * 64 + i]; s += x; sa += fabs(x); mx = fmax(mx, x); mn = fmin(mn, x); }
    for (int i = 0; i < 64; ++i) {
      const double* r = &o[(w * 64 + i) * 3];
      worst = fmax(worst, fabs(r[0] - s) / sa);
      if (r[1] != mx || r[2] != mn || fabs(r[0] - s) > 1e-14 * sa) ++bad;
    }
  }
  printf("dpp wave_reduce: %d waves, worst relative sum error %.2e, mismatches %d -> %s\n", W, worst, bad, bad ? "FAIL" : "OK");
  return bad != 0;
}
