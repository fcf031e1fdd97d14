// Developer micro-benchmark (not part of the product): cost of the inner operation of the register-resident
// LDL' -- "every lane subtracts l_i times the value lane k holds" -- with the broadcast done three ways.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../omg-tools_amd/csrc/omgx_core.h"
using namespace omgx;

#define NK 32
#define REPS 16

__global__ __launch_bounds__(512) void bc_kernel(long long* out, double* dout, int n_active_waves) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4096; i += 512) lds[i] = 1.0 + 0.001 * i;
  __syncthreads();
  const bool active = wave < n_active_waves;
  double a[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) a[k] = lds[lane * 33 + k];
  double src = lds[lane + 2048], li = lds[lane + 1024] * 1e-3;
  long long t[8];
  double* col = lds + 3000 + wave * 64;
  t[0] = clock64();
  if (active) {                                  // V0: FMA only (operands in VGPRs)
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
      for (int k = 0; k < NK; ++k) a[k] = fma(-li, src, a[k]);
      li += 1e-9;
    }
  }
  t[1] = clock64();
  if (active) {                                  // V1: v_readlane x2 + FMA
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
      for (int k0 = 0; k0 < NK; k0 += 4) {
        double s[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = readlane_d(src, k0 + q);
#pragma unroll
        for (int q = 0; q < 4; ++q) a[k0 + q] = fma(-li, s[q], a[k0 + q]);
      }
      src += 1e-9;
    }
  }
  t[2] = clock64();
  if (active) {                                  // V2: column through LDS, b64 broadcast reads
    for (int r = 0; r < REPS; ++r) {
      col[lane] = src;
      wave_fence();
#pragma unroll
      for (int k = 0; k < NK; ++k) a[k] = fma(-li, col[k], a[k]);
      src += 1e-9;
      wave_fence();
    }
  }
  t[3] = clock64();
  if (active) {                                  // V3: column through LDS, b128 broadcast reads
    typedef double v2d __attribute__((ext_vector_type(2)));
    for (int r = 0; r < REPS; ++r) {
      col[lane] = src;
      wave_fence();
      const v2d* c2 = (const v2d*)col;
#pragma unroll
      for (int k = 0; k < NK; k += 2) { const v2d s = c2[k >> 1]; a[k] = fma(-li, s.x, a[k]); a[k + 1] = fma(-li, s.y, a[k + 1]); }
      src += 1e-9;
      wave_fence();
    }
  }
  t[4] = clock64();
  if (active) {                                  // V4: v_readlane only (xor-accumulated so it is not removed)
    int acc = 0;
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
      for (int k = 0; k < NK; ++k) acc ^= __builtin_amdgcn_readlane(__double2loint(src), k) + __builtin_amdgcn_readlane(__double2hiint(src), k);
      src += 1e-9;
    }
    a[0] += acc;
  }
  t[5] = clock64();
  if (active) {                                  // V5: ds_bpermute-free alternative: __shfl broadcast (ds_bpermute) + FMA
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
      for (int k = 0; k < NK; ++k) a[k] = fma(-li, __shfl(src, k, 64), a[k]);
      src += 1e-9;
    }
  }
  t[6] = clock64();
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < NK; ++k) acc += a[k];
  dout[blockIdx.x * 512 + tid] = acc + src;
  if (tid == 0) for (int i = 0; i < 6; ++i) out[blockIdx.x * 8 + i] = t[i + 1] - t[i];
}

int main() {
  long long* d_out; double* d_d;
  const int nb = 256;
  hipMalloc(&d_out, nb * 8 * sizeof(long long));
  hipMalloc(&d_d, nb * 512 * sizeof(double));
  const char* names[6] = {"FMA only (VGPR operands)", "2 x v_readlane + FMA", "LDS column, b64 broadcast read + FMA", "LDS column, b128 broadcast read + 2 FMA",
                          "2 x v_readlane only", "__shfl (ds_bpermute) + FMA"};
  for (int nw = 8; nw >= 1; nw >>= 1) {
    hipLaunchKernelGGL(bc_kernel, dim3(nb), dim3(512), 64 * 1024, 0, d_out, d_d, nw);
    hipLaunchKernelGGL(bc_kernel, dim3(nb), dim3(512), 64 * 1024, 0, d_out, d_d, nw);
    hipDeviceSynchronize();
    long long ho[256 * 8];
    hipMemcpy(ho, d_out, sizeof(ho), hipMemcpyDeviceToHost);
    printf("active waves %d (cycles per broadcast-and-FMA, %d per repetition)\n", nw, NK);
    for (int k = 0; k < 6; ++k) { double s = 0; for (int b = 0; b < nb; ++b) s += ho[b * 8 + k]; printf("  %-42s %8.2f\n", names[k], s / nb / (NK * REPS)); }
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
