// Developer micro-benchmark (not part of the product): latencies of the building blocks of the
// per-agent solve on one CU with the solve kernel's shape (512 threads, 1 workgroup per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../omg-tools_amd/csrc/omgx_core.h"
using namespace omgx;

__global__ __launch_bounds__(512) void lat_kernel(long long* out, double* dout, int n_active_waves) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 512) lds[i] = 1.0 + 0.001 * i;
  int* ilds = (int*)(lds + 4096);
  for (int i = tid; i < 1024; i += 512) ilds[i] = (i * 37 + 11) & 1023;
  __syncthreads();
  long long t[13];
  double acc = 0.0;
  const bool active = (tid >> 6) < n_active_waves;
  // 0: barrier x 64
  t[0] = clock64();
  for (int i = 0; i < 64; ++i) __syncthreads();
  t[1] = clock64();
  // 1: blk4_from chain x 16 (dependent through g00)
  if (active) {
    double g = lds[tid];
    for (int i = 0; i < 16; ++i) {
      Blk4 B = blk4_from(g + 4.0, 0.1, g + 5.0, 0.2, 0.3, g + 6.0, 0.1, 0.2, 0.3, g + 7.0);
      g = B.i3 + B.l32;
    }
    acc += g;
  }
  t[2] = clock64();
  // 2: dependent LDS chase x 64
  if (active) {
    int j = tid & 1023;
    for (int i = 0; i < 64; ++i) j = ilds[j];
    acc += j;
  }
  t[3] = clock64();
  // 3: 64 independent b64 reads (row-per-lane, stride 37) + 64 fma
  if (active) {
    double s = 0.0;
#pragma unroll 16
    for (int i = 0; i < 64; ++i) s += lds[(tid & 63) * 37 + i];
    acc += s;
  }
  t[4] = clock64();
  // 4: 64 LDS atomic adds to distinct addresses
  if (active) {
    for (int i = 0; i < 64; ++i) atomicAdd(&lds[(tid * 7 + i * 513) & 4095], 1.0);
  }
  __syncthreads();
  t[5] = clock64();
  // 5: rcp_pivot chain x 64
  if (active) {
    double g = lds[tid] + 2.0;
    for (int i = 0; i < 64; ++i) g = rcp_pivot(g) + 1.5;
    acc += g;
  }
  t[6] = clock64();
  // 6: dependent fma chain x 256
  if (active) {
    double g = lds[tid];
#pragma unroll 16
    for (int i = 0; i < 256; ++i) g = fma(g, 0.999, 0.5);
    acc += g;
  }
  t[7] = clock64();
  // 7: L2 global dependent load chain x 32 (pointer chase in global memory)
  if (active) {
    int j = tid;
    const int* gp = (const int*)(dout + 65536);
    for (int i = 0; i < 32; ++i) j = gp[j & 65535];
    acc += j;
  }
  t[8] = clock64();
  // 8: wave barrier + fence x 64 (wave_sync)
  if (active) {
    for (int i = 0; i < 64; ++i) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      lds[tid] += 1.0;
    }
  }
  t[9] = clock64();
  // 9: 64 dependent f64 MFMA 16x16x4 (one accumulator chain)
  if (active) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    v4d a4 = {0.0, 0.0, 0.0, 0.0};
    double av = lds[tid], bv = lds[tid + 1];
    for (int i = 0; i < 64; ++i) a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, a4, 0, 0, 0);
    acc += a4[0] + a4[1] + a4[2] + a4[3];
  }
  t[10] = clock64();
  // 10: 16 x (3 LDS loads -> mul -> MFMA) serialised like the panel update
  if (active) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    v4d a4 = {0.0, 0.0, 0.0, 0.0};
    int j = tid & 255;
    for (int i = 0; i < 16; ++i) {
      const double a_l = lds[j + 4 * i], b_l = lds[j + 300 + 4 * i], d_l = lds[(4 * i + (tid >> 4 & 3)) * 38];
      a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_l * d_l, b_l, a4, 0, 0, 0);
    }
    acc += a4[0] + a4[1] + a4[2] + a4[3];
  }
  t[11] = clock64();
  dout[blockIdx.x * 512 + tid] = acc;
  if (tid == 0) for (int i = 0; i < 11; ++i) out[blockIdx.x * 16 + i] = t[i + 1] - t[i];
}

int main() {
  long long* d_out; double* d_d;
  const int nb = 256;
  hipMalloc(&d_out, nb * 16 * sizeof(long long));
  hipMalloc(&d_d, (65536 + 65536) * sizeof(double));
  int* h = (int*)malloc(65536 * sizeof(int));
  for (int i = 0; i < 65536; ++i) h[i] = (int)(((long long)i * 40503 + 12345) & 65535);
  hipMemcpy(d_d + 65536, h, 65536 * sizeof(int), hipMemcpyHostToDevice);
  const char* names[11] = {"64 x __syncthreads", "16 x blk4_from chain", "64 x dependent LDS load", "64 indep b64 reads + add",
                          "64 x LDS atomicAdd f64", "64 x rcp_pivot chain", "256 x dependent fma", "32 x dependent global (L2) load", "64 x wave_sync + LDS rmw", "64 x dependent f64 MFMA 16x16x4", "16 x (3 LDS loads, mul, MFMA)"};
  for (int nw = 8; nw >= 1; nw >>= 1) {
    hipLaunchKernelGGL(lat_kernel, dim3(nb), dim3(512), 150 * 1024, 0, d_out, d_d, nw);
    hipLaunchKernelGGL(lat_kernel, dim3(nb), dim3(512), 150 * 1024, 0, d_out, d_d, nw);
    hipDeviceSynchronize();
    long long ho[256 * 16];
    hipMemcpy(ho, d_out, sizeof(ho), hipMemcpyDeviceToHost);
    printf("active waves %d\n", nw);
    for (int k = 0; k < 11; ++k) { double s = 0; for (int b = 0; b < nb; ++b) s += ho[b * 16 + k]; printf("  %-34s %10.0f cycles\n", names[k], s / nb); }
  }
  return 0;
}
