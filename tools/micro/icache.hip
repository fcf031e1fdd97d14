// Developer micro-benchmark (not part of the product): cost of executing straight-line code of a given
// size repeatedly on one CU (instruction cache: 64 KB shared by two CUs) -- cycles per instruction
// against the size of the loop body, with 1 and with 8 waves per workgroup.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int N>
__global__ __launch_bounds__(512) void ic_kernel(long long* out, double* dout, int reps, int n_active) {
  const int wave = threadIdx.x >> 6;
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1.0, x2 = x0 + 2.0, x3 = x0 + 3.0;
  const double a = 0.999, b = 1e-3;
  long long t0 = 0, t1 = 0;
  if (wave < n_active) {
    for (int r = 0; r < reps; ++r) {
      if (r == 1) t0 = clock64();
#pragma unroll
      for (int i = 0; i < N / 4; ++i) { x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b); }
      asm volatile("" ::: "memory");
    }
    t1 = clock64();
  }
  dout[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int N>
void run(long long* d_out, double* d_d, int nb) {
  const int reps = 9;
  for (int nw = 1; nw <= 8; nw *= 8) {
    hipLaunchKernelGGL(ic_kernel<N>, dim3(nb), dim3(512), 0, 0, d_out, d_d, reps, nw);
    hipLaunchKernelGGL(ic_kernel<N>, dim3(nb), dim3(512), 0, 0, d_out, d_d, reps, nw);
    hipDeviceSynchronize();
    long long ho[256];
    hipMemcpy(ho, d_out, nb * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < nb; ++i) s += ho[i];
    printf("body %4d KB (%6d v_fma_f64), %d active wave(s): %7.2f cycles per instruction (passes 2..%d)\n", N * 8 / 1024, N, nw, s / nb / ((reps - 1) * (double)N), reps);
  }
}

int main() {
  long long* d_out; double* d_d;
  const int nb = 256;
  hipMalloc(&d_out, nb * sizeof(long long));
  hipMalloc(&d_d, nb * 512 * sizeof(double));
  run<1024>(d_out, d_d, nb);
  run<2048>(d_out, d_d, nb);
  run<4096>(d_out, d_d, nb);
  run<6144>(d_out, d_d, nb);
  run<8192>(d_out, d_d, nb);
  run<12288>(d_out, d_d, nb);
  run<16384>(d_out, d_d, nb);
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
