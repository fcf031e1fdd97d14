// Developer experiment (round 5, NOT part of the product; profiles/r05_micro_root_ldl.txt): the packed root block in panels of
// 16 columns -- each factorised in registers like wave_ldl, the rank-16 update of the columns right of it on the matrix pipe
// (v_mfma_f64_16x16x4), operands and result tiles changing layout through 240 doubles of LDS scratch at `soff`:
//   [0, 96)     the current four panel columns of the rows below the panel, row-major (stride 4)
//   [96, 112)   inverse pivots of the panel's columns
//   [112, 240)  half a result tile (8 rows x 16)
// Same contract as wave_ldl<NC, NC, false>.  Correct (<= 3e-15 against a host LDL'), and no faster: kept for the record.
#pragma once
namespace omgx {
template <int NC>
__device__ __forceinline__ int wave_ldl_packed16(int off, const WPanel Pin, int soff) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  const double* A = omgx_lds + off;
  double* Aw = omgx_lds + off;
  double* S = omgx_lds + soff;
  const WPanel P = wpanel_uniform(Pin);
  const int lane = threadIdx.x & 63;
  const int n = P.n;
  const bool has_row = lane < P.nreg;
  const int rl = has_row ? lane : 0;
  const int ra = wsym<false>(P, rl < n ? rl : 0);
  const int khi = rl < n ? rl : n - 1;
  double a[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int kk = k > khi ? khi : k;
    const double v = A[ra + kk];
    a[k] = (has_row && rl < n && k <= khi) ? v : 0.0;
  }
  const int v0a = wcarried<false>(P, P.vrow);
  double yv0;
  {
    const int c = lane < n ? lane : 0;
    const double v0 = A[v0a + c];
    yv0 = (P.nvec > 0 && lane < n) ? v0 : 0.0;
  }
  int bad = 0;
#pragma unroll
  for (int c0 = 0; c0 < NC; c0 += 16) {
    const int c1 = c0 + 16 < NC ? c0 + 16 : NC;
#pragma unroll
    for (int j = c0; j < c1; ++j) {
      const double dj = readlane_d(a[j], j);
      const bool okp = (j < P.npos) ? (dj > 0.0) : (dj < 0.0);
      bad |= (j < n && !okp) ? 1 : 0;
      const double inv = rcp_pivot(dj);
      const double li = a[j] * inv;
      {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const double lm = ln > j ? li : 0.0;
        const double y0j = readlane_d(yv0, j);
        yv0 = fma(-lm, j < n ? y0j : 0.0, yv0);
        if (ln == 0) S[96 + (j - c0)] = (j < n) ? inv : 0.0;
      }
#pragma unroll
      for (int c = j / 4; c < (c1 + 3) / 4; ++c) {
        double s[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = readlane_d(a[j], 4 * c + q);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = 4 * c + q;
          if (k > j && k < c1) a[k] = fma(-li, s[q], a[k]);
        }
        asm volatile("" : "+v"(a[4 * c]), "+v"(a[4 * c + 1]), "+v"(a[4 * c + 2]), "+v"(a[4 * c + 3]));
      }
    }
#ifndef OMGX_DBG_SKIP_TRAIL
    if (c1 < NC) {
      const int NT = (NC - c1 + 15) / 16;
      v4d acc[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
      const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
      for (int kb = 0; kb < (c1 - c0) / 4; ++kb) {
        wave_fence();
        {
          int ln = lane;
          asm volatile("" : "+v"(ln));
          const int row = ln - c1;
          if (row >= 0 && ln < n) {
#pragma unroll
            for (int q = 0; q < 4; ++q) S[row * 4 + q] = a[c0 + 4 * kb + q];
          }
        }
        wave_fence();
        const double dk = S[96 + 4 * kb + kq];
        double u[2], l[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int row = 16 * b + r16;
          const bool on = b < NT && c1 + row < n;
          const double v = S[(on ? row : 0) * 4 + kq];
          u[b] = on ? v : 0.0;
          l[b] = u[b] * dk;
        }
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(l[0], u[0], acc[0], 0, 0, 0);
        if (NT > 1) {
          acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(l[1], u[0], acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(l[1], u[1], acc[2], 0, 0, 0);
        }
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (t > 0 && NT < 2) continue;
        const int rb = t == 0 ? 0 : 1, cb = t == 2 ? 1 : 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (c1 + 16 * rb + 8 * h >= NC) continue;
          wave_fence();
          S[112 + (kq + 0) * 16 + r16] = acc[t][2 * h];
          S[112 + (kq + 4) * 16 + r16] = acc[t][2 * h + 1];
          wave_fence();
          int ln = lane;
          asm volatile("" : "+v"(ln));
          const int rr = ln - (c1 + 16 * rb + 8 * h);
          const bool mine = rr >= 0 && rr < 8 && ln < n;
          const double* Drow = S + 112 + (mine ? rr : 0) * 16;
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            const int k = c1 + 16 * cb + c;
            if (k < NC) {
              const double dv = Drow[c];
              a[k] = (mine && k <= ln) ? a[k] - dv : a[k];
            }
          }
        }
      }
    }
#endif
  }
  if (!bad) {
    if (has_row && rl < n) {
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if (k < n && k <= ln) Aw[ra + k] = a[k];
      }
    }
    if (lane < n && P.nvec > 0) Aw[v0a + lane] = yv0;
  }
  return bad;
}
}  // namespace omgx
