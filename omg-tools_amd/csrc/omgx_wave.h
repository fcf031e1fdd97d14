// omgx_wave.h -- wave-level, register-resident linear algebra for the block-arrow KKT solve (gfx950 only).
//
// One wave owns one small symmetric matrix (a leaf panel or the root block of the KKT store): lane i
// holds row i in registers, a column is a register index.  The right-looking LDL' needs, per pair
// (j, k > j), the entry u_kj of the pivot column in every lane -- a v_readlane broadcast from lane k --
// and one fp64 FMA on the whole wave; there is no LDS traffic and no barrier inside the factorisation
// (the LDS-resident routines it replaces, ldl_left4 / ldl_blocked, re-read every operand from LDS for
// every block step and were bound by LDS issue: 61 k + 69 k cycles for four 36-column leaves and the
// 39-column root of config 2).  All register indices are compile-time constants (full unrolling), the
// run-time order n only cuts the loops short through wave-uniform branches.
//
// Storage convention (same as the routines replaced): factorised rows hold U = L D
// (u_ij = l_ij d_j, the pivot d_i on the diagonal), carried rows hold W = B L^{-T}, inverse pivots in a
// side array.  "Vector rows" are carried rows kept as one value per lane (lane k = column k) instead of
// one row per lane: the phase-I coupling row and the right-hand side of a leaf, the right-hand side of
// the root -- so that config 2's 36 + 28 register rows fill the 64 lanes exactly.
#pragma once

namespace omgx {

// (lane: any wave-uniform value)
__device__ __forceinline__ double readlane_d(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A panel of the KKT store as one wave sees it.  Rows [0, n): the symmetric block (lower part stored);
// rows [n, nreg): carried rows held one per lane; rows [nreg, nreg + nvec): carried rows held as lane
// vectors.  Pivots j < npos must be positive, the others negative.  Two storage forms:
//   packed (root block, band < 0): row r starts at base + r (r + 1) / 2;
//   banded (leaf panels, band >= 0): row i of the symmetric block keeps its entries (i, i - band) .. (i, i) at
//     base + i * ldb (a leaf in reverse Cuthill-McKee order has no fill outside its band), the carried rows are
//     dense rows of n entries at wbase + (r - n) * ld.
struct WPanel { int base, ld, n, nreg, nvec, npos, bw, vrow, band, ldb, wbase; };    // bw: half bandwidth the factorisation works with (n - 1: dense); vrow: index of the first vector row

// The descriptor fields are the same in every lane but come from LDS (per-lane loads): without this the
// compiler has to treat every loop bound of the routines below as divergent (exec-masked regions
// instead of scalar branches, broadcast values spilled across them).
__device__ __forceinline__ WPanel wpanel_uniform(const WPanel& Q) {
  WPanel P;
  P.base = __builtin_amdgcn_readfirstlane(Q.base); P.ld = __builtin_amdgcn_readfirstlane(Q.ld);
  P.n = __builtin_amdgcn_readfirstlane(Q.n);
  P.nreg = __builtin_amdgcn_readfirstlane(Q.nreg); P.nvec = __builtin_amdgcn_readfirstlane(Q.nvec);
  P.npos = __builtin_amdgcn_readfirstlane(Q.npos); P.bw = __builtin_amdgcn_readfirstlane(Q.bw);
  P.vrow = __builtin_amdgcn_readfirstlane(Q.vrow); P.band = __builtin_amdgcn_readfirstlane(Q.band);
  P.ldb = __builtin_amdgcn_readfirstlane(Q.ldb); P.wbase = __builtin_amdgcn_readfirstlane(Q.wbase);
  return P;
}

// start of carried row r >= n (a dense row of n entries)
template <bool BANDED>
__device__ __forceinline__ int wcarried(const WPanel& P, int r) {
  return BANDED ? P.wbase + (r - P.n) * P.ld : P.base + ((r * (r + 1)) >> 1);
}
// offset such that entry (i, k) of the symmetric block sits at wsym(P, i) + k (banded: only i - band <= k <= i exist)
template <bool BANDED>
__device__ __forceinline__ int wsym(const WPanel& P, int i) {
  return BANDED ? P.base + i * P.ldb + P.band - i : P.base + ((i * (i + 1)) >> 1);
}

// All dynamic LDS of the workgroup (HIP: every `extern __shared__` array starts at the same address): the
// out-of-line routines below take offsets into it, so that their accesses stay ds_* instructions (a
// pointer argument would be a generic pointer: flat_* instructions).
extern __shared__ double omgx_lds[];

// In-place LDL' of a panel by one wave, the matrix in registers: lane i holds row i, a[k] = A[i][k].
// Straight-line code: all NC columns are processed whatever the order n is (the columns >= n work on
// garbage that is never stored and raise no flag), the half bandwidth BW is a compile-time bound (columns further than
// BW right of the pivot are skipped: u_kj = 0 there, nothing to subtract from any row), so all register
// indices and all broadcast lanes are constants and there is no branch inside the factorisation --
// every run-time loop bound tried (uniform scalar branches per chunk, rolled block loops with a shifting
// register window and run-time broadcast lanes) made the compiler hoist, spill or relocate the chunks,
// at 30-40 cycles per broadcast-and-FMA instead of the 12 of this form (tools/micro/bcast.hip: two
// v_readlane + one fp64 FMA).  Details that matter (each measured on the device):
//   * the descriptor must be provably wave-uniform (wpanel_uniform);
//   * the four FMAs of a chunk are pinned behind their broadcasts by an empty volatile asm: without it
//     the compiler issues the broadcasts of a whole column first and spills their SGPRs (v_writelane);
//   * lane masks are formed from a laundered lane id where they are used, otherwise all NC of them are
//     hoisted to the top and spilled;
//   * a pivot of the wrong sign is only recorded; the caller looks at the flag at the end (the panel is
//     garbage then, but nothing traps).
// On return the panel holds U = L D (rows < n, lower part and diagonal) / W = B L^{-T} (carried rows), the
// vector rows are stored forward-substituted.  Returns 1 if a pivot had the wrong sign.  `off`: offset of
// the KKT store in the dynamic LDS (doubles).  BANDED: storage form of the panel (WPanel).
template <int NC, int BW, bool BANDED>
__device__ __forceinline__ int wave_ldl(int off, const WPanel Pin) {
  const double* A = omgx_lds + off;
  double* Aw = omgx_lds + off;
  const WPanel P = wpanel_uniform(Pin);
  const int lane = threadIdx.x & 63;
  const int n = P.n;
  const bool has_row = lane < P.nreg;
  const bool sym_row = lane < n;                       // a row of the symmetric block: only its lower part is stored
  const int rl = has_row ? lane : 0;
  const int ra = rl < n ? wsym<BANDED>(P, rl) : wcarried<BANDED>(P, rl);
  // stored columns of this lane's row: [klo, khi]
  const int klo = (BANDED && rl < n && rl > P.band) ? rl - P.band : 0;
  const int khi = rl < n ? rl : n - 1;
  double a[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {                       // unconditional loads (all in flight), masked afterwards
    const int kk = k < klo ? klo : (k > khi ? khi : k);
    const double v = A[ra + kk];
    a[k] = (has_row && k >= klo && k <= khi) ? v : 0.0;
  }
  const int v0a = wcarried<BANDED>(P, P.vrow), v1a = wcarried<BANDED>(P, P.vrow + (P.nvec > 1 ? 1 : 0));
  double yv0, yv1;
  {
    const int c = sym_row ? lane : 0;
    const double v0 = A[v0a + c], v1 = A[v1a + c];
    yv0 = (P.nvec > 0 && sym_row) ? v0 : 0.0;
    yv1 = (P.nvec > 1 && sym_row) ? v1 : 0.0;
  }
  int bad = 0;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const double dj = readlane_d(a[j], j);
    const bool okp = (j < P.npos) ? (dj > 0.0) : (dj < 0.0);
    bad |= (j < n && !okp) ? 1 : 0;                    // (columns >= n: padding, whatever they compute is never stored)
    const double inv = rcp_pivot(dj);
    const double li = a[j] * inv;                      // l_ij of every row i > j (lanes <= j: unused upper part)
    {
      // the vector rows live on the lanes of the block rows: only rows below the pivot take part
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const double lm = ln > j ? li : 0.0;
      // (columns >= n broadcast zero: their garbage may hold NaN, and 0 * NaN would reach the real lanes)
      const double y0j = readlane_d(yv0, j), y1j = readlane_d(yv1, j);
      yv0 = fma(-lm, j < n ? y0j : 0.0, yv0);
      yv1 = fma(-lm, j < n ? y1j : 0.0, yv1);
    }
#pragma unroll
    for (int c = j / 4; c < NC / 4; ++c) {             // columns in chunks of four
      if (4 * c > j + BW) continue;                    // (compile time)
      double s[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) s[q] = readlane_d(a[j], 4 * c + q);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = 4 * c + q;
        if (k > j && k <= j + BW) a[k] = fma(-li, s[q], a[k]);
      }
      asm volatile("" : "+v"(a[4 * c]), "+v"(a[4 * c + 1]), "+v"(a[4 * c + 2]), "+v"(a[4 * c + 3]));
    }
  }
  if (!bad) {
    if (has_row) {
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // (a row of the symmetric block: its stored part [klo, lane]; a carried row: all n columns)
        if (k < n && (sym_row ? (k <= ln && k >= klo) : true)) Aw[ra + k] = a[k];
      }
    }
    if (sym_row) {
      if (P.nvec > 0) Aw[v0a + lane] = yv0;
      if (P.nvec > 1) Aw[v1a + lane] = yv1;
    }
  }
  return bad;
}

// Round 6: the packed root block by FOUR waves.  wave_ldl<40, 40, false> on one wave is a chain of 39 pivots at ~770 cycles each (two
// v_readlane + one FMA per pair, 741 pairs, three waves parked: 35 k of the 76 k factorisation cycles of a solve).  Here the columns are
// dealt round-robin to the waves (wave w holds columns w, w + 4, ...: NC / 4 registers per lane, lane i = row i as before); per pivot
// the owner publishes its column and the reciprocal of its pivot in LDS (double-buffered: ONE workgroup barrier per pivot), every wave reads its
// row's entry and the reciprocal, and the entries u_jk of its own columns as broadcast reads (all lanes one address: no v_readlane, no scalar-register traffic),
// and updates its <= NC / 4 columns.  The arithmetic of every stored entry is that of wave_ldl -- a_ik <- fma(-(a_ij / d_j), u_kj, a_ik), the
// pivots in the same order, the right-hand-side row as a lane vector updated with the same operands (kept by every wave) -- so the
// factor is the SAME BITS.  All threads of the workgroup call it (waves beyond the fourth only take part in the barriers); `xoff`:
// offset of 145 doubles of exchange buffer in the dynamic LDS; returns 1 (in every thread) if a pivot had the wrong sign -- nothing is
// stored then -- and ends with a barrier (the factor is visible to the workgroup).
// (W: this wave's index as a compile-time constant -- which columns it holds, whether it publishes the next one: nothing is decided at
// run time but k < n; with the wave index in a register every pivot was a thicket of scalar branches, 730 cycles instead of 600 for the
// un-pipelined form.  W = 4: a wave beyond the fourth, barriers only.)
template <int NC, int W>
__device__ __forceinline__ int root_ldl_4w_wave(int off, const WPanel P, int xoff) {
  constexpr int NL = NC / 4;
  constexpr int BS = 72;                               // a buffer: [0, 64) the column, [64] the reciprocal of its pivot
  constexpr bool act = W < 4;
  const double* A = omgx_lds + off;
  double* Aw = omgx_lds + off;
  double* X = omgx_lds + xoff;
  const int lane = threadIdx.x & 63;
  const int n = P.n;
  const bool has_row = lane < n;                       // (the root has no carried register rows: nreg == n)
  const int rl = has_row ? lane : 0;
  const int ra = wsym<false>(P, rl);
  double a[NL];
#pragma unroll
  for (int c = 0; c < NL; ++c) {                       // unconditional loads (all in flight), masked afterwards
    const int j = 4 * c + (act ? W : 0);
    const int jj = j > rl ? rl : j;
    const double v = A[ra + jj];
    a[c] = (has_row && j <= rl) ? v : 0.0;
  }
  const int v0a = wcarried<false>(P, P.vrow);
  double yv0;
  {
    const double v0 = A[v0a + rl];
    yv0 = (P.nvec > 0 && has_row) ? v0 : 0.0;
  }
  int bad = 0;
  if (W == 0) {
    X[lane] = a[0];
    const double d0 = readlane_d(a[0], 0);
    if (lane == 0) X[64] = rcp_pivot(d0);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    if (k < n) {                                        // (wave-uniform; the only run-time decision)
      const double* buf = X + (k & 1) * BS;
      double* nxt = X + ((k + 1) & 1) * BS;
      if (act) {
        const bool pub = (k + 1 < NC) && (((k + 1) & 3) == W);      // (compile time) this wave owns column k + 1
        const int c1 = pub ? (k + 1) >> 2 : 0;
        const double colk = buf[lane], inv = buf[64], s1 = buf[pub ? k + 1 : 0], dk = buf[k];
        const double li = colk * inv;
        if (pub) {                                      // the next pivot's column first, published at once (with the reciprocal of its pivot)
          a[c1] = fma(-li, s1, a[c1]);
          nxt[lane] = a[c1];
          const double d1 = readlane_d(a[c1], k + 1);
          if (lane == 0) nxt[64] = rcp_pivot(d1);
        }
        const bool okp = (k < P.npos) ? (dk > 0.0) : (dk < 0.0);
        bad |= okp ? 0 : 1;
        {
          int ln = lane;
          asm volatile("" : "+v"(ln));
          const double lm = ln > k ? li : 0.0;
          const double y0k = readlane_d(yv0, k);
          yv0 = fma(-lm, y0k, yv0);
        }
#pragma unroll
        for (int c = k >> 2; c < NL; ++c) {
          const int j = 4 * c + W;
          if (j > k + 1) {                              // (compile time; column k + 1: done above by its owner)
            const double s = buf[j];                    // u_jk: every lane reads the same address
            a[c] = fma(-li, s, a[c]);
          }
        }
      }
      __syncthreads();
    }
  }
  if (!bad && act) {
    if (has_row) {
#pragma unroll
      for (int c = 0; c < NL; ++c) {
        const int j = 4 * c + W;
        if (j < n && j <= lane) Aw[ra + j] = a[c];
      }
    }
    if (W == 0 && has_row && P.nvec > 0) Aw[v0a + lane] = yv0;
  }
  // (every wave saw the same pivots: `bad` is the same in all of them; waves beyond the fourth learn it through the buffer)
  if (W == 0 && lane == 0) X[2 * BS] = bad ? 1.0 : 0.0;
  __syncthreads();
  const int bad_all = X[2 * BS] != 0.0 ? 1 : 0;
  __syncthreads();
  return bad_all;
}

template <int NC>
__device__ __forceinline__ int root_ldl_4w(int off, const WPanel Pin, int xoff) {
  static_assert(NC % 4 == 0, "columns are dealt to four waves");
  const WPanel P = wpanel_uniform(Pin);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  switch (wv) {
    case 0: return root_ldl_4w_wave<NC, 0>(off, P, xoff);
    case 1: return root_ldl_4w_wave<NC, 1>(off, P, xoff);
    case 2: return root_ldl_4w_wave<NC, 2>(off, P, xoff);
    case 3: return root_ldl_4w_wave<NC, 3>(off, P, xoff);
    default: return root_ldl_4w_wave<NC, 4>(off, P, xoff);
  }
}

// inverse pivot of column `lane`, read back from the stored diagonal (after wave_ldl + wave_fence)
template <bool BANDED>
__device__ __forceinline__ double wave_dinv(const double* A, const WPanel P) {
  const int lane = threadIdx.x & 63;
  const int c = lane < P.n ? lane : 0;
  const double d = A[wsym<BANDED>(P, c) + c];
  return lane < P.n ? rcp_pivot(d) : 0.0;
}

// x <- L^{-T} z for the factor stored in the panel (U = L D in LDS, left by wave_ldl): lane j reads
// column j (u_ij, i > j), scales it by its own inverse pivot and the wave substitutes backwards with one
// broadcast per row.  z: component `lane` (lanes >= n ignored); returns x_lane.
template <int NC, bool BANDED>
__device__ __forceinline__ double wave_bwd(int off, const WPanel Pin, double dinvl, double z) {
  const double* A = omgx_lds + off;
  const WPanel P = wpanel_uniform(Pin);
  const int lane = threadIdx.x & 63;
  const int n = P.n;
  double lt[NC];
  const int c = lane < n ? lane : 0;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int ic = i < n ? i : n - 1;
    // (banded: entry (i, c) exists for i - c <= band; outside the band the factor is zero)
    const bool in = !BANDED || ic - c <= P.band;
    const int cc = in ? c : ic - P.band;
    const double v = A[wsym<BANDED>(P, ic) + cc];
    lt[i] = (i < n && lane < i && in) ? v * dinvl : 0.0;
  }
  double x = lane < n ? z : 0.0;
#pragma unroll
  for (int i = NC - 1; i >= 1; --i) {
    if (i < n) x = fma(-lt[i], readlane_d(x, i), x);
  }
  return x;
}

// Substitutions for a right-hand side that did NOT ride through the factorisation (second-order correction of a trial
// step: one more solve with the factors of the iteration).  Rolled forms -- a run-time loop over chunks of eight columns,
// the broadcast lane a scalar register -- on purpose: they are called from inside the line search, where the scalars of
// the step are alive on top of the iteration's, and n steps of ~15 cycles do not need the straight-line treatment of the
// n^2 / 2 steps of wave_ldl (a second fully unrolled instance of wave_bwd there cost 35 KB of code and spilled).
//
// wave_fwd_r: y <- L^{-1} r.  Lane i reads row i of the symmetric block (u_ij, j < i), scales column j by the inverse
// pivot of lane j (dinvl: lane c holds 1 / d_c).  r: component `lane` (lanes >= n ignored); returns y_lane.
template <bool BANDED>
__device__ __forceinline__ double wave_fwd_r(int off, const WPanel Pin, double dinvl, double r) {
  const double* A = omgx_lds + off;
  const WPanel P = wpanel_uniform(Pin);
  const int lane = threadIdx.x & 63;
  const int n = P.n;
  const int rl = lane < n ? lane : 0;
  const int ra = wsym<BANDED>(P, rl);
  const int klo = (BANDED && rl > P.band) ? rl - P.band : 0;
  double y = lane < n ? r : 0.0;
  for (int j0 = 0; j0 < n; j0 += 8) {
    double l[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {                      // unconditional loads on clamped addresses, masked afterwards
      const int j = j0 + q;
      const int kk = j < klo ? klo : (j > rl ? rl : j);
      const double v = A[ra + kk];
      const double dk = readlane_d(dinvl, j);          // (j < 48: n <= OMGX_WAVE_COLS)
      l[q] = (lane < n && j >= klo && j < rl) ? v * dk : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = j0 + q;
      const double yj = readlane_d(y, j);
      y = fma(-l[q], j < n ? yj : 0.0, y);
    }
  }
  return y;
}

// wave_bwd_r: x <- L^{-T} z, the rolled twin of wave_bwd (lane j reads column j and scales it by its own inverse pivot)
template <bool BANDED>
__device__ __forceinline__ double wave_bwd_r(int off, const WPanel Pin, double dinvl, double z) {
  const double* A = omgx_lds + off;
  const WPanel P = wpanel_uniform(Pin);
  const int lane = threadIdx.x & 63;
  const int n = P.n;
  const int c = lane < n ? lane : 0;
  double x = lane < n ? z : 0.0;
  for (int i0 = ((n + 7) & ~7) - 8; i0 >= 0; i0 -= 8) {
    double l[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q;
      const int ic = i < n ? i : n - 1;
      const bool in = !BANDED || ic - c <= P.band;     // (banded: entry (i, c) exists for i - c <= band)
      const int cc = in ? c : ic - P.band;
      const double v = A[wsym<BANDED>(P, ic) + cc];
      l[q] = (i < n && lane < i && in) ? v * dinvl : 0.0;
    }
#pragma unroll
    for (int q = 7; q >= 0; --q) {
      const int i = i0 + q;
      const double xi = readlane_d(x, i);
      x = fma(-l[q], i < n ? xi : 0.0, x);
    }
  }
  return x;
}

}  // namespace omgx
