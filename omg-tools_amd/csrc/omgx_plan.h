// omgx_plan.h -- from the flat NLP (omgx_template, host pointers) to everything the solve kernel works
// from: the static block-arrow structure of the KKT matrix (which variables form leaves, which the
// root), the permuted Jacobian structure, KKT addresses, and the owner-computes tables that make every
// sum of the solve a fixed-order sum (no floating-point atomics anywhere).  Host-side only; shared by
// the HIP library (which uploads the tables) and by the CPU port of the oracle.
//
// What it replaces in the reference: nothing one-to-one -- CasADi derives the sparsity of the
// Lagrangian Hessian / constraint Jacobian (`basics/optilayer.py:49-60`, `expand=True`) and MUMPS does its
// own symbolic analysis; this is the symbolic phase of the hand-written solver.
#pragma once
#include <stdlib.h>
#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <functional>
#include <numeric>
#include <queue>
#include <set>
#include <string>
#include <vector>
#include "../../include/omgx.h"
#include "omgx_core.h"

namespace omgx {

struct HostPlan {
  Dims dims;
  Tables tables;            // host pointers (into tpl and the vectors below)
  int kkt_doubles;
  // ---- structure (built here; a caller-provided plan is not needed) ----------------------------
  std::vector<int32_t> order, pos, blk, leaf_off, leaf_bw, eq_rows, eq_index;
  std::vector<int32_t> jr_ptr, jr_pos, t_jidx, row_leaf, cpl_ptr, cpl_idx, cpl_map;
  std::vector<int32_t> d_off, b_off;
  std::vector<int32_t> lf_w, lf_ldb, lf_band, lf_kind, dl_pos;      // compact store of the wave path (Tables)
  std::vector<std::vector<std::vector<int>>> var_cpl;                // [leaf][variable]: root positions (but t) its entries of B_l can touch
  // ---- addresses and packed records ---------------------------------------------------------------
  std::vector<int32_t> pair4, eqe3, je_row, jt_addr, diag_addr, t_row, tq_addr;
  std::vector<double> reg_w;
  std::vector<MonoRec> pm_rec;
  std::vector<MonoRec8> pm_rec8, sl_ell8;      // (Dims::mono_packed == 2: Tables::pm_rec / sl_ell point at these)
  std::vector<int32_t> je_rp, slot_rng;
  // host only: the terms with >= 2 factors and, per pair of factors, the KKT address of their Hessian entry
  struct HessTerm { double coef; int32_t slot, row; int nv; int32_t v[OMGX_TV], p[OMGX_TV], ha[OMGX_TV * (OMGX_TV - 1) / 2]; };
  std::vector<HessTerm> hrec;
  // ---- owner-computes tables ----------------------------------------------------------------------
  std::vector<int32_t> je_ptr, jv_list, row_perm, cs_ptr, cs_rec, obj_ent;
  std::vector<JItem> je_item;
  std::vector<int32_t> ka_rec, ka_fix, kg_fix, kh_fix;
  std::vector<RowTerm> rt_ell;
  std::vector<JItem> jv_ell, ja_ell;
  std::vector<MonoRec> sl_ell;
  std::vector<int32_t> ja_own, jv_own, ja_glen, sl_list, sl_glen;
  std::vector<int32_t> rt_glen, jp_ell, jp_glen, cs_ell, cs_glen, cs_col, cs_own, jv_glen;
  std::vector<HItem> kh_rec, kg_rec;
  std::vector<int32_t> lift_rec, lift_lev;
  std::string error;

  bool fail(const char* msg) { error = msg; return false; }

  // ------------------------------------------------------------------------------------------------
  // Block-arrow structure.  Variables are permuted into [leaf_0 | leaf_1 | ... | root | t]: no constraint
  // row couples two different leaves (each obstacle's hyperplane variables form a leaf, the
  // trajectory coefficients the root), so the condensed KKT matrix is block-arrow.  Root = variables of
  // the equality rows + the caller's hint (`root_vars`: e.g. the vehicle's spline coefficients); without
  // a hint, the highest-degree variables are moved to the root until every component fits one wave.
  // Components smaller than 8 variables (every coefficient of the terminal slacks g*, which meets the rest
  // of the problem through one trajectory coefficient only) are gathered into one extra leaf.  Inside a
  // leaf the variables are ordered by reverse Cuthill-McKee: the hyperplane block of an obstacle is block
  // tridiagonal by knot (degree-1 splines), its factor keeps that band and the factorisation skips
  // everything outside it.
  // ------------------------------------------------------------------------------------------------
  bool build_structure(const omgx_template& t) {
    const int n = t.n_var, m = t.n_con;
    std::vector<std::vector<int>> rows_vars(m + 1);
    for (int r = 0; r <= m; ++r) {
      std::set<int> s;
      for (int tt = t.row_ptr[r]; tt < t.row_ptr[r + 1]; ++tt)
        for (int k = 0; k < OMGX_TV; ++k) { const int v = t.t_var[OMGX_TV * tt + k]; if (v >= n) return fail("term variable out of range"); if (v >= 0) s.insert(v); }
      rows_vars[r].assign(s.begin(), s.end());
    }
    std::vector<std::vector<int>> obj_cpl;       // nonlinear objective terms couple their variables like a row
    for (int tt = t.row_ptr[m]; tt < t.row_ptr[m + 1]; ++tt) {
      std::set<int> s;
      for (int k = 0; k < OMGX_TV; ++k) if (t.t_var[OMGX_TV * tt + k] >= 0) s.insert(t.t_var[OMGX_TV * tt + k]);
      if (s.size() > 1) obj_cpl.emplace_back(s.begin(), s.end());
    }
    eq_rows.assign(t.eq_rows, t.eq_rows + t.n_eq);
    eq_index.assign(m, -1);
    for (int k = 0; k < t.n_eq; ++k) { if (eq_rows[k] < 0 || eq_rows[k] >= m) return fail("equality row out of range"); eq_index[eq_rows[k]] = k; }
    std::vector<char> in_root(n, 0);
    for (int r : eq_rows) for (int v : rows_vars[r]) in_root[v] = 1;
    for (int i = 0; i < t.n_root_vars; ++i) { const int v = t.root_vars[i]; if (v < 0 || v >= n) return fail("root hint out of range"); in_root[v] = 1; }
    // adjacency of the variables (same row or same nonlinear objective term)
    std::vector<std::set<int>> adj(n);
    auto couple = [&](const std::vector<int>& vs) { for (int a : vs) for (int b : vs) if (a != b) adj[a].insert(b); };
    for (int r = 0; r < m; ++r) couple(rows_vars[r]);
    for (auto& vs : obj_cpl) couple(vs);
    auto components = [&]() {
      std::vector<int> comp(n, -1); std::vector<std::vector<int>> out;
      for (int v = 0; v < n; ++v) {
        if (in_root[v] || comp[v] >= 0) continue;
        std::vector<int> cur; std::queue<int> q; q.push(v); comp[v] = (int)out.size();
        while (!q.empty()) { const int a = q.front(); q.pop(); cur.push_back(a);
          for (int b : adj[a]) if (!in_root[b] && comp[b] < 0) { comp[b] = (int)out.size(); q.push(b); } }
        std::sort(cur.begin(), cur.end());
        out.push_back(cur);
      }
      return out;
    };
    std::vector<std::vector<int>> comps = components();
    if (t.n_root_vars == 0) {
      // no hint: vertex separator by greedy removal of the highest-degree variable of an oversized component
      for (;;) {
        int big = -1;
        for (size_t c = 0; c < comps.size(); ++c) if ((int)comps[c].size() > OMGX_WAVE_ROWS && (big < 0 || comps[c].size() > comps[big].size())) big = (int)c;
        if (big < 0) break;
        int best = -1, bdeg = -1;
        for (int v : comps[big]) { int deg = 0; for (int b : adj[v]) if (!in_root[b]) ++deg; if (deg > bdeg) { bdeg = deg; best = v; } }
        in_root[best] = 1;
        comps = components();
      }
    }
    std::vector<std::vector<int>> leaves; std::vector<int> small;
    for (auto& c : comps) { if ((int)c.size() < OMGX_MIN_LEAF) small.insert(small.end(), c.begin(), c.end()); else leaves.push_back(c); }
    std::sort(small.begin(), small.end());
    if ((int)small.size() >= OMGX_MIN_LEAF) leaves.push_back(small); else for (int v : small) in_root[v] = 1;
    if ((int)leaves.size() > OMGX_MAX_LEAF) return fail("more leaves than OMGX_MAX_LEAF");
    // reverse Cuthill-McKee inside each leaf (restricted to the leaf's own variables)
    leaf_bw.assign(leaves.size(), 0);
    for (size_t l = 0; l < leaves.size(); ++l) {
      std::vector<int>& lv = leaves[l];
      std::set<int> inl(lv.begin(), lv.end());
      auto deg = [&](int v) { int dg = 0; for (int b : adj[v]) if (inl.count(b)) ++dg; return dg; };
      std::vector<int> out; std::set<int> seen;
      while (out.size() < lv.size()) {
        int start = -1;
        for (int v : lv) if (!seen.count(v) && (start < 0 || deg(v) < deg(start))) start = v;
        std::queue<int> q; q.push(start); seen.insert(start);
        while (!q.empty()) {
          const int a = q.front(); q.pop(); out.push_back(a);
          std::vector<int> nb;
          for (int b : adj[a]) if (inl.count(b) && !seen.count(b)) nb.push_back(b);
          std::sort(nb.begin(), nb.end(), [&](int x, int y) { const int dx = deg(x), dy = deg(y); return dx != dy ? dx < dy : x < y; });
          for (int b : nb) { seen.insert(b); q.push(b); }
        }
      }
      std::reverse(out.begin(), out.end());
      std::vector<int> where(n, -1);
      for (size_t i = 0; i < out.size(); ++i) where[out[i]] = (int)i;
      int bw = 0;
      for (int v : out) for (int b : adj[v]) if (where[b] >= 0) bw = std::max(bw, std::abs(where[v] - where[b]));
      lv = out; leaf_bw[l] = bw;
    }
    // order / positions
    order.clear();
    leaf_off.assign(1, 0);
    for (auto& lv : leaves) { order.insert(order.end(), lv.begin(), lv.end()); leaf_off.push_back((int)order.size()); }
    for (int v = 0; v < n; ++v) if (in_root[v]) order.push_back(v);
    order.push_back(n);                               // t last
    Dims& d = dims;
    d.n_leaf = (int)leaves.size(); d.root_off = leaf_off.back(); d.n_root = n + 1 - d.root_off;
    pos.assign(n + 1, -1);
    for (int q = 0; q <= n; ++q) pos[order[q]] = q;
    std::vector<int> leaf_of(n + 1, -1);
    for (size_t l = 0; l < leaves.size(); ++l) for (int v : leaves[l]) leaf_of[v] = (int)l;
    // per-row Jacobian structure in position order, term -> entry index
    jr_ptr.assign(1, 0); jr_pos.clear(); t_jidx.assign(OMGX_TV * (size_t)std::max(1, t.n_terms), -1); row_leaf.assign(m + 1, -1);
    for (int r = 0; r <= m; ++r) {
      std::vector<int> vs = rows_vars[r];
      std::sort(vs.begin(), vs.end(), [&](int a, int b) { return pos[a] < pos[b]; });
      std::set<int> ls;
      for (int v : vs) if (leaf_of[v] >= 0) ls.insert(leaf_of[v]);
      if (r < m && ls.size() > 1) return fail("a constraint row couples two leaves");
      if (r < m && eq_index[r] >= 0 && !ls.empty()) return fail("an equality row touches a leaf variable");
      row_leaf[r] = (r < m && !ls.empty()) ? *ls.begin() : -1;
      const int base = (int)jr_pos.size();
      for (int v : vs) jr_pos.push_back(pos[v]);
      for (int tt = t.row_ptr[r]; tt < t.row_ptr[r + 1]; ++tt)
        for (int k = 0; k < OMGX_TV; ++k) {
          const int v = t.t_var[OMGX_TV * tt + k];
          if (v >= 0) t_jidx[OMGX_TV * tt + k] = base + (int)(std::find(vs.begin(), vs.end(), v) - vs.begin());
        }
      jr_ptr.push_back((int)jr_pos.size());
    }
    for (auto& vs : obj_cpl) { std::set<int> ls; for (int v : vs) if (leaf_of[v] >= 0) ls.insert(leaf_of[v]); if (ls.size() > 1) return fail("the objective couples two leaves"); }
    d.nnz_j = (int)jr_pos.size();
    // root positions coupled to each leaf (rows of the B_l blocks); t always
    cpl_ptr.assign(1, 0); cpl_idx.clear();
    cpl_map.assign((size_t)std::max(1, d.n_leaf) * d.n_root, -1);
    for (int l = 0; l < d.n_leaf; ++l) {
      std::set<int> s;
      for (int r = 0; r < m; ++r) if (row_leaf[r] == l) for (int v : rows_vars[r]) if (leaf_of[v] < 0) s.insert(pos[v] - d.root_off);
      for (auto& vs : obj_cpl) { bool mine = false; for (int v : vs) if (leaf_of[v] == l) mine = true; if (mine) for (int v : vs) if (leaf_of[v] < 0) s.insert(pos[v] - d.root_off); }
      s.insert(d.n_root - 1);
      int k = 0;
      for (int i : s) { cpl_map[(size_t)l * d.n_root + i] = k++; cpl_idx.push_back(i); }
      cpl_ptr.push_back((int)cpl_idx.size());
    }
    // which root positions every leaf variable meets in a row (or a nonlinear objective term): the entries of B_l that
    // can be non-zero (the phase-I variable t apart: it meets every variable)
    var_cpl.assign(d.n_leaf, {});
    for (int l = 0; l < d.n_leaf; ++l) var_cpl[l].assign(leaf_off[l + 1] - leaf_off[l], {});
    auto meet = [&](const std::vector<int>& vs) {
      for (int a : vs) { if (leaf_of[a] < 0) continue;
        std::vector<int>& dst = var_cpl[leaf_of[a]][pos[a] - leaf_off[leaf_of[a]]];
        for (int b : vs) if (leaf_of[b] < 0 && std::find(dst.begin(), dst.end(), pos[b] - d.root_off) == dst.end()) dst.push_back(pos[b] - d.root_off); }
    };
    for (int r = 0; r < m; ++r) meet(rows_vars[r]);
    for (auto& vs : obj_cpl) meet(vs);
    for (auto& lv : var_cpl) for (auto& v : lv) std::sort(v.begin(), v.end());
    return true;
  }

  // LPT assignment of weighted items to OMGX_NBIN bins (heaviest first, to the lightest bin)
  int owners = OMGX_NBIN;      // owner bins the assembly records are dealt to (= threads of the solve workgroup; set before build)
  std::vector<int> lpt_bins(const std::vector<int>& weight) const {
    std::vector<int> idx(weight.size()), bin(weight.size(), 0);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return weight[a] > weight[b]; });
    typedef std::pair<long, int> LB;     // (load, bin)
    std::priority_queue<LB, std::vector<LB>, std::greater<LB>> heap;
    for (int b = 0; b < owners; ++b) heap.push(LB(0, b));
    for (int i : idx) { LB lb = heap.top(); heap.pop(); bin[i] = lb.second; lb.first += weight[i]; heap.push(lb); }
    return bin;
  }

  // address of entry (p, q), p >= q (positions), inside the KKT store; -1: the structure has no such entry
  int32_t kkt_addr(int p, int q) const {
  const Dims& d = dims;
  auto tri = [](int i, int k) { return i * (i + 1) / 2 + k; };
    const int ro = d.root_off;
    if (compact && q < ro) {
      const int l = blk[q], o = leaf_off[l], nl = leaf_off[l + 1] - o;
      if (p < ro) {
        if (blk[p] != l) return -1;
        if (lf_kind[l]) return p == q ? d_off[l] + (p - o) : -1;
        return (p - q) <= lf_band[l] ? d_off[l] + (p - o) * lf_ldb[l] + (q - p + lf_band[l]) : -1;
      }
      const int arow = cpl_map[(size_t)l * d.n_root + (p - ro)];
      if (arow < 0) return -1;
      if (!lf_kind[l]) return lf_w[l] + arow * b_off[l] + (q - o);
      if (p - ro == d.n_root - 1) return d_off[l] + (1 + lf_w[l]) * nl + (q - o);       // coupling with t
      for (int s2 = 0; s2 < lf_w[l]; ++s2) if (dl_pos[4 * (size_t)o + s2 * nl + (q - o)] == p - ro) return d_off[l] + (1 + s2) * nl + (q - o);
      return -1;
    }
    if (p < ro) { const int l = blk[p], o = leaf_off[l]; if (blk[q] != l) return -1;
                  return col_major ? d_off[l] + (q - o) * b_off[l] + (p - o) : d_off[l] + (p - o) * b_off[l] + (q - o); }
    if (q < ro) {
      const int l = blk[q], n = leaf_off[l + 1] - leaf_off[l];
      const int arow = cpl_map[(size_t)l * d.n_root + (p - ro)];
      if (arow < 0) return -1;
      return col_major ? d_off[l] + (q - leaf_off[l]) * b_off[l] + (n + arow) : d_off[l] + (n + arow) * b_off[l] + (q - leaf_off[l]);
    }
    return d_off[d.n_leaf] + tri(p - ro, q - ro);
  }

  // col_major: leaf panels column by column (the spill modes: one thread per panel row then reads consecutive
  // addresses of the HBM slab; the LDS modes keep rows with an odd leading dimension, conflict-free there)
  bool col_major = false;
  // compact: the store of the register-resident wave path (device, LDS): leaf blocks by their band, the carried rows
  // dense behind it, a diagonal leaf whose variables meet the root through a few entries each (the terminal slacks)
  // as plain arrays -- config 2: 80 KB -> 42 KB, which is what lets two agents share a CU
  bool compact = false;
  // (developer print, OMGX_PLAN_DEBUG: how full an ELL table is -- owners, real records, records the groups of 64 walk)
  static void ell_stats(const char* name, int owners, long real, const std::vector<int32_t>& glen) {
    if (!getenv("OMGX_PLAN_DEBUG")) return;
    long walked = 0; std::string g;
    for (int i = 0; i < (owners + 63) / 64; ++i) { walked += 64L * glen[i]; g += std::to_string(glen[i]) + " "; }
    fprintf(stderr, "[plan] %-8s owners %5d real %6ld walked %6ld (%.0f %%) group lengths %s\n", name, owners, real, walked, walked ? 100.0 * real / walked : 0.0, g.c_str());
  }
  bool build(const omgx_template& t, bool col_major_panels = false, bool compact_store = false) {
    col_major = col_major_panels; compact = compact_store;
    Dims& d = dims;
    d = Dims();
    d.n_var = t.n_var; d.n_par = t.n_par; d.n_con = t.n_con; d.n_atoms = t.n_atoms;
    d.n_slots = t.n_slots; d.n_terms = t.n_terms; d.n_prog = t.n_prog;
    if (t.n_slots > 32767) return fail("more than 32767 parameter slots (16-bit slot indices in the item records)");
    d.N = t.n_var + 1; d.n_eq = t.n_eq; d.n_knots = t.n_knots;
    d.atoms_alias = (getenv("OMGX_NO_ATOMS_ALIAS") == nullptr && d.n_atoms + d.n_knots <= 2 * d.N + d.n_eq) ? 1 : 0;
    if (t.n_var <= 0 || t.n_con < 0 || t.n_terms < 0 || t.n_var >= 32767) return fail("bad dimensions");
    if (t.row_ptr[t.n_con + 1] != t.n_terms) return fail("row_ptr does not cover the terms");
    for (int k = 0; k < d.n_prog; ++k) {
      if (t.prog[6 * k] == OP_BSPL && (t.prog[6 * k + 3] < 0 || t.prog[6 * k + 3] > 32)) return fail("basis degree > 32");
      if (t.prog[6 * k] < OP_DIV || t.prog[6 * k] > OP_SIN) return fail("unknown op in the derived-atom program");
    }
    if (!build_structure(t)) return false;
    d.nr = d.n_root + d.n_eq;
    blk.assign(d.N, -1);
    d.max_leaf = 0; d.max_cpl = 0;
    d_off.assign(d.n_leaf + 1, 0); b_off.assign(d.n_leaf > 0 ? d.n_leaf : 1, 0);
    lf_w.assign(b_off.size(), 0); lf_ldb.assign(b_off.size(), 0); lf_band.assign(b_off.size(), 0); lf_kind.assign(b_off.size(), 0);
    dl_pos.assign(4 * (size_t)std::max(1, d.root_off), -1);
    int off = 0;
    for (int l = 0; l < d.n_leaf; ++l) {
      const int n = leaf_off[l + 1] - leaf_off[l], nc = cpl_ptr[l + 1] - cpl_ptr[l];
      for (int q = leaf_off[l]; q < leaf_off[l + 1]; ++q) blk[q] = l;
      if (n > d.max_leaf) d.max_leaf = n;
      if (nc > d.max_cpl) d.max_cpl = nc;
      const int ld = n | 1;                        // odd leading dimension: conflict-free row-per-lane access
      if (compact) {
        int S = 0; bool sparse = leaf_bw[l] == 0;
        if (sparse) {
          std::vector<int> seen(d.n_root, 0);
          for (auto& v : var_cpl[l]) { S = std::max(S, (int)v.size()); for (int q : v) if (seen[q]++) sparse = false; }     // a root position met by two variables: shared targets
          if (S > 4 || n > OMGX_WAVE_ROWS) sparse = false;
          if (getenv("OMGX_NO_SPARSE")) sparse = false;      // (developer knob)
        }
        lf_kind[l] = sparse ? 1 : 0;
        d_off[l] = off;
        if (sparse) {
          // arrays of n: diagonal | S coupling slots | coupling with t | right-hand side
          b_off[l] = n; lf_w[l] = S; lf_ldb[l] = 0; lf_band[l] = 0;
          for (int j = 0; j < n; ++j) for (size_t s2 = 0; s2 < var_cpl[l][j].size(); ++s2) dl_pos[4 * (size_t)leaf_off[l] + s2 * n + j] = var_cpl[l][j][s2];
          off += n * (S + 3);
        } else {
          const int band = leaf_bw[l] <= 8 ? leaf_bw[l] : n - 1;      // (wider than the compile-time band of the wave routine: dense)
          const int ldb = (band + 1) | 1;
          lf_band[l] = band; lf_ldb[l] = ldb; b_off[l] = ld;
          off += n * ldb;
          lf_w[l] = off;
          off += (nc + 1) * ld;
        }
        continue;
      }
      // rows [0,n): D_l, rows [n,n+nc): coupling rows B_l, row n+nc: the right-hand side of the leaf
      // (carried through the factorisation like a coupling row: it comes out as L^{-1} r)
      if (col_major) { const int ldc = (n + nc + 2) & ~1; d_off[l] = off; b_off[l] = ldc; off += n * ldc; }      // (even: columns start 16-byte aligned)
      else { d_off[l] = off; b_off[l] = ld; off += (n + nc + 1) * ld; }
    }
    d_off[d.n_leaf] = off; off += (d.nr + 1) * (d.nr + 2) / 2;      // packed root + its right-hand-side row
    {
      int leaf_rows = 0;
      for (int l = 0; l < d.n_leaf; ++l) leaf_rows += (leaf_off[l + 1] - leaf_off[l]) + (cpl_ptr[l + 1] - cpl_ptr[l]) + 1;
      const int pr = leaf_rows > d.nr + 1 ? leaf_rows : d.nr + 1;
      d.col_doubles = (OMGX_BMAT_DOUBLES + OMGX_STAGE_LD) * (OMGX_MAX_LEAF + 1) + OMGX_PAN_LD * pr;
      const int wave_scratch = OMGX_BMAT_DOUBLES * (OMGX_MAX_LEAF + 1) + 8 * 64;      // kkt_solve_wave: 64 doubles per wave
      if (d.col_doubles < wave_scratch) d.col_doubles = wave_scratch;
      if (compact) d.col_doubles = wave_scratch;      // (the wave path needs no panel buffers: descriptors + the gathered root solutions)
      int small = 0;
      for (int l = 0; l < d.n_leaf; ++l) small += OMGX_PAN_SMALL(leaf_off[l + 1] - leaf_off[l]);
      // the root's panel buffer starts at the same offset, followed by the LDS copy of the root block (Work::root)
      { const int rootpan = OMGX_PAN_LD * (d.nr + 1); d.col_small_noroot = (OMGX_BMAT_DOUBLES + OMGX_STAGE_LD) * (OMGX_MAX_LEAF + 1) + (small < rootpan ? rootpan : small); }
      { const int rootpart = OMGX_PAN_LD * (d.nr + 1) + (int)root_doubles(d); if (small < rootpart) small = rootpart; }
      d.col_small = (OMGX_BMAT_DOUBLES + OMGX_STAGE_LD) * (OMGX_MAX_LEAF + 1) + small;
    }
    kkt_doubles = off;                           // (side and dump slots of the assembly are appended below)
    // the wave-level register routines apply when every panel fits one wave
    d.wave_ok = d.nr <= OMGX_WAVE_COLS ? 1 : 0;
    for (int l = 0; l < d.n_leaf; ++l) {
      const int n = leaf_off[l + 1] - leaf_off[l], nc = cpl_ptr[l + 1] - cpl_ptr[l];
      if (n > OMGX_WAVE_COLS || n + nc - 1 > OMGX_WAVE_ROWS || nc + 1 > 32) d.wave_ok = 0;
    }
    // packed parameter monomials
    {
      int maxq = 0;
      for (int mm = 0; mm < t.n_mono; ++mm) maxq = std::max(maxq, t.pm_ptr[mm + 1] - t.pm_ptr[mm]);
      d.mono_packed = (d.n_atoms >= 32768 || maxq > 8) ? 0 : (maxq > 4 ? 2 : 1);
    }
    pm_rec.assign(t.n_mono > 0 ? t.n_mono : 1, MonoRec{0.0, -1, -1, -1, -1});
    pm_rec8.assign(d.mono_packed == 2 ? std::max(1, t.n_mono) : 1, MonoRec8{0.0, -1, -1, -1, -1, -1, -1, -1, -1});
    for (int mm = 0; mm < t.n_mono && d.mono_packed; ++mm) {
      const int q0 = t.pm_ptr[mm], nq = t.pm_ptr[mm + 1] - q0;
      int16_t a[8];
      for (int k = 0; k < 8; ++k) a[k] = k < nq ? (int16_t)t.pm_atom[q0 + k] : (int16_t)-1;
      if (d.mono_packed == 1) pm_rec[mm] = MonoRec{t.pm_coef[mm], a[0], a[1], a[2], a[3]};
      else pm_rec8[mm] = MonoRec8{t.pm_coef[mm], a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]};
    }
    Tables& T = tables;
    T = Tables();
    T.prog = t.prog; T.knots = t.knots; T.pp_ptr = t.pp_ptr; T.pm_coef = t.pm_coef;
    T.pm_ptr = t.pm_ptr; T.pm_atom = t.pm_atom; T.slot_pp = t.slot_pp; T.row_ptr = t.row_ptr;
    T.t_coef = t.t_coef; T.t_slot = t.t_slot; T.t_var = t.t_var; T.order = order.data();
    T.leaf_off = leaf_off.data(); T.leaf_bw = leaf_bw.data(); T.blk = blk.data(); T.eq_rows = eq_rows.data();
    T.eq_index = eq_index.data();
    T.cpl_ptr = cpl_ptr.data(); T.cpl_idx = cpl_idx.data(); T.cpl_map = cpl_map.data();
    T.d_off = d_off.data(); T.b_off = b_off.data();
    T.lf_w = lf_w.data(); T.lf_ldb = lf_ldb.data(); T.lf_band = lf_band.data(); T.lf_kind = lf_kind.data(); T.dl_pos = dl_pos.data();
    // ---- precomputed addresses -------------------------------------------------------
    auto tri = [](int i, int k) { return i * (i + 1) / 2 + k; };
    auto addr = [&](int p, int q) -> int32_t { return kkt_addr(p, q); };
    const int m = d.n_con, n = d.n_var;
    je_row.assign(d.nnz_j > 0 ? d.nnz_j : 1, 0);
    jt_addr.assign(d.nnz_j > 0 ? d.nnz_j : 1, 0);
    for (int r = 0; r <= m; ++r)
      for (int e = jr_ptr[r]; e < jr_ptr[r + 1]; ++e) {
        je_row[e] = r;
        jt_addr[e] = (r < m && eq_index[r] < 0) ? addr(d.N - 1, jr_pos[e]) : 0;
        if (jt_addr[e] < 0) return fail("no slot for a phase-I column entry");
      }
    tq_addr.assign(n, 0);
    for (int q = 0; q < n; ++q) { tq_addr[q] = addr(d.N - 1, q); if (tq_addr[q] < 0) return fail("no slot for (t, q)"); }
    pair4.clear();
    for (int r = 0; r < m; ++r) {
      if (eq_index[r] >= 0) continue;
      for (int a = jr_ptr[r]; a < jr_ptr[r + 1]; ++a)
        for (int b2 = jr_ptr[r]; b2 <= a; ++b2) {
          const int32_t ad = addr(jr_pos[a], jr_pos[b2]);
          if (ad < 0) return fail("no slot for a Jacobian pair");
          pair4.push_back(a); pair4.push_back(b2); pair4.push_back(ad); pair4.push_back(r);
        }
    }
    eqe3.clear();
    for (int k = 0; k < d.n_eq; ++k) {
      const int r = eq_rows[k];
      for (int a = jr_ptr[r]; a < jr_ptr[r + 1]; ++a) {
        eqe3.push_back(a); eqe3.push_back(d_off[d.n_leaf] + tri(d.n_root + k, jr_pos[a] - d.root_off)); eqe3.push_back(r);
      }
    }
    d.n_eqe = (int)eqe3.size() / 3;
    if (eqe3.empty()) eqe3.assign(3, 0);
    d.n_pairs = (int)pair4.size() / 4;
    diag_addr.assign(d.N, 0);
    for (int q = 0; q < d.N; ++q) diag_addr[q] = addr(q, q);
    // terms with >= 2 factors: every pair of factors (i < j, OMGX_TV (OMGX_TV - 1) / 2 of them) owns a Hessian entry
    t_row.assign(d.n_terms > 0 ? d.n_terms : 1, 0);
    hrec.clear();
    for (int r = 0; r <= m; ++r)
      for (int tt = t.row_ptr[r]; tt < t.row_ptr[r + 1]; ++tt) {
        t_row[tt] = r;
        const int32_t* tv = t.t_var + OMGX_TV * tt;
        int nv = 0; while (nv < OMGX_TV && tv[nv] >= 0) ++nv;
        for (int k = nv; k < OMGX_TV; ++k) if (tv[k] >= 0) return fail("unused term variables must come last");
        if (nv < 2) continue;
        HessTerm h; h.coef = t.t_coef[tt]; h.slot = t.t_slot[tt]; h.row = r; h.nv = nv;
        for (int k = 0; k < OMGX_TV; ++k) { h.v[k] = tv[k]; h.p[k] = tv[k] >= 0 ? pos[tv[k]] : -1; }
        int q = 0;
        for (int i = 0; i < OMGX_TV; ++i) for (int j = i + 1; j < OMGX_TV; ++j, ++q) {
          h.ha[q] = -1;
          if (j >= nv) continue;
          const int pa = h.p[i], pb = h.p[j];
          const int32_t ad = pa >= pb ? addr(pa, pb) : addr(pb, pa);
          if (ad < 0) return fail("no slot for a Hessian entry");
          h.ha[q] = ad;
        }
        hrec.push_back(h);
      }
    d.n_hess = (int)hrec.size();
    d.quartic = 0;
    for (int tt = 0; tt < d.n_terms; ++tt) if (t.t_var[OMGX_TV * tt + OMGX_TV - 1] >= 0) d.quartic = 1;
    d.general = d.quartic || t.n_lift > 0;      // (lifted auxiliaries: the general instance carries their projection)
    for (int k = 0; k < d.n_prog; ++k)
      if (t.prog[6 * k] == OP_COS || t.prog[6 * k] == OP_SIN || (t.prog[6 * k] == OP_BSPL && t.prog[6 * k + 3] > 5)) d.general = 1;
    // flat tables of the parameter stage
    slot_rng.assign(2 * (size_t)(d.n_slots > 0 ? d.n_slots : 1), 0);
    for (int sl = 0; sl < d.n_slots; ++sl) { slot_rng[2 * sl] = t.pp_ptr[t.slot_pp[sl]]; slot_rng[2 * sl + 1] = t.pp_ptr[t.slot_pp[sl] + 1]; }
    T.slot_rng = slot_rng.data();
    {
      // monomials of the slots in ELL form (setup pass), slots by decreasing monomial count
      const int no = d.n_slots;
      sl_list.resize(std::max(1, no));
      std::iota(sl_list.begin(), sl_list.end(), 0);
      auto cnt = [&](int sl) { return slot_rng[2 * sl + 1] - slot_rng[2 * sl]; };
      std::stable_sort(sl_list.begin(), sl_list.begin() + no, [&](int x, int y) { return cnt(x) > cnt(y); });
      sl_glen.assign((no + 63) / 64 + 1, 0);
      int steps = 0;
      // (the table holds the first OMGX_SLOT_CAP monomials of a slot; the rest of a longer slot -- the constant term of
      // an ADMM objective has hundreds -- is summed by a whole wave, eval_params: the long slots are the first n_long
      // entries of sl_list)
      d.n_long = 0;
      for (int i = 0; i < no; ++i) {
        const int c_ = std::min(cnt(sl_list[i]), OMGX_SLOT_CAP);
        if (cnt(sl_list[i]) > OMGX_SLOT_CAP) ++d.n_long;
        const int len = (c_ + 3) / 4 * 4; sl_glen[i >> 6] = std::max(sl_glen[i >> 6], len); steps = std::max(steps, len);
      }
      const size_t ell_n = (size_t)std::max(1, steps) * std::max(1, no);
      sl_ell.assign(d.mono_packed == 1 ? ell_n : 1, MonoRec{0.0, -1, -1, -1, -1});
      sl_ell8.assign(d.mono_packed == 2 ? ell_n : 1, MonoRec8{0.0, -1, -1, -1, -1, -1, -1, -1, -1});
      if (d.mono_packed)
        for (int i = 0; i < no; ++i) {
          const int k0 = slot_rng[2 * sl_list[i]], k1 = std::min(slot_rng[2 * sl_list[i] + 1], k0 + OMGX_SLOT_CAP);
          for (int k = k0; k < k1; ++k) {
            if (d.mono_packed == 1) sl_ell[(size_t)(k - k0) * no + i] = pm_rec[k];
            else sl_ell8[(size_t)(k - k0) * no + i] = pm_rec8[k];
          }
        }
      T.sl_list = sl_list.data(); T.sl_glen = sl_glen.data();
      T.sl_ell = d.mono_packed == 2 ? (const MonoRec*)sl_ell8.data() : sl_ell.data();
    }
    // packed (row, position) of the Jacobian entries
    d.rp_packed = (m < 65535 && d.N < 65536) ? 1 : 0;
    je_rp.assign(d.nnz_j > 0 ? d.nnz_j : 1, 0);
    if (d.rp_packed)
      for (int e = 0; e < d.nnz_j; ++e) je_rp[e] = (int32_t)(((uint32_t)je_row[e] << 16) | (uint32_t)jr_pos[e]);
    T.je_rp = je_rp.data();
    T.eqe3 = eqe3.data();
    T.je_row = je_row.data(); T.diag_addr = diag_addr.data(); T.tq_addr = tq_addr.data();
    T.pm_rec = dims.mono_packed == 2 ? (const MonoRec*)pm_rec8.data() : pm_rec.data();
    reg_w.assign(d.N, OMGX_DW_LINEAR);
    for (int tt = 0; tt < t.row_ptr[m + 1]; ++tt) {
      const int32_t* tv = t.t_var + OMGX_TV * tt;
      if (tv[1] < 0) continue;                                  // constant or linear term
      // class markers, turned into weights by the kernel: -1 nonlinear leaf variable, +1 nonlinear root variable
      for (int k = 0; k < OMGX_TV; ++k) if (tv[k] >= 0) reg_w[pos[tv[k]]] = (pos[tv[k]] < d.root_off) ? -1.0 : 1.0;
    }
    T.reg_w = reg_w.data();

    // ---- owner-computes tables: every sum of the solve has one owner and a fixed order ---------------
    // (1) Jacobian entries: entry e = sum of its items coef * slot * x[va] * x[vb] (term order)
    {
      std::vector<std::vector<JItem>> items(d.nnz_j > 0 ? d.nnz_j : 1);
      for (int tt = 0; tt < d.n_terms; ++tt) {
        const int32_t* tv = t.t_var + OMGX_TV * tt;
        int nv = 0; while (nv < OMGX_TV && tv[nv] >= 0) ++nv;
        for (int k = 0; k < nv; ++k) {
          JItem it; it.coef = t.t_coef[tt]; it.slot = (int16_t)t.t_slot[tt]; it.va = -1; it.vb = -1; it.vc = -1;
          int o = 0;
          for (int k2 = 0; k2 < nv; ++k2) if (k2 != k) { (o == 0 ? it.va : (o == 1 ? it.vb : it.vc)) = (int16_t)tv[k2]; ++o; }
          items[t_jidx[OMGX_TV * tt + k]].push_back(it);
        }
      }
      je_ptr.assign(1, 0); je_item.clear(); jv_list.clear();
      for (int e = 0; e < d.nnz_j; ++e) {
        bool varying = false;
        for (auto& it : items[e]) { je_item.push_back(it); if (it.va >= 0) varying = true; }
        je_ptr.push_back((int)je_item.size());
        if (varying) jv_list.push_back(e);
      }
      d.n_jv = (int)jv_list.size();
      // item tables of jac_entries4: entries by decreasing item count, four in a row per owner (homogeneous groups of 64
      // owners), {entry, row} beside them
      auto table4 = [&](std::vector<int>& list, std::vector<JItem>& ell, std::vector<int32_t>& own, std::vector<int32_t>& glen, const char* name) {
        std::stable_sort(list.begin(), list.end(), [&](int x, int y) { return (je_ptr[x + 1] - je_ptr[x]) > (je_ptr[y + 1] - je_ptr[y]); });
        const int ne = (int)list.size(), no = (ne + OMGX_JE_OWN - 1) / OMGX_JE_OWN;
        glen.assign((no + 63) / 64 + 1, 0);
        int steps = 1; long real = 0;
        for (int i = 0; i < ne; ++i) {
          const int cnt = je_ptr[list[i] + 1] - je_ptr[list[i]], len = cnt <= 1 ? 1 : (cnt + 1) / 2 * 2, g = (i / OMGX_JE_OWN) >> 6;
          glen[g] = std::max(glen[g], len); steps = std::max(steps, len); real += cnt;
        }
        ell.assign((size_t)steps * OMGX_JE_OWN * std::max(1, no), JItem{0.0, -1, -1, -1, -1});
        own.assign((size_t)2 * OMGX_JE_OWN * std::max(1, no), -1);
        for (int i = 0; i < ne; ++i) {
          const int o = i / OMGX_JE_OWN, k = i % OMGX_JE_OWN, e = list[i];
          own[2 * ((size_t)k * no + o)] = e; own[2 * ((size_t)k * no + o) + 1] = je_row[e];
          for (int q = je_ptr[e]; q < je_ptr[e + 1]; ++q) ell[((size_t)(q - je_ptr[e]) * OMGX_JE_OWN + k) * no + o] = je_item[q];
        }
        if (getenv("OMGX_PLAN_DEBUG")) {
          long walked = 0; for (int g = 0; g < (no + 63) / 64; ++g) walked += 64L * OMGX_JE_OWN * glen[g];
          fprintf(stderr, "[plan] %-8s entries %5d owners %5d items %6ld walked %6ld\n", name, ne, no, real, walked);
        }
        return no;
      };
      d.n_jv4 = table4(jv_list, jv_ell, jv_own, jv_glen, "jv");
      {
        std::vector<int> all(d.nnz_j);
        std::iota(all.begin(), all.end(), 0);
        d.n_ja4 = table4(all, ja_ell, ja_own, ja_glen, "ja");
      }
      T.jv_ell = jv_ell.data(); T.jv_own = jv_own.data(); T.jv_glen = jv_glen.data();
      T.ja_ell = ja_ell.data(); T.ja_own = ja_own.data(); T.ja_glen = ja_glen.data();
      if (je_item.empty()) je_item.push_back(JItem{0.0, -1, -1, -1, -1});
    }
    // (2) rows by decreasing term count: the threads of the first pass get the long rows
    {
      row_perm.resize(m > 0 ? m : 1, 0);
      std::iota(row_perm.begin(), row_perm.end(), 0);
      std::stable_sort(row_perm.begin(), row_perm.begin() + m, [&](int a, int b) {
        return (t.row_ptr[a + 1] - t.row_ptr[a]) > (t.row_ptr[b + 1] - t.row_ptr[b]); });
      // The row passes walk the slots i = tid, tid + threads, ...: wave w of pass p owns the group of 64 slots p * waves + w and
      // walks its longest row.  The sorted groups are dealt to the waves so that no wave collects all the long ones (config 2,
      // four waves: term steps 16 16 8 8 | 8 8 8 8 | 8 gave wave 0 four batches of eight and wave 3 two; dealt, three at most):
      // longest group first, to the wave with the least work that still has a free position; the last, partial group keeps
      // the last position.
      {
        const int G = (m + 63) / 64, nw = std::max(1, owners / 64);
        if (G > nw) {
          auto steps8 = [](int k) { return (k + 7) / 8 * 8; };
          std::vector<int> wgt(G, 0);
          for (int i = 0; i < m; ++i) { const int r = row_perm[i]; wgt[i >> 6] = std::max(wgt[i >> 6], steps8(t.row_ptr[r + 1] - t.row_ptr[r]) + steps8(jr_ptr[r + 1] - jr_ptr[r])); }
          std::vector<int> room(nw, 0), load(nw, 0);
          for (int gp = 0; gp < G; ++gp) ++room[gp % nw];
          std::vector<std::vector<int>> mine(nw);
          const bool partial = (m % 64) != 0;
          const int w_last = (G - 1) % nw;
          if (partial) { --room[w_last]; load[w_last] += wgt[G - 1]; }
          std::vector<int> gs(partial ? G - 1 : G);
          std::iota(gs.begin(), gs.end(), 0);
          std::stable_sort(gs.begin(), gs.end(), [&](int a, int b) { return wgt[a] > wgt[b]; });
          for (int g : gs) {
            int best = -1;
            for (int w2 = 0; w2 < nw; ++w2) if (room[w2] > 0 && (best < 0 || load[w2] < load[best])) best = w2;
            mine[best].push_back(g); --room[best]; load[best] += wgt[g];
          }
          if (partial) mine[w_last].push_back(G - 1);
          std::vector<int32_t> perm2(row_perm.size(), 0);
          for (int gp = 0; gp < G; ++gp) {
            const int g = mine[gp % nw][gp / nw];
            for (int k = 0; k < 64 && 64 * g + k < m; ++k) perm2[64 * gp + k] = row_perm[64 * g + k];
          }
          row_perm = perm2;
        }
      }
      T.row_perm = row_perm.data();
      // terms of every row (slot i = row row_perm[i]) in ELL form, eight per step
      {
        rt_glen.assign((m + 63) / 64 + 1, 0);
        int steps = 0;
        for (int i = 0; i < m; ++i) { const int r = row_perm[i]; const int len = (t.row_ptr[r + 1] - t.row_ptr[r] + 7) / 8 * 8; rt_glen[i >> 6] = std::max(rt_glen[i >> 6], len); steps = std::max(steps, len); }
        rt_ell.assign((size_t)std::max(1, steps) * std::max(1, m), RowTerm{0.0, -1, -1, -1, -1, -1});
        for (int i = 0; i < m; ++i) {
          const int r = row_perm[i];
          for (int tt = t.row_ptr[r]; tt < t.row_ptr[r + 1]; ++tt) {
            RowTerm q; q.coef = t.t_coef[tt]; q.slot = t.t_slot[tt];
            q.v0 = (int16_t)t.t_var[OMGX_TV * tt]; q.v1 = (int16_t)t.t_var[OMGX_TV * tt + 1]; q.v2 = (int16_t)t.t_var[OMGX_TV * tt + 2];
            q.v3 = (int16_t)t.t_var[OMGX_TV * tt + 3];
            rt_ell[(size_t)(tt - t.row_ptr[r]) * m + i] = q;
          }
        }
        T.rt_ell = rt_ell.data(); T.rt_glen = rt_glen.data();
        ell_stats("rt", m, (long)t.row_ptr[m], rt_glen);
        // {Jacobian entry, position} of every row for J dx; padding points at the zero slot jval[nnz_j]
        jp_glen.assign((m + 63) / 64 + 1, 0);
        steps = 0;
        for (int i = 0; i < m; ++i) { const int r = row_perm[i]; const int len = (jr_ptr[r + 1] - jr_ptr[r] + 7) / 8 * 8; jp_glen[i >> 6] = std::max(jp_glen[i >> 6], len); steps = std::max(steps, len); }
        jp_ell.assign((size_t)2 * std::max(1, steps) * std::max(1, m), 0);
        for (int i = 0; i < m; ++i) {
          const int r = row_perm[i];
          for (int st = 0; st < steps; ++st) {
            int32_t* q = jp_ell.data() + 2 * ((size_t)st * m + i);
            const int e = jr_ptr[r] + st;
            if (e < jr_ptr[r + 1]) { q[0] = e; q[1] = jr_pos[e]; } else { q[0] = d.nnz_j; q[1] = 0; }
          }
        }
        T.jp_ell = jp_ell.data(); T.jp_glen = jp_glen.data();
        ell_stats("jp", m, (long)jr_ptr[m], jp_glen);
      }
      // lifted auxiliaries: the defining rows level by level (a row of level L reads auxiliaries of lower levels only)
      {
        d.n_lift = t.n_lift > 0 ? t.n_lift : 0;
        d.lift_depth = 0;
        lift_rec.assign(2 * (size_t)std::max(1, d.n_lift), 0);
        lift_lev.assign(2, 0);
        if (d.n_lift) {
          const int v0 = d.n_var - d.n_lift, r0 = t.lift_row0;
          if (v0 < 0 || r0 < 0 || r0 + d.n_lift > m) return fail("lifted rows outside the template");
          std::vector<int> slot_of(m, 0), lev(d.n_lift, 0);
          for (int i = 0; i < m; ++i) slot_of[row_perm[i]] = i;
          for (int k = 0; k < d.n_lift; ++k) {
            if (eq_index[r0 + k] < 0) return fail("the defining row of a lifted auxiliary is not an equality row");
            int own = 0, lv = 0;
            for (int tt = t.row_ptr[r0 + k]; tt < t.row_ptr[r0 + k + 1]; ++tt) {
              int mine = 0;
              for (int q = 0; q < OMGX_TV; ++q) {
                const int v = t.t_var[OMGX_TV * tt + q];
                if (v < v0) continue;
                if (v == v0 + k) ++mine;
                else if (v > v0 + k) return fail("a lifted row reads an auxiliary defined after it");
                else lv = std::max(lv, lev[v - v0] + 1);
              }
              if (mine > 1) return fail("a lifted row is not linear in its auxiliary");
              own += mine;
            }
            if (!own) return fail("a lifted row does not contain its auxiliary");
            lev[k] = lv;
            d.lift_depth = std::max(d.lift_depth, lv + 1);
          }
          lift_lev.assign(d.lift_depth + 1, 0);
          int out = 0;
          for (int L = 0; L < d.lift_depth; ++L) {
            lift_lev[L] = out;
            for (int k = 0; k < d.n_lift; ++k) if (lev[k] == L) { lift_rec[2 * out] = slot_of[r0 + k]; lift_rec[2 * out + 1] = v0 + k; ++out; }
          }
          lift_lev[d.lift_depth] = out;
        }
        T.lift_rec = lift_rec.data(); T.lift_lev = lift_lev.data();
      }
    }
    // (3) column sums J'w: entries of every column (position) in row order, objective entry apart
    {
      std::vector<std::vector<std::pair<int, int>>> cols(n);
      obj_ent.assign(n, -1);
      for (int r = 0; r <= m; ++r)
        for (int e = jr_ptr[r]; e < jr_ptr[r + 1]; ++e) {
          const int q = jr_pos[e];
          if (q >= n) return fail("Jacobian entry in the phase-I column");
          if (r == m) obj_ent[q] = e; else cols[q].push_back(std::make_pair(e, r));
        }
      cs_ptr.assign(1, 0); cs_rec.clear();
      for (int q = 0; q < n; ++q) { for (auto& er : cols[q]) { cs_rec.push_back(er.first); cs_rec.push_back(er.second); } cs_ptr.push_back((int)cs_rec.size() / 2); }
      if (cs_rec.empty()) cs_rec.assign(2, 0);
      T.cs_ptr = cs_ptr.data(); T.cs_rec = cs_rec.data(); T.obj_ent = obj_ent.data();
      // owners: the columns by decreasing length, each in chunks of at most `cap` records (a multiple of the batch of
      // eight the loop loads at a time) -- the smallest cap for which the owners fit the workgroup and their partial sums
      // the (idle) KKT store; records in ELL form
      {
        cs_col.resize(n);
        std::iota(cs_col.begin(), cs_col.end(), 0);
        std::stable_sort(cs_col.begin(), cs_col.end(), [&](int x, int y) { return (cs_ptr[x + 1] - cs_ptr[x]) > (cs_ptr[y + 1] - cs_ptr[y]); });
        auto owners_at = [&](int cp) { int k = 0; for (int q = 0; q < n; ++q) k += std::max(1, (cs_ptr[q + 1] - cs_ptr[q] + cp - 1) / cp); return k; };
        auto fits = [&](int k, int lim) { return k <= lim && (size_t)4 * k <= (size_t)kkt_doubles; };
        int cap = 8;
        while (cap < 32 && !fits(owners_at(cap), owners)) cap += 8;                    // one pass of the workgroup if chunks of <= 32 records allow it
        if (!fits(owners_at(cap), owners)) { cap = 8; while (cap < (1 << 20) && !fits(owners_at(cap), std::max(owners, 2 * n))) cap += 8; }
        const int no = owners_at(cap);
        d.n_cs_own = no;
        cs_own.assign(n + 1, 0);
        std::vector<std::vector<std::pair<int, int>>> own;
        for (int j = 0; j < n; ++j) {
          const int q = cs_col[j], len = cs_ptr[q + 1] - cs_ptr[q], np = std::max(1, (len + cap - 1) / cap);
          for (int k = 0; k < np; ++k) {
            own.push_back({});
            for (int i = cs_ptr[q] + k * len / np; i < cs_ptr[q] + (k + 1) * len / np; ++i) own.back().push_back(std::make_pair(cs_rec[2 * i], cs_rec[2 * i + 1]));
          }
          cs_own[j + 1] = (int)own.size();
        }
        cs_glen.assign((no + 63) / 64 + 1, 0);
        int steps = 0;
        for (int o = 0; o < no; ++o) { const int len = ((int)own[o].size() + 7) / 8 * 8; cs_glen[o >> 6] = std::max(cs_glen[o >> 6], len); steps = std::max(steps, len); }
        cs_ell.assign((size_t)2 * std::max(1, steps) * std::max(1, no), 0);
        for (int o = 0; o < no; ++o) for (int st = 0; st < steps; ++st) {
          int32_t* q = cs_ell.data() + 2 * ((size_t)st * no + o);
          if (st < (int)own[o].size()) { q[0] = own[o][st].first; q[1] = own[o][st].second; } else { q[0] = d.nnz_j; q[1] = 0; }
        }
        T.cs_own = cs_own.data();
        T.cs_ell = cs_ell.data(); T.cs_glen = cs_glen.data(); T.cs_col = cs_col.data();
        ell_stats("cs", no, (long)cs_rec.size() / 2, cs_glen);
      }
    }
    // (4) KKT assembly.  Every target (a KKT address; for the Gershgorin sums a position) has its records
    // summed in table order.  A long run (the diagonal entry of a trajectory coefficient collects 47 pairs,
    // the average owner has 21 records) is cut into segments of at most OMGX_RUN_CAP records; the first
    // segment goes to the target, the others to side slots behind the store, and a fix-up list says which
    // slots to add to which target, in order: still one fixed order per sum, but the longest owner loop is
    // ~24 records instead of 47.  Segments are dealt to the OMGX_NBIN owners by LPT; an owner's records are
    // stored "transposed" (ELL layout): record r of owner b sits at index r * OMGX_NBIN + b, so that thread b
    // = lane b reads next to its neighbours; owners shorter than the longest are padded with null records.
    // The target is stored only in the last record of a segment (-1 elsewhere): the owner adds every record
    // to a running sum and stores / restarts where it sees one.
    {
      struct Seg { int target; std::vector<int> items; };
      // items of one target in table order -> segments, side slots and fix-ups
      auto cut = [&](const std::vector<std::pair<int, std::vector<int>>>& runs, std::vector<Seg>& segs, std::vector<int32_t>& fix, int& n_side, int cap = OMGX_RUN_CAP) {
        for (auto& run : runs) {
          const std::vector<int>& it = run.second;
          const int nseg = ((int)it.size() + cap - 1) / cap;
          if (nseg > 1) { fix.push_back(run.first); fix.push_back(n_side); fix.push_back(nseg - 1); }
          for (int sg = 0; sg < nseg; ++sg) {
            Seg g; g.target = sg == 0 ? run.first : -2 - (n_side++);       // (-2 - k: side slot k)
            const int lo = sg * (int)it.size() / nseg, hi = (sg + 1) * (int)it.size() / nseg;
            g.items.assign(it.begin() + lo, it.begin() + hi);
            segs.push_back(g);
          }
        }
      };
      auto deal = [&](const std::vector<Seg>& segs, std::vector<std::vector<std::pair<int, int>>>& per) {   // per[bin] = (item, target or -1)
        std::vector<int> wgt(segs.size());
        for (size_t i = 0; i < segs.size(); ++i) wgt[i] = (int)segs[i].items.size();
        std::vector<int> bin = lpt_bins(wgt);
        per.assign(OMGX_NBIN, {});
        for (size_t i = 0; i < segs.size(); ++i)
          for (size_t k = 0; k < segs[i].items.size(); ++k)
            per[bin[i]].push_back(std::make_pair(segs[i].items[k], k + 1 == segs[i].items.size() ? segs[i].target : -1));
      };
      auto ell_len = [&](const std::vector<std::vector<std::pair<int, int>>>& per) {
        size_t len = 0; for (auto& v : per) len = std::max(len, v.size());
        return (int)((len + OMGX_REC_BATCH - 1) / OMGX_REC_BATCH * OMGX_REC_BATCH);      // (owner loops load OMGX_REC_BATCH records at a time)
      };
      const int side0 = kkt_doubles;                  // side slots start here; the store grows by them below
      int n_side = 0;
      // ---- pairs of J' Sigma J
      {
        std::vector<int> ord(d.n_pairs);
        std::iota(ord.begin(), ord.end(), 0);
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return pair4[4 * x + 2] < pair4[4 * y + 2]; });
        std::vector<std::pair<int, std::vector<int>>> runs;
        for (int e : ord) { const int ad = pair4[4 * e + 2]; if (runs.empty() || runs.back().first != ad) runs.push_back(std::make_pair(ad, std::vector<int>())); runs.back().second.push_back(e); }
        std::vector<Seg> segs; cut(runs, segs, ka_fix, n_side);
        std::vector<std::vector<std::pair<int, int>>> per; deal(segs, per);
        d.ka_len = ell_len(per);
        ka_rec.assign((size_t)4 * OMGX_NBIN * std::max(1, d.ka_len), 0);
        for (int bb = 0; bb < OMGX_NBIN; ++bb)
          for (int r = 0; r < d.ka_len; ++r) {
            int32_t* q = ka_rec.data() + 4 * ((size_t)r * OMGX_NBIN + bb);
            q[0] = 0; q[1] = 0; q[2] = -1; q[3] = 0;
            if (r < (int)per[bb].size()) {
              const int e = per[bb][r].first, tg = per[bb][r].second;
              q[0] = pair4[4 * e]; q[1] = pair4[4 * e + 1]; q[3] = pair4[4 * e + 3];
              q[2] = tg <= -2 ? side0 + (-2 - tg) : tg;
            }
          }
      }
      const int n_side_pairs = n_side;      // (side slots of the pairs: consumed by their fix-up before the Hessian / Gershgorin passes run)
      // ---- Hessian items: one per (nonlinear term, variable pair), added to the store after the pairs
      struct HI { HItem it; int pa, pb; };
      std::vector<HI> his;
      for (const HessTerm& h : hrec) {
        int q = 0;
        for (int i = 0; i < OMGX_TV; ++i) for (int j = i + 1; j < OMGX_TV; ++j, ++q) {
          if (j >= h.nv) continue;
          HI x; x.it.coef = h.coef; x.it.slot = (int16_t)h.slot; x.it.row = h.row; x.it.target = h.ha[q];
          // the factors that stay: the term without factors i and j
          x.it.vthird = -1; x.it.vfourth = -1;
          int o = 0;
          for (int k = 0; k < h.nv; ++k) if (k != i && k != j) { (o == 0 ? x.it.vthird : x.it.vfourth) = (int16_t)h.v[k]; ++o; }
          x.it.kind = (h.v[i] == h.v[j]) ? 1 : 0;
          x.pa = h.p[i]; x.pb = h.p[j];
          his.push_back(x);
        }
      }
      {
        std::vector<int> oh(his.size());
        std::iota(oh.begin(), oh.end(), 0);
        std::stable_sort(oh.begin(), oh.end(), [&](int x, int y) { return his[x].it.target < his[y].it.target; });
        std::vector<std::pair<int, std::vector<int>>> runs;
        for (int i : oh) { const int ad = his[i].it.target; if (runs.empty() || runs.back().first != ad) runs.push_back(std::make_pair(ad, std::vector<int>())); runs.back().second.push_back(i); }
        // (the benchmark classes: at most a handful of terms meet in one Hessian entry and nothing is cut.  Templates with
        // lifted auxiliaries: thousands do -- an auxiliary that stands for a product spline meets every row it was substituted
        // into --, one owner walked 14,520 records where the average has 1,500 (round 5): runs longer than OMGX_RUN_CAP_H are
        // cut like the pairs' -- the first segment goes to the entry, the others to side slots (zero before the pass: the
        // owners add to what they find), and kh_fix says which slots to add to which entry, in order)
        std::vector<Seg> segs; cut(runs, segs, kh_fix, n_side, OMGX_RUN_CAP_H);
        std::vector<std::vector<std::pair<int, int>>> per; deal(segs, per);
        d.kh_len = ell_len(per);
        kh_rec.assign((size_t)OMGX_NBIN * std::max(1, d.kh_len), HItem{0.0, 0, -1, -1, -1, -1, 0});
        for (int bb = 0; bb < OMGX_NBIN; ++bb) for (size_t r = 0; r < per[bb].size(); ++r) {
          HItem it = his[per[bb][r].first].it;
          const int tg = per[bb][r].second;
          it.target = tg <= -2 ? side0 + (-2 - tg) : tg;
          kh_rec[r * OMGX_NBIN + bb] = it;
        }
      }
      // ---- Gershgorin row sums: target = position (w.xt), side sums in w.dinv (free between the dual residual and the
      // factorisation)
      {
        struct GI { HItem it; int p; };
        std::vector<GI> gis;
        // Off-diagonal items contribute |lambda coef slot x3 x4| to the sums of both their positions: items of one row that
        // differ only in the coefficient (a bilinear row of two splines meets position q once per coefficient of the
        // other spline) are one record with the sum of the |coef| -- the same sum, a fraction of the records (config 2:
        // 16 -> 8 per owner).  Diagonal items (kind 1) count by sign and stay as they are.
        std::map<std::array<int, 5>, int> seen;
        for (auto& x : his) {
          for (int side = 0; side < (x.it.kind ? 1 : 2); ++side) {
            GI g; g.it = x.it; g.p = side ? x.pb : x.pa;
            if (!x.it.kind) {
              const std::array<int, 5> key = {g.p, x.it.row, x.it.slot, x.it.vthird, x.it.vfourth};
              auto f = seen.find(key);
              if (f != seen.end()) { gis[f->second].it.coef += std::fabs(x.it.coef); continue; }
              seen[key] = (int)gis.size();
              g.it.coef = std::fabs(x.it.coef);
            }
            gis.push_back(g);
          }
        }
        std::vector<int> og(gis.size());
        std::iota(og.begin(), og.end(), 0);
        std::stable_sort(og.begin(), og.end(), [&](int x, int y) { return gis[x].p < gis[y].p; });
        std::vector<std::pair<int, std::vector<int>>> runs;
        for (int i : og) { const int q = gis[i].p; if (runs.empty() || runs.back().first != q) runs.push_back(std::make_pair(q, std::vector<int>())); runs.back().second.push_back(i); }
        // (a hundred positions with a dozen records each leave most owners idle: the runs are cut shorter while that
        // shortens the longest owner and the side sums fit, never below four records)
        int n_side_g = 0;
        std::vector<Seg> segs; std::vector<std::vector<std::pair<int, int>>> per;
        for (int cap = 1 << 20; cap >= 4; cap = cap > 4 * OMGX_RUN_CAP ? 4 * OMGX_RUN_CAP : cap / 2) {      // uncut, 64, 32, 16, 8, 4
          int ns = 0; std::vector<Seg> sg; std::vector<int32_t> fx; std::vector<std::vector<std::pair<int, int>>> pr;
          cut(runs, sg, fx, ns, cap); deal(sg, pr);
          // (the side sums of this pass live in w.dinv [N] -- free between the dual residual and the factorisation -- or,
          // when they are more, in the side slots of the pairs, which the pair fix-up has consumed by then)
          if (ns > d.N && ns > n_side_pairs) break;      // (the slots behind them hold the cut runs of the Hessian items during that pass)
          if (per.empty() || ell_len(pr) < ell_len(per)) { segs = sg; kg_fix = fx; per = pr; n_side_g = ns; }
        }
        d.kg_side_dinv = n_side_g <= d.N ? 1 : 0;
        d.kg_len = ell_len(per);
        if (getenv("OMGX_PLAN_DEBUG")) { size_t mx = 0; for (auto& v : per) mx = std::max(mx, v.size()); fprintf(stderr, "[plan] hessian items %zu, gershgorin records %zu (longest owner %zu), runs %zu, side slots %d (pairs %d, with the Hessian's %d), owners %d\n", his.size(), gis.size(), mx, runs.size(), n_side_g, n_side_pairs, n_side, owners); }
        kg_rec.assign((size_t)OMGX_NBIN * std::max(1, d.kg_len), HItem{0.0, 0, -1, -1, -1, -1, 0});
        for (int bb = 0; bb < OMGX_NBIN; ++bb) for (size_t r = 0; r < per[bb].size(); ++r) {
          HItem it = gis[per[bb][r].first].it;
          const int tg = per[bb][r].second;
          it.target = tg <= -2 ? d.N + (-2 - tg) : tg;             // side slot k is encoded as N + k
          kg_rec[r * OMGX_NBIN + bb] = it;
        }
      }
      d.n_kafix = (int)ka_fix.size() / 3; d.n_kgfix = (int)kg_fix.size() / 3; d.n_khfix = (int)kh_fix.size() / 3;
      if (kh_fix.empty()) kh_fix.assign(3, 0);
      if (ka_fix.empty()) ka_fix.assign(3, 0);
      if (kg_fix.empty()) kg_fix.assign(3, 0);
      d.side_off = side0;
      d.n_owner = owners;
      kkt_doubles = side0 + n_side + 64;              // + side slots + one dump slot per lane (owner passes store there
                                                      //   whatever is not the end of a segment; distinct banks)
      d.dump_off = side0 + n_side;
      T.ka_rec = ka_rec.data(); T.kh_rec = kh_rec.data(); T.kg_rec = kg_rec.data();
      T.ka_fix = ka_fix.data(); T.kg_fix = kg_fix.data(); T.kh_fix = kh_fix.data();
    }
    if (pair4.empty()) pair4.assign(4, 0);
    T.pair4 = pair4.data();
    return true;
  }
};

}  // namespace omgx
