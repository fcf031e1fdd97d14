// omgx_plan.h -- turn an omgx_template (host pointers) into Dims + derived
// tables (position maps, KKT block offsets).  Host-side only; shared by the HIP
// library (which then uploads the tables) and by the CPU port.
#pragma once
#include <vector>
#include "../../include/omgx.h"
#include "omgx_core.h"

namespace omgx {

struct HostPlan {
  Dims dims;
  Tables tables;            // host pointers (into tpl and the vectors below)
  int kkt_doubles;
  std::vector<int32_t> pos, blk, eq_index, d_off, b_off;
  std::vector<int32_t> pair4, eqe3, je_row, jt_addr, diag_addr, h_addr, t_row, t_pos;
  std::vector<double> reg_w;
  std::vector<MonoRec> pm_rec;
  std::vector<int32_t> je_rp, slot_rng;
  std::vector<TermRec> trec;
  std::vector<HessRec> hrec;

  bool build(const omgx_template& t) {
    Dims& d = dims;
    d.n_var = t.n_var; d.n_par = t.n_par; d.n_con = t.n_con; d.n_atoms = t.n_atoms;
    d.n_slots = t.n_slots; d.n_terms = t.n_terms; d.n_prog = t.n_prog;
    d.N = t.n_var + 1; d.n_leaf = t.n_leaf; d.n_root = t.n_root; d.n_eq = t.n_eq;
    d.nnz_j = t.nnz_j; d.root_off = t.leaf_off[t.n_leaf]; d.nr = t.n_root + t.n_eq; d.n_knots = t.n_knots;
    if (d.root_off + d.n_root != d.N) return false;
    for (int k = 0; k < d.n_prog; ++k)                          // bspl_entry holds a triangle of degree <= 5
      if (t.prog[6 * k] == OP_BSPL && (t.prog[6 * k + 3] < 0 || t.prog[6 * k + 3] > 5)) return false;
    pos.assign(d.N, -1); blk.assign(d.N, -1);
    for (int q = 0; q < d.N; ++q) { if (t.order[q] < 0 || t.order[q] >= d.N) return false; pos[t.order[q]] = q; }
    if (t.order[d.N - 1] != t.n_var) return false;       // t must be the last position
    d.max_leaf = 0; d.max_cpl = 0;
    d_off.assign(d.n_leaf + 1, 0); b_off.assign(d.n_leaf > 0 ? d.n_leaf : 1, 0);
    int off = 0;
    for (int l = 0; l < d.n_leaf; ++l) {
      const int n = t.leaf_off[l + 1] - t.leaf_off[l], nc = t.cpl_ptr[l + 1] - t.cpl_ptr[l];
      for (int q = t.leaf_off[l]; q < t.leaf_off[l + 1]; ++q) blk[q] = l;
      if (n > d.max_leaf) d.max_leaf = n;
      if (nc > d.max_cpl) d.max_cpl = nc;
      const int ld = n | 1;                        // odd leading dimension: conflict-free row-per-lane access
      // rows [0,n): D_l, rows [n,n+nc): coupling rows B_l, row n+nc: the right-hand side of the leaf
      // (carried through the factorisation like a coupling row: it comes out as L^{-1} r)
      d_off[l] = off; b_off[l] = ld; off += (n + nc + 1) * ld;
    }
    d_off[d.n_leaf] = off; off += (d.nr + 1) * (d.nr + 2) / 2;      // packed root + its right-hand-side row
    {
      int leaf_rows = 0;
      for (int l = 0; l < d.n_leaf; ++l) leaf_rows += (t.leaf_off[l + 1] - t.leaf_off[l]) + (t.cpl_ptr[l + 1] - t.cpl_ptr[l]) + 1;
      const int pr = leaf_rows > d.nr + 1 ? leaf_rows : d.nr + 1;
      d.col_doubles = (OMGX_BMAT_DOUBLES + OMGX_STAGE_LD) * (OMGX_MAX_LEAF + 1) + OMGX_PAN_LD * pr;
      if (d.n_leaf > OMGX_MAX_LEAF) return false;
    }
    kkt_doubles = off;
    // packed parameter monomials
    d.mono_packed = (d.n_atoms < 32768) ? 1 : 0;
    pm_rec.assign(t.n_mono > 0 ? t.n_mono : 1, MonoRec{0.0, -1, -1, -1, -1});
    for (int mm = 0; mm < t.n_mono; ++mm) {
      const int q0 = t.pm_ptr[mm], nq = t.pm_ptr[mm + 1] - q0;
      if (nq > 4) { d.mono_packed = 0; break; }
      MonoRec& r = pm_rec[mm];
      r.coef = t.pm_coef[mm];
      if (nq > 0) r.a0 = (int16_t)t.pm_atom[q0];
      if (nq > 1) r.a1 = (int16_t)t.pm_atom[q0 + 1];
      if (nq > 2) r.a2 = (int16_t)t.pm_atom[q0 + 2];
      if (nq > 3) r.a3 = (int16_t)t.pm_atom[q0 + 3];
    }
    eq_index.assign(d.n_con, -1);
    for (int k = 0; k < d.n_eq; ++k) eq_index[t.eq_rows[k]] = k;
    Tables& T = tables;
    T.prog = t.prog; T.knots = t.knots; T.pp_ptr = t.pp_ptr; T.pm_coef = t.pm_coef;
    T.pm_ptr = t.pm_ptr; T.pm_atom = t.pm_atom; T.slot_pp = t.slot_pp; T.row_ptr = t.row_ptr;
    T.t_coef = t.t_coef; T.t_slot = t.t_slot; T.t_var = t.t_var; T.order = t.order;
    T.pos = pos.data(); T.leaf_off = t.leaf_off; T.blk = blk.data(); T.eq_rows = t.eq_rows;
    T.eq_index = eq_index.data(); T.jr_ptr = t.jr_ptr; T.jr_pos = t.jr_pos; T.t_jidx = t.t_jidx;
    T.row_leaf = t.row_leaf; T.jc_ptr = t.jc_ptr; T.jc_row = t.jc_row; T.jc_ent = t.jc_ent;
    T.cpl_ptr = t.cpl_ptr; T.cpl_idx = t.cpl_idx; T.cpl_map = t.cpl_map;
    T.d_off = d_off.data(); T.b_off = b_off.data();
    // ---- precomputed addresses -------------------------------------------------------
    auto tri = [](int i, int k) { return i * (i + 1) / 2 + k; };
    auto addr = [&](int p, int q) -> int32_t {               // p >= q, positions
      const int ro = d.root_off;
      if (p < ro) { const int l = blk[p], o = t.leaf_off[l]; return d_off[l] + (p - o) * b_off[l] + (q - o); }
      if (q < ro) {
        const int l = blk[q], n = t.leaf_off[l + 1] - t.leaf_off[l];
        const int arow = t.cpl_map[l * d.n_root + (p - ro)];
        if (arow < 0) return -1;
        return d_off[l] + (n + arow) * b_off[l] + (q - t.leaf_off[l]);
      }
      return d_off[d.n_leaf] + tri(p - ro, q - ro);
    };
    const int m = d.n_con;
    je_row.assign(d.nnz_j > 0 ? d.nnz_j : 1, 0);
    jt_addr.assign(d.nnz_j > 0 ? d.nnz_j : 1, 0);
    for (int r = 0; r <= m; ++r)
      for (int e = t.jr_ptr[r]; e < t.jr_ptr[r + 1]; ++e) {
        je_row[e] = r;
        jt_addr[e] = (r < m && eq_index[r] < 0) ? addr(d.N - 1, t.jr_pos[e]) : 0;
        if (jt_addr[e] < 0) return false;
      }
    pair4.clear();
    for (int r = 0; r < m; ++r) {
      if (eq_index[r] >= 0) continue;
      for (int a = t.jr_ptr[r]; a < t.jr_ptr[r + 1]; ++a)
        for (int b2 = t.jr_ptr[r]; b2 <= a; ++b2) {
          const int32_t ad = addr(t.jr_pos[a], t.jr_pos[b2]);
          if (ad < 0) return false;
          pair4.push_back(a); pair4.push_back(b2); pair4.push_back(ad); pair4.push_back(r);
        }
    }
    eqe3.clear();
    for (int k = 0; k < d.n_eq; ++k) {
      const int r = t.eq_rows[k];
      for (int a = t.jr_ptr[r]; a < t.jr_ptr[r + 1]; ++a) {
        eqe3.push_back(a); eqe3.push_back(d_off[d.n_leaf] + tri(d.n_root + k, t.jr_pos[a] - d.root_off)); eqe3.push_back(r);
      }
    }
    d.n_eqe = (int)eqe3.size() / 3;
    if (eqe3.empty()) eqe3.assign(3, 0);
    d.n_pairs = (int)pair4.size() / 4;
    if (pair4.empty()) pair4.assign(4, 0);
    diag_addr.assign(d.N, 0);
    for (int q = 0; q < d.N; ++q) diag_addr[q] = addr(q, q);
    t_row.assign(d.n_terms > 0 ? d.n_terms : 1, 0);
    h_addr.assign(3 * (d.n_terms > 0 ? d.n_terms : 1), 0);
    t_pos.assign(3 * (d.n_terms > 0 ? d.n_terms : 1), -1);
    for (int tt = 0; tt < d.n_terms; ++tt)
      for (int k = 0; k < 3; ++k) if (t.t_var[3 * tt + k] >= 0) t_pos[3 * tt + k] = pos[t.t_var[3 * tt + k]];
    for (int r = 0; r <= m; ++r)
      for (int tt = t.row_ptr[r]; tt < t.row_ptr[r + 1]; ++tt) {
        t_row[tt] = r;
        const int32_t* tv = t.t_var + 3 * tt;
        const int pr[3][2] = {{0, 1}, {0, 2}, {1, 2}};
        for (int k = 0; k < 3; ++k) {
          const int va = tv[pr[k][0]], vb = tv[pr[k][1]];
          if (va < 0 || vb < 0) continue;
          const int pa = pos[va], pb = pos[vb];
          const int32_t ad = pa >= pb ? addr(pa, pb) : addr(pb, pa);
          if (ad < 0) return false;
          h_addr[3 * tt + k] = ad;
        }
      }
    // flat tables of the parameter stage
    slot_rng.assign(2 * (d.n_slots > 0 ? d.n_slots : 1), 0);
    for (int sl = 0; sl < d.n_slots; ++sl) { slot_rng[2 * sl] = t.pp_ptr[t.slot_pp[sl]]; slot_rng[2 * sl + 1] = t.pp_ptr[t.slot_pp[sl] + 1]; }
    T.slot_rng = slot_rng.data();
    // packed (row, position) of the Jacobian entries
    d.rp_packed = (m < 65535 && d.N < 65536) ? 1 : 0;
    je_rp.assign(d.nnz_j > 0 ? d.nnz_j : 1, 0);
    if (d.rp_packed)
      for (int e = 0; e < d.nnz_j; ++e) je_rp[e] = (int32_t)(((uint32_t)je_row[e] << 16) | (uint32_t)t.jr_pos[e]);
    T.je_rp = je_rp.data();
    // packed term records
    if (d.n_var >= 32767) return false;
    trec.assign(d.n_terms > 0 ? d.n_terms : 1, TermRec{0.0, -1, 0, 0, 0, 0, -1, -1, -1, 0});
    hrec.clear();
    for (int tt = 0; tt < d.n_terms; ++tt) {
      const int32_t* tv = t.t_var + 3 * tt;
      const int32_t* je = t.t_jidx + 3 * tt;
      TermRec& q = trec[tt];
      q.coef = t.t_coef[tt]; q.slot = t.t_slot[tt]; q.row = t_row[tt];
      q.j0 = je[0] < 0 ? 0 : je[0]; q.j1 = je[1] < 0 ? 0 : je[1]; q.j2 = je[2] < 0 ? 0 : je[2];
      q.v0 = (int16_t)tv[0]; q.v1 = (int16_t)tv[1]; q.v2 = (int16_t)tv[2];
      if (tv[1] >= 0) {
        HessRec h;
        h.coef = q.coef; h.slot = q.slot; h.row = q.row;
        h.ha0 = h_addr[3 * tt]; h.ha1 = h_addr[3 * tt + 1]; h.ha2 = h_addr[3 * tt + 2];
        h.v0 = q.v0; h.v1 = q.v1; h.v2 = q.v2;
        h.p0 = (int16_t)t_pos[3 * tt]; h.p1 = (int16_t)t_pos[3 * tt + 1]; h.p2 = (int16_t)(tv[2] >= 0 ? t_pos[3 * tt + 2] : 0);
        hrec.push_back(h);
      }
    }
    d.n_hess = (int)hrec.size();
    if (hrec.empty()) hrec.push_back(HessRec{0.0, -1, 0, 0, 0, 0, -1, -1, -1, 0, 0, 0});
    T.trec = trec.data(); T.hrec = hrec.data();
    T.pair4 = pair4.data(); T.eqe3 = eqe3.data();
    T.je_row = je_row.data(); T.jt_addr = jt_addr.data(); T.diag_addr = diag_addr.data();
    T.h_addr = h_addr.data(); T.t_row = t_row.data(); T.t_pos = t_pos.data(); T.pm_rec = pm_rec.data();
    reg_w.assign(d.N, OMGX_DW_LINEAR);
    for (int tt = 0; tt < t.row_ptr[m + 1]; ++tt) {
      const int32_t* tv = t.t_var + 3 * tt;
      if (tv[1] < 0) continue;                                  // constant or linear term
      // class markers, turned into weights by the kernel: -1 nonlinear leaf variable, +1 nonlinear root variable
      for (int k = 0; k < 3; ++k) if (tv[k] >= 0) reg_w[pos[tv[k]]] = (pos[tv[k]] < d.root_off) ? -1.0 : 1.0;
    }
    T.reg_w = reg_w.data();
    return true;
  }
};

}  // namespace omgx
