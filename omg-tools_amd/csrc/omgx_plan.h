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

  bool build(const omgx_template& t) {
    Dims& d = dims;
    d.n_var = t.n_var; d.n_par = t.n_par; d.n_con = t.n_con; d.n_atoms = t.n_atoms;
    d.n_slots = t.n_slots; d.n_terms = t.n_terms; d.n_prog = t.n_prog;
    d.N = t.n_var + 1; d.n_leaf = t.n_leaf; d.n_root = t.n_root; d.n_eq = t.n_eq;
    d.nnz_j = t.nnz_j; d.root_off = t.leaf_off[t.n_leaf]; d.nr = t.n_root + t.n_eq;
    if (d.root_off + d.n_root != d.N) return false;
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
      d_off[l] = off; off += n * (n + 1) / 2;
      b_off[l] = off; off += nc * n;
    }
    d_off[d.n_leaf] = off; off += d.nr * (d.nr + 1) / 2;
    kkt_doubles = off;
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
    return true;
  }
};

}  // namespace omgx
