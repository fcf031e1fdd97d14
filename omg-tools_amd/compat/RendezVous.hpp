// omg::RendezVous -- header-compatible with `export/point2point/admm/rendezvous/RendezVous.hpp:27-44`: the ADMM class
// whose shared variable is the meeting point of the fleet, the vehicle's free end point conT0 + rel_pos_c
// (`problems/rendezvous.py:26-67`; the x-update is a FreeEndPoint2point, `problems/point2point.py:376-418`).
#ifndef OMG_COMPAT_RENDEZVOUS
#define OMG_COMPAT_RENDEZVOUS

#include "ADMMPoint2Point.hpp"

namespace omg {

class RendezVous : public ADMMPoint2Point {
  private:
    vec_t rel_pos_c;
    std::string free_lbl;                                 // label of the problem that owns conT0
    void fillParameterDict(obstacles_t&, dict_t&);
    void retrieveSharedVariables(dict_t&);
    void initVariables();

  public:
    RendezVous(Vehicle* vehicle, double update_time, double sample_time, double horizon_time);
    RendezVous(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length);
    RendezVous(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, int init_iter);
    RendezVous(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, int init_iter, double rho);
    bool update1(vec_t& condition0, vec_t& conditionT, mat_t& state_trajectory, mat_t& input_trajectory, vec_t& x_i,
                 mat_t& z_ji, mat_t& l_ji, obstacles_t& obstacles, vec_t& rel_pos_c);
    bool update1(vec_t& condition0, vec_t& conditionT, mat_t& state_trajectory, mat_t& input_trajectory, vec_t& x_i,
                 mat_t& z_ji, mat_t& l_ji, obstacles_t& obstacles, vec_t& rel_pos_c, int predict_shift);
    bool update2(mat_t& x_j, mat_t& z_ij, mat_t& l_ij, vec_t& residuals);
};

}  // namespace omg
#endif
