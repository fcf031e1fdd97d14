// omg::FormationPoint2Point -- header-compatible with `export/point2point/admm/formation/FormationPoint2Point.hpp:27-44`:
// the ADMM class whose shared variable is the fleet centre, trajectory + rel_pos_c (`vehicles/vehicle.py:234-240`).
#ifndef OMG_COMPAT_FORMATIONPOINT2POINT
#define OMG_COMPAT_FORMATIONPOINT2POINT

#include "ADMMPoint2Point.hpp"

namespace omg {

class FormationPoint2Point : public ADMMPoint2Point {
  private:
    vec_t rel_pos_c;
    void fillParameterDict(obstacles_t&, dict_t&);
    void retrieveSharedVariables(dict_t&);

  public:
    FormationPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time);
    FormationPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length);
    FormationPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, int init_iter);
    FormationPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, int init_iter, double rho);
    bool update1(vec_t& condition0, vec_t& conditionT, mat_t& state_trajectory, mat_t& input_trajectory, vec_t& x_i,
                 mat_t& z_ji, mat_t& l_ji, obstacles_t& obstacles, vec_t& rel_pos_c);
    bool update1(vec_t& condition0, vec_t& conditionT, mat_t& state_trajectory, mat_t& input_trajectory, vec_t& x_i,
                 mat_t& z_ji, mat_t& l_ji, obstacles_t& obstacles, vec_t& rel_pos_c, int predict_shift);
    bool update2(mat_t& x_j, mat_t& z_ij, mat_t& l_ij, vec_t& residuals);
};

}  // namespace omg
#endif
