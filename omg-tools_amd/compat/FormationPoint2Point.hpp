// omg::FormationPoint2Point -- header-compatible with `export/point2point/admm/formation/FormationPoint2Point.hpp:27-44`:
// the ADMM class whose shared variable is the fleet centre, trajectory + rel_pos_c (`vehicles/vehicle.py:234-240`).
#ifndef OMG_COMPAT_FORMATIONPOINT2POINT
#define OMG_COMPAT_FORMATIONPOINT2POINT

#include "ADMMPoint2Point.hpp"

namespace omg {

class FormationPoint2Point : public ADMMPoint2Point {
  private:
    std::vector<double> rel_pos_c;
    void fillParameterDict(std::vector<obstacle_t>&, std::map<std::string, std::map<std::string, std::vector<double>>>&);
    void retrieveSharedVariables(std::map<std::string, std::map<std::string, std::vector<double>>>&);

  public:
    FormationPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time);
    FormationPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length);
    FormationPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, int init_iter);
    FormationPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, int init_iter, double rho);
    bool update1(std::vector<double>&, std::vector<double>&, std::vector<std::vector<double>>&, std::vector<std::vector<double>>&, std::vector<double>&, std::vector<std::vector<double>>&, std::vector<std::vector<double>>&, std::vector<obstacle_t>&, std::vector<double>&);
    bool update1(std::vector<double>&, std::vector<double>&, std::vector<std::vector<double>>&, std::vector<std::vector<double>>&, std::vector<double>&, std::vector<std::vector<double>>&, std::vector<std::vector<double>>&, std::vector<obstacle_t>&, std::vector<double>&, int);
    bool update2(std::vector<std::vector<double>>&, std::vector<std::vector<double>>&, std::vector<std::vector<double>>&, std::vector<double>&);
};

}  // namespace omg
#endif
