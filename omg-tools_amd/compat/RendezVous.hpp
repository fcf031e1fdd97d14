// omg::RendezVous -- header-compatible with `export/point2point/admm/rendezvous/RendezVous.hpp:27-44`: the ADMM class
// whose shared variable is the meeting point of the fleet, the vehicle's free end point conT0 + rel_pos_c
// (`problems/rendezvous.py:26-67`; the x-update is a FreeEndPoint2point, `problems/point2point.py:376-418`).
#ifndef OMG_COMPAT_RENDEZVOUS
#define OMG_COMPAT_RENDEZVOUS

#include "ADMMPoint2Point.hpp"

namespace omg {

class RendezVous : public ADMMPoint2Point {
  private:
    std::vector<double> rel_pos_c;
    std::string free_lbl;                                 // label of the problem that owns conT0
    void fillParameterDict(std::vector<obstacle_t>&, std::map<std::string, std::map<std::string, std::vector<double>>>&);
    void retrieveSharedVariables(std::map<std::string, std::map<std::string, std::vector<double>>>&);
    void initVariables();

  public:
    RendezVous(Vehicle* vehicle, double update_time, double sample_time, double horizon_time);
    RendezVous(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length);
    RendezVous(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, int init_iter);
    RendezVous(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, int init_iter, double rho);
    bool update1(std::vector<double>&, std::vector<double>&, std::vector<std::vector<double>>&, std::vector<std::vector<double>>&, std::vector<double>&, std::vector<std::vector<double>>&, std::vector<std::vector<double>>&, std::vector<obstacle_t>&, std::vector<double>&);
    bool update1(std::vector<double>&, std::vector<double>&, std::vector<std::vector<double>>&, std::vector<std::vector<double>>&, std::vector<double>&, std::vector<std::vector<double>>&, std::vector<std::vector<double>>&, std::vector<obstacle_t>&, std::vector<double>&, int);
    bool update2(std::vector<std::vector<double>>&, std::vector<std::vector<double>>&, std::vector<std::vector<double>>&, std::vector<double>&);
};

}  // namespace omg
#endif
