// omg::Holonomic -- header-compatible with `export/vehicles/Holonomic.hpp:27-44`: two states (position), two inputs
// (velocity), two cubic splines; the plan is the position, its derivative the input.
#ifndef OMG_COMPAT_HOLONOMIC
#define OMG_COMPAT_HOLONOMIC

#include "Vehicle.hpp"

namespace omg {

class Holonomic : public Vehicle {
  private:
    std::vector<double> poseT;

  public:
    Holonomic();
    void setInitialConditions(std::vector<double>& conditions);
    void setTerminalConditions(std::vector<double>& conditions);
    void setParameters(std::map<std::string, std::vector<double>>& par_dict);
    void ode(std::vector<double>& state, std::vector<double>& input, std::vector<double>& dstate);
    void getInitSplineValue(std::vector<std::vector<double>>& init_value);
    void splines2State(std::vector<std::vector<double>>& spline_coeffs, std::vector<double> time,
                       std::vector<std::vector<double>>& state);
    void splines2Input(std::vector<std::vector<double>>& spline_coeffs, std::vector<double> time,
                       std::vector<std::vector<double>>& input);
};

}  // namespace omg
#endif
