// omg::Holonomic -- header-compatible with `export/vehicles/Holonomic.hpp:27-44`: two states (position), two inputs
// (velocity), two cubic splines; the plan is the position, its derivative the input.
#ifndef OMG_COMPAT_HOLONOMIC
#define OMG_COMPAT_HOLONOMIC

#include "Vehicle.hpp"

namespace omg {

class Holonomic : public Vehicle {
  private:
    vec_t poseT;

  public:
    Holonomic();
    void setInitialConditions(vec_t& conditions);
    void setTerminalConditions(vec_t& conditions);
    void setParameters(std::map<std::string, vec_t>& par_dict);
    void ode(vec_t& state, vec_t& input, vec_t& dstate);
    void getInitSplineValue(mat_t& init_value);
    void splines2State(mat_t& spline_coeffs, vec_t time,
                       mat_t& state);
    void splines2Input(mat_t& spline_coeffs, vec_t time,
                       mat_t& input);
};

}  // namespace omg
#endif
