// omg::Vehicle -- header-compatible with the reference's C++ export (`export/vehicles/Vehicle.hpp:27-81`: the same
// public methods and pure virtuals, so that a vehicle class written for the exported library compiles against this
// one), implemented from scratch for the compat layer over libomgx.so (no CasADi).  What a vehicle does here:
// holds the predicted state / input the next solve starts from, turns spline coefficients into sampled state and
// input trajectories (clamped uniform B-spline basis of `degree` with `knot_intervals` intervals over the
// horizon), and predicts the state at the next update (`Vehicle.cpp:61-110`: ideal prediction from the stored
// trajectories, or the caller's state integrated with classic Runge-Kutta over the stored inputs).
#ifndef OMG_COMPAT_VEHICLE
#define OMG_COMPAT_VEHICLE

#include <map>
#include <string>
#include <vector>

namespace omg {

class Vehicle {
  private:
    int n_st, n_in, n_spl, degree, len_basis, knot_intervals;
    bool ideal_prediction, provide_prediction;
    double horizon_time;
    std::vector<double> knots;                       // on [0, 1]: degree + 1 zeros, the interior breaks, degree + 1 ones
    std::vector<double> predicted_state, predicted_input;
    // coefficients of the o-th derivative (w.r.t. the normalised time) = derivative_T[o] * coefficients
    std::vector<std::vector<std::vector<double>>> derivative_T;

    void integrate(std::vector<double>& state0, std::vector<std::vector<double>>& input, std::vector<double>& stateT,
                   double sample_time, int steps);
    void createDerivativeMatrices();

  protected:
    double evalSpline(double x, const std::vector<double>& knots, const std::vector<double>& coeffs, int degree);
    void sampleSplines(std::vector<std::vector<double>>& spline_coeffs, std::vector<double> time, int derivative,
                       std::vector<std::vector<double>>& spline_sampled);
    void sampleSplines(std::vector<std::vector<double>>& spline_coeffs, std::vector<double> time,
                       std::vector<std::vector<double>>& spline_sampled);
    void getPrediction(std::vector<double>& state, std::vector<double>& input);
    void setPrediction(std::vector<double>& state, std::vector<double>& input);

  public:
    virtual void setInitialConditions(std::vector<double>& conditions) = 0;
    virtual void setTerminalConditions(std::vector<double>& conditions) = 0;
    virtual void getInitSplineValue(std::vector<std::vector<double>>& init_value) = 0;
    virtual void setParameters(std::map<std::string, std::vector<double>>& par_dict) = 0;
    virtual void ode(std::vector<double>& state, std::vector<double>& input, std::vector<double>& dstate) = 0;
    virtual void splines2State(std::vector<std::vector<double>>& spline_coeffs, std::vector<double> time,
                               std::vector<std::vector<double>>& state) = 0;
    virtual void splines2Input(std::vector<std::vector<double>>& spline_coeffs, std::vector<double> time,
                               std::vector<std::vector<double>>& input) = 0;
    virtual ~Vehicle() {}

    Vehicle(int n_st, int n_in, int n_spl, int degree, int knot_intervals);
    Vehicle(int n_st, int n_in, int n_spl, int degree);

    void predict(std::vector<double>& state0, std::vector<std::vector<double>>& state_trajectory,
                 std::vector<std::vector<double>>& input_trajectory, double predict_time, double sample_time,
                 int predict_shift);
    void setKnotHorizon(double horizon_time);
    void setIdealPrediction(bool ideal_prediction);
    void setProvidePrediction(bool provide_prediction);
    int getNSplines();
    int getNState();
    int getNInput();
    int getLenBasis();
    int getDegree();
    int getKnotIntervals();
};

// B-spline helpers shared with Point2Point (clamped uniform basis on [0, 1])
std::vector<double> clampedUniformKnots(int degree, int knot_intervals);
// all basis functions of the span that holds x (reference convention `basics/spline.py:131-136`: spans are
// (k_j, k_{j+1}], closed on the left at the first knot); returns the index of the first non-zero function
int basisFunctions(const std::vector<double>& knots, int degree, double x, std::vector<double>& values);
// Horizon shift by one knot interval (`basics/spline_extra.py:165-191` shiftoverknot_T): c' = T c describes
// s(tau + 1 / knot_intervals) on the same knots, the last span continuing the last polynomial piece.  Row-major L x L.
std::vector<double> shiftOverKnot(int degree, int knot_intervals);

}  // namespace omg
#endif
