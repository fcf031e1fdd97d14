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

// (shorthands of this layer; the reference's headers spell the types out, callers' code compiles against either)
typedef std::vector<double> vec_t;
typedef std::vector<std::vector<double>> mat_t;
typedef std::map<std::string, std::map<std::string, std::vector<double>>> dict_t;

class Vehicle {
  private:
    int n_st, n_in, n_spl, degree, len_basis, knot_intervals;
    bool ideal_prediction, provide_prediction;
    double horizon_time;
    vec_t knots;                       // on [0, 1]: degree + 1 zeros, the interior breaks, degree + 1 ones
    vec_t predicted_state, predicted_input;
    // coefficients of the o-th derivative (w.r.t. the normalised time) = derivative_T[o] * coefficients
    std::vector<mat_t> derivative_T;

    void integrate(vec_t& state0, mat_t& input, vec_t& stateT,
                   double sample_time, int steps);
    void createDerivativeMatrices();

  protected:
    double evalSpline(double x, const vec_t& knots, const vec_t& coeffs, int degree);
    void sampleSplines(mat_t& spline_coeffs, vec_t time, int derivative,
                       mat_t& spline_sampled);
    void sampleSplines(mat_t& spline_coeffs, vec_t time,
                       mat_t& spline_sampled);
    void getPrediction(vec_t& state, vec_t& input);
    void setPrediction(vec_t& state, vec_t& input);

  public:
    virtual void setInitialConditions(vec_t& conditions) = 0;
    virtual void setTerminalConditions(vec_t& conditions) = 0;
    virtual void getInitSplineValue(mat_t& init_value) = 0;
    virtual void setParameters(std::map<std::string, vec_t>& par_dict) = 0;
    virtual void ode(vec_t& state, vec_t& input, vec_t& dstate) = 0;
    virtual void splines2State(mat_t& spline_coeffs, vec_t time,
                               mat_t& state) = 0;
    virtual void splines2Input(mat_t& spline_coeffs, vec_t time,
                               mat_t& input) = 0;
    virtual ~Vehicle() {}

    Vehicle(int n_st, int n_in, int n_spl, int degree, int knot_intervals);
    Vehicle(int n_st, int n_in, int n_spl, int degree);

    void predict(vec_t& state0, mat_t& state_trajectory,
                 mat_t& input_trajectory, double predict_time, double sample_time,
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
vec_t clampedUniformKnots(int degree, int knot_intervals);
// all basis functions of the span that holds x (reference convention `basics/spline.py:131-136`: spans are
// (k_j, k_{j+1}], closed on the left at the first knot); returns the index of the first non-zero function
int basisFunctions(const vec_t& knots, int degree, double x, vec_t& values);
// Horizon shift by one knot interval (`basics/spline_extra.py:165-191` shiftoverknot_T): c' = T c describes
// s(tau + 1 / knot_intervals) on the same knots, the last span continuing the last polynomial piece.  Row-major L x L.
vec_t shiftOverKnot(int degree, int knot_intervals);

}  // namespace omg
#endif
