// omg::Vehicle / omg::Holonomic for the compat layer (see Vehicle.hpp); behaviour of `export/vehicles/Vehicle.cpp`
// and `Holonomic.cpp`, written from scratch.
#include "Vehicle.hpp"
#include "Holonomic.hpp"
#include <cmath>
#include <stdexcept>

namespace omg {

std::vector<double> clampedUniformKnots(int degree, int knot_intervals) {
    std::vector<double> k;
    for (int i = 0; i < degree; ++i) k.push_back(0.0);
    for (int i = 0; i <= knot_intervals; ++i) k.push_back((double)i / knot_intervals);
    for (int i = 0; i < degree; ++i) k.push_back(1.0);
    return k;
}

// span j with k_j < x <= k_{j+1} (first span closed on the left), then the degree + 1 functions that live on it
// (the triangular scheme of The NURBS Book A2.2); values[r] = B_{j - degree + r}(x)
static int findSpan(const std::vector<double>& knots, int degree, double x) {
    const int n_fun = (int)knots.size() - degree - 1;
    int j = degree;
    for (int q = degree + 1; q < n_fun; ++q) if (knots[q] < x) j = q;
    return j;
}

int basisFunctions(const std::vector<double>& knots, int degree, double x, std::vector<double>& values) {
    const int j = findSpan(knots, degree, x);
    values.assign(degree + 1, 0.0);
    std::vector<double> left(degree + 1), right(degree + 1);
    values[0] = 1.0;
    for (int r = 1; r <= degree; ++r) {
        left[r] = x - knots[j + 1 - r];
        right[r] = knots[j + r] - x;
        double saved = 0.0;
        for (int q = 0; q < r; ++q) {
            const double den = right[q + 1] + left[r - q];
            const double tmp = den != 0.0 ? values[q] / den : 0.0;
            values[q] = saved + right[q + 1] * tmp;
            saved = left[r - q] * tmp;
        }
        values[r] = saved;
    }
    return j - degree;
}

// value at x of the polynomial piece that lives on span j (for x outside the span: its analytic continuation --
// the recursion is a polynomial identity)
static double piece(const std::vector<double>& knots, int degree, int j, const std::vector<double>& c, double x) {
    std::vector<double> d(c.begin() + (j - degree), c.begin() + j + 1);
    for (int r = 1; r <= degree; ++r)
        for (int q = degree; q >= r; --q) {
            const double den = knots[j + 1 + q - r] - knots[j - degree + q];
            const double a = den != 0.0 ? (x - knots[j - degree + q]) / den : 0.0;
            d[q] = (1.0 - a) * d[q - 1] + a * d[q];
        }
    return d[degree];
}

std::vector<double> shiftOverKnot(int degree, int knot_intervals) {
    const std::vector<double> knots = clampedUniformKnots(degree, knot_intervals);
    const int L = knot_intervals + degree;
    const double delta = 1.0 / knot_intervals;
    // collocation at the Greville abscissae: B c' = s(x + delta), the last piece continued beyond 1
    std::vector<double> x(L);
    for (int i = 0; i < L; ++i) {
        double sum = 0.0;
        for (int q = 1; q <= degree; ++q) sum += knots[i + q];
        x[i] = degree > 0 ? sum / degree : 0.5 * (knots[i] + knots[i + 1]);
    }
    std::vector<double> A(L * L, 0.0), R(L * L, 0.0);       // A: collocation matrix; R[:, m]: target values of unit vector m
    std::vector<double> vals;
    for (int i = 0; i < L; ++i) {
        const int f0 = basisFunctions(knots, degree, x[i], vals);
        for (int r = 0; r <= degree; ++r) A[i * L + f0 + r] = vals[r];
        const double xs = x[i] + delta;
        const int j = xs <= 1.0 ? findSpan(knots, degree, xs) : L - 1;      // (beyond the horizon: the last span's piece)
        std::vector<double> e(L, 0.0);
        for (int m = j - degree; m <= j; ++m) { e[m] = 1.0; R[i * L + m] = piece(knots, degree, j, e, xs); e[m] = 0.0; }
    }
    // solve A T = R by Gaussian elimination with partial pivoting
    for (int c = 0; c < L; ++c) {
        int piv = c;
        for (int r = c + 1; r < L; ++r) if (std::fabs(A[r * L + c]) > std::fabs(A[piv * L + c])) piv = r;
        if (A[piv * L + c] == 0.0) throw std::runtime_error("shiftOverKnot: singular collocation matrix");
        if (piv != c) for (int k = 0; k < L; ++k) { std::swap(A[c * L + k], A[piv * L + k]); std::swap(R[c * L + k], R[piv * L + k]); }
        for (int r = c + 1; r < L; ++r) {
            const double f = A[r * L + c] / A[c * L + c];
            if (f == 0.0) continue;
            for (int k = c; k < L; ++k) A[r * L + k] -= f * A[c * L + k];
            for (int k = 0; k < L; ++k) R[r * L + k] -= f * R[c * L + k];
        }
    }
    for (int c = L - 1; c >= 0; --c)
        for (int k = 0; k < L; ++k) {
            double v = R[c * L + k];
            for (int q = c + 1; q < L; ++q) v -= A[c * L + q] * R[q * L + k];
            R[c * L + k] = v / A[c * L + c];
        }
    for (double& v : R) if (std::fabs(v) < 1e-10) v = 0.0;        // (`spline_extra.py`: entries below 1e-10 are zeros)
    return R;
}

Vehicle::Vehicle(int n_st, int n_in, int n_spl, int degree, int knot_intervals)
    : n_st(n_st), n_in(n_in), n_spl(n_spl), degree(degree), len_basis(knot_intervals + degree), knot_intervals(knot_intervals),
      ideal_prediction(false), provide_prediction(false), horizon_time(1.0), predicted_state(n_st), predicted_input(n_in) {
    knots = clampedUniformKnots(degree, knot_intervals);
    createDerivativeMatrices();
}

Vehicle::Vehicle(int n_st, int n_in, int n_spl, int degree) : Vehicle(n_st, n_in, n_spl, degree, 10) {}

void Vehicle::createDerivativeMatrices() {
    // d/dx sum c_i B_{i,p} = sum p (c_{i+1} - c_i) / (k_{i+p+1} - k_{i+1}) B_{i+1,p-1}: products of bidiagonal matrices
    derivative_T.assign(degree + 1, {});
    std::vector<std::vector<double>> cur(len_basis, std::vector<double>(len_basis, 0.0));
    for (int i = 0; i < len_basis; ++i) cur[i][i] = 1.0;
    derivative_T[0] = cur;
    for (int o = 1; o <= degree; ++o) {
        const int p = degree - o + 1, rows = len_basis - o;
        std::vector<std::vector<double>> nxt(rows, std::vector<double>(len_basis, 0.0));
        for (int i = 0; i < rows; ++i) {
            // knots of the order-(o-1) derivative basis: the original ones with o - 1 dropped at either end
            const double den = knots[i + p + o] - knots[i + o];
            const double f = den != 0.0 ? p / den : 0.0;
            for (int m = 0; m < len_basis; ++m) nxt[i][m] = f * (cur[i + 1][m] - cur[i][m]);
        }
        derivative_T[o] = nxt;
        cur = nxt;
    }
}

double Vehicle::evalSpline(double x, const std::vector<double>& kn, const std::vector<double>& coeffs, int deg) {
    std::vector<double> vals;
    const int f0 = basisFunctions(kn, deg, x, vals);
    double v = 0.0;
    for (int r = 0; r <= deg; ++r) v += vals[r] * coeffs[f0 + r];
    return v;
}

void Vehicle::sampleSplines(std::vector<std::vector<double>>& spline_coeffs, std::vector<double> time, int derivative,
                            std::vector<std::vector<double>>& sampled) {
    const std::vector<double> kn(knots.begin() + derivative, knots.end() - derivative);
    const double scale = 1.0 / std::pow(horizon_time, derivative);
    for (int i = 0; i < n_spl; ++i) {
        std::vector<double> c(len_basis - derivative, 0.0);
        for (int l = 0; l < len_basis - derivative; ++l)
            for (int m = 0; m < len_basis; ++m) c[l] += scale * derivative_T[derivative][l][m] * spline_coeffs[i][m];
        for (size_t k = 0; k < time.size(); ++k) sampled[k][i] = evalSpline(time[k] / horizon_time, kn, c, degree - derivative);
    }
}

void Vehicle::sampleSplines(std::vector<std::vector<double>>& spline_coeffs, std::vector<double> time,
                            std::vector<std::vector<double>>& sampled) {
    sampleSplines(spline_coeffs, time, 0, sampled);
}

void Vehicle::getPrediction(std::vector<double>& state, std::vector<double>& input) { state = predicted_state; input = predicted_input; }
void Vehicle::setPrediction(std::vector<double>& state, std::vector<double>& input) { predicted_state = state; predicted_input = input; }

void Vehicle::predict(std::vector<double>& state0, std::vector<std::vector<double>>& state_trajectory,
                      std::vector<std::vector<double>>& input_trajectory, double predict_time, double sample_time,
                      int predict_shift) {
    const int steps = (int)(predict_time / sample_time);
    if (ideal_prediction) {                       // the plan itself at the time of the next update
        std::vector<double> st = state_trajectory[steps + predict_shift], in = input_trajectory[steps + predict_shift];
        setPrediction(st, in);
    } else if (provide_prediction) {              // the caller's state, the planned input
        std::vector<double> in = input_trajectory[steps + predict_shift];
        setPrediction(state0, in);
    } else {                                      // the caller's state integrated over the planned inputs
        std::vector<std::vector<double>> input(steps + 1);
        for (int k = 0; k <= steps; ++k) input[k] = input_trajectory[k + predict_shift];
        std::vector<double> stT(state0.size());
        integrate(state0, input, stT, sample_time, steps);
        setPrediction(stT, input[steps]);
    }
}

void Vehicle::integrate(std::vector<double>& state0, std::vector<std::vector<double>>& input, std::vector<double>& stateT,
                        double h, int steps) {
    // classic Runge-Kutta, the statements of `Vehicle.cpp:82-110` (every stage starts from state0; the input is held
    // over the first three stages and taken at the end of the interval for the fourth)
    std::vector<double> k1(n_st), k2(n_st), k3(n_st), k4(n_st), st(n_st);
    stateT = state0;
    for (int i = 0; i < steps; ++i) {
        ode(state0, input[i], k1);
        for (int j = 0; j < n_st; ++j) st[j] = state0[j] + 0.5 * h * k1[j];
        ode(st, input[i], k2);
        for (int j = 0; j < n_st; ++j) st[j] = state0[j] + 0.5 * h * k2[j];
        ode(st, input[i], k3);
        for (int j = 0; j < n_st; ++j) st[j] = state0[j] + h * k3[j];
        ode(st, input[i + 1], k4);
        for (int j = 0; j < n_st; ++j) stateT[j] += (h / 6.0) * (k1[j] + 2 * k2[j] + 2 * k3[j] + k4[j]);
    }
}

void Vehicle::setKnotHorizon(double T) { horizon_time = T; }
void Vehicle::setIdealPrediction(bool v) { ideal_prediction = v; }
void Vehicle::setProvidePrediction(bool v) { provide_prediction = v; }
int Vehicle::getNSplines() { return n_spl; }
int Vehicle::getNState() { return n_st; }
int Vehicle::getNInput() { return n_in; }
int Vehicle::getLenBasis() { return len_basis; }
int Vehicle::getDegree() { return degree; }
int Vehicle::getKnotIntervals() { return knot_intervals; }

// ---- Holonomic ----------------------------------------------------------------------------------------------
Holonomic::Holonomic() : Vehicle(2, 2, 2, 3), poseT(2) {}

void Holonomic::setInitialConditions(std::vector<double>& conditions) {
    std::vector<double> zeros(2, 0.0);
    setPrediction(conditions, zeros);
}
void Holonomic::setTerminalConditions(std::vector<double>& conditions) { poseT = conditions; }

void Holonomic::getInitSplineValue(std::vector<std::vector<double>>& init_value) {
    // coefficients on the straight line from the predicted state to the target (`vehicles/holonomic.py:107-114`)
    std::vector<double> s0(2), i0(2);
    getPrediction(s0, i0);
    const int L = getLenBasis();
    for (int k = 0; k < getNSplines(); ++k)
        for (int j = 0; j < L; ++j) init_value[k][j] = s0[k] + j * (poseT[k] - s0[k]) / (L - 1);
}

void Holonomic::setParameters(std::map<std::string, std::vector<double>>& par) {
    std::vector<double> s0(2), i0(2);
    getPrediction(s0, i0);
    par["state0"] = s0; par["input0"] = i0; par["poseT"] = poseT;
}

void Holonomic::splines2State(std::vector<std::vector<double>>& c, std::vector<double> time, std::vector<std::vector<double>>& state) { sampleSplines(c, time, state); }
void Holonomic::splines2Input(std::vector<std::vector<double>>& c, std::vector<double> time, std::vector<std::vector<double>>& input) { sampleSplines(c, time, 1, input); }
void Holonomic::ode(std::vector<double>&, std::vector<double>& input, std::vector<double>& dstate) { dstate = input; }

}  // namespace omg
