// Replay test of the compat layer, shaped like the reference's `export/tests/point2point/test.cpp:30-141`: the scenario
// of `examples/p2p_holonomic_export.py:31-45` (Holonomic vehicle from (0, 0) to (3.5, 3.5), two rectangular obstacles),
// n_iter updates of omg::Point2Point with ideal prediction, the state and input trajectories of every update compared
// with CSV files the Python path wrote (one row per signal and update, trajectory_length values) -- here with a
// two-sided tolerance (the reference's assert is one-sided), absolute for values near zero.
//   replay <data_state.csv> <data_input.csv> <n_iter> <tolerance>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <vector>
#include "Holonomic.hpp"
#include "Point2Point.hpp"

using namespace std;

static bool read_csv(const char* path, int n_iter, int len, vector<vector<vector<double>>>& data) {
    ifstream f(path);
    if (!f) return false;
    for (int i = 0; i < 2 * n_iter; ++i) {
        string line, val;
        if (!getline(f, line)) return false;
        stringstream ss(line);
        for (int j = 0; j < len; ++j) { if (!getline(ss, val, ',')) return false; data[i / 2][j][i % 2] = atof(val.c_str()); }
    }
    return true;
}

int main(int argc, char** argv) {
    if (argc < 5) { cerr << "usage: replay data_state.csv data_input.csv n_iter tolerance" << endl; return 2; }
    const int n_iter = atoi(argv[3]);
    const double tol = atof(argv[4]);
    const double horizon_time = 10, sample_time = 0.01, update_time = 0.1;
    const int trajectory_length = 5;
    omg::Holonomic* vehicle = new omg::Holonomic();
    vehicle->setIdealPrediction(true);       // prediction of the initial state from the plan itself
    omg::Point2Point p2p(vehicle, update_time, sample_time, horizon_time, trajectory_length);

    vector<double> state0(2, 0.0), stateT(2, 3.5);
    vector<vector<double>> input_trajectory(trajectory_length, vector<double>(2)), state_trajectory(trajectory_length, vector<double>(2));
    vector<omg::obstacle_t> obstacles(p2p.n_obs);
    const double width = 3.0, height = 0.2, radius = 0.001;
    for (int k = 0; k < p2p.n_obs; ++k) {
        obstacles[k].position.assign(2, 0.0); obstacles[k].velocity.assign(2, 0.0); obstacles[k].acceleration.assign(2, 0.0);
        obstacles[k].checkpoints = {0.5 * width, 0.5 * height, 0.5 * width, -0.5 * height, -0.5 * width, -0.5 * height, -0.5 * width, 0.5 * height};
        obstacles[k].radii.assign(4, radius);
        obstacles[k].avoid = true;
    }
    obstacles[0].position = {-0.6, 1.0};
    obstacles[1].position = {3.2, 1.0};

    vector<vector<vector<double>>> data_state(n_iter, vector<vector<double>>(trajectory_length, vector<double>(2)));
    vector<vector<vector<double>>> data_input(n_iter, vector<vector<double>>(trajectory_length, vector<double>(2)));
    if (!read_csv(argv[1], n_iter, trajectory_length, data_state) || !read_csv(argv[2], n_iter, trajectory_length, data_input)) {
        cerr << "cannot read the csv files" << endl; return 2;
    }
    double worst = 0.0;
    for (int i = 0; i < n_iter; ++i) {
        if (!p2p.update(state0, stateT, state_trajectory, input_trajectory, obstacles)) { cerr << "update " << i << " failed" << endl; return 1; }
        for (int k = 0; k < 2; ++k)
            for (int j = 0; j < trajectory_length; ++j) {
                const double es = fabs(data_state[i][j][k] - state_trajectory[j][k]) / max(1.0, fabs(data_state[i][j][k]));
                const double ei = fabs(data_input[i][j][k] - input_trajectory[j][k]) / max(1.0, fabs(data_input[i][j][k]));
                worst = max(worst, max(es, ei));
                if (es > tol || ei > tol) {
                    cerr << "update " << i << " sample " << j << " axis " << k << ": state " << state_trajectory[j][k] << " vs " << data_state[i][j][k]
                         << ", input " << input_trajectory[j][k] << " vs " << data_input[i][j][k] << endl;
                    return 1;
                }
            }
        cout << "it: " << i << ", iterations: " << p2p.getIterations() << ", x: " << state_trajectory[0][0] << " " << state_trajectory[0][1] << endl;
    }
    vector<double> coeffs;
    p2p.getCoefficients(coeffs);
    cout << "replayed " << n_iter << " updates, worst deviation " << worst << ", " << coeffs.size() << " coefficients, basis length " << p2p.getLenBasis() << endl;
    delete vehicle;
    return 0;
}
