// omg::Point2Point over libomgx.so (see Point2Point.hpp).  The flow of `update` follows the reference's exported class
// (`export/point2point/Point2Point.cpp:124-205`): time bookkeeping, warm-start shift on a knot crossing, terminal
// conditions, prediction, solve, trajectories out.  Nothing here is generated: the template file carries what the
// exporter would have baked in.
#include "Point2Point.hpp"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include "../../include/omgx.h"

#ifndef OMG_TEMPLATE_FILE
#define OMG_TEMPLATE_FILE "p2p.omgx"
#endif

namespace omg {

typedef std::map<std::string, std::map<std::string, std::vector<double>>> Dict;

Point2Point::Point2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length,
                         bool initialize)
    : _recover(false), tpl(nullptr), problem(nullptr) {
    if (trajectory_length > int(horizon_time / sample_time)) std::cerr << "trajectory_length > (horizon_time/sample_time)!" << std::endl;
    const int n_samp = std::max(trajectory_length, int(update_time / sample_time)) + 1;
    time.resize(n_samp);
    state_trajectory.assign(n_samp, std::vector<double>(vehicle->getNState()));
    input_trajectory.assign(n_samp, std::vector<double>(vehicle->getNInput()));
    this->vehicle = vehicle;
    this->update_time = update_time;
    this->sample_time = sample_time;
    this->horizon_time = horizon_time;
    this->trajectory_length = trajectory_length;
    for (size_t k = 0; k < time.size(); ++k) time[k] = k * sample_time;
    n_var = n_par = n_con = n_dim = n_obs = 0;
    if (initialize) this->initialize();
}

Point2Point::Point2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time)
    : Point2Point(vehicle, update_time, sample_time, horizon_time, int(update_time / sample_time), true) {}

Point2Point::Point2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length)
    : Point2Point(vehicle, update_time, sample_time, horizon_time, trajectory_length, true) {}

Point2Point::~Point2Point() {
    if (problem) omgx_batch_destroy(problem);
    if (tpl) omgx_template_free(tpl);
}

void Point2Point::initialize() {
    generateProblem();
    readBlockTable();
    parameters.assign(n_par, 0.0);
    variables.assign(n_var, 0.0);
    multipliers.assign(n_con, 0.0);
    lbg.assign(tpl->lbg_def, tpl->lbg_def + n_con);
    ubg.assign(tpl->ubg_def, tpl->ubg_def + n_con);
    vehicle->setKnotHorizon(horizon_time);
}

// `Point2Point.cpp:80-91` loads the generated nlp.so into an IPOPT instance; here: the template file into a one-agent batch
void Point2Point::generateProblem() {
    const char* path = getenv("OMG_TEMPLATE") ? getenv("OMG_TEMPLATE") : OMG_TEMPLATE_FILE;
    if (omgx_template_read(path, &tpl) != OMGX_OK) throw std::runtime_error(std::string("omg::Point2Point: ") + omgx_last_error());
    if (!tpl->has_bounds || tpl->n_blocks == 0) throw std::runtime_error("omg::Point2Point: the template file carries no bounds / block table");
    if (omgx_batch_create(tpl, 1, 0, &problem) != OMGX_OK) throw std::runtime_error(std::string("omg::Point2Point: ") + omgx_last_error());
    omgx_options opt;
    omgx_default_options(&opt);
    opt.tol = getenv("OMG_TOL") ? atof(getenv("OMG_TOL")) : 1e-3;          // (the exporter's TOL: ipopt.tol of the problem)
    opt.max_iter = 500;
    opt.warm_start = 0;          // like the reference: every update starts from the transformed plan, no multipliers carried over
    if (omgx_batch_set_options(problem, &opt) != OMGX_OK) throw std::runtime_error(omgx_last_error());
    n_var = tpl->n_var; n_par = tpl->n_par; n_con = tpl->n_con;
}

// the block table replaces the offsets `export/export.py:302-353` generates
void Point2Point::readBlockTable() {
    blocks.clear();
    for (int kind = 0; kind < 3; ++kind)
        for (int i = 0; i < omgx_template_n_blocks(tpl, kind); ++i) {
            const char* nm; int32_t off, rows, cols;
            if (omgx_template_block_at(tpl, kind, i, &nm, &off, &rows, &cols) != OMGX_OK) throw std::runtime_error(omgx_last_error());
            const std::string full(nm);
            const size_t dot = full.find('.');
            blocks.push_back(Block{full.substr(0, dot), full.substr(dot + 1), kind, off, rows, cols});
        }
    obstacle_lbl.clear();
    for (const Block& b : blocks) {
        if (b.kind == OMGX_BLOCK_VAR && b.name == "splines_seg0") vehicle_lbl = b.label;
        if (b.kind == OMGX_BLOCK_PAR && b.name == "T") p2p_lbl = b.label;
        if (b.kind == OMGX_BLOCK_PAR && b.name == "checkpoints") obstacle_lbl.push_back(b.label);
    }
    if (vehicle_lbl.empty() || p2p_lbl.empty()) throw std::runtime_error("omg::Point2Point: not a fixed-T point-to-point template");
    const Block* spl = find(OMGX_BLOCK_VAR, vehicle_lbl, "splines_seg0");
    if (spl->rows != vehicle->getLenBasis() || spl->cols != vehicle->getNSplines())
        throw std::runtime_error("omg::Point2Point: the vehicle does not match the template (basis length / number of splines)");
    n_dim = find(OMGX_BLOCK_PAR, obstacle_lbl.empty() ? vehicle_lbl : obstacle_lbl[0], obstacle_lbl.empty() ? "poseT" : "x")->rows;
    n_obs = (int)obstacle_lbl.size();
    // every spline variable is shifted on a knot crossing; its degree follows from the length of its basis
    const int K = vehicle->getKnotIntervals();
    for (const Block& b : blocks)
        if (b.kind == OMGX_BLOCK_VAR && b.rows > K && !shift_T.count(b.rows - K)) shift_T[b.rows - K] = shiftOverKnot(b.rows - K, K);
}

const Point2Point::Block* Point2Point::find(int kind, const std::string& label, const std::string& name) const {
    for (const Block& b : blocks) if (b.kind == kind && b.label == label && b.name == name) return &b;
    throw std::runtime_error("omg::Point2Point: the template has no entry " + label + "." + name);
}

void Point2Point::reset() {
    for (auto& row : input_trajectory) for (double& v : row) v = 0.0;
}

void Point2Point::resetTime() { current_time = 0.0; current_time_prev = 0.0; }
void Point2Point::recover() { _recover = true; }

bool Point2Point::update(std::vector<double>& condition0, std::vector<double>& conditionT,
                         std::vector<std::vector<double>>& state_traj, std::vector<std::vector<double>>& input_traj,
                         std::vector<obstacle_t>& obstacles) {
    return update(condition0, conditionT, state_traj, input_traj, obstacles, 0);
}

bool Point2Point::update(std::vector<double>& condition0, std::vector<double>& conditionT,
                         std::vector<std::vector<double>>& state_traj, std::vector<std::vector<double>>& input_traj,
                         std::vector<obstacle_t>& obstacles, int predict_shift) {
    current_time += predict_shift * sample_time;
    transformSplines(current_time, current_time_prev);            // warm start of this update
    vehicle->setTerminalConditions(conditionT);
    if (std::fabs(current_time) <= 1.e-6) vehicle->setInitialConditions(condition0);
    else vehicle->predict(condition0, this->state_trajectory, this->input_trajectory, update_time, sample_time, predict_shift);
    if (!solve(current_time, obstacles)) {
        current_time_prev = current_time;                         // (no second transformation after an infeasible update)
        return false;                                             // the caller retries
    }
    extractData();
    for (int k = 0; k < trajectory_length; ++k) {
        for (size_t j = 0; j < state_traj[0].size(); ++j) state_traj[k][j] = this->state_trajectory[k][j];
        for (size_t j = 0; j < input_traj[0].size(); ++j) input_traj[k][j] = this->input_trajectory[k][j];
    }
    current_time_prev = current_time;
    current_time += update_time;
    return true;
}

bool Point2Point::solve(double now, std::vector<obstacle_t>& obstacles) {
    if (std::fabs(now) <= 1.e-6 || _recover) { initVariables(); _recover = false; }
    updateBounds(now, obstacles);
    setParameters(obstacles);
    int32_t status = 1, iters = 0;
    std::vector<double> x(n_var);
    if (omgx_batch_solve(problem, parameters.data(), variables.data(), lbg.data(), ubg.data(), x.data(), multipliers.data(),
                         &status, &iters, OMGX_BOUNDS_SHARED) != OMGX_OK) {
        solver_output = omgx_last_error();
        std::cout << solver_output << std::endl;
        return false;
    }
    last_iters = iters;
    solver_output = omgx_status_string(status);
    if (status != OMGX_SOLVE_SUCCEEDED) { std::cout << solver_output << std::endl; return false; }
    variables = x;
    return true;
}

void Point2Point::getCoefficients(std::vector<double>& coeffs) { coeffs = spline_coeffs_vec; }
int Point2Point::getLenBasis() { return vehicle->getLenBasis(); }

void Point2Point::initVariables() {
    const int n_spl = vehicle->getNSplines(), L = vehicle->getLenBasis();
    std::vector<std::vector<double>> init(n_spl, std::vector<double>(L));
    vehicle->getInitSplineValue(init);
    Dict var_dict;
    std::vector<double>& flat = var_dict[vehicle_lbl]["splines_seg0"];
    for (int k = 0; k < n_spl; ++k) flat.insert(flat.end(), init[k].begin(), init[k].end());
    std::fill(variables.begin(), variables.end(), 0.0);
    getVariableVector(variables, var_dict);
}

void Point2Point::setParameters(std::vector<obstacle_t>& obstacles) {
    Dict par_dict;
    fillParameterDict(obstacles, par_dict);
    getParameterVector(parameters, par_dict);
}

// `Point2Point.cpp:263-277` + the generated obstacle part: vehicle conditions, time since the last knot, obstacles
void Point2Point::fillParameterDict(std::vector<obstacle_t>& obstacles, Dict& par_dict) {
    vehicle->setParameters(par_dict[vehicle_lbl]);
    par_dict[p2p_lbl]["t"] = {std::fmod(std::round(current_time * 1000.) / 1000., horizon_time / vehicle->getKnotIntervals())};
    par_dict[p2p_lbl]["T"] = {horizon_time};
    for (int k = 0; k < n_obs && k < (int)obstacles.size(); ++k) {
        std::map<std::string, std::vector<double>>& d = par_dict[obstacle_lbl[k]];
        d["x"] = obstacles[k].position; d["v"] = obstacles[k].velocity; d["a"] = obstacles[k].acceleration;
        d["checkpoints"] = obstacles[k].checkpoints; d["rad"] = obstacles[k].radii;
    }
}

void Point2Point::getParameterVector(std::vector<double>& vec, Dict& dict) {
    for (const Block& b : blocks) {
        if (b.kind != OMGX_BLOCK_PAR || !dict.count(b.label) || !dict[b.label].count(b.name)) continue;
        const std::vector<double>& v = dict[b.label][b.name];
        for (int k = 0; k < b.rows * b.cols && k < (int)v.size(); ++k) vec[b.off + k] = v[k];
    }
}

void Point2Point::getVariableVector(std::vector<double>& vec, Dict& dict) {
    for (const Block& b : blocks) {
        if (b.kind != OMGX_BLOCK_VAR || !dict.count(b.label) || !dict[b.label].count(b.name)) continue;
        const std::vector<double>& v = dict[b.label][b.name];
        for (int k = 0; k < b.rows * b.cols && k < (int)v.size(); ++k) vec[b.off + k] = v[k];
    }
}

void Point2Point::getVariableDict(std::vector<double>& vec, Dict& dict) {
    for (const Block& b : blocks)
        if (b.kind == OMGX_BLOCK_VAR) dict[b.label][b.name].assign(vec.begin() + b.off, vec.begin() + b.off + b.rows * b.cols);
}

// an obstacle that is not to be avoided lifts the bounds of its constraints (`export/export.py:355-404` generates the same
// per obstacle); with every obstacle avoided the defaults of the template stand
void Point2Point::updateBounds(double, std::vector<obstacle_t>& obstacles) {
    lbg.assign(tpl->lbg_def, tpl->lbg_def + n_con);
    ubg.assign(tpl->ubg_def, tpl->ubg_def + n_con);
    for (int k = 0; k < n_obs && k < (int)obstacles.size(); ++k) {
        if (obstacles[k].avoid) continue;
        for (const Block& b : blocks) {
            // the obstacle's own rows, and the vehicle's rows against it (their names carry the obstacle's index)
            const bool own = b.kind == OMGX_BLOCK_CON && b.label == obstacle_lbl[k];
            if (own) for (int r = 0; r < b.rows * b.cols; ++r) { lbg[b.off + r] = -inf; ubg[b.off + r] = inf; }
        }
    }
}

void Point2Point::extractData() {
    Dict var_dict;
    getVariableDict(variables, var_dict);
    spline_coeffs_vec = var_dict[vehicle_lbl]["splines_seg0"];
    vehicle->setKnotHorizon(horizon_time);
    const int n_spl = vehicle->getNSplines(), L = vehicle->getLenBasis();
    std::vector<std::vector<double>> c(n_spl, std::vector<double>(L));
    for (int k = 0; k < n_spl; ++k) for (int j = 0; j < L; ++j) c[k][j] = spline_coeffs_vec[k * L + j];
    retrieveTrajectories(c);
}

void Point2Point::retrieveTrajectories(std::vector<std::vector<double>>& spline_coeffs) {
    std::vector<double> t(time);
    const double t_rel = std::fmod(std::round(current_time * 1000.) / 1000., horizon_time / vehicle->getKnotIntervals());
    for (double& v : t) v += t_rel;
    vehicle->splines2State(spline_coeffs, t, state_trajectory);
    vehicle->splines2Input(spline_coeffs, t, input_trajectory);
}

// `export/export.py:406-444` generates, per spline variable, the product with its shift matrix when the horizon start
// passes a knot; here one loop over the block table
void Point2Point::transformSplines(double now, double prev) {
    const double knot_time = horizon_time / vehicle->getKnotIntervals();
    const int interval_prev = (int)std::floor(std::round(prev * 1e6) / 1e6 / knot_time + 1e-9);
    const int interval_now = (int)std::floor(std::round(now * 1e6) / 1e6 / knot_time + 1e-9);
    if (interval_now <= interval_prev) return;
    const int K = vehicle->getKnotIntervals();
    for (const Block& b : blocks) {
        if (b.kind != OMGX_BLOCK_VAR || b.rows <= K) continue;
        const std::vector<double>& T = shift_T[b.rows - K];
        const int L = b.rows;
        for (int k = 0; k < b.cols; ++k) {
            std::vector<double> old(variables.begin() + b.off + k * L, variables.begin() + b.off + (k + 1) * L);
            for (int i = 0; i < L; ++i) {
                double v = 0.0;
                for (int m = 0; m < L; ++m) v += T[i * L + m] * old[m];
                variables[b.off + k * L + i] = v;
            }
        }
    }
}

}  // namespace omg

// C entry for the tests: the shift matrix the class computes for a basis (compared with the front end's)
extern "C" int omg_compat_shift_matrix(int degree, int knot_intervals, double* out) {
    const std::vector<double> T = omg::shiftOverKnot(degree, knot_intervals);
    std::memcpy(out, T.data(), T.size() * sizeof(double));
    return (int)T.size();
}
