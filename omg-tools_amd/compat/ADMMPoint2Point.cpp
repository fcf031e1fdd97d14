// omg::ADMMPoint2Point / omg::FormationPoint2Point over libomgx.so (see the headers).  The flow of update1 / update2
// follows the reference's exported class (`export/point2point/admm/ADMMPoint2Point.cpp:104-268`): update1 takes the
// neighbours' z_ji / l_ji, shifts the plan and the consensus splines on a knot crossing, predicts, solves the x-update and
// returns x_i; update2 takes the neighbours' x_j and does the z-update, the multiplier update and the residuals.
// One deliberate difference: the export code advances current_time at the end of update1 and then evaluates the
// z-update at that new time; the Python classes (`problems/admm.py:584-611`) use the time of the x-update for both, and so
// does this class (it is what the test compares with).
#include "ADMMPoint2Point.hpp"
#include "FormationPoint2Point.hpp"
#include "RendezVous.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include "../../include/omgx.h"

#ifndef OMG_INITITER
#define OMG_INITITER 5
#endif
#ifndef OMG_RHO
#define OMG_RHO 1.0
#endif

namespace omg {

typedef std::map<std::string, std::map<std::string, std::vector<double>>> Dict;

ADMMPoint2Point::ADMMPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time,
                                 int trajectory_length, int init_iter, double rho)
    : Point2Point(vehicle, update_time, sample_time, horizon_time, trajectory_length, false), residuals(3) {
    this->init_iter = init_iter;
    this->rho = rho;
    initialize();                       // (virtual: generateProblem below)
    for (const Block& b : blocks) if (b.kind == OMGX_BLOCK_PAR && b.name == "z_ji") admm_lbl = b.label;
    if (admm_lbl.empty()) throw std::runtime_error("omg::ADMMPoint2Point: the template has no consensus parameters (z_i, z_ji, l_i, l_ji, rho)");
    n_shared = find(OMGX_BLOCK_PAR, admm_lbl, "z_i")->rows;
    n_nghb = find(OMGX_BLOCK_PAR, admm_lbl, "z_ji")->rows / n_shared;
    for (const char* nm : {"x_i", "z_i", "l_i"}) variables_admm[nm].assign(n_shared, 0.0);
    for (const char* nm : {"x_j", "z_ij", "l_ij", "z_ji", "l_ji"}) variables_admm[nm].assign(n_nghb * n_shared, 0.0);
    loadTables();
}

ADMMPoint2Point::ADMMPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time)
    : ADMMPoint2Point(vehicle, update_time, sample_time, horizon_time, int(update_time / sample_time), OMG_INITITER, OMG_RHO) {}
ADMMPoint2Point::ADMMPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length)
    : ADMMPoint2Point(vehicle, update_time, sample_time, horizon_time, trajectory_length, OMG_INITITER, OMG_RHO) {}
ADMMPoint2Point::ADMMPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, int init_iter)
    : ADMMPoint2Point(vehicle, update_time, sample_time, horizon_time, trajectory_length, init_iter, OMG_RHO) {}

// the x-update NLP: consecutive x-updates are neighbouring problems and start from the previous one's primal-dual point
// (`ADMMPoint2Point.cpp:78-91`: ipopt.warm_start_init_point)
void ADMMPoint2Point::generateProblem() {
    Point2Point::generateProblem();
    omgx_options opt;
    omgx_default_options(&opt);
    // (x-updates are solved three digits tighter than ipopt.tol = 1e-3, like `FormationPoint2point.xupdate_tol` of the Python
    // classes: at 1e-3 the consensus keeps a residual velocity at the goal and the run never ends)
    opt.tol = getenv("OMG_TOL") ? atof(getenv("OMG_TOL")) : 1e-6;
    opt.max_iter = 500;
    opt.warm_start = 1;
    if (opt.tol < 1e-4) { opt.warm_z_cap = 0.0; opt.max_soc = 0; }          // (as `formation.FormationPoint2point` sets it for tight x-updates)
    if (omgx_batch_set_options(problem, &opt) != OMGX_OK) throw std::runtime_error(omgx_last_error());
}

// the z-update tables: "OMGXADM1", int32 {na, n_keys}, then per key: t_rel, M [na x na], F [na x na] (row-major)
void ADMMPoint2Point::loadTables() {
    const char* path = getenv("OMG_ADMM_TABLES") ? getenv("OMG_ADMM_TABLES") : "admm_tables.omgx";
    FILE* fp = fopen(path, "rb");
    char magic[8];
    int32_t hdr[2] = {0, 0};
    if (!fp || fread(magic, 1, 8, fp) != 8 || memcmp(magic, "OMGXADM1", 8) != 0 || fread(hdr, sizeof(int32_t), 2, fp) != 2)
        throw std::runtime_error(std::string("omg::ADMMPoint2Point: cannot read the z-update tables ") + path);
    const int na = hdr[0], nk = hdr[1];
    if (na != (1 + n_nghb) * n_shared || nk <= 0) { fclose(fp); throw std::runtime_error("omg::ADMMPoint2Point: z-update tables of another problem"); }
    tab_t.resize(nk); tab_M.resize((size_t)nk * na * na); tab_F.resize((size_t)nk * na * na);
    bool ok = true;
    for (int k = 0; k < nk && ok; ++k)
        ok = fread(&tab_t[k], sizeof(double), 1, fp) == 1 && fread(&tab_M[(size_t)k * na * na], sizeof(double), (size_t)na * na, fp) == (size_t)na * na &&
             fread(&tab_F[(size_t)k * na * na], sizeof(double), (size_t)na * na, fp) == (size_t)na * na;
    fclose(fp);
    if (!ok) throw std::runtime_error("omg::ADMMPoint2Point: truncated z-update tables");
}

const double* ADMMPoint2Point::table(const std::vector<double>& tab) const {
    const double knot_time = horizon_time / vehicle->getKnotIntervals();
    // the same rounding as the keys (omgtools/backend.py admm_table_keys): time since the last knot to 1e-6, a time on a knot is 0
    double t_rel = std::fmod(t_update, knot_time);
    if (t_rel < 0.) t_rel += knot_time;
    t_rel = std::round(t_rel * 1e6) / 1e6;
    if (knot_time - t_rel < 5e-7) t_rel = 0.;
    const int na = (1 + n_nghb) * n_shared;
    for (size_t k = 0; k < tab_t.size(); ++k) if (std::fabs(tab_t[k] - t_rel) < 1.5e-6) return &tab[k * na * na];
    char msg[256];
    snprintf(msg, sizeof(msg), "omg::ADMMPoint2Point: no z-update table for %.6f s since the last knot (%zu tables: written for another "
             "update_time / sample_time? -- omgtools.backend.save_admm_tables covers the multiples of one step)", t_rel, tab_t.size());
    throw std::runtime_error(msg);
}

void ADMMPoint2Point::reset() { Point2Point::reset(); }
void ADMMPoint2Point::resetTime() { Point2Point::resetTime(); iteration = 0; status = 1; }
void ADMMPoint2Point::stepBack() { iteration--; current_time = current_time_prev; }
int ADMMPoint2Point::getIteration() { return iteration; }
double ADMMPoint2Point::getCurrentTime() { return current_time; }

bool ADMMPoint2Point::update1(std::vector<double>& condition0, std::vector<double>& conditionT,
                              std::vector<std::vector<double>>& state_traj, std::vector<std::vector<double>>& input_traj,
                              std::vector<double>& x_var, std::vector<std::vector<double>>& z_ji_var,
                              std::vector<std::vector<double>>& l_ji_var, std::vector<obstacle_t>& obstacles) {
    return update1(condition0, conditionT, state_traj, input_traj, x_var, z_ji_var, l_ji_var, obstacles, 0);
}

bool ADMMPoint2Point::update1(std::vector<double>& condition0, std::vector<double>& conditionT,
                              std::vector<std::vector<double>>& state_traj, std::vector<std::vector<double>>& input_traj,
                              std::vector<double>& x_var, std::vector<std::vector<double>>& z_ji_var,
                              std::vector<std::vector<double>>& l_ji_var, std::vector<obstacle_t>& obstacles, int predict_shift) {
    if (iteration > 0)                                              // (before: the own initial guess, initVariablesADMM)
        for (int i = 0; i < n_nghb; ++i)
            for (int j = 0; j < n_shared; ++j) {
                variables_admm["z_ji"][i * n_shared + j] = z_ji_var[i][j];
                variables_admm["l_ji"][i * n_shared + j] = l_ji_var[i][j];
            }
    current_time += predict_shift * sample_time;
    transformSplines(current_time, current_time_prev);             // warm start of this update ...
    transformSharedSplines(current_time, current_time_prev);       // ... and the consensus splines (`admm.py:477-491`)
    vehicle->setTerminalConditions(conditionT);
    if (std::fabs(current_time) <= 1.e-6) vehicle->setInitialConditions(condition0);
    else vehicle->predict(condition0, this->state_trajectory, this->input_trajectory, update_time, sample_time, predict_shift);
    t_update = current_time;
    if (!solveUpdx(current_time, obstacles)) {
        current_time_prev = current_time;
        return false;                                               // the caller retries
    }
    extractData();
    for (int k = 0; k < trajectory_length; ++k) {
        for (size_t j = 0; j < state_traj[0].size(); ++j) state_traj[k][j] = this->state_trajectory[k][j];
        for (size_t j = 0; j < input_traj[0].size(); ++j) input_traj[k][j] = this->input_trajectory[k][j];
    }
    x_var = variables_admm["x_i"];
    if (iteration >= init_iter) {                                   // the first init_iter iterations stay at the start time
        current_time_prev = current_time;
        current_time += update_time;
    } else {
        current_time_prev = current_time;
    }
    iteration++;
    return true;
}

bool ADMMPoint2Point::update2(std::vector<std::vector<double>>& x_j_var, std::vector<std::vector<double>>& z_ij_var,
                              std::vector<std::vector<double>>& l_ij_var, std::vector<double>& res) {
    for (int i = 0; i < n_nghb; ++i)
        for (int j = 0; j < n_shared; ++j) variables_admm["x_j"][i * n_shared + j] = x_j_var[i][j];
    solveUpdz();
    solveUpdl();
    computeResiduals();
    for (int i = 0; i < n_nghb; ++i)
        for (int j = 0; j < n_shared; ++j) {
            z_ij_var[i][j] = variables_admm["z_ij"][i * n_shared + j];
            l_ij_var[i][j] = variables_admm["l_ij"][i * n_shared + j];
        }
    res = residuals;
    return true;
}

bool ADMMPoint2Point::solveUpdx(double now, std::vector<obstacle_t>& obstacles) {
    Point2Point::setParameters(obstacles);                          // (before initVariablesADMM: the vehicle's conditions)
    if (iteration == 0) {
        initVariables();                                            // (virtual: RendezVous adds its free end point)
        initVariablesADMM();
        Point2Point::setParameters(obstacles);                      // again, with the initial consensus variables
    }
    Point2Point::updateBounds(now, obstacles);
    if (const char* dir = getenv("OMG_DEBUG_DUMP")) {               // (developer: the inputs of this x-update)
        char path[512];
        snprintf(path, sizeof path, "%s/updx_%p_%03d.bin", dir, (void*)this, iteration);
        if (FILE* fd = fopen(path, "wb")) {
            fwrite(parameters.data(), 8, parameters.size(), fd); fwrite(variables.data(), 8, variables.size(), fd);
            fwrite(multipliers.data(), 8, multipliers.size(), fd); fclose(fd);
        }
    }
    int32_t iters = 0;
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

// z_all = M (x_all + l_all / rho): the closed form of the equality-constrained least-squares problem `admm.py:117-168`
bool ADMMPoint2Point::solveUpdz() {
    variables_admm["z_i_p"] = variables_admm["z_i"];
    variables_admm["z_ij_p"] = variables_admm["z_ij"];
    const int ns = n_shared, na = (1 + n_nghb) * ns;
    const double* M = table(tab_M);
    std::vector<double> va(na);
    for (int q = 0; q < ns; ++q) va[q] = variables_admm["x_i"][q] + variables_admm["l_i"][q] / rho;
    for (int q = 0; q < n_nghb * ns; ++q) va[ns + q] = variables_admm["x_j"][q] + variables_admm["l_ij"][q] / rho;
    for (int r = 0; r < na; ++r) {
        double z = 0.0;
        for (int c = 0; c < na; ++c) z += M[(size_t)r * na + c] * va[c];
        (r < ns ? variables_admm["z_i"][r] : variables_admm["z_ij"][r - ns]) = z;
    }
    return true;
}

// l += rho (x - z)  (`admm.py:447-466`)
bool ADMMPoint2Point::solveUpdl() {
    const int ns = n_shared;
    for (int q = 0; q < ns; ++q) variables_admm["l_i"][q] += rho * (variables_admm["x_i"][q] - variables_admm["z_i"][q]);
    for (int q = 0; q < n_nghb * ns; ++q) variables_admm["l_ij"][q] += rho * (variables_admm["x_j"][q] - variables_admm["z_ij"][q]);
    return true;
}

// primal residual |F (x - z)|^2, dual residual rho |F (z - z_prev)|^2, combined rho pr + dr (`admm.py:493-508`)
bool ADMMPoint2Point::computeResiduals() {
    const int ns = n_shared, na = (1 + n_nghb) * ns;
    const double* F = table(tab_F);
    std::vector<double> d1(na), d2(na);
    for (int q = 0; q < ns; ++q) {
        d1[q] = variables_admm["x_i"][q] - variables_admm["z_i"][q];
        d2[q] = variables_admm["z_i"][q] - variables_admm["z_i_p"][q];
    }
    for (int q = 0; q < n_nghb * ns; ++q) {
        d1[ns + q] = variables_admm["x_j"][q] - variables_admm["z_ij"][q];
        d2[ns + q] = variables_admm["z_ij"][q] - variables_admm["z_ij_p"][q];
    }
    double pr = 0.0, dr = 0.0;
    for (int r = 0; r < na; ++r) {
        double a1 = 0.0, a2 = 0.0;
        for (int c = 0; c < na; ++c) { a1 += F[(size_t)r * na + c] * d1[c]; a2 += F[(size_t)r * na + c] * d2[c]; }
        pr += a1 * a1; dr += a2 * a2;
    }
    dr *= rho;
    residuals[0] = pr; residuals[1] = dr; residuals[2] = rho * pr + dr;
    return true;
}

void ADMMPoint2Point::initVariablesADMM() {
    Dict var_dict;
    getVariableDict(variables, var_dict);
    retrieveSharedVariables(var_dict);                              // x_i of the initial guess
    variables_admm["z_i"] = variables_admm["x_i"];
    for (int i = 0; i < n_nghb; ++i)
        for (int j = 0; j < n_shared; ++j) variables_admm["z_ji"][i * n_shared + j] = variables_admm["x_i"][j];
}

void ADMMPoint2Point::fillParameterDict(std::vector<obstacle_t>& obstacles, Dict& par_dict) {
    Point2Point::fillParameterDict(obstacles, par_dict);
    par_dict[admm_lbl]["z_i"] = variables_admm["z_i"];
    par_dict[admm_lbl]["z_ji"] = variables_admm["z_ji"];
    par_dict[admm_lbl]["l_i"] = variables_admm["l_i"];
    par_dict[admm_lbl]["l_ji"] = variables_admm["l_ji"];
    par_dict[admm_lbl]["rho"] = {rho};
}

void ADMMPoint2Point::extractData() {
    Point2Point::extractData();
    Dict var_dict;
    getVariableDict(variables, var_dict);
    retrieveSharedVariables(var_dict);
}

void ADMMPoint2Point::retrieveSharedVariables(Dict& var_dict) {
    variables_admm["x_i"] = var_dict[vehicle_lbl]["splines_seg0"];
}

// on a knot crossing the consensus splines move with the horizon like the plan itself (`admm.py:477-491`)
void ADMMPoint2Point::transformSharedSplines(double now, double prev) {
    const double knot_time = horizon_time / vehicle->getKnotIntervals();
    const int interval_prev = (int)std::floor(std::round(prev * 1e6) / 1e6 / knot_time + 1e-9);
    const int interval_now = (int)std::floor(std::round(now * 1e6) / 1e6 / knot_time + 1e-9);
    if (interval_now <= interval_prev) return;
    const int L = vehicle->getLenBasis();
    if (n_shared % L != 0) return;                                  // (a shared vector that is no spline: RendezVous)
    const std::vector<double>& T = shift_T[vehicle->getDegree()];
    for (const char* nm : {"z_i", "l_i", "z_ji", "l_ji", "z_ij", "l_ij"}) {
        std::vector<double>& v = variables_admm[nm];
        for (size_t off = 0; off + L <= v.size(); off += L) {
            std::vector<double> old(v.begin() + off, v.begin() + off + L);
            for (int i = 0; i < L; ++i) {
                double acc = 0.0;
                for (int m = 0; m < L; ++m) acc += T[i * L + m] * old[m];
                v[off + i] = acc;
            }
        }
    }
}

// ---- FormationPoint2Point ----------------------------------------------------------------------------------------------
FormationPoint2Point::FormationPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time,
                                           int trajectory_length, int init_iter, double rho)
    : ADMMPoint2Point(vehicle, update_time, sample_time, horizon_time, trajectory_length, init_iter, rho), rel_pos_c(n_dim, 0.0) {}
FormationPoint2Point::FormationPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time)
    : FormationPoint2Point(vehicle, update_time, sample_time, horizon_time, int(update_time / sample_time), OMG_INITITER, OMG_RHO) {}
FormationPoint2Point::FormationPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length)
    : FormationPoint2Point(vehicle, update_time, sample_time, horizon_time, trajectory_length, OMG_INITITER, OMG_RHO) {}
FormationPoint2Point::FormationPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, int init_iter)
    : FormationPoint2Point(vehicle, update_time, sample_time, horizon_time, trajectory_length, init_iter, OMG_RHO) {}

bool FormationPoint2Point::update1(std::vector<double>& condition0, std::vector<double>& conditionT,
                                   std::vector<std::vector<double>>& state_traj, std::vector<std::vector<double>>& input_traj,
                                   std::vector<double>& x_var, std::vector<std::vector<double>>& z_ji_var,
                                   std::vector<std::vector<double>>& l_ji_var, std::vector<obstacle_t>& obstacles,
                                   std::vector<double>& rel_pos_c) {
    return update1(condition0, conditionT, state_traj, input_traj, x_var, z_ji_var, l_ji_var, obstacles, rel_pos_c, 0);
}

bool FormationPoint2Point::update1(std::vector<double>& condition0, std::vector<double>& conditionT,
                                   std::vector<std::vector<double>>& state_traj, std::vector<std::vector<double>>& input_traj,
                                   std::vector<double>& x_var, std::vector<std::vector<double>>& z_ji_var,
                                   std::vector<std::vector<double>>& l_ji_var, std::vector<obstacle_t>& obstacles,
                                   std::vector<double>& rel_pos_c, int predict_shift) {
    this->rel_pos_c = rel_pos_c;
    return ADMMPoint2Point::update1(condition0, conditionT, state_traj, input_traj, x_var, z_ji_var, l_ji_var, obstacles, predict_shift);
}

bool FormationPoint2Point::update2(std::vector<std::vector<double>>& x_j_var, std::vector<std::vector<double>>& z_ij_var,
                                   std::vector<std::vector<double>>& l_ij_var, std::vector<double>& res) {
    return ADMMPoint2Point::update2(x_j_var, z_ij_var, l_ij_var, res);
}

void FormationPoint2Point::fillParameterDict(std::vector<obstacle_t>& obstacles, Dict& par_dict) {
    ADMMPoint2Point::fillParameterDict(obstacles, par_dict);
    par_dict[vehicle_lbl]["rel_pos_c"] = rel_pos_c;
}

// the fleet centre this vehicle believes in: its trajectory + its position relative to the centre (`vehicle.py:234-240`)
void FormationPoint2Point::retrieveSharedVariables(Dict& var_dict) {
    const std::vector<double>& c = var_dict[vehicle_lbl]["splines_seg0"];
    const int L = vehicle->getLenBasis();
    std::vector<double>& x_i = variables_admm["x_i"];
    x_i.resize(c.size());
    for (size_t q = 0; q < c.size(); ++q) x_i[q] = c[q] + rel_pos_c[q / L];
}

// ---- RendezVous ------------------------------------------------------------------------------------------------------
RendezVous::RendezVous(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length,
                       int init_iter, double rho)
    : ADMMPoint2Point(vehicle, update_time, sample_time, horizon_time, trajectory_length, init_iter, rho), rel_pos_c(n_dim, 0.0) {
    for (const Block& b : blocks) if (b.kind == OMGX_BLOCK_VAR && b.name == "conT0") free_lbl = b.label;
    if (free_lbl.empty()) throw std::runtime_error("omg::RendezVous: the template has no free end point (conT0)");
}
RendezVous::RendezVous(Vehicle* vehicle, double update_time, double sample_time, double horizon_time)
    : RendezVous(vehicle, update_time, sample_time, horizon_time, int(update_time / sample_time), OMG_INITITER, OMG_RHO) {}
RendezVous::RendezVous(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length)
    : RendezVous(vehicle, update_time, sample_time, horizon_time, trajectory_length, OMG_INITITER, OMG_RHO) {}
RendezVous::RendezVous(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, int init_iter)
    : RendezVous(vehicle, update_time, sample_time, horizon_time, trajectory_length, init_iter, OMG_RHO) {}

bool RendezVous::update1(std::vector<double>& condition0, std::vector<double>& conditionT,
                         std::vector<std::vector<double>>& state_traj, std::vector<std::vector<double>>& input_traj,
                         std::vector<double>& x_var, std::vector<std::vector<double>>& z_ji_var,
                         std::vector<std::vector<double>>& l_ji_var, std::vector<obstacle_t>& obstacles, std::vector<double>& rel_pos_c) {
    return update1(condition0, conditionT, state_traj, input_traj, x_var, z_ji_var, l_ji_var, obstacles, rel_pos_c, 0);
}

bool RendezVous::update1(std::vector<double>& condition0, std::vector<double>& conditionT,
                         std::vector<std::vector<double>>& state_traj, std::vector<std::vector<double>>& input_traj,
                         std::vector<double>& x_var, std::vector<std::vector<double>>& z_ji_var,
                         std::vector<std::vector<double>>& l_ji_var, std::vector<obstacle_t>& obstacles,
                         std::vector<double>& rel_pos_c, int predict_shift) {
    this->rel_pos_c = rel_pos_c;
    return ADMMPoint2Point::update1(condition0, conditionT, state_traj, input_traj, x_var, z_ji_var, l_ji_var, obstacles, predict_shift);
}

bool RendezVous::update2(std::vector<std::vector<double>>& x_j_var, std::vector<std::vector<double>>& z_ij_var,
                         std::vector<std::vector<double>>& l_ij_var, std::vector<double>& res) {
    return ADMMPoint2Point::update2(x_j_var, z_ij_var, l_ij_var, res);
}

void RendezVous::fillParameterDict(std::vector<obstacle_t>& obstacles, Dict& par_dict) {
    ADMMPoint2Point::fillParameterDict(obstacles, par_dict);
    par_dict[vehicle_lbl]["rel_pos_c"] = rel_pos_c;
}

// the first guess of the free end point is the terminal condition the caller handed in (`point2point.py:391-399`)
void RendezVous::initVariables() {
    Point2Point::initVariables();
    std::map<std::string, std::vector<double>> veh;
    vehicle->setParameters(veh);
    Dict var_dict;
    var_dict[free_lbl]["conT0"] = veh["poseT"];
    getVariableVector(variables, var_dict);
}

// the meeting point this vehicle believes in: its free end point + its position relative to it
void RendezVous::retrieveSharedVariables(Dict& var_dict) {
    const std::vector<double>& c = var_dict[free_lbl]["conT0"];
    std::vector<double>& x_i = variables_admm["x_i"];
    x_i.resize(c.size());
    for (size_t q = 0; q < c.size(); ++q) x_i[q] = c[q] + rel_pos_c[q];
}

}  // namespace omg
