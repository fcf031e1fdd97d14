// omg::Point2Point -- header-compatible with the reference's exported C++ class
// (`export/point2point/Point2Point.hpp:34-107`: obstacle_t, the three constructors, reset / resetTime / recover,
// update(condition0, conditionT, state_trajectory, input_trajectory, obstacles[, predict_shift]), getCoefficients,
// getLenBasis, n_dim, n_obs), backed by libomgx.so instead of CasADi + IPOPT.  Where the reference's exporter bakes
// the problem into generated code -- N_VAR / N_PAR / N_CON, LBG_DEF / UBG_DEF, the labels, the offsets of every entry,
// the spline transformations (`export/export.py:236-444`) -- this class reads a template file written once by the
// Python front end (`omgtools.backend.save_template`): sizes, default bounds and the block table come from there,
// the shift matrices are computed from the bases.  Template path: environment variable OMG_TEMPLATE, else the macro
// OMG_TEMPLATE_FILE, else "p2p.omgx"; solver tolerance: OMG_TOL (default 1e-3 = the exporter's TOL).
#ifndef OMG_COMPAT_POINT2POINT
#define OMG_COMPAT_POINT2POINT

#include <limits>
#include <map>
#include <string>
#include <vector>
#include "Vehicle.hpp"

struct omgx_template;
struct omgx_batch;

namespace omg {

const double inf = std::numeric_limits<double>::infinity();

typedef struct obstacle {
    vec_t position;
    vec_t velocity;
    vec_t acceleration;
    vec_t checkpoints;
    vec_t radii;
    vec_t traj_coeffs;     // (kept for datatype compatibility; not used by fixed-T problems)
    bool avoid;
} obstacle_t;
typedef std::vector<obstacle_t> obstacles_t;

class Point2Point {
  private:
    bool _recover;
    void readBlockTable();

  protected:
    // (protected rather than private: the ADMM classes of the export derive from this one and add entries of their own)
    omgx_template* tpl;
    omgx_batch* problem;
    bool solve(double, obstacles_t&);
    struct Block { std::string label, name; int kind, off, rows, cols; };
    std::vector<Block> blocks;
    std::string vehicle_lbl, p2p_lbl;
    std::vector<std::string> obstacle_lbl;
    std::map<int, vec_t> shift_T;          // spline degree -> shift matrix of its basis
    const Block* find(int kind, const std::string& label, const std::string& name) const;
    int last_iters = 0;

    Vehicle* vehicle;
    vec_t spline_coeffs_vec;
    double current_time = 0.0;
    double current_time_prev = 0.0;
    double horizon_time;
    double update_time;
    double sample_time;
    int trajectory_length;
    vec_t parameters;
    vec_t variables;
    vec_t multipliers;
    vec_t lbg;
    vec_t ubg;
    vec_t time;
    mat_t state_trajectory;
    mat_t input_trajectory;
    std::string solver_output;
    int n_var, n_par, n_con;
    const int freeT = 0;

    void setParameters(obstacles_t&);
    virtual void initVariables();          // (virtual here: RendezVous also initialises its free end point)
    void updateBounds(double, obstacles_t&);
    void retrieveTrajectories(mat_t&);
    void getParameterVector(vec_t&, dict_t&);
    void getVariableVector(vec_t&, dict_t&);
    void getVariableDict(vec_t&, dict_t&);
    void transformSplines(double, double);

    virtual void generateProblem();
    virtual void fillParameterDict(obstacles_t&, dict_t&);
    virtual void extractData();
    virtual void initialize();

  public:
    int n_dim;
    int n_obs;
    Point2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time);
    Point2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length);
    Point2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, bool initialize);
    virtual ~Point2Point();
    virtual void reset();
    virtual void resetTime();
    virtual void recover();
    bool update(vec_t& condition0, vec_t& conditionT, mat_t& state_trajectory, mat_t& input_trajectory, obstacles_t& obstacles);
    bool update(vec_t& condition0, vec_t& conditionT, mat_t& state_trajectory, mat_t& input_trajectory, obstacles_t& obstacles,
                int predict_shift);
    void getCoefficients(vec_t& coeffs);
    int getLenBasis();
    int getIterations() const { return last_iters; }        // (extension: interior-point iterations of the last update)
};

}  // namespace omg
#endif
