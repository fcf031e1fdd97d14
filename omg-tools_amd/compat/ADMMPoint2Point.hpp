// omg::ADMMPoint2Point -- header-compatible with the reference's exported class
// (`export/point2point/admm/ADMMPoint2Point.hpp:27-69`: one object per vehicle; update1 = x-update of this vehicle, its
// shared variable x_i out; update2 = the neighbours' x_j in, z- and lambda-update, residuals, z_ij / l_ij out for the
// neighbours, who hand them back as z_ji / l_ji to the next update1), over libomgx.so.  The x-update NLP comes from a
// template file (OMG_TEMPLATE, written by `omgtools.backend.save_template` from `formation.build_updx_template`); what the
// reference's exporter generates as updz.so / updl.so / updres.so -- the closed-form consensus projection, the multiplier
// update and the residuals (`problems/admm.py:117-168, 407-466, 493-508`) -- is a table of the projector M and the
// knot transform F per time since the last knot (OMG_ADMM_TABLES, written by `omgtools.backend.save_admm_tables`) and
// three matrix-vector products here.
#ifndef OMG_COMPAT_ADMMPOINT2POINT
#define OMG_COMPAT_ADMMPOINT2POINT

#include <cstdint>
#include "Point2Point.hpp"

namespace omg {

class ADMMPoint2Point : public Point2Point {
  private:
    int iteration = 0;
    int init_iter;
    int32_t status = 1;                                   // of the previous x-update (1: the next one starts cold)
    double t_update = 0.0;                                // time of the last x-update (the z-update's)
    vec_t residuals;
    vec_t tab_t, tab_M, tab_F;              // [n_keys], [n_keys][na x na] each
    bool solveUpdx(double, obstacles_t&);
    bool solveUpdz();
    bool solveUpdl();
    bool computeResiduals();
    void initVariablesADMM();
    void loadTables();
    const double* table(const vec_t& tab) const;
    void transformSharedSplines(double, double);

  protected:
    double rho;
    int n_nghb = 0;
    std::string admm_lbl;
    std::map<std::string, vec_t> variables_admm;
    virtual void generateProblem();
    virtual void extractData();
    virtual void fillParameterDict(obstacles_t&, dict_t&);
    // the shared variable of this vehicle from its solution (`@retrieveSharedVariables@` of the exporter): the trajectory
    // splines themselves here, the fleet centre in FormationPoint2Point
    virtual void retrieveSharedVariables(dict_t&);
    virtual bool update1(vec_t& condition0, vec_t& conditionT, mat_t& state_trajectory, mat_t& input_trajectory, vec_t& x_i,
                         mat_t& z_ji, mat_t& l_ji, obstacles_t& obstacles);
    virtual bool update1(vec_t& condition0, vec_t& conditionT, mat_t& state_trajectory, mat_t& input_trajectory, vec_t& x_i,
                         mat_t& z_ji, mat_t& l_ji, obstacles_t& obstacles, int predict_shift);
    virtual bool update2(mat_t& x_j, mat_t& z_ij, mat_t& l_ij, vec_t& residuals);

  public:
    int n_shared = 0;
    ADMMPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time);
    ADMMPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length);
    ADMMPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, int init_iter);
    ADMMPoint2Point(Vehicle* vehicle, double update_time, double sample_time, double horizon_time, int trajectory_length, int init_iter, double rho);
    virtual void reset();
    virtual void resetTime();
    int getIteration();
    double getCurrentTime();
    void stepBack();
};

}  // namespace omg
#endif
