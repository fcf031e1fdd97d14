// Formation test of the compat layer: N omg::FormationPoint2Point objects (one per vehicle, as the reference's exported
// class is deployed: `export/point2point/admm/formation/FormationPoint2Point.hpp`) run the two-phase ADMM update in one
// process -- update1 of every vehicle, the x_i handed to the neighbours, update2 of every vehicle, the z_ij / l_ij handed
// back as z_ji / l_ji -- and write the shared variables and residuals of every iteration for the Python test to compare
// with the batched path.
//   formation <scenario.bin> <out.bin> [rendezvous]      (third argument: omg::RendezVous objects instead)
// scenario: int32 {N, n_nghb, n_iter, init_iter, n_obs}, double rho, then per vehicle start[2] goal[2] rel_pos_c[2], the
// neighbour table int32 [N][n_nghb], per obstacle pos[2] vel[2], int32 n_chk, checkpoints[2 n_chk], radii[n_chk].
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>
#include "FormationPoint2Point.hpp"
#include "RendezVous.hpp"
#include "Holonomic.hpp"

using namespace std;

template <class Problem>
static int run(int argc, char** argv) {
    if (argc < 3) { cerr << "usage: formation scenario.bin out.bin" << endl; return 2; }
    FILE* fp = fopen(argv[1], "rb");
    int32_t hdr[5];
    double rho;
    if (!fp || fread(hdr, sizeof(int32_t), 5, fp) != 5 || fread(&rho, sizeof(double), 1, fp) != 1) { cerr << "bad scenario" << endl; return 2; }
    const int N = hdr[0], nn = hdr[1], n_iter = hdr[2], init_iter = hdr[3], n_obs = hdr[4];
    vector<vector<double>> start(N, vector<double>(2)), goal(N, vector<double>(2)), rel(N, vector<double>(2));
    for (int i = 0; i < N; ++i)
        if (fread(start[i].data(), 8, 2, fp) != 2 || fread(goal[i].data(), 8, 2, fp) != 2 || fread(rel[i].data(), 8, 2, fp) != 2) return 2;
    vector<int32_t> nbr(N * nn);
    if (fread(nbr.data(), sizeof(int32_t), N * nn, fp) != (size_t)(N * nn)) return 2;
    vector<omg::obstacle_t> obstacles(n_obs);
    for (int k = 0; k < n_obs; ++k) {
        omg::obstacle_t& o = obstacles[k];
        o.position.resize(2); o.velocity.resize(2); o.acceleration.assign(2, 0.0); o.avoid = true;
        int32_t nc;
        if (fread(o.position.data(), 8, 2, fp) != 2 || fread(o.velocity.data(), 8, 2, fp) != 2 || fread(&nc, 4, 1, fp) != 1) return 2;
        o.checkpoints.resize(2 * nc); o.radii.resize(nc);
        if (fread(o.checkpoints.data(), 8, 2 * nc, fp) != (size_t)(2 * nc) || fread(o.radii.data(), 8, nc, fp) != (size_t)nc) return 2;
    }
    fclose(fp);

    const double horizon_time = 10, sample_time = 0.01, update_time = 0.1;
    const int trajectory_length = 20;
    vector<omg::Holonomic*> vehicles(N);
    vector<Problem*> problems(N);
    for (int i = 0; i < N; ++i) {
        vehicles[i] = new omg::Holonomic();
        vehicles[i]->setIdealPrediction(true);
        problems[i] = new Problem(vehicles[i], update_time, sample_time, horizon_time, trajectory_length, init_iter, rho);
    }
    const int ns = problems[0]->n_shared;
    vector<vector<double>> x_var(N, vector<double>(ns));
    vector<vector<vector<double>>> z_ij(N, vector<vector<double>>(nn, vector<double>(ns))), l_ij(z_ij), z_ji(z_ij), l_ji(z_ij), x_j(z_ij);
    vector<vector<vector<double>>> st(N, vector<vector<double>>(trajectory_length, vector<double>(2))), in(st);
    vector<vector<double>> residuals(N, vector<double>(3));
    FILE* fo = fopen(argv[2], "wb");
    if (!fo) return 2;
    for (int it = 0; it < n_iter; ++it) {
        const double t_before = problems[0]->getCurrentTime();
        for (int i = 0; i < N; ++i)
            if (!problems[i]->update1(start[i], goal[i], st[i], in[i], x_var[i], z_ji[i], l_ji[i], obstacles, rel[i])) {
                cerr << "update1 of vehicle " << i << " failed in iteration " << it << endl; return 1;
            }
        for (int i = 0; i < N; ++i) for (int k = 0; k < nn; ++k) x_j[i][k] = x_var[nbr[i * nn + k]];      // communicate x
        for (int i = 0; i < N; ++i) problems[i]->update2(x_j[i], z_ij[i], l_ij[i], residuals[i]);
        // communicate z_ij / l_ij: what neighbour j keeps for vehicle i comes back as z_ji (`admm.py:468-475`)
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < nn; ++k) {
                const int j = nbr[i * nn + k];
                int slot = -1;
                for (int q = 0; q < nn; ++q) if (nbr[j * nn + q] == i) slot = q;
                if (slot < 0) { cerr << "the neighbour table is not symmetric" << endl; return 2; }
                z_ji[i][k] = z_ij[j][slot]; l_ji[i][k] = l_ij[j][slot];
            }
        for (int i = 0; i < N; ++i) { fwrite(x_var[i].data(), 8, ns, fo); fwrite(residuals[i].data(), 8, 3, fo); }
        // the world moves on when the problems' clocks did (after the init_iter iterations at the start time)
        if (problems[0]->getCurrentTime() > t_before + 1e-9)
            for (int k = 0; k < n_obs; ++k) for (int d = 0; d < 2; ++d) obstacles[k].position[d] += update_time * obstacles[k].velocity[d];
    }
    fclose(fo);
    cout << "ran " << n_iter << " ADMM iterations of " << N << " vehicles, time " << problems[0]->getCurrentTime() << endl;
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 3 && string(argv[3]) == "rendezvous") return run<omg::RendezVous>(argc, argv);
    return run<omg::FormationPoint2Point>(argc, argv);
}
