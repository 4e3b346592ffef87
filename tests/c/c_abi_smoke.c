/* c_abi_smoke.c -- a consumer of libgusto_hip.so that is not Python: plain C through include/gusto_hip.h.
 * create -> set_env -> set_problems -> solve -> get_status / get_history / get_traj on the notebook problem
 * (examples/freeflyerSE2.ipynb cell 2: x_init (0.2, 2.4), goal (3.0, 0.5, 0, 0.05, -0.05, 0), tf = 200; N = 50 here),
 * printed as one line of numbers that tests/test_gpu_seam.py compares with the oracle's run of the same problem.
 *   gcc -std=c99 -Iinclude tests/c/c_abi_smoke.c -o c_abi_smoke -L gusto.jl_amd -lgusto_hip -Wl,-rpath,$PWD/gusto.jl_amd
 *   ./c_abi_smoke boxes.txt      (boxes.txt: n_box, then 6 doubles per box: min xyz, max xyz)                        */
#include <stdio.h>
#include <stdlib.h>
#include "gusto_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != GUSTO_OK) { \
    fprintf(stderr, "%s -> %d: %s\n", #call, rc_, gusto_last_error(h)); return 2; } } while (0)

int main(int argc, char** argv) {
    enum { N = 50, CAP = 96 };
    gusto_handle h = 0;
    int n_box = 0, n = 0, m = 0;
    double* boxes = 0;
    if (argc < 2) { fprintf(stderr, "usage: c_abi_smoke boxes.txt\n"); return 1; }
    FILE* f = fopen(argv[1], "r");
    if (!f || fscanf(f, "%d", &n_box) != 1) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    boxes = (double*)malloc(sizeof(double) * 6 * (size_t)n_box);
    for (int i = 0; i < 6 * n_box; i++)
        if (fscanf(f, "%lf", &boxes[i]) != 1) { fprintf(stderr, "short box table\n"); return 1; }
    fclose(f);

    CHECK(gusto_model_dims(GUSTO_FREEFLYER_SE2, &n, &m));
    CHECK(gusto_create(&h, GUSTO_FREEFLYER_SE2, N, 1, CAP, 0));
    gusto_scp_params sp;
    gusto_model_params mp;
    CHECK(gusto_default_params(GUSTO_FREEFLYER_SE2, &sp, &mp));
    CHECK(gusto_set_params(h, &sp, &mp));
    CHECK(gusto_set_env(h, n_box, boxes, 0, 0));
    const double x_init[6] = {0.2, 2.4, 0.0, 0.0, 0.0, 0.0}, goal[6] = {3.0, 0.5, 0.0, 0.05, -0.05, 0.0}, tf = 200.0;
    CHECK(gusto_set_problems(h, 1, x_init, goal, goal, &tf, 0, 0));          /* NULL X0/U0: init_traj_straightline */
    CHECK(gusto_solve(h, 30, 0));

    int iterations = 0, converged = 0, successful = 0, stop = 0, ipm = 0, cap = 0;
    CHECK(gusto_get_status(h, &iterations, &converged, &successful, &stop, &ipm));
    CHECK(gusto_get_hist_cap(h, &cap));
    if (cap != CAP) { fprintf(stderr, "hist_cap %d\n", cap); return 3; }
    int n_hist = 0, nJ = 0, n_rho = 0, accept[CAP], scp_status[CAP];
    double J_true[CAP], Delta[CAP], omega[CAP], conv[CAP];
    gusto_history hist = {0};
    hist.hist_cap = CAP; hist.n_hist = &n_hist; hist.nJ = &nJ; hist.n_rho = &n_rho;
    hist.J_true = J_true; hist.Delta = Delta; hist.omega = omega; hist.convergence_measure = conv;
    hist.accept_solution = accept; hist.scp_status = scp_status;
    CHECK(gusto_get_history(h, &hist));
    static double X[N * 6], U[N * 3], dual[6];
    CHECK(gusto_get_traj(h, X, U));
    CHECK(gusto_get_dual(h, dual));

    printf("%d %d %d %d %d %d %d %d", iterations, converged, successful, stop, ipm, n_hist, nJ, n_rho);
    for (int i = 0; i < nJ; i++) printf(" %.17g", J_true[i]);
    for (int i = 0; i < n_hist; i++) printf(" %.17g %.17g %.17g %d %d", Delta[i], omega[i], conv[i], accept[i], scp_status[i]);
    for (int i = 0; i < 6; i++) printf(" %.17g", X[(N - 1) * 6 + i]);
    for (int i = 0; i < 3; i++) printf(" %.17g", U[(N / 2) * 3 + i]);
    printf("\n");
    CHECK(gusto_destroy(h));
    free(boxes);
    return 0;
}
