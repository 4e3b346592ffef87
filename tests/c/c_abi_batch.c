/* c_abi_batch.c -- the call sequences of the BATCH wrappers of gusto.jl_amd/julia/GuSTOHIPBatch.jl (solve_SCP_batch! with one
 * environment per problem and two shards gathered by gusto_gather_peer; solve_SCPshooting_batch!: gusto_set_active, one-trip
 * solves and gusto_shoot over the live problems; solve_trajopt_batch!: gusto_solve_trajopt_async + gusto_wait), replayed by a
 * plain C program -- Julia is not available in the build image, so this is the non-Python consumer that exercises those
 * entry points in that order.  Prints "ok" and a few numbers; every check failing returns a nonzero code.
 *   gcc -std=c99 -Iinclude tests/c/c_abi_batch.c -o c_abi_batch -L gusto.jl_amd -lgusto_hip -lm -Wl,-rpath,$PWD/gusto.jl_amd
 *   ./c_abi_batch boxes.txt                                                                                              */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gusto_hip.h"

#define CHECKH(hh, call) do { int rc_ = (call); if (rc_ != GUSTO_OK) { \
    fprintf(stderr, "%s -> %d: %s\n", #call, rc_, gusto_last_error(hh)); return 2; } } while (0)
#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "check failed: %s (line %d)\n", #c, __LINE__); return 3; } } while (0)

enum { N = 50, B = 6, NX = 6, NU = 3 };

int main(int argc, char** argv) {
    int n_box = 0;
    if (argc < 2) { fprintf(stderr, "usage: c_abi_batch boxes.txt\n"); return 1; }
    FILE* f = fopen(argv[1], "r");
    if (!f || fscanf(f, "%d", &n_box) != 1) return 1;
    double* boxes = (double*)malloc(sizeof(double) * 6 * (size_t)n_box);
    for (int i = 0; i < 6 * n_box; i++) if (fscanf(f, "%lf", &boxes[i]) != 1) return 1;
    fclose(f);
    /* six problems on a line of start points, the notebook goal; every problem brings its own environment: the first b + 4 boxes */
    double x0[B * NX] = {0}, goal[B * NX], tf[B];
    for (int b = 0; b < B; b++) {
        x0[b * NX] = 0.35 + 0.1 * b; x0[b * NX + 1] = 2.3 - 0.05 * b;
        const double g[NX] = {3.0, 0.5, 0.0, 0.05, -0.05, 0.0};
        memcpy(goal + b * NX, g, sizeof(g)); tf[b] = 200.0;
    }
    int nb[B], ns[B] = {0};
    size_t tot = 0;
    for (int b = 0; b < B; b++) { nb[b] = (b + 4 < n_box) ? b + 4 : n_box; tot += (size_t)nb[b]; }
    double* envs = (double*)malloc(sizeof(double) * 6 * tot);
    for (size_t at = 0, b = 0; b < B; b++) { memcpy(envs + 6 * at, boxes, sizeof(double) * 6 * (size_t)nb[b]); at += (size_t)nb[b]; }

    /* ---- solve_SCP_batch!: two shards (3 + 3) on one GPU, per-problem environments, asynchronous, device-side gather ------- */
    gusto_handle h[2] = {0, 0};
    size_t off = 0;
    for (int r = 0; r < 2; r++) {
        CHECKH(h[r], gusto_create(&h[r], GUSTO_FREEFLYER_SE2, N, 3, 96, 0));
        CHECKH(h[r], gusto_set_env(h[r], n_box, boxes, 0, 0));
        CHECKH(h[r], gusto_set_env_batch(h[r], 3, nb + 3 * r, envs + 6 * off, ns + 3 * r, 0));
        for (int b = 3 * r; b < 3 * r + 3; b++) off += (size_t)nb[b];
        CHECKH(h[r], gusto_set_problems(h[r], 3, x0 + 3 * r * NX, goal + 3 * r * NX, goal + 3 * r * NX, tf + 3 * r, 0, 0));
        CHECKH(h[r], gusto_solve_async(h[r], 30, 0));
    }
    static double Xall[B * N * NX], Uall[B * N * NU], X1[3 * N * NX], U1[3 * N * NU];
    int bt = 0;
    CHECKH(h[0], gusto_gather_peer(h[0], 2, h, 0, 0, Xall, Uall, &bt));
    REQUIRE(bt == B);
    int conv_total = 0;
    for (int r = 0; r < 2; r++) {
        int it[3], cv[3], su[3], st[3], ipm[3];
        CHECKH(h[r], gusto_wait(h[r]));
        CHECKH(h[r], gusto_get_traj(h[r], X1, U1));
        REQUIRE(memcmp(X1, Xall + 3 * r * N * NX, sizeof(X1)) == 0 && memcmp(U1, Uall + 3 * r * N * NU, sizeof(U1)) == 0);
        CHECKH(h[r], gusto_get_status(h[r], it, cv, su, st, ipm));
        for (int b = 0; b < 3; b++) conv_total += cv[b];
    }
    REQUIRE(conv_total >= 4);

    /* ---- solve_SCPshooting_batch!'s use of gusto_set_active: one-trip solves over a subset leave the rest untouched -------- */
    gusto_handle d = 0;
    enum { ND = 3, BD = 8, NDK = 30 };
    double xd[BD * ND], gd[BD * ND] = {0}, tfd[BD];
    for (int b = 0; b < BD; b++) { xd[b * ND] = -2.0 + 0.5 * b; xd[b * ND + 1] = 1.5 - 0.3 * b; xd[b * ND + 2] = 0.2 * b; tfd[b] = 10.0; }
    CHECKH(d, gusto_create(&d, GUSTO_DUBINS_CAR, NDK, BD, 96, 0));
    CHECKH(d, gusto_set_problems(d, BD, xd, gd, gd, tfd, 0, 0));
    CHECKH(d, gusto_solve(d, 1, 0));
    int it0[BD], it1[BD], cv[BD], su[BD], st[BD], ipm[BD], act[BD];
    CHECKH(d, gusto_get_status(d, it0, cv, su, st, ipm));
    for (int b = 0; b < BD; b++) act[b] = (b % 2 == 0) && st[b] != GUSTO_STOP_SUBPROBLEM_FAILED;
    CHECKH(d, gusto_set_active(d, act));
    gusto_shoot_opts so;
    CHECKH(d, gusto_default_shoot_opts(&so));
    REQUIRE(so.no_group_pass == 0);
    CHECKH(d, gusto_shoot(d, 0, &so));                       /* seeds = the SCP duals, on the device */
    int sst[BD];
    CHECKH(d, gusto_get_shoot(d, sst, 0, 0, 0, 0, 0));
    for (int b = 0; b < BD; b++) if (!act[b]) REQUIRE(sst[b] == 0);      /* inactive: reported :Diverged, not integrated */
    CHECKH(d, gusto_solve(d, 1, 0));
    CHECKH(d, gusto_get_status(d, it1, cv, su, st, ipm));
    for (int b = 0; b < BD; b++) REQUIRE(it1[b] == it0[b] + (act[b] ? 1 : 0));
    CHECKH(d, gusto_set_active(d, 0));

    /* ---- solve_trajopt_batch!: two TrajOpt handles, launches in flight together, then gusto_wait ---------------------------- */
    gusto_handle t[2] = {0, 0};
    gusto_trajopt_params tp;
    CHECKH(t[0], gusto_default_trajopt_params(GUSTO_FREEFLYER_SE2, &tp));
    const int cap = 2 * tp.max_penalty_iteration * tp.max_convex_iteration * tp.max_trust_iteration + 16;
    for (int r = 0; r < 2; r++) {
        CHECKH(t[r], gusto_create_trajopt(&t[r], GUSTO_FREEFLYER_SE2, N, 3, cap, 0));
        CHECKH(t[r], gusto_set_env(t[r], n_box, boxes, 0, 0));
        CHECKH(t[r], gusto_set_problems(t[r], 3, x0 + 3 * r * NX, goal + 3 * r * NX, goal + 3 * r * NX, tf + 3 * r, 0, 0));
    }
    for (int r = 0; r < 2; r++) CHECKH(t[r], gusto_solve_trajopt_async(t[r], 125));
    int solves = 0;
    for (int r = 0; r < 2; r++) {
        int it[3];
        CHECKH(t[r], gusto_wait(t[r]));
        CHECKH(t[r], gusto_get_status(t[r], it, 0, 0, 0, 0));
        for (int b = 0; b < 3; b++) solves += it[b];
        CHECKH(t[r], gusto_get_traj(t[r], X1, U1));
        for (int i = 0; i < 3 * N * NX; i++) REQUIRE(isfinite(X1[i]));
    }
    REQUIRE(solves >= 6);
    REQUIRE(gusto_solve_trajopt_async(h[0], 10) != GUSTO_OK);   /* a GuSTO handle is refused */
    REQUIRE(gusto_set_active(t[0], act) != GUSTO_OK);           /* ... and so is a TrajOpt handle by gusto_set_active */
    printf("ok %d %d\n", conv_total, solves);
    for (int r = 0; r < 2; r++) { gusto_destroy(h[r]); gusto_destroy(t[r]); }
    gusto_destroy(d);
    free(boxes); free(envs);
    return 0;
}
