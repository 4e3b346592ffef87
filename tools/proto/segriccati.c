/* tools/proto/segriccati.c -- PROTOTYPE (round 6, CPU only): the KKT solve of the interior point method as S independent
 * Riccati segments plus a coarse recursion over the S-1 interface states, in place of the oracle's sequential
 * riccati_factor / riccati_solve.  Built by tools/proto/build.sh into scratch/proto/libgusto_oracle_seg.so from a
 * GENERATED copy of oracle/gusto_oracle.c in which the two functions are renamed *_seq and this file is appended; the
 * committed oracle is not touched and nothing here is mirrored into it.  Purpose: show, before any kernel work, that the
 * segmented solve leaves the interior point iterations where they are (tools/proto/compare.py).
 *
 * Stage k maps dy_{k-1} -> dy_k.  Segment j owns the stages lo_j .. hi_j; all but the last treat their end state
 * dy_{hi_j} = xi_j as a terminal equality with multiplier lam_j (the machinery the goal rows already use: Pi starts as I, no
 * E term, n columns); the last segment is the oracle's own recursion with the goal rows.  Per segment the backward pass
 * leaves  V_j(xi_in; lam) = 1/2 xi' Pf xi + xi' (pf + Pif lam) - 1/2 lam' Gdf lam + lam' th,  and the interfaces obey
 *     xi_j    = th_j + Pif_j' xi_{j-1} - Gdf_j lam_j            (end state of segment j)
 *     lam_j-1 = Pf_j xi_{j-1} + pf_j + Pif_j lam_j              (costate at its start)
 * Maximised over lam, segment j costs 1/2 w' Gdf^-1 w, w = xi_j - (th_j + Pif_j' xi_{j-1}): a coarse LQR stage with transition Pif',
 * a FREE n-dim control w weighted by Gdf^-1, and the cost-to-go Pt_{j+1}, pt_{j+1} of what follows.  Its stage solve
 *     Sig_j = (Gdf_j^-1 + Pt_{j+1})^-1  (two SPD inversions),  a = th_j + Pif_j' xi_{j-1},  w = -Sig_j (Pt_{j+1} a + pt_{j+1}),
 *     xi_j = a + w,  lam_j = -Gdf_j^-1 w,      Pt_j = Pf_j + Pif_j (Gdf_j^-1 Sig_j Pt_{j+1}) Pif_j',
 * never forms I + Gdf Pt (whose inverse loses lam = Pt xi + pt to cancellation once barrier weights ~1/mu make Pt stiff:
 * GO_SEG_FORM=1 keeps that first attempt for the record).                                                              */
#define SEG_MAXS 8
#define SEG_MAXN 256
typedef struct {
    int S, lo[SEG_MAXS], hi[SEG_MAXS];
    double Pf[SEG_MAXS][NX * NX], Pif[SEG_MAXS][NX * NX], Gdf[SEG_MAXS][NX * NX];
    double Pt[SEG_MAXS][NX * NX], Ginv[SEG_MAXS][NX * NX], GdBinv[NX * NX], Gdinv[SEG_MAXS][NX * NX], Sig[SEG_MAXS][NX * NX], Pc[SEG_MAXS][NX * NX], Pic[SEG_MAXS][NX * NX], Gdc[SEG_MAXS][NX * NX], Ta[SEG_MAXS][NX * NX], Pa[SEG_MAXS][NX * NX];
    double pc[SEG_MAXS][NX], thc[SEG_MAXS][NX], lam0[SEG_MAXS][NX];
    double pf[SEG_MAXS][NX], th[SEG_MAXS][NX], pt[SEG_MAXS][NX], tt[SEG_MAXS][NX], xi[SEG_MAXS][NX], lam[SEG_MAXS][NX];
} seg_ws;
static __thread seg_ws SW;
static int seg_count(void) { const char* e = getenv("GO_SEG_S"); int s = e ? atoi(e) : 2; return s < 1 ? 1 : (s > SEG_MAXS ? SEG_MAXS : s); }
static int seg_nu_mode(void) { const char* e = getenv("GO_SEG_NU"); return e ? atoi(e) : 0; }       /* 1: interface costate from the coarse value function at the ACTUAL end state */
static int seg_form(void) { const char* e = getenv("GO_SEG_FORM"); return e ? atoi(e) : 0; }
static int seg_shift(void) { const char* e = getenv("GO_SEG_SHIFT"); return e ? atoi(e) : 1; }
#ifndef SEG_PIV
#define SEG_PIV 1e-13
#endif
/* lower Cholesky factor of a positive SEMI-definite matrix with floored pivots, and its inverse */
static int seg_chol_floor(double* L, double* Li, const double* A, int n) {
    double dmax = 0;
    for (int i = 0; i < n; i++) if (A[i * n + i] > dmax) dmax = A[i * n + i];
    if (!(dmax > 0)) return -1;
    const char* e = getenv("GO_SEG_PIV");
    const double flo = (e ? atof(e) : SEG_PIV) * dmax;
    memset(L, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int l = 0; l < j; l++) d -= L[j * n + l] * L[j * n + l];
        if (!(d > flo)) d = flo;
        d = sqrt(d);
        L[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int l = 0; l < j; l++) s -= L[i * n + l] * L[j * n + l];
            L[i * n + j] = s / d;
        }
    }
    memset(Li, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; j++) {
        Li[j * n + j] = 1.0 / L[j * n + j];
        for (int i = j + 1; i < n; i++) {
            double s = 0;
            for (int l = j; l < i; l++) s -= L[i * n + l] * Li[l * n + j];
            Li[i * n + j] = s / L[i * n + i];
        }
    }
    return 0;
}
/* Gauss-Jordan inverse WITHOUT pivoting (what a row-per-lane kernel version would like to do) */
static int seg_inv_nopiv(double* Ainv, const double* A, int n) {
    double W[NX * 2 * NX];
    int w = 2 * n;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) { W[i * w + j] = A[i * n + j]; W[i * w + n + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < n; c++) {
        if (W[c * w + c] == 0.0) return -1;
        double d = 1.0 / W[c * w + c];
        for (int j = 0; j < w; j++) W[c * w + j] *= d;
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            double f = W[r * w + c];
            for (int j = 0; j < w; j++) W[r * w + j] -= f * W[c * w + j];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ainv[i * n + j] = W[i * w + n + j];
    return 0;
}
static int seg_fwd_mode(void) { const char* e = getenv("GO_SEG_FWD"); return e ? atoi(e) : 0; }     /* 1: a segment starts from the actual end state of the one before (sequential forward pass) */

/* the oracle's stage recursion over lo..hi; last != 0: goal rows (ngc = ng, E term at knot N-1), else Pi starts as I (ngc = n) */
static int seg_factor_range(go_problem* p, int lo, int hi, int last, int ng, const int* gidx, double* Pf, double* Pif, double* Gdf) {
    const int n = p->n, m = p->m, N = p->N, nz = n + m, ngc = last ? ng : n;
    double P[NX * NX], Pi[NX * NX], T[NX * NZ], PG[NX * NZ], Hh[NZ * NZ], Z[NZ * NX];
    double Qt[NX * NX], Qb[NX * NU], tmp[NX * NX], S[NU * NU];
    memset(P, 0, sizeof(P));
    memset(Pi, 0, sizeof(Pi));
    memset(Gdf, 0, sizeof(double) * NX * NX);
    if (!last) for (int i = 0; i < n; i++) Pi[i * ngc + i] = 1.0;
    for (int k = hi; k >= lo; k--) {
        const double *Phi = p->Phi + k * n * n, *Gam = p->Gam + k * n * m, *M = p->Mm + k * n * n, *b = p->bm + k * n * m;
        double* QQ = p->QQ + k * nz * nz;
        memcpy(p->Ps + k * n * n, P, sizeof(double) * n * n);
        memcpy(p->Pis + k * n * n, Pi, sizeof(double) * n * ngc);
        memset(QQ, 0, sizeof(double) * nz * nz);
        if (k > 0) {
            mm(tmp, p->Hx + k * n * n, M, n, n, n);
            mtm(Qt, M, tmp, n, n, n);
            mm(Qb, Qt, b, n, n, m);
            for (int i = 0; i < n; i++) {
                for (int j = 0; j < n; j++) QQ[i * nz + j] = Qt[i * n + j];
                for (int j = 0; j < m; j++) { QQ[i * nz + n + j] = Qb[i * m + j]; QQ[(n + j) * nz + i] = Qb[i * m + j]; }
            }
            mtm(S, b, Qb, m, n, m);
        } else {
            memset(S, 0, sizeof(S));
        }
        for (int i = 0; i < m; i++)
            for (int j = 0; j < m; j++) QQ[(n + i) * nz + n + j] = p->Hu[k * m * m + i * m + j] + S[i * m + j];
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < n; j++) PG[i * nz + j] = Phi[i * n + j];
            for (int j = 0; j < m; j++) PG[i * nz + n + j] = Gam[i * m + j];
        }
        mm(T, P, PG, n, n, nz);
        mtm(Hh, PG, T, nz, n, nz);
        for (int i = 0; i < nz * nz; i++) Hh[i] += QQ[i];
        mtm(Z, PG, Pi, nz, n, ngc);
        if (last && k == N - 1 && ng > 0) {
            for (int j = 0; j < ng; j++) {
                double ey[NX];
                for (int i = 0; i < n; i++) { ey[i] = M[gidx[j] * n + i]; Z[i * ngc + j] += ey[i]; }
                for (int i = 0; i < m; i++) {
                    double s = 0;
                    for (int l = 0; l < n; l++) s += b[l * m + i] * ey[l];
                    Z[(n + i) * ngc + j] += s;
                }
            }
        }
        for (int i = 0; i < m; i++)
            for (int j = 0; j < m; j++) S[i * m + j] = 0.5 * (Hh[(n + i) * nz + n + j] + Hh[(n + j) * nz + n + i]);
        double* Sinv = p->Sinv + k * m * m;
        double Li[NU * NU], Wm[NU * NX], Vm[NU * NX];
        if (chol_inv(Sinv, Li, S, m)) { if (getenv("GO_DEBUG")) fprintf(stderr, "seg: S not PD at k=%d (range %d..%d)\n", k, lo, hi); return -1; }
        for (int i = 0; i < m; i++) {
            for (int j = 0; j < n; j++) {
                double s = 0;
                for (int l = 0; l <= i; l++) s += Li[i * m + l] * Hh[j * nz + n + l];
                Wm[i * n + j] = s;
            }
            for (int j = 0; j < ngc; j++) {
                double s = 0;
                for (int l = 0; l <= i; l++) s += Li[i * m + l] * Z[(n + l) * ngc + j];
                Vm[i * ngc + j] = s;
            }
        }
        double* K = p->Ks + k * m * n, *D = p->Ds + k * m * n;
        for (int i = 0; i < m; i++) {
            for (int j = 0; j < n; j++) {
                double s = 0;
                for (int l = i; l < m; l++) s += Li[l * m + i] * Wm[l * n + j];
                K[i * n + j] = s;
            }
            for (int j = 0; j < ngc; j++) {
                double s = 0;
                for (int l = i; l < m; l++) s += Li[l * m + i] * Vm[l * ngc + j];
                D[i * ngc + j] = s;
            }
        }
        for (int i = 0; i < ngc; i++)
            for (int j = 0; j < ngc; j++) {
                double s = 0;
                for (int l = 0; l < m; l++) s += Vm[l * ngc + i] * Vm[l * ngc + j];
                Gdf[i * ngc + j] += s;
            }
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < n; j++) {
                double s = 0.5 * (Hh[i * nz + j] + Hh[j * nz + i]);
                for (int l = 0; l < m; l++) s -= Wm[l * n + i] * Wm[l * n + j];
                P[i * n + j] = s;
            }
            for (int j = 0; j < ngc; j++) {
                double s = Z[i * ngc + j];
                for (int l = 0; l < m; l++) s -= Wm[l * n + i] * Vm[l * ngc + j];
                Pi[i * ngc + j] = s;
            }
        }
    }
    memcpy(Pf, P, sizeof(double) * n * n);
    memcpy(Pif, Pi, sizeof(double) * n * ngc);
    return 0;
}

static void seg_check(go_problem* p, int ng, const int* gidx, const double* rg);
static int riccati_factor(go_problem* p, int ng, const int* gidx) {
    const int n = p->n, N = p->N;
    seg_ws* w = &SW;
    int S = seg_count();
    if (S > N / 2) S = N / 2;
    w->S = S;
    for (int j = 0; j < S; j++) { w->lo[j] = (int)((long)N * j / S); w->hi[j] = (int)((long)N * (j + 1) / S) - 1; }
    for (int j = 0; j < S; j++)
        if (seg_factor_range(p, w->lo[j], w->hi[j], j == S - 1, ng, gidx, w->Pf[j], w->Pif[j], w->Gdf[j])) return -1;
    /* coarse backward recursion over the interfaces: the same Riccati recursion with goal sensitivities, one stage per
     * segment j < S-1 -- transition Pif_j', a free n-dim control w weighted by Gdf_j^-1 -- carrying (Pc, Pic, Gdc) */
    const int form = seg_form();
    memcpy(w->Pc[S - 1], w->Pf[S - 1], sizeof(double) * n * n);
    memcpy(w->Pic[S - 1], w->Pif[S - 1], sizeof(double) * n * (ng > 0 ? ng : 1));
    memcpy(w->Gdc[S - 1], w->Gdf[S - 1], sizeof(double) * NX * NX);
    for (int j = S - 2; j >= 0; j--) {
        double H[NX * NX], t1[NX * NX], t2[NX * NX];
        const double *Pc = w->Pc[j + 1], *Pic = w->Pic[j + 1];
        /* Gdf_j = G G' (Cholesky; a segment with an uncontrollable direction -- the quaternion norm of the manifold model -- has a
         * singular Gdf: pivots are floored at SEG_PIV x the largest diagonal entry, i.e. the rigid direction gets a compliance
         * far below anything a barrier weight can resolve), M = I + G' Pc G (SPD, eigenvalues >= 1), and with them
         *   Sig = (Gdf^-1 + Pc)^-1 = G M^-1 G',   Ta = (I + Pc Gdf)^-1 = G^-T M^-1 G',   Pa = Ta Pc
         * -- products only: Gdf^-1 itself (pivots squared) is never formed */
        double G[NX * NX], Gi[NX * NX], Mi[NX * NX];
        if (form == 2) {   /* GO_SEG_FORM=2: Ta = (I + Pc Gd)^-1 by Gauss-Jordan with partial pivoting, Sig = Gd Ta, Pa = Ta Pc: no factor of Gd at all */
            mm(H, Pc, w->Gdf[j], n, n, n);
            for (int i = 0; i < n; i++) H[i * n + i] += 1.0;
            if (getenv("GO_SEG_NOPIV") ? seg_inv_nopiv(w->Ta[j], H, n) : inv_gj(w->Ta[j], H, n)) return -3;
            mm(w->Sig[j], w->Gdf[j], w->Ta[j], n, n, n);
            for (int i = 0; i < n; i++) for (int c = 0; c < i; c++) { const double a = 0.5 * (w->Sig[j][i * n + c] + w->Sig[j][c * n + i]); w->Sig[j][i * n + c] = a; w->Sig[j][c * n + i] = a; }
            mm(w->Pa[j], w->Ta[j], Pc, n, n, n);
        } else {
        if (seg_chol_floor(G, Gi, w->Gdf[j], n)) { if (getenv("GO_DEBUG")) fprintf(stderr, "seg: Gd of segment %d not PSD\n", j); return -3; }
        mm(t1, Pc, G, n, n, n);
        mtm(H, G, t1, n, n, n);
        for (int i = 0; i < n; i++) H[i * n + i] += 1.0;
        for (int i = 0; i < n; i++) for (int c = 0; c < i; c++) { const double a = 0.5 * (H[i * n + c] + H[c * n + i]); H[i * n + c] = a; H[c * n + i] = a; }
        if (inv_spd(Mi, H, n)) { if (getenv("GO_DEBUG")) fprintf(stderr, "seg: I + G' Pc G of segment %d not PD\n", j); return -4; }
        mm(t1, Mi, Gi, n, n, n);                 /* M^-1 G^-1 */
        mm(t2, G, t1, n, n, n);                  /* Ta' = G M^-1 G^-1 */
        for (int i = 0; i < n; i++) for (int c = 0; c < n; c++) w->Ta[j][i * n + c] = t2[c * n + i];
        for (int i = 0; i < n; i++) for (int c = 0; c < n; c++) t1[i * n + c] = G[c * n + i];   /* G' */
        mm(t2, Mi, t1, n, n, n);
        mm(w->Sig[j], G, t2, n, n, n);           /* Sig = G M^-1 G' */
        mm(w->Pa[j], w->Ta[j], Pc, n, n, n);
        }
        for (int i = 0; i < n; i++)
            for (int c = 0; c < i; c++) { const double a = 0.5 * (w->Pa[j][i * n + c] + w->Pa[j][c * n + i]); w->Pa[j][i * n + c] = a; w->Pa[j][c * n + i] = a; }
        /* Gdc_j = Gdc + Pic' Sig Pic */
        mm(t1, w->Sig[j], Pic, n, n, ng);
        for (int i = 0; i < ng; i++)
            for (int c = 0; c < ng; c++) {
                double sacc = w->Gdc[j + 1][i * ng + c];
                for (int l = 0; l < n; l++) sacc += Pic[l * ng + i] * t1[l * ng + c];
                w->Gdc[j][i * ng + c] = sacc;
            }
        /* Pc_j = Pf_j + Pif_j Pa Pif_j',  Pic_j = Pif_j Ta Pic */
        mm(t1, w->Pif[j], w->Pa[j], n, n, n);
        for (int i = 0; i < n; i++)
            for (int c = 0; c < n; c++) {
                double sacc = w->Pf[j][i * n + c];
                for (int l = 0; l < n; l++) sacc += t1[i * n + l] * w->Pif[j][c * n + l];
                w->Pc[j][i * n + c] = sacc;
            }
        mm(t1, w->Ta[j], Pic, n, n, ng);
        mm(w->Pic[j], w->Pif[j], t1, n, n, ng);
        (void)t2; (void)form;
    }
    if (ng > 0) {
        if (inv_spd(w->GdBinv, w->Gdc[0], ng)) { if (getenv("GO_DEBUG")) fprintf(stderr, "seg: coarse Gd not PD\n"); return -2; }
        memcpy(p->Gd, w->GdBinv, sizeof(double) * ng * ng);
    }
    return 0;
}

static void riccati_solve(go_problem* p, int ng, const int* gidx, const double* rg) {
    const int n = p->n, m = p->m, N = p->N, nz = n + m;
    seg_ws* w = &SW;
    const int S = w->S;
    /* backward vector sweeps, one per segment */
    for (int sg = 0; sg < S; sg++) {
        const int last = sg == S - 1, ngc = last ? ng : n;
        double pv[NX], tp[NX], l[NZ], gy[NX], t1[NX], th[NX], lu[NU];
        memset(pv, 0, sizeof(pv));
        memset(th, 0, sizeof(th));
        /* the interface multiplier is solved for as an INCREMENT on the current costate iterate lam0 = nu[hi + 1]: the segment's
         * backward pass starts from it, so that its feed-forward terms lu_k -- and with them th, the end state the segment would
         * reach -- vanish with the Newton step instead of cancelling against Gdf lam afterwards (GO_SEG_SHIFT=0: from zero) */
        if (!last && seg_shift()) memcpy(pv, p->nu + (w->hi[sg] + 1) * n, sizeof(double) * n);
        memcpy(w->lam0[sg], pv, sizeof(double) * n);
        for (int k = w->hi[sg]; k >= w->lo[sg]; k--) {
            const double *Phi = p->Phi + k * n * n, *Gam = p->Gam + k * n * m, *M = p->Mm + k * n * n, *b = p->bm + k * n * m;
            const double* QQ = p->QQ + k * nz * nz;
            const double* rd = p->rd + k * n;
            double* c = p->cc + k * n;
            memcpy(p->ps + k * n, pv, sizeof(double) * n);
            if (k > 0) {
                mv(c, Phi, rd, n, n);
                mtv(t1, M, p->gx + k * n, n, n);
                for (int i = 0; i < n; i++) {
                    double s = t1[i];
                    for (int j = 0; j < n; j++) s += QQ[i * nz + j] * rd[j];
                    gy[i] = s;
                }
            } else {
                memset(c, 0, sizeof(double) * n);
                memset(gy, 0, sizeof(gy));
            }
            for (int i = 0; i < n; i++) l[i] = gy[i];
            for (int i = 0; i < m; i++) {
                double s = p->gu[k * m + i];
                for (int j = 0; j < n; j++) s += b[j * m + i] * gy[j];
                l[n + i] = s;
            }
            mv(t1, p->Ps + k * n * n, c, n, n);
            for (int i = 0; i < n; i++) tp[i] = pv[i] + t1[i];
            for (int i = 0; i < n; i++) {
                double s = 0;
                for (int j = 0; j < n; j++) s += Phi[j * n + i] * tp[j];
                l[i] += s;
            }
            for (int i = 0; i < m; i++) {
                double s = 0;
                for (int j = 0; j < n; j++) s += Gam[j * m + i] * tp[j];
                l[n + i] += s;
                lu[i] = l[n + i];
            }
            const double *Pi = p->Pis + k * n * n, *D = p->Ds + k * m * n, *K = p->Ks + k * m * n;
            for (int j = 0; j < ngc; j++) {
                double s = 0;
                for (int i = 0; i < n; i++) s += Pi[i * ngc + j] * c[i];
                for (int i = 0; i < m; i++) s -= D[i * ngc + j] * lu[i];
                th[j] += s;
            }
            mv(p->d0s + k * m, p->Sinv + k * m * m, lu, m, m);
            for (int i = 0; i < n; i++) {
                double s = l[i];
                for (int j = 0; j < m; j++) s -= K[j * n + i] * lu[j];
                pv[i] = s;
            }
        }
        memcpy(w->pf[sg], pv, sizeof(double) * n);
        memcpy(w->th[sg], th, sizeof(double) * NX);
    }
    /* coarse backward: (pc, thc) of the last segment, then one coarse stage per segment */
    memcpy(w->pc[S - 1], w->pf[S - 1], sizeof(double) * n);
    memset(w->thc[S - 1], 0, sizeof(double) * NX);
    if (ng > 0) {
        const double* M = p->Mm + (N - 1) * n * n;
        for (int j = 0; j < ng; j++) {
            double s = w->th[S - 1][j] - rg[j];
            for (int i = 0; i < n; i++) s += M[gidx[j] * n + i] * p->rd[(N - 1) * n + i];
            w->thc[S - 1][j] = s;
        }
    }
    for (int j = S - 2; j >= 0; j--) {
        double t1[NX], t2[NX], t3[NX];
        double pch[NX];
        for (int i = 0; i < n; i++) pch[i] = w->pc[j + 1][i] - w->lam0[j][i];   /* cost-to-go gradient seen from lam0 */
        const double *pc = pch, *Pic = w->Pic[j + 1];
        /* thc_j = thc + Pic' (Ta' th_j - Sig pc);  pc_j = pf_j + Pif_j (Pa th_j + Ta pc) */
        mtv(t1, w->Ta[j], w->th[j], n, n);
        mv(t2, w->Sig[j], pc, n, n);
        for (int i = 0; i < n; i++) t1[i] -= t2[i];
        for (int c = 0; c < ng; c++) {
            double s = w->thc[j + 1][c];
            for (int l = 0; l < n; l++) s += Pic[l * ng + c] * t1[l];
            w->thc[j][c] = s;
        }
        mv(t1, w->Pa[j], w->th[j], n, n);
        mv(t2, w->Ta[j], pc, n, n);
        for (int i = 0; i < n; i++) t1[i] += t2[i];
        mv(t3, w->Pif[j], t1, n, n);
        for (int i = 0; i < n; i++) w->pc[j][i] = w->pf[j][i] + t3[i];   /* (pf_j was formed with lam0 at the segment's end) */
    }
    if (ng > 0) mv(p->mugn, w->GdBinv, w->thc[0], ng, ng);
    /* coarse forward: interface states and costates */
    double xprev[NX];
    memset(xprev, 0, sizeof(xprev));
    for (int j = 0; j + 1 < S; j++) {
        double t1[NX], a[NX], r[NX], wv[NX];
        mtv(t1, w->Pif[j], xprev, n, n);
        for (int i = 0; i < n; i++) a[i] = w->th[j][i] + t1[i];
        /* each output by the formula without a subtraction of large terms:
         *   xi  = Ta' a - Sig (pc + Pic mu)        (a soft segment against a stiff cost-to-go: |a| >> |xi|)
         *   lam = Ta (Pc a + pc + Pic mu)          (lam = Pc xi + pc + Pic mu would multiply the rounding of xi by Pc) */
        double q[NX];
        for (int i = 0; i < n; i++) {
            double s = w->pc[j + 1][i] - w->lam0[j][i];
            for (int c = 0; c < ng; c++) s += w->Pic[j + 1][i * ng + c] * p->mugn[c];
            q[i] = s;
        }
        mtv(t1, w->Ta[j], a, n, n);
        mv(wv, w->Sig[j], q, n, n);
        for (int i = 0; i < n; i++) w->xi[j][i] = t1[i] - wv[i];
        mv(t1, w->Pc[j + 1], a, n, n);
        for (int i = 0; i < n; i++) r[i] = t1[i] + q[i];
        mv(w->lam[j], w->Ta[j], r, n, n);                  /* (the increment on lam0) */
        memcpy(xprev, w->xi[j], sizeof(double) * n);
    }
    /* forward sweeps */
    const int nu_mode = seg_nu_mode(), fwd_mode = seg_fwd_mode();
    double dyend[NX];
    memset(dyend, 0, sizeof(dyend));
    for (int sg = 0; sg < S; sg++) {
        const int last = sg == S - 1, ngc = last ? ng : n;
        const double* mult = last ? p->mugn : w->lam[sg];
        double dy[NX], dyn[NX], du[NU], a[NX];
        if (sg == 0) memset(dy, 0, sizeof(dy));
        else memcpy(dy, fwd_mode ? dyend : w->xi[sg - 1], sizeof(double) * n);
        for (int k = w->lo[sg]; k <= w->hi[sg]; k++) {
            const double *Phi = p->Phi + k * n * n, *Gam = p->Gam + k * n * m, *M = p->Mm + k * n * n, *b = p->bm + k * n * m;
            const double *D = p->Ds + k * m * n, *K = p->Ks + k * m * n;
            for (int i = 0; i < m; i++) {
                double s = p->d0s[k * m + i];
                for (int j = 0; j < ngc; j++) s += D[i * ngc + j] * mult[j];
                for (int j = 0; j < n; j++) s += K[i * n + j] * dy[j];
                du[i] = -s;
                p->dU[k * m + i] = du[i];
            }
            if (k == 0) {
                memset(p->dX, 0, sizeof(double) * n);
            } else {
                for (int i = 0; i < n; i++) {
                    double s = dy[i] + p->rd[k * n + i];
                    for (int j = 0; j < m; j++) s += b[i * m + j] * du[j];
                    a[i] = s;
                }
                mv(p->dX + k * n, M, a, n, n);
            }
            for (int i = 0; i < n; i++) {
                double s = p->cc[k * n + i];
                for (int j = 0; j < n; j++) s += Phi[i * n + j] * dy[j];
                for (int j = 0; j < m; j++) s += Gam[i * m + j] * du[j];
                dyn[i] = s;
            }
            if (k + 1 < N) {
                if (k == w->hi[sg] && nu_mode) {   /* interface: the coarse value function at the actual end state */
                    double t1[NX];
                    mv(t1, w->Pc[sg + 1], dyn, n, n);
                    for (int i = 0; i < n; i++) {
                        double s = t1[i] + w->pc[sg + 1][i];
                        for (int c = 0; c < ng; c++) s += w->Pic[sg + 1][i * ng + c] * p->mugn[c];
                        p->nun[(k + 1) * n + i] = s;
                    }
                } else {
                    const double *P = p->Ps + k * n * n, *Pi = p->Pis + k * n * n;
                    for (int i = 0; i < n; i++) {
                        double s = p->ps[k * n + i];
                        for (int j = 0; j < n; j++) s += P[i * n + j] * dyn[j];
                        for (int j = 0; j < ngc; j++) s += Pi[i * ngc + j] * mult[j];
                        p->nun[(k + 1) * n + i] = s;
                    }
                }
            }
            memcpy(dy, dyn, sizeof(double) * n);
        }
        memcpy(dyend, dy, sizeof(double) * n);
    }
    for (int i = 0; i < n; i++) {
        double s = p->gx[i];
        if (N > 1)
            for (int j = 0; j < n; j++) s += p->Fm[j * n + i] * p->nun[n + j];
        p->nun[i] = -s;
    }
    { static __thread int busy = 0; if (!busy && getenv("GO_SEG_CHECK")) { busy = 1; seg_check(p, ng, gidx, rg); busy = 0; } }
}

/* GO_SEG_CHECK=1: every right-hand side is also solved by the oracle's sequential recursion and the two directions compared */
static void seg_check(go_problem* p, int ng, const int* gidx, const double* rg) {
    const int n = p->n, m = p->m, N = p->N;
    static __thread double dX[SEG_MAXN * NX], dU[SEG_MAXN * NU], nun[SEG_MAXN * NX], mg[NX];
    memcpy(dX, p->dX, sizeof(double) * n * N); memcpy(dU, p->dU, sizeof(double) * m * N); memcpy(nun, p->nun, sizeof(double) * n * N);
    memcpy(mg, p->mugn, sizeof(mg));
    seg_ws keep = SW;
    if (riccati_factor_seq(p, ng, gidx)) { fprintf(stderr, "check: sequential factor failed\n"); }
    riccati_solve_seq(p, ng, gidx, rg);
    double ex = 0, eu = 0, en = 0, sx = 0, su = 0, sn = 0; int kx = -1, ku = -1, kn = -1;
    for (int k = 0; k < N; k++) {
        for (int i = 0; i < n; i++) { double d = fabs(dX[k * n + i] - p->dX[k * n + i]); if (d > ex) { ex = d; kx = k; } if (fabs(p->dX[k * n + i]) > sx) sx = fabs(p->dX[k * n + i]);
                                      d = fabs(nun[k * n + i] - p->nun[k * n + i]); if (d > en) { en = d; kn = k; } if (fabs(p->nun[k * n + i]) > sn) sn = fabs(p->nun[k * n + i]); }
        for (int i = 0; i < m; i++) { double d = fabs(dU[k * m + i] - p->dU[k * m + i]); if (d > eu) { eu = d; ku = k; } if (fabs(p->dU[k * m + i]) > su) su = fabs(p->dU[k * m + i]); }
    }
    double pmax = 0, gmin = 1e300, gmax = 0;
    for (int i = 0; i < n; i++) { if (keep.Pc[1][i * n + i] > pmax) pmax = keep.Pc[1][i * n + i]; if (keep.Gdf[0][i * n + i] < gmin) gmin = keep.Gdf[0][i * n + i]; if (keep.Gdf[0][i * n + i] > gmax) gmax = keep.Gdf[0][i * n + i]; }
    fprintf(stderr, "check: dX %.2e/%.2e (k %d)  dU %.2e/%.2e (k %d)  nu %.2e/%.2e (k %d)   max diag Pt %.2e  diag GdA %.2e..%.2e\n", ex, sx, kx, eu, su, ku, en, sn, kn, pmax, gmin, gmax);
    /* leave the SEGMENTED direction in place */
    memcpy(p->dX, dX, sizeof(double) * n * N); memcpy(p->dU, dU, sizeof(double) * m * N); memcpy(p->nun, nun, sizeof(double) * n * N);
    memcpy(p->mugn, mg, sizeof(mg));
    SW = keep;
    riccati_factor(p, ng, gidx);   /* (the records of the segmented factorisation back in p) */
}
