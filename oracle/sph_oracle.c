/*
 * oracle/sph_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, IEEE double) of the reference's per-step SPH hot path:
 *   - mex/sph_neighbor_search_mex.c  (cell linked list + periodic ghost entries -> flat pair list)
 *   - mex/sph_physics_mex.c          (8 string-dispatched physics modes on that pair list)
 *   - SPH_Poiseuille.m:246-302,519-577 (time loop, dt rule, periodic wrap, cell re-sort)
 * Every function cites the reference file:line it follows.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path (libsphx.so) never does.
 *
 * PARITY STATUS: "parity unpinned" in the strict sense -- the reference ships no tests, golden
 * vectors or fixtures for this path, and its two C files need MATLAB's mex.h (absent in this image)
 * so they are unbuildable here.  The restatement is anchored on (a) the analytic Poiseuille profile
 * and (b) the figures BASELINE.md section 2 records from the reference itself (pair counts on the
 * lattice, steps-to-20 s, L2 at 20 s); see tests/test_oracle_anchor.py and DESIGN.md.
 *
 * The pair loops keep the reference's serial pair order and expression order so that a
 * single-threaded run reproduces the reference's OMP_NUM_THREADS=1 arithmetic.  Build with
 * -ffp-contract=off.  With -fopenmp the pair loops use the same parallel-for + atomic scatter
 * scheme as the reference (sph_physics_mex.c:186-212 etc.); the neighbour search stays serial
 * exactly as in the reference (no pragma in sph_neighbor_search_mex.c).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_API __attribute__((visibility("default")))
#define ORC_EPS_REG 1e-8 /* sph_physics_mex.c:30 */

#ifdef _OPENMP
#define ORC_PARALLEL_FOR _Pragma("omp parallel for schedule(static)")
#define ORC_ATOMIC _Pragma("omp atomic")
#else
#define ORC_PARALLEL_FOR
#define ORC_ATOMIC
#endif

typedef struct {
    size_t count;
    size_t capacity;
    double *pair_i; /* 1-based, stored as double (sph_neighbor_search_mex.c:375-376) */
    double *pair_j;
    double *dx;
    double *dy;
    double *r;
    double *W;
    double *dW;
} orc_pairs;

static double now_seconds(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------ */
/* kernel: sph_neighbor_search_mex.c:116-133, sph_physics_mex.c:33-38                          */
/* ------------------------------------------------------------------------------------------ */
static void spline2d(double r, double h, double *W, double *dW)
{
    const double pi = 3.14159265358979323846;
    const double sigma = 10.0 / (7.0 * pi * h * h);
    const double q = r / h;
    if (q < 1.0) {
        *W = sigma * (1.0 - 1.5 * q * q + 0.75 * q * q * q);
        *dW = sigma * (-3.0 * q + 2.25 * q * q) / h;
    } else if (q < 2.0) {
        const double tq = 2.0 - q;
        *W = sigma * 0.25 * tq * tq * tq;
        *dW = -sigma * 0.75 * tq * tq / h;
    } else {
        *W = 0.0;
        *dW = 0.0;
    }
}

static double spline2d_w0(double h)
{
    const double pi = 3.14159265358979323846;
    return 10.0 / (7.0 * pi * h * h);
}

ORC_API void orc_kernel(double r, double h, double *W, double *dW) { spline2d(r, h, W, dW); }

/* ------------------------------------------------------------------------------------------ */
/* pair buffer (sph_neighbor_search_mex.c:34-44,136-183: initial 64*n_fluid+1024, doubling)     */
/* ------------------------------------------------------------------------------------------ */
static int pairs_reserve(orc_pairs *p, size_t cap)
{
    double **cols[7] = {&p->pair_i, &p->pair_j, &p->dx, &p->dy, &p->r, &p->W, &p->dW};
    for (int c = 0; c < 7; ++c) {
        double *q = (double *)realloc(*cols[c], (cap ? cap : 1) * sizeof(double));
        if (!q) return -1;
        *cols[c] = q;
    }
    p->capacity = cap;
    return 0;
}

ORC_API void orc_pairs_free(orc_pairs *p)
{
    if (!p) return;
    free(p->pair_i); free(p->pair_j); free(p->dx); free(p->dy); free(p->r); free(p->W); free(p->dW);
    memset(p, 0, sizeof(*p));
}

static int wrap_cell(int c, int n)
{ /* sph_neighbor_search_mex.c:57-67 */
    if (n <= 0) return 0;
    c %= n;
    if (c < 0) c += n;
    return c;
}

/*
 * orc_neighbor_search -- sph_neighbor_search_mex.c:185-421.
 * Returns 0 on success; negative = the reference's error ids in order of appearance
 * (-3 "SPH:Neighbor:pos" is a shape check done by the caller, -4 count, -5 param, -6 entryCapacity).
 * The cell structure is a CSR table filled in insertion order and walked backwards, which visits
 * entries in exactly the order of the reference's LIFO linked list (:100-101,:340-388).
 */
ORC_API int orc_neighbor_search(const double *pos, int n_fluid, int n_total, double h, double DL,
                                orc_pairs *out)
{
    if (n_total <= 0 || n_fluid <= 0 || n_fluid > n_total) return -4; /* :237-239 */
    if (h <= 0.0 || DL <= 0.0) return -5;                              /* :240-242 */
    const double *x = pos;
    const double *y = pos + n_total;

    double y_min = y[0], y_max = y[0]; /* :245-250 */
    for (int i = 1; i < n_total; ++i) {
        if (y[i] < y_min) y_min = y[i];
        if (y[i] > y_max) y_max = y[i];
    }
    const double cell_size = 2.0 * h; /* :253-258 */
    int n_cell_x = (int)ceil(DL / cell_size);
    if (n_cell_x < 1) n_cell_x = 1;
    int n_cell_y = (int)ceil((y_max - y_min + 1e-12) / cell_size) + 1;
    if (n_cell_y < 1) n_cell_y = 1;
    const int n_cells = n_cell_x * n_cell_y;
    const double cutoff = 2.0 * h; /* :264 */

    /* entries: up to 3 per particle (:265); record (cell, particle, x, y) in insertion order */
    const size_t cap = (size_t)n_total * 3;
    int *e_cell = (int *)malloc(cap * sizeof(int));
    int *e_part = (int *)malloc(cap * sizeof(int));
    double *e_x = (double *)malloc(cap * sizeof(double));
    double *e_y = (double *)malloc(cap * sizeof(double));
    int *cell_x = (int *)malloc((size_t)n_total * sizeof(int));
    int *cell_y = (int *)malloc((size_t)n_total * sizeof(int));
    int *seen = (int *)malloc((size_t)n_total * sizeof(int));
    int *start = (int *)calloc((size_t)n_cells + 1, sizeof(int));
    size_t n_e = 0;

    for (int i = 0; i < n_total; ++i) { /* :269-296 */
        const double xw = x[i] - floor(x[i] / DL) * DL;
        const double yi = y[i];
        const int cxi = wrap_cell((int)floor(xw / cell_size), n_cell_x);
        int cyi = (int)floor((yi - y_min) / cell_size);
        if (cyi < 0) cyi = 0;
        if (cyi >= n_cell_y) cyi = n_cell_y - 1;
        cell_x[i] = cxi;
        cell_y[i] = cyi;
        e_cell[n_e] = cyi * n_cell_x + cxi; e_part[n_e] = i; e_x[n_e] = xw; e_y[n_e] = yi; ++n_e;
        if (xw > DL - cutoff) {
            const double gx = xw - DL;
            const int gcx = wrap_cell((int)floor(gx / cell_size), n_cell_x);
            if (gcx != cxi) {
                e_cell[n_e] = cyi * n_cell_x + gcx; e_part[n_e] = i; e_x[n_e] = gx; e_y[n_e] = yi; ++n_e;
            }
        }
        if (xw < cutoff) {
            const double gx = xw + DL;
            const int gcx = wrap_cell((int)floor(gx / cell_size), n_cell_x);
            if (gcx != cxi) {
                e_cell[n_e] = cyi * n_cell_x + gcx; e_part[n_e] = i; e_x[n_e] = gx; e_y[n_e] = yi; ++n_e;
            }
        }
    }
    /* CSR by cell, entries of one cell kept in insertion order */
    for (size_t e = 0; e < n_e; ++e) start[e_cell[e] + 1]++;
    for (int c = 0; c < n_cells; ++c) start[c + 1] += start[c];
    int *cursor = (int *)malloc(((size_t)n_cells + 1) * sizeof(int));
    memcpy(cursor, start, ((size_t)n_cells + 1) * sizeof(int));
    int *order = (int *)malloc((n_e ? n_e : 1) * sizeof(int));
    for (size_t e = 0; e < n_e; ++e) order[cursor[e_cell[e]]++] = (int)e;

    for (int i = 0; i < n_total; ++i) seen[i] = -1; /* :298-300 */

    memset(out, 0, sizeof(*out));
    int rc = pairs_reserve(out, (size_t)n_fluid * 64 + 1024); /* :305 */
    const double r_cut_sq = (2.0 * h) * (2.0 * h);            /* :306 */

    for (int i = 0; i < n_fluid && rc == 0; ++i) { /* :312-392 */
        const int cxi = cell_x[i], cyi = cell_y[i];
        for (int oy = -1; oy <= 1; ++oy) {
            const int cy = cyi + oy;
            if (cy < 0 || cy >= n_cell_y) continue;
            for (int ox = -1; ox <= 1; ++ox) {
                int cx = cxi + ox;
                if (cx < 0) cx += n_cell_x; else if (cx >= n_cell_x) cx -= n_cell_x;
                if (cx < 0 || cx >= n_cell_x) continue;
                const int cid = cy * n_cell_x + cx;
                for (int k = start[cid + 1] - 1; k >= start[cid]; --k) { /* LIFO order */
                    const int e = order[k];
                    const int j = e_part[e];
                    if (j == i || seen[j] == i) continue;
                    if (j < n_fluid && j < i) continue; /* :353-355 */
                    const double xw = x[i] - floor(x[i] / DL) * DL; /* :350 */
                    double dxij = xw - e_x[e];
                    if (dxij > 0.5 * DL) dxij -= DL; else if (dxij < -0.5 * DL) dxij += DL;
                    const double dyij = y[i] - e_y[e];
                    const double r2 = dxij * dxij + dyij * dyij;
                    if (r2 > 1e-24 && r2 < r_cut_sq) { /* :368 */
                        const double rij = sqrt(r2);
                        double Wij, dWij;
                        spline2d(rij, h, &Wij, &dWij);
                        if (Wij > 0.0 || fabs(dWij) > 0.0) { /* :372 */
                            if (out->count >= out->capacity) {
                                size_t nc = out->capacity * 2;
                                if (nc < 1024) nc = 1024;
                                rc = pairs_reserve(out, nc);
                                if (rc) break;
                            }
                            const size_t c = out->count++;
                            out->pair_i[c] = (double)(i + 1);
                            out->pair_j[c] = (double)(j + 1);
                            out->dx[c] = dxij; out->dy[c] = dyij; out->r[c] = rij;
                            out->W[c] = Wij; out->dW[c] = dWij;
                            seen[j] = i; /* :383 */
                        }
                    }
                }
            }
        }
    }
    free(e_cell); free(e_part); free(e_x); free(e_y); free(cell_x); free(cell_y); free(seen);
    free(start); free(cursor); free(order);
    if (rc) { orc_pairs_free(out); return -6; }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* physics modes                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* sph_physics_mex.c:95-374 */
ORC_API void orc_density_correction(size_t n_pairs, const double *pair_i, const double *pair_j,
                                    const double *dx, const double *dy, const double *r,
                                    const double *W, const double *dW, const double *mass,
                                    int n_fluid, int n_total, double rho0, double h,
                                    double inv_sigma0, double *rho_out, double *Vol_out,
                                    double *B_out)
{
    double *sigma_inner = (double *)calloc((size_t)n_fluid, sizeof(double));
    double *sigma_contact = (double *)calloc((size_t)n_fluid, sizeof(double));
    double *A11 = (double *)calloc((size_t)n_fluid, sizeof(double));
    double *A12 = (double *)calloc((size_t)n_fluid, sizeof(double));
    double *A21 = (double *)calloc((size_t)n_fluid, sizeof(double));
    double *A22 = (double *)calloc((size_t)n_fluid, sizeof(double));
    const double W0 = spline2d_w0(h);
    const long np = (long)n_pairs;
    for (int i = 0; i < n_fluid; ++i) sigma_inner[i] = W0; /* :178-181 */

    ORC_PARALLEL_FOR
    for (long k = 0; k < np; ++k) { /* :188-212 */
        const int ii = (int)pair_i[k] - 1, jj = (int)pair_j[k] - 1;
        const double wk = W[k];
        if (ii < 0 || ii >= n_fluid || jj < 0 || jj >= n_total) continue;
        if (jj < n_fluid) {
            ORC_ATOMIC
            sigma_inner[ii] += wk;
            ORC_ATOMIC
            sigma_inner[jj] += wk;
        } else {
            ORC_ATOMIC
            sigma_contact[ii] += wk * (mass[jj] / rho0);
        }
    }
    for (int i = 0; i < n_total; ++i) rho_out[i] = rho0; /* :214-216 */
    for (int i = 0; i < n_fluid; ++i) {                    /* :218-225 */
        double rhoi = sigma_inner[i] * rho0 * inv_sigma0;
        rhoi += sigma_contact[i] * rho0 * rho0 * inv_sigma0 / mass[i];
        if (rhoi <= 1e-12) rhoi = rho0;
        rho_out[i] = rhoi;
    }
    for (int i = 0; i < n_total; ++i) { /* :227-234 */
        double rhoi = rho_out[i];
        if (rhoi <= 1e-12) { rhoi = rho0; rho_out[i] = rhoi; }
        Vol_out[i] = mass[i] / rhoi;
    }

    ORC_PARALLEL_FOR
    for (long k = 0; k < np; ++k) { /* :239-312 */
        const int ii = (int)pair_i[k] - 1, jj = (int)pair_j[k] - 1;
        const double rk = r[k], dWk = dW[k];
        if (ii < 0 || ii >= n_fluid || jj < 0 || jj >= n_total || rk <= 1e-12) continue;
        const double ex = dx[k] / rk, ey = dy[k] / rk;
        const double fxj = dWk * Vol_out[jj];
        ORC_ATOMIC
        A11[ii] -= dx[k] * (fxj * ex);
        ORC_ATOMIC
        A12[ii] -= dx[k] * (fxj * ey);
        ORC_ATOMIC
        A21[ii] -= dy[k] * (fxj * ex);
        ORC_ATOMIC
        A22[ii] -= dy[k] * (fxj * ey);
        if (jj < n_fluid) {
            const double fxi = dWk * Vol_out[ii];
            ORC_ATOMIC
            A11[jj] -= dx[k] * (fxi * ex);
            ORC_ATOMIC
            A12[jj] -= dx[k] * (fxi * ey);
            ORC_ATOMIC
            A21[jj] -= dy[k] * (fxi * ex);
            ORC_ATOMIC
            A22[jj] -= dy[k] * (fxi * ey);
        }
    }
    for (int i = 0; i < n_total; ++i) { /* :314-319 */
        B_out[i] = 1.0; B_out[i + n_total] = 0.0;
        B_out[i + 2 * (size_t)n_total] = 0.0; B_out[i + 3 * (size_t)n_total] = 1.0;
    }
    for (int i = 0; i < n_fluid; ++i) { /* :321-366 */
        const double a11 = A11[i], a12 = A12[i], a21 = A21[i], a22 = A22[i];
        const double ata11 = a11 * a11 + a21 * a21 + ORC_EPS_REG;
        const double ata12 = a11 * a12 + a21 * a22;
        const double ata22 = a12 * a12 + a22 * a22 + ORC_EPS_REG;
        const double det_m = ata11 * ata22 - ata12 * ata12;
        double p11, p12, p21, p22;
        if (fabs(det_m) < 1e-20) {
            p11 = 1.0; p12 = 0.0; p21 = 0.0; p22 = 1.0;
        } else {
            const double im11 = ata22 / det_m, im12 = -ata12 / det_m, im22 = ata11 / det_m;
            p11 = im11 * a11 + im12 * a12;
            p12 = im11 * a21 + im12 * a22;
            p21 = im12 * a11 + im22 * a12;
            p22 = im12 * a21 + im22 * a22;
        }
        const double det_a = a11 * a22 - a12 * a21;
        const double det_sqr = fmax(1.0 - det_a, 0.0);
        const double denom = det_a + det_sqr;
        double w1, w2;
        if (fabs(denom) < 1e-12) { w1 = 0.0; w2 = 1.0; }
        else { w1 = det_a / denom; w2 = det_sqr / denom; }
        B_out[i] = w1 * p11 + w2;
        B_out[i + n_total] = w1 * p12;
        B_out[i + 2 * (size_t)n_total] = w1 * p21;
        B_out[i + 3 * (size_t)n_total] = w1 * p22 + w2;
    }
    free(sigma_inner); free(sigma_contact); free(A11); free(A12); free(A21); free(A22);
}

/* sph_physics_mex.c:396-550 */
ORC_API void orc_viscous_force(size_t n_pairs, const double *pair_i, const double *pair_j,
                               const double *dx, const double *dy, const double *r,
                               const double *dW, const double *vel, const double *Vol,
                               const double *B, double mu, double h, int n_fluid, int n_total,
                               const double *mass, const double *wall_vel, double *force_out)
{
    (void)mass; /* :547 */
    const size_t nt = (size_t)n_total;
    const double *vx = vel, *vy = vel + nt, *wvx = wall_vel, *wvy = wall_vel + nt;
    double *acc_x = (double *)calloc(nt, sizeof(double));
    double *acc_y = (double *)calloc(nt, sizeof(double));
    const long np = (long)n_pairs;

    ORC_PARALLEL_FOR
    for (long k = 0; k < np; ++k) { /* :469-536 */
        const int ii = (int)pair_i[k] - 1, jj = (int)pair_j[k] - 1;
        const double rk = r[k], dWk = dW[k];
        if (ii < 0 || ii >= n_fluid || jj < 0 || jj >= n_total || rk <= 1e-12) continue;
        const double ex = dx[k] / rk, ey = dy[k] / rk;
        const double b11i = B[ii], b12i = B[ii + nt], b21i = B[ii + 2 * nt], b22i = B[ii + 3 * nt];
        if (jj < n_fluid) {
            const double bs11 = b11i + B[jj], bs12 = b12i + B[jj + nt];
            const double bs21 = b21i + B[jj + 2 * nt], bs22 = b22i + B[jj + 3 * nt];
            const double eBe = ex * (bs11 * ex + bs12 * ey) + ey * (bs21 * ex + bs22 * ey);
            const double denom = rk + 0.01 * h;
            const double dvx = vx[ii] - vx[jj], dvy = vy[ii] - vy[jj];
            const double coeff_i = eBe * mu * dWk * Vol[jj] / denom;
            const double coeff_j = eBe * mu * dWk * Vol[ii] / denom;
            ORC_ATOMIC
            acc_x[ii] += coeff_i * dvx;
            ORC_ATOMIC
            acc_y[ii] += coeff_i * dvy;
            ORC_ATOMIC
            acc_x[jj] -= coeff_j * dvx;
            ORC_ATOMIC
            acc_y[jj] -= coeff_j * dvy;
        } else {
            const double eBe = ex * (b11i * ex + b12i * ey) + ey * (b21i * ex + b22i * ey);
            const double denom = rk + 0.01 * h;
            const double dvx = vx[ii] - wvx[jj], dvy = vy[ii] - wvy[jj];
            const double coeff = 4.0 * eBe * mu * dWk * Vol[jj] / denom;
            ORC_ATOMIC
            acc_x[ii] += coeff * dvx;
            ORC_ATOMIC
            acc_y[ii] += coeff * dvy;
        }
    }
    for (int i = 0; i < n_fluid; ++i) { /* :538-541 */
        force_out[i] = acc_x[i] * Vol[i];
        force_out[i + nt] = acc_y[i] * Vol[i];
    }
    for (int i = n_fluid; i < n_total; ++i) { force_out[i] = 0.0; force_out[i + nt] = 0.0; }
    free(acc_x); free(acc_y);
}

/* sph_physics_mex.c:569-714 (transport_coeff default 0.2 at :584 is applied by the caller) */
ORC_API void orc_transport_correction(size_t n_pairs, const double *pair_i, const double *pair_j,
                                      const double *dx, const double *dy, const double *r,
                                      const double *dW, const double *Vol, const double *B,
                                      const double *pos, double h, int n_fluid, int n_total,
                                      double transport_coeff, double *pos_out)
{
    const size_t nt = (size_t)n_total;
    double *inc_x = (double *)calloc(nt, sizeof(double));
    double *inc_y = (double *)calloc(nt, sizeof(double));
    const long np = (long)n_pairs;
    memcpy(pos_out, pos, 2 * nt * sizeof(double)); /* :627-628 */

    ORC_PARALLEL_FOR
    for (long k = 0; k < np; ++k) { /* :636-700 */
        const int ii = (int)pair_i[k] - 1, jj = (int)pair_j[k] - 1;
        const double rk = r[k], dWk = dW[k];
        if (ii < 0 || ii >= n_fluid || jj < 0 || jj >= n_total || rk <= 1e-12) continue;
        const double ex = dx[k] / rk, ey = dy[k] / rk;
        const double b11i = B[ii], b12i = B[ii + nt], b21i = B[ii + 2 * nt], b22i = B[ii + 3 * nt];
        if (jj < n_fluid) {
            const double bs11 = b11i + B[jj], bs12 = b12i + B[jj + nt];
            const double bs21 = b21i + B[jj + 2 * nt], bs22 = b22i + B[jj + 3 * nt];
            const double tx = bs11 * ex + bs12 * ey, ty = bs21 * ex + bs22 * ey;
            const double coeff_i = -dWk * Vol[jj], coeff_j = dWk * Vol[ii];
            ORC_ATOMIC
            inc_x[ii] += coeff_i * tx;
            ORC_ATOMIC
            inc_y[ii] += coeff_i * ty;
            ORC_ATOMIC
            inc_x[jj] += coeff_j * tx;
            ORC_ATOMIC
            inc_y[jj] += coeff_j * ty;
        } else {
            const double tx = b11i * ex + b12i * ey, ty = b21i * ex + b22i * ey;
            const double coeff = -2.0 * dWk * Vol[jj];
            ORC_ATOMIC
            inc_x[ii] += coeff * tx;
            ORC_ATOMIC
            inc_y[ii] += coeff * ty;
        }
    }
    for (int i = 0; i < n_fluid; ++i) { /* :702-710 */
        const double n2 = inc_x[i] * inc_x[i] + inc_y[i] * inc_y[i];
        double limiter = 100.0 * n2 / (h * h);
        const double scale = transport_coeff * h * h;
        if (limiter > 1.0) limiter = 1.0;
        if (limiter < 0.0) limiter = 0.0;
        pos_out[i] += scale * limiter * inc_x[i];
        pos_out[i + nt] += scale * limiter * inc_y[i];
    }
    free(inc_x); free(inc_y);
}

static double riemann_beta(double un_l, double un_r, double c_f)
{ /* sph_physics_mex.c:1121-1129 */
    double compression = un_l - un_r;
    if (compression < 0.0) compression = 0.0;
    return fmin(3.0 * compression, c_f);
}

/* sph_physics_mex.c:736-967 */
ORC_API void orc_integration_1st(size_t n_pairs, const double *pair_i, const double *pair_j,
                                 const double *dx, const double *dy, const double *r,
                                 const double *dW, const double *Vol, const double *B,
                                 const double *rho_in, const double *mass, const double *pos_in,
                                 const double *vel, const double *drho_in,
                                 const double *force_prior, double dt, int n_fluid, int n_total,
                                 double rho0, double p0, double c_f, const double *wall_vel,
                                 double *rho_out, double *p_out, double *pos_out,
                                 double *force_out, double *drho_out)
{
    (void)wall_vel; /* :964-965 */
    const size_t nt = (size_t)n_total;
    const double *vx = vel, *vy = vel + nt, *fpx = force_prior, *fpy = force_prior + nt;
    double *fx = force_out, *fy = force_out + nt;
    double *diss = (double *)calloc(nt, sizeof(double));
    const long np = (long)n_pairs;

    memcpy(rho_out, rho_in, nt * sizeof(double)); /* :847-853 */
    memcpy(pos_out, pos_in, 2 * nt * sizeof(double));
    memset(p_out, 0, nt * sizeof(double));
    memset(force_out, 0, 2 * nt * sizeof(double));
    memset(drho_out, 0, nt * sizeof(double));

    for (int i = 0; i < n_fluid; ++i) { /* :857-865 */
        rho_out[i] = rho_out[i] + 0.5 * dt * drho_in[i];
        if (rho_out[i] < 1e-10) rho_out[i] = rho0;
        p_out[i] = p0 * (rho_out[i] / rho0 - 1.0);
        pos_out[i] += 0.5 * dt * vx[i];
        pos_out[i + nt] += 0.5 * dt * vy[i];
    }

    ORC_PARALLEL_FOR
    for (long k = 0; k < np; ++k) { /* :870-951 */
        const int ii = (int)pair_i[k] - 1, jj = (int)pair_j[k] - 1;
        const double rk = r[k], dWk = dW[k];
        if (ii < 0 || ii >= n_fluid || jj < 0 || jj >= n_total || rk <= 1e-12) continue;
        const double ex = dx[k] / rk, ey = dy[k] / rk;
        const double b11i = B[ii], b12i = B[ii + nt], b21i = B[ii + 2 * nt], b22i = B[ii + 3 * nt];
        if (jj < n_fluid) {
            const double p_i = p_out[ii], p_j = p_out[jj];
            const double rho_bar = 0.5 * (rho_out[ii] + rho_out[jj]);
            const double un_l = vx[ii] * ex + vy[ii] * ey;
            const double un_r = vx[jj] * ex + vy[jj] * ey;
            const double beta = riemann_beta(un_l, un_r, c_f);
            const double p_star = 0.5 * (p_i + p_j) + 0.5 * beta * rho_bar * (un_l - un_r);
            const double p_face = 0.5 * (0.5 * (p_i + p_j) + p_star); /* :892 */
            const double b11j = B[jj], b12j = B[jj + nt], b21j = B[jj + 2 * nt], b22j = B[jj + 3 * nt];
            const double tx = p_face * ((b11i + b11j) * ex + (b12i + b12j) * ey);
            const double ty = p_face * ((b21i + b21j) * ex + (b22i + b22j) * ey);
            const double dWVj = dWk * Vol[jj], dWVi = dWk * Vol[ii];
            const double p_diff = p_i - p_j;
            ORC_ATOMIC
            fx[ii] -= tx * dWVj;
            ORC_ATOMIC
            fy[ii] -= ty * dWVj;
            ORC_ATOMIC
            fx[jj] += tx * dWVi;
            ORC_ATOMIC
            fy[jj] += ty * dWVi;
            ORC_ATOMIC
            diss[ii] += (p_diff / (rho0 * c_f)) * dWVj;
            ORC_ATOMIC
            diss[jj] += (-p_diff / (rho0 * c_f)) * dWVi;
        } else {
            const double p_i = p_out[ii], rho_i = rho_out[ii];
            const double dWVj = dWk * Vol[jj];
            const double ax = fpx[ii] / mass[ii], ay = fpy[ii] / mass[ii];
            const double face_wall_ext_acc = -(ax * ex + ay * ey);
            const double p_wall = p_i + rho_i * rk * fmax(0.0, face_wall_ext_acc);
            const double tx = b11i * ex + b12i * ey, ty = b21i * ex + b22i * ey;
            ORC_ATOMIC
            fx[ii] -= (p_i + p_wall) * dWVj * tx;
            ORC_ATOMIC
            fy[ii] -= (p_i + p_wall) * dWVj * ty;
            ORC_ATOMIC
            diss[ii] += ((p_i - p_wall) / (rho0 * c_f)) * dWVj;
        }
    }
    for (int i = 0; i < n_fluid; ++i) { /* :953-957 */
        fx[i] *= Vol[i];
        fy[i] *= Vol[i];
        drho_out[i] = diss[i] * rho_out[i];
    }
    for (int i = n_fluid; i < n_total; ++i) { fx[i] = 0.0; fy[i] = 0.0; drho_out[i] = 0.0; }
    free(diss);
}

/* sph_physics_mex.c:987-1119 ; third output is all zeros (:1050,:1064) */
ORC_API void orc_integration_2nd(size_t n_pairs, const double *pair_i, const double *pair_j,
                                 const double *dx, const double *dy, const double *r,
                                 const double *dW, const double *Vol, const double *rho,
                                 const double *pos_in, const double *vel, double dt, int n_fluid,
                                 int n_total, const double *wall_vel, double *pos_out,
                                 double *drho_out, double *zeros_out)
{
    const size_t nt = (size_t)n_total;
    const double *vx = vel, *vy = vel + nt, *wvx = wall_vel, *wvy = wall_vel + nt;
    const long np = (long)n_pairs;
    memcpy(pos_out, pos_in, 2 * nt * sizeof(double));
    memset(drho_out, 0, nt * sizeof(double));
    if (zeros_out) memset(zeros_out, 0, 2 * nt * sizeof(double));
    for (int i = 0; i < n_fluid; ++i) { /* :1066-1069 */
        pos_out[i] += 0.5 * dt * vx[i];
        pos_out[i + nt] += 0.5 * dt * vy[i];
    }
    double *rate = (double *)calloc(nt, sizeof(double));

    ORC_PARALLEL_FOR
    for (long k = 0; k < np; ++k) { /* :1076-1109 */
        const int ii = (int)pair_i[k] - 1, jj = (int)pair_j[k] - 1;
        const double rk = r[k], dWk = dW[k];
        if (ii < 0 || ii >= n_fluid || jj < 0 || jj >= n_total || rk <= 1e-12) continue;
        const double ex = dx[k] / rk, ey = dy[k] / rk;
        if (jj < n_fluid) {
            const double u_jump = (vx[ii] - vx[jj]) * ex + (vy[ii] - vy[jj]) * ey;
            ORC_ATOMIC
            rate[ii] += u_jump * dWk * Vol[jj];
            ORC_ATOMIC
            rate[jj] += u_jump * dWk * Vol[ii];
        } else {
            const double vjx = 2.0 * wvx[jj] - vx[ii], vjy = 2.0 * wvy[jj] - vy[ii];
            const double jump = (vx[ii] - vjx) * ex + (vy[ii] - vjy) * ey;
            ORC_ATOMIC
            rate[ii] += jump * dWk * Vol[jj];
        }
    }
    for (int i = 0; i < n_fluid; ++i) drho_out[i] = rate[i] * rho[i]; /* :1111-1116 */
    for (int i = n_fluid; i < n_total; ++i) drho_out[i] = 0.0;
    free(rate);
}

/* sph_physics_mex.c:1316-1469 */
ORC_API void orc_integration_verlet(size_t n_pairs, const double *pair_i, const double *pair_j,
                                    const double *dx, const double *dy, const double *r,
                                    const double *dW, const double *Vol, const double *B,
                                    const double *rho_in, const double *mass, const double *pos_in,
                                    const double *vel_in, const double *drho_in,
                                    const double *force_prior, double dt, int n_fluid,
                                    int n_total, double rho0, double p0, double c_f,
                                    const double *wall_vel, double *rho_out, double *p_out,
                                    double *pos_out, double *vel_out, double *drho_out,
                                    double *force_out)
{
    const size_t nt = (size_t)n_total;
    double *rho_h = (double *)malloc(nt * sizeof(double));
    double *p_h = (double *)malloc(nt * sizeof(double));
    double *pos_h = (double *)malloc(2 * nt * sizeof(double));
    double *diss_unused = (double *)malloc(nt * sizeof(double));
    orc_integration_1st(n_pairs, pair_i, pair_j, dx, dy, r, dW, Vol, B, rho_in, mass, pos_in,
                        vel_in, drho_in, force_prior, dt, n_fluid, n_total, rho0, p0, c_f, wall_vel,
                        rho_h, p_h, pos_h, force_out, diss_unused); /* :1386 */
    memcpy(vel_out, vel_in, 2 * nt * sizeof(double)); /* :1388-1409 */
    for (int i = 0; i < n_fluid; ++i) {
        const double inv_mass = 1.0 / mass[i];
        vel_out[i] += (force_prior[i] + force_out[i]) * inv_mass * dt;
        vel_out[i + nt] += (force_prior[i + nt] + force_out[i + nt]) * inv_mass * dt;
    }
    for (int i = n_fluid; i < n_total; ++i) { vel_out[i] = 0.0; vel_out[i + nt] = 0.0; }
    orc_integration_2nd(n_pairs, pair_i, pair_j, dx, dy, r, dW, Vol, rho_h, pos_h, vel_out, dt,
                        n_fluid, n_total, wall_vel, pos_out, drho_out, NULL); /* :1427 */
    memcpy(rho_out, rho_h, nt * sizeof(double)); /* :1429-1451 */
    for (int i = 0; i < n_fluid; ++i) {
        rho_out[i] += drho_out[i] * (0.5 * dt);
        if (rho_out[i] < 1e-10) rho_out[i] = rho0;
        p_out[i] = p0 * (rho_out[i] / rho0 - 1.0);
    }
    for (int i = n_fluid; i < n_total; ++i) { rho_out[i] = rho_h[i]; p_out[i] = 0.0; }
    free(rho_h); free(p_h); free(pos_h); free(diss_unused);
}

/* sph_physics_mex.c:1490-1639 ; transport runs with the 13-argument default coeff 0.2 (:584,:1596) */
ORC_API void orc_advance_shell_step(size_t n_pairs, const double *pair_i, const double *pair_j,
                                    const double *dx, const double *dy, const double *r,
                                    const double *W, const double *dW, const double *mass,
                                    const double *pos, const double *vel, const double *wall_vel,
                                    const double *rho, const double *drho_dt, double dt,
                                    int n_fluid, int n_total, double rho0, double p0, double c_f,
                                    double mu, double h, double inv_sigma0, double gravity_g,
                                    double *rho_out, double *p_out, double *pos_out,
                                    double *vel_out, double *drho_out, double *force_out,
                                    double *force_prior_out, double *Vol_out, double *B_out)
{
    (void)rho; /* validated for size only (:1532); density is re-summed (:1554) */
    const size_t nt = (size_t)n_total;
    double *rho_d = (double *)malloc(nt * sizeof(double));
    double *pos_t = (double *)malloc(2 * nt * sizeof(double));
    orc_density_correction(n_pairs, pair_i, pair_j, dx, dy, r, W, dW, mass, n_fluid, n_total, rho0,
                           h, inv_sigma0, rho_d, Vol_out, B_out);
    orc_viscous_force(n_pairs, pair_i, pair_j, dx, dy, r, dW, vel, Vol_out, B_out, mu, h, n_fluid,
                      n_total, mass, wall_vel, force_prior_out);
    for (int i = 0; i < n_fluid; ++i) force_prior_out[i] += mass[i] * gravity_g; /* :1575-1580 */
    orc_transport_correction(n_pairs, pair_i, pair_j, dx, dy, r, dW, Vol_out, B_out, pos, h,
                             n_fluid, n_total, 0.2, pos_t);
    orc_integration_verlet(n_pairs, pair_i, pair_j, dx, dy, r, dW, Vol_out, B_out, rho_d, mass,
                           pos_t, vel, drho_dt, force_prior_out, dt, n_fluid, n_total, rho0, p0,
                           c_f, wall_vel, rho_out, p_out, pos_out, vel_out, drho_out, force_out);
    free(rho_d); free(pos_t);
}

/* sph_physics_mex.c:1653-1743 (serial in the reference too) */
ORC_API void orc_wall_shear_monitor(size_t n_pairs, const double *pair_i, const double *pair_j,
                                    const double *dx, const double *dy, const double *r,
                                    const double *dW, const double *pos, const double *vel,
                                    const double *wall_vel, const double *Vol, const double *B,
                                    int n_fluid, int n_total, double DL, double DH, double mu,
                                    double h, double *tau_bottom, double *tau_top)
{
    const size_t nt = (size_t)n_total;
    const double *pos_y = pos + nt, *vx = vel, *wvx = wall_vel;
    double sb = 0.0, st = 0.0;
    for (size_t k = 0; k < n_pairs; ++k) {
        const int ii = (int)pair_i[k] - 1, jj = (int)pair_j[k] - 1;
        const double rk = r[k];
        if (ii < 0 || ii >= n_fluid || jj < n_fluid || rk <= 1e-12) continue; /* :1722 */
        const double ex = dx[k] / rk, ey = dy[k] / rk;
        const double eBe = ex * (B[ii] * ex + B[ii + nt] * ey) +
                           ey * (B[ii + 2 * nt] * ex + B[ii + 3 * nt] * ey);
        const double dv_x = vx[ii] - wvx[jj];
        const double f_pair = 4.0 * mu * eBe * dW[k] * Vol[jj] * dv_x / (rk + 0.01 * h) * Vol[ii];
        if (pos_y[jj] <= 0.0) sb += f_pair;
        else if (pos_y[jj] >= DH) st += f_pair;
    }
    *tau_bottom = -sb / DL;
    *tau_top = -st / DL;
}

/* ------------------------------------------------------------------------------------------ */
/* host-driver pieces restated from SPH_Poiseuille.m                                           */
/* ------------------------------------------------------------------------------------------ */

/* SPH_Poiseuille.m:519-527 */
ORC_API double orc_verlet_time_step(const double *vel, int n_fluid, int n_total, double c_max,
                                    double h, double nu, double gravity_g, double remain)
{
    double v_max = 0.0;
    for (int i = 0; i < n_fluid; ++i) {
        const double vx = vel[i], vy = vel[i + (size_t)n_total];
        const double v = sqrt(vx * vx + vy * vy);
        if (v > v_max) v_max = v;
    }
    const double dt_acoustic = 0.25 * h / fmax(c_max + v_max, 1e-12);
    const double dt_viscous = 0.125 * h * h / fmax(nu, 1e-12);
    const double dt_body = 0.25 * sqrt(h / fmax(fabs(gravity_g), 1e-12));
    double dt = fmin(fmin(dt_acoustic, dt_viscous), fmin(dt_body, remain));
    return fmax(dt, 1e-12);
}

/* SPH_Poiseuille.m:570-577 ; mod(x,DL) written as x - floor(x/DL)*DL */
ORC_API void orc_periodic_bounding(double *pos, int n_fluid, double DL)
{
    for (int i = 0; i < n_fluid; ++i) pos[i] = pos[i] - floor(pos[i] / DL) * DL;
}

/* SPH_Poiseuille.m:555-568 : stable sort of the fluid block by (cy, cx), 2h cells, y0 = min fluid y */
typedef struct { int cy, cx, idx; } sort_key;
static int sort_key_cmp(const void *a, const void *b)
{
    const sort_key *p = (const sort_key *)a, *q = (const sort_key *)b;
    if (p->cy != q->cy) return p->cy < q->cy ? -1 : 1;
    if (p->cx != q->cx) return p->cx < q->cx ? -1 : 1;
    return p->idx < q->idx ? -1 : (p->idx > q->idx ? 1 : 0);
}
ORC_API void orc_sort_subset_indices(const double *pos, int n_fluid, int n_total, double DL,
                                     double h, int *idx_out)
{
    const double cell_size = 2.0 * h;
    const double *x = pos, *y = pos + (size_t)n_total;
    double y0 = y[0];
    for (int i = 1; i < n_fluid; ++i) if (y[i] < y0) y0 = y[i];
    sort_key *keys = (sort_key *)malloc((size_t)n_fluid * sizeof(sort_key));
    for (int i = 0; i < n_fluid; ++i) {
        const double xm = x[i] - floor(x[i] / DL) * DL;
        keys[i].cx = (int)floor(xm / cell_size);
        keys[i].cy = (int)floor((y[i] - y0) / cell_size);
        keys[i].idx = i;
    }
    qsort(keys, (size_t)n_fluid, sizeof(sort_key), sort_key_cmp);
    for (int i = 0; i < n_fluid; ++i) idx_out[i] = keys[i].idx;
    free(keys);
}

static void permute_rows(double *a, int ncol, int n_fluid, int n_total, const int *idx, double *tmp)
{
    for (int c = 0; c < ncol; ++c) {
        double *col = a + (size_t)c * n_total;
        for (int i = 0; i < n_fluid; ++i) tmp[i] = col[idx[i]];
        memcpy(col, tmp, (size_t)n_fluid * sizeof(double));
    }
}

typedef struct {
    double DL, DH, rho0, mu, c_f, h, p0, inv_sigma0, gravity_g, transport_coeff;
    double t_end, output_interval;
    int sort_interval;
    int enable_sort;  /* SPH_Poiseuille.m:272-278 on/off */
    long max_steps;   /* <=0: unlimited */
    int log_every;    /* 0 = silent */
} orc_run_config;

typedef struct {
    long steps;
    double t;
    double seconds_neighbor, seconds_physics, seconds_total;
    double tau_bottom, tau_top, vmax, dt_last;
    double n_pairs_last;
} orc_run_stats;

/*
 * orc_run -- the time loop of SPH_Poiseuille.m:246-302 on caller-owned column-major state.
 * `order` (int[n_total], may be NULL) tracks the permutation applied by the re-sort so callers can
 * map rows back to their initial identity.  State in/out: pos, vel, drho_dt (+ mass, wall_vel which the
 * sort permutes too, :273-277); outputs of the last step: rho, p, force, force_prior, Vol, B.
 */
ORC_API int orc_run(const orc_run_config *cfg, int n_fluid, int n_total, double *pos, double *vel,
                    double *drho_dt, double *mass, double *wall_vel, double *rho, double *p,
                    double *force, double *force_prior, double *Vol, double *B, int *order,
                    double t0, long step0, orc_run_stats *stats)
{
    const size_t nt = (size_t)n_total;
    const double nu = cfg->mu / cfg->rho0;
    orc_pairs nb;
    double t = t0;
    long step = step0;
    double *pos2 = (double *)malloc(2 * nt * sizeof(double));
    double *vel2 = (double *)malloc(2 * nt * sizeof(double));
    double *rho2 = (double *)malloc(nt * sizeof(double));
    double *p2 = (double *)malloc(nt * sizeof(double));
    double *drho2 = (double *)malloc(nt * sizeof(double));
    double *tmp = (double *)malloc(nt * sizeof(double));
    int *idx = (int *)malloc(nt * sizeof(int));
    int *ord2 = (int *)malloc(nt * sizeof(int));
    memset(stats, 0, sizeof(*stats));
    const double t_begin = now_seconds();
    double t_a = now_seconds();
    int rc = orc_neighbor_search(pos, n_fluid, n_total, cfg->h, cfg->DL, &nb); /* :167 */
    stats->seconds_neighbor += now_seconds() - t_a;
    if (rc) goto done;

    while (t < cfg->t_end - 1e-12) { /* :247 */
        double target_time = fmin(t + cfg->output_interval, cfg->t_end);
        while (t < target_time - 1e-12) { /* :250 */
            if (cfg->max_steps > 0 && step - step0 >= cfg->max_steps) goto finished;
            step += 1;
            const double remain = fmin(target_time - t, cfg->t_end - t);
            t_a = now_seconds();
            orc_density_correction(nb.count, nb.pair_i, nb.pair_j, nb.dx, nb.dy, nb.r, nb.W, nb.dW,
                                   mass, n_fluid, n_total, cfg->rho0, cfg->h, cfg->inv_sigma0, rho,
                                   Vol, B); /* :254 */
            orc_viscous_force(nb.count, nb.pair_i, nb.pair_j, nb.dx, nb.dy, nb.r, nb.dW, vel, Vol,
                              B, cfg->mu, cfg->h, n_fluid, n_total, mass, wall_vel, force_prior);
            for (int i = 0; i < n_fluid; ++i) force_prior[i] += mass[i] * cfg->gravity_g; /* :392 */
            for (int i = n_fluid; i < n_total; ++i) { force_prior[i] = 0.0; force_prior[i + nt] = 0.0; }
            orc_transport_correction(nb.count, nb.pair_i, nb.pair_j, nb.dx, nb.dy, nb.r, nb.dW, Vol,
                                     B, pos, cfg->h, n_fluid, n_total, cfg->transport_coeff, pos2);
            memcpy(pos, pos2, 2 * nt * sizeof(double)); /* :257 */
            const double dt = orc_verlet_time_step(vel, n_fluid, n_total, cfg->c_f, cfg->h, nu,
                                                   cfg->gravity_g, remain); /* :259 */
            if (dt < 1e-14) { rc = -20; goto done; } /* :260-263 */
            orc_integration_verlet(nb.count, nb.pair_i, nb.pair_j, nb.dx, nb.dy, nb.r, nb.dW, Vol,
                                   B, rho, mass, pos, vel, drho_dt, force_prior, dt, n_fluid,
                                   n_total, cfg->rho0, cfg->p0, cfg->c_f, wall_vel, rho2, p2, pos2,
                                   vel2, drho2, force); /* :265 */
            memcpy(rho, rho2, nt * sizeof(double));
            memcpy(p, p2, nt * sizeof(double));
            memcpy(pos, pos2, 2 * nt * sizeof(double));
            memcpy(vel, vel2, 2 * nt * sizeof(double));
            memcpy(drho_dt, drho2, nt * sizeof(double));
            stats->seconds_physics += now_seconds() - t_a;
            t += dt; /* :267 */
            stats->dt_last = dt;
            orc_periodic_bounding(pos, n_fluid, cfg->DL); /* :269 */
            for (int i = n_fluid; i < n_total; ++i) { vel[i] = 0.0; vel[i + nt] = 0.0; } /* :270 */

            if (cfg->enable_sort && cfg->sort_interval > 0 && step % cfg->sort_interval == 0 &&
                step != 1 && n_fluid < n_total) { /* :272-278, :535 */
                orc_sort_subset_indices(pos, n_fluid, n_total, cfg->DL, cfg->h, idx);
                permute_rows(pos, 2, n_fluid, n_total, idx, tmp);
                permute_rows(vel, 2, n_fluid, n_total, idx, tmp);
                permute_rows(rho, 1, n_fluid, n_total, idx, tmp);
                permute_rows(mass, 1, n_fluid, n_total, idx, tmp);
                permute_rows(wall_vel, 2, n_fluid, n_total, idx, tmp);
                permute_rows(drho_dt, 1, n_fluid, n_total, idx, tmp);
                permute_rows(force_prior, 2, n_fluid, n_total, idx, tmp);
                permute_rows(force, 2, n_fluid, n_total, idx, tmp);
                permute_rows(p, 1, n_fluid, n_total, idx, tmp);
                permute_rows(Vol, 1, n_fluid, n_total, idx, tmp);
                permute_rows(B, 4, n_fluid, n_total, idx, tmp);
                if (order) {
                    for (int i = 0; i < n_fluid; ++i) ord2[i] = order[idx[i]];
                    memcpy(order, ord2, (size_t)n_fluid * sizeof(int));
                }
            }
            t_a = now_seconds();
            orc_pairs_free(&nb);
            rc = orc_neighbor_search(pos, n_fluid, n_total, cfg->h, cfg->DL, &nb); /* :280 */
            stats->seconds_neighbor += now_seconds() - t_a;
            if (rc) goto done;
            orc_wall_shear_monitor(nb.count, nb.pair_i, nb.pair_j, nb.dx, nb.dy, nb.r, nb.dW, pos,
                                   vel, wall_vel, Vol, B, n_fluid, n_total, cfg->DL, cfg->DH,
                                   cfg->mu, cfg->h, &stats->tau_bottom, &stats->tau_top); /* :281 */
            if (cfg->log_every > 0 && step % cfg->log_every == 0) { /* :285-291 */
                double vmax = 0.0;
                for (int i = 0; i < n_fluid; ++i) {
                    const double v = sqrt(vel[i] * vel[i] + vel[i + nt] * vel[i + nt]);
                    if (v > vmax) vmax = v;
                }
                fprintf(stderr, "step=%ld, t=%.6f/%.6f, dt=%.4e, pairs=%zu, vmax=%.4f tau=%.4f/%.4f\n",
                        step, t, cfg->t_end, dt, nb.count, vmax, stats->tau_bottom, stats->tau_top);
            }
        }
    }
finished:
    {
        double vmax = 0.0;
        for (int i = 0; i < n_fluid; ++i) {
            const double v = sqrt(vel[i] * vel[i] + vel[i + nt] * vel[i + nt]);
            if (v > vmax) vmax = v;
        }
        stats->vmax = vmax;
    }
done:
    stats->steps = step - step0;
    stats->t = t;
    stats->n_pairs_last = (double)nb.count;
    stats->seconds_total = now_seconds() - t_begin;
    orc_pairs_free(&nb);
    free(pos2); free(vel2); free(rho2); free(p2); free(drho2); free(tmp); free(idx); free(ord2);
    return rc;
}

/* SPH_Poiseuille.m:579-590 : discretize() semantics -- bins [e_k, e_{k+1}), last bin closed. */
ORC_API void orc_binned_profile_mean(const double *y_values, const double *u_values, int n,
                                     double y_min, double y_max, int n_bins, double *y_mid,
                                     double *u_mean, double *count)
{
    double *edges = (double *)malloc(((size_t)n_bins + 1) * sizeof(double));
    for (int k = 0; k <= n_bins; ++k) /* linspace */
        edges[k] = (k == n_bins) ? y_max : y_min + (y_max - y_min) * ((double)k / (double)n_bins);
    for (int k = 0; k < n_bins; ++k) { y_mid[k] = 0.5 * (edges[k] + edges[k + 1]); u_mean[k] = 0.0; count[k] = 0.0; }
    for (int i = 0; i < n; ++i) {
        const double yv = y_values[i];
        if (!(yv >= edges[0] && yv <= edges[n_bins])) continue;
        int lo = 0, hi = n_bins; /* largest k with edges[k] <= yv */
        while (hi - lo > 1) { const int mid = (lo + hi) / 2; if (edges[mid] <= yv) lo = mid; else hi = mid; }
        if (lo >= n_bins) lo = n_bins - 1;
        u_mean[lo] += u_values[i];
        count[lo] += 1.0;
    }
    for (int k = 0; k < n_bins; ++k) u_mean[k] = count[k] > 0.0 ? u_mean[k] / count[k] : NAN;
    free(edges);
}

ORC_API void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
