// sphx_resident.hip -- device-resident SPH step for MI355X (gfx950): the whole inner loop of
// SPH_Poiseuille.m:250-292 (density_correction -> viscous_force + gravity -> transport_correction ->
// verlet_time_step -> integration_verlet -> periodic wrap -> neighbour rebuild) as seven HIP kernels
// per step with no host round trip, plus the monitors and the MEX-convention pair-list emitter.
//
// Design (DESIGN.md has the long form):
//   * particles live in SoA double arrays sorted by cell, cell id = cx*ncy + cy (y fastest): the 3x3
//     neighbourhood of a cell is three contiguous index ranges, and an x-slab of the channel is one
//     contiguous range (multi-GPU halo = contiguous copies);
//   * cells are exactly periodic in x: ncx = floor(DL/2h), width DL/ncx >= 2h, so the wrapped 3x3
//     sweep sees every min-image neighbour and the reference's ghost entries + seen_neighbor
//     (mex/sph_neighbor_search_mex.c:282-295,342,383) are not needed; the accepted set is the same
//     {r^2 > 1e-24, r^2 < (2h)^2} (:368);
//   * every pair sum is written as a per-particle gather (each fluid-fluid update of
//     mex/sph_physics_mex.c is symmetric under i<->j), so there are no atomics in the physics and the
//     result is bitwise reproducible run to run;
//   * LPP lanes of a wavefront cooperate on one particle's neighbour ring and combine with
//     __shfl_xor (wave64), which is what fills the chip at the 5-60 k particle configs;
//   * geometry is frozen for a step: all four passes use the positions the cell grid was built from
//     (the reference keeps dx,dy,r,W,dW of the list built at the end of the previous step);
//   * dt, t, step count and the stop condition live in a device-side clock, so steps can be captured
//     into a hipGraph and replayed.
#include <algorithm>
#include <cmath>
#include <map>

#include "sphx_common.hpp"
#include "sphx_device.hpp"

namespace sphx {
namespace {

constexpr int kBlock = 256;
constexpr int kScanBlock = 1024;

struct Grid {
    int ncx, ncy, ncells;
    double DL, y0, inv_csx, inv_csy;
};

struct Phys {
    KernelConst kc;
    double rho0, inv_sigma0, mu, p0, c_f, g, tc, nu, DL, DH, w0;
};

// Device-side clock: replaces the host variables state.t/state.step/dt_step/remain of
// SPH_Poiseuille.m:247-267 so the loop needs no host decisions.
struct Clock {
    double t, dt, dt_last, t_target, t_end, vmax;
    long long step, steps_left;  // steps_left < 0: unlimited
    int run[2];                  // run[q]: the step slot of parity q executes
    int status;
    int pad;
};

struct FluidSet {  // persistent per-particle state, sorted by cell
    double *x, *y, *vx, *vy, *drho, *mass;
    int *id;
    int *start;  // [ncells+1] cell ranges of this ordering
};

struct FluidTmp {
    double *xn, *yn, *vxn, *vyn, *drhon;                   // end-of-step state, pre-sort order
    double *rho, *Vol, *rhoh, *ph, *b11, *b12, *b21, *b22;  // per-step fields
    double *fpx, *fpy, *fx, *fy, *rho_out, *p_out;         // outputs of the step
    int *cellid, *count, *perm, *src_of;
    double *vpart;  // per-block max |v|^2 of pass E
};

struct Walls {
    const double *x, *y, *Vol, *vx, *vy;
    const int *id;
    const int *start;  // [ncells+1]
    const int *row_any;  // [ncy] 1 when rows cy-1..cy+1 hold any wall particle
    int n;
};

__device__ __forceinline__ void cell_of(const Grid &g, double x, double y, int &cx, int &cy)
{
    cx = (int)(x * g.inv_csx);
    cx = min(max(cx, 0), g.ncx - 1);
    cy = (int)floor((y - g.y0) * g.inv_csy);
    cy = min(max(cy, 0), g.ncy - 1);
}

__device__ __forceinline__ double wrap_x(double x, double DL) { return x - floor(x / DL) * DL; }

template <int LPP>
__device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
    for (int off = LPP / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Visit the candidates of the three cell columns around (cx,cy): body(k, xi_shifted) with
// dx = xi_shifted - x[k] being the min-image separation.
template <int LPP, typename Body>
__device__ __forceinline__ void sweep(const Grid &g, const int *__restrict__ start, int cx, int cy,
                                      int sub, double xi, Body &&body)
{
    const int cylo = max(cy - 1, 0), cyhi = min(cy + 1, g.ncy - 1);
#pragma unroll
    for (int ox = -1; ox <= 1; ++ox) {
        int col = cx + ox;
        double shift = 0.0;
        if (col < 0) { col += g.ncx; shift = -g.DL; }
        else if (col >= g.ncx) { col -= g.ncx; shift = g.DL; }
        const int base = col * g.ncy;
        const int lo = start[base + cylo], hi = start[base + cyhi + 1];
        const double xis = xi - shift;
        for (int k = lo + sub; k < hi; k += LPP) body(k, xis);
    }
}

__device__ __forceinline__ double next_dt(const Clock &c, const Phys &ph)
{  // SPH_Poiseuille.m:519-527 with remain of :252
    const double remain = fmin(c.t_target - c.t, c.t_end - c.t);
    const double h = ph.kc.h;
    const double dt_acoustic = 0.25 * h / fmax(ph.c_f + c.vmax, 1e-12);
    const double dt_viscous = 0.125 * h * h / fmax(ph.nu, 1e-12);
    const double dt_body = 0.25 * sqrt(h / fmax(fabs(ph.g), 1e-12));
    const double dt = fmin(fmin(dt_acoustic, dt_viscous), fmin(dt_body, remain));
    return fmax(dt, 1e-12);
}

__device__ __forceinline__ bool loop_continues(const Clock &c)
{  // while state.t < target_time - 1e-12 (SPH_Poiseuille.m:250) and step budget left
    return (c.t < c.t_target - 1e-12) && (c.steps_left != 0) && (c.status == 0);
}

// one thread: arm the clock for an advance call
__global__ void k_prepare(Clock *clk, Phys ph, double t_target, long long max_steps, int q0)
{
    Clock c = *clk;
    c.t_target = fmin(t_target, c.t_end);  // target_time = min(t + output_interval, t_end), SPH_Poiseuille.m:248
    c.steps_left = max_steps > 0 ? max_steps : -1;
    if (!(c.vmax == c.vmax) || isinf(c.vmax)) c.status = SPHX_ERR_DIVERGED;
    c.dt = next_dt(c, ph);
    const int go = loop_continues(c) ? 1 : 0;
    c.run[q0] = go;
    c.run[1 - q0] = 0;
    *clk = c;
}

// ---------------------------------------------------------------------------------------------
// pass A: number-density summation -> rho, Vol (mex/sph_physics_mex.c:188-234) and the half-step
// density/pressure of integration_1st's pre-pass (:857-862), which only needs own-particle data.
// ---------------------------------------------------------------------------------------------
template <int LPP>
__global__ __launch_bounds__(kBlock) void k_density(const Clock *clk, int q, Grid g, Phys ph, int nf,
                                                    FluidSet s, FluidTmp t, Walls w)
{
    if (!clk->run[q]) return;
    const int tid = blockIdx.x * kBlock + threadIdx.x;
    const int i = tid / LPP, sub = tid % LPP;
    const bool active = i < nf;
    double s_in = 0.0, s_ct = 0.0;
    if (active) {
        const double xi = s.x[i], yi = s.y[i];
        int cx, cy;
        cell_of(g, xi, yi, cx, cy);
        sweep<LPP>(g, s.start, cx, cy, sub, xi, [&](int k, double xis) {
            const double dx = xis - s.x[k], dy = yi - s.y[k];
            const double r2 = dx * dx + dy * dy;
            if (r2 > kR2Min && r2 < ph.kc.rcut2) s_in += spline_W(ph.kc, sqrt(r2));
        });
        if (w.row_any[cy]) {
            sweep<LPP>(g, w.start, cx, cy, sub, xi, [&](int k, double xis) {
                const double dx = xis - w.x[k], dy = yi - w.y[k];
                const double r2 = dx * dx + dy * dy;
                if (r2 > kR2Min && r2 < ph.kc.rcut2) s_ct += spline_W(ph.kc, sqrt(r2)) * w.Vol[k];
            });
        }
    }
    s_in = group_sum<LPP>(s_in);
    s_ct = group_sum<LPP>(s_ct);
    if (active && sub == 0) {
        const double m = s.mass[i];
        const double rho = density_from_sigma(ph.w0 + s_in, s_ct, m, ph.rho0, ph.inv_sigma0);
        const double dt = clk->dt;
        double rhoh = rho + 0.5 * dt * s.drho[i];
        if (rhoh < 1e-10) rhoh = ph.rho0;
        t.rho[i] = rho;
        t.Vol[i] = m / rho;
        t.rhoh[i] = rhoh;
        t.ph[i] = eos_pressure(rhoh, ph.rho0, ph.p0);
    }
}

// ---------------------------------------------------------------------------------------------
// pass B: kernel-gradient-correction matrix A -> blended pseudo-inverse B
// (mex/sph_physics_mex.c:239-366)
// ---------------------------------------------------------------------------------------------
template <int LPP>
__global__ __launch_bounds__(kBlock) void k_kgc(const Clock *clk, int q, Grid g, Phys ph, int nf,
                                                FluidSet s, FluidTmp t, Walls w)
{
    if (!clk->run[q]) return;
    const int tid = blockIdx.x * kBlock + threadIdx.x;
    const int i = tid / LPP, sub = tid % LPP;
    const bool active = i < nf;
    double a11 = 0.0, a12 = 0.0, a21 = 0.0, a22 = 0.0;
    if (active) {
        const double xi = s.x[i], yi = s.y[i];
        int cx, cy;
        cell_of(g, xi, yi, cx, cy);
        auto term = [&](double dx, double dy, double r2, double Volj) {
            const double r = sqrt(r2), inv_r = 1.0 / r;
            const double ex = dx * inv_r, ey = dy * inv_r;
            const double fxj = spline_dW(ph.kc, r) * Volj;
            a11 -= dx * (fxj * ex);
            a12 -= dx * (fxj * ey);
            a21 -= dy * (fxj * ex);
            a22 -= dy * (fxj * ey);
        };
        sweep<LPP>(g, s.start, cx, cy, sub, xi, [&](int k, double xis) {
            const double dx = xis - s.x[k], dy = yi - s.y[k];
            const double r2 = dx * dx + dy * dy;
            if (r2 > kR2Min && r2 < ph.kc.rcut2) term(dx, dy, r2, t.Vol[k]);
        });
        if (w.row_any[cy]) {
            sweep<LPP>(g, w.start, cx, cy, sub, xi, [&](int k, double xis) {
                const double dx = xis - w.x[k], dy = yi - w.y[k];
                const double r2 = dx * dx + dy * dy;
                if (r2 > kR2Min && r2 < ph.kc.rcut2) term(dx, dy, r2, w.Vol[k]);
            });
        }
    }
    a11 = group_sum<LPP>(a11);
    a12 = group_sum<LPP>(a12);
    a21 = group_sum<LPP>(a21);
    a22 = group_sum<LPP>(a22);
    if (active && sub == 0) {
        const Mat2 B = kgc_from_A(a11, a12, a21, a22);
        t.b11[i] = B.m11;
        t.b12[i] = B.m12;
        t.b21[i] = B.m21;
        t.b22[i] = B.m22;
    }
}

// ---------------------------------------------------------------------------------------------
// pass CD: viscous force (+gravity) [sph_physics_mex.c:469-545, SPH_Poiseuille.m:392], transport
// shift [:636-710], Riemann pressure force of integration_1st [:870-957], velocity kick
// [:1400-1408] and both position half-drifts [:863-864,:1066-1069] + periodic wrap
// [SPH_Poiseuille.m:570-577].  One sweep over the fluid ring serves all three operators because
// they share e, dW, B_i+B_j; the wall ring is swept twice because the wall pressure needs the
// complete viscous+gravity force of the particle first (p_wall uses force_prior_i, :931-934).
// ---------------------------------------------------------------------------------------------
template <int LPP>
__global__ __launch_bounds__(kBlock) void k_forces(const Clock *clk, int q, Grid g, Phys ph, int nf,
                                                   FluidSet s, FluidTmp t, Walls w)
{
    if (!clk->run[q]) return;
    const int tid = blockIdx.x * kBlock + threadIdx.x;
    const int i = tid / LPP, sub = tid % LPP;
    const bool active = i < nf;
    const double h = ph.kc.h;
    double ax = 0.0, ay = 0.0, ix = 0.0, iy = 0.0, px = 0.0, py = 0.0;
    double xi = 0.0, yi = 0.0, vxi = 0.0, vyi = 0.0, Voli = 0.0, mi = 1.0, p_i = 0.0, rhoh_i = 0.0;
    double b11i = 1.0, b12i = 0.0, b21i = 0.0, b22i = 1.0;
    int cx = 0, cy = 0;
    bool near_wall = false;
    if (active) {
        xi = s.x[i]; yi = s.y[i]; vxi = s.vx[i]; vyi = s.vy[i];
        Voli = t.Vol[i]; mi = s.mass[i]; p_i = t.ph[i]; rhoh_i = t.rhoh[i];
        b11i = t.b11[i]; b12i = t.b12[i]; b21i = t.b21[i]; b22i = t.b22[i];
        cell_of(g, xi, yi, cx, cy);
        near_wall = w.row_any[cy] != 0;
        sweep<LPP>(g, s.start, cx, cy, sub, xi, [&](int k, double xis) {
            const double dx = xis - s.x[k], dy = yi - s.y[k];
            const double r2 = dx * dx + dy * dy;
            if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                const double r = sqrt(r2), inv_r = 1.0 / r;
                const double ex = dx * inv_r, ey = dy * inv_r;
                const double dW = spline_dW(ph.kc, r);
                const double Volj = t.Vol[k];
                const double tx = (b11i + t.b11[k]) * ex + (b12i + t.b12[k]) * ey;
                const double ty = (b21i + t.b21[k]) * ex + (b22i + t.b22[k]) * ey;
                const double eBe = ex * tx + ey * ty;
                const double vxj = s.vx[k], vyj = s.vy[k];
                const double dWVj = dW * Volj;
                // viscous
                const double coeff = eBe * ph.mu * dWVj / (r + 0.01 * h);
                ax += coeff * (vxi - vxj);
                ay += coeff * (vyi - vyj);
                // transport
                ix -= dWVj * tx;
                iy -= dWVj * ty;
                // pressure (Riemann-dissipated face pressure)
                const double p_j = t.ph[k];
                const double rho_bar = 0.5 * (rhoh_i + t.rhoh[k]);
                const double un_l = vxi * ex + vyi * ey, un_r = vxj * ex + vyj * ey;
                const double beta = riemann_beta(un_l, un_r, ph.c_f);
                const double p_avg = 0.5 * (p_i + p_j);
                const double p_star = p_avg + 0.5 * beta * rho_bar * (un_l - un_r);
                const double p_face = 0.5 * (p_avg + p_star);
                px -= (p_face * tx) * dWVj;
                py -= (p_face * ty) * dWVj;
            }
        });
        if (near_wall) {
            sweep<LPP>(g, w.start, cx, cy, sub, xi, [&](int k, double xis) {
                const double dx = xis - w.x[k], dy = yi - w.y[k];
                const double r2 = dx * dx + dy * dy;
                if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                    const double r = sqrt(r2), inv_r = 1.0 / r;
                    const double ex = dx * inv_r, ey = dy * inv_r;
                    const double dWVj = spline_dW(ph.kc, r) * w.Vol[k];
                    const double tx = b11i * ex + b12i * ey, ty = b21i * ex + b22i * ey;
                    const double eBe = ex * tx + ey * ty;
                    const double coeff = 4.0 * eBe * ph.mu * dWVj / (r + 0.01 * h);
                    ax += coeff * (vxi - w.vx[k]);
                    ay += coeff * (vyi - w.vy[k]);
                    ix -= 2.0 * dWVj * tx;
                    iy -= 2.0 * dWVj * ty;
                }
            });
        }
    }
    ax = group_sum<LPP>(ax);
    ay = group_sum<LPP>(ay);
    ix = group_sum<LPP>(ix);
    iy = group_sum<LPP>(iy);
    const double fpx = ax * Voli + mi * ph.g;  // + gravity, SPH_Poiseuille.m:392
    const double fpy = ay * Voli;
    if (active && near_wall) {
        const double acx = fpx / mi, acy = fpy / mi;
        sweep<LPP>(g, w.start, cx, cy, sub, xi, [&](int k, double xis) {
            const double dx = xis - w.x[k], dy = yi - w.y[k];
            const double r2 = dx * dx + dy * dy;
            if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                const double r = sqrt(r2), inv_r = 1.0 / r;
                const double ex = dx * inv_r, ey = dy * inv_r;
                const double dWVj = spline_dW(ph.kc, r) * w.Vol[k];
                const double face = -(acx * ex + acy * ey);
                const double p_wall = p_i + rhoh_i * r * fmax(0.0, face);
                const double tx = b11i * ex + b12i * ey, ty = b21i * ex + b22i * ey;
                px -= (p_i + p_wall) * dWVj * tx;
                py -= (p_i + p_wall) * dWVj * ty;
            }
        });
    }
    px = group_sum<LPP>(px);
    py = group_sum<LPP>(py);
    if (active && sub == 0) {
        const double dt = clk->dt;
        const double fx = px * Voli, fy = py * Voli;
        const double inv_m = 1.0 / mi;
        const double vxn = vxi + (fpx + fx) * inv_m * dt;
        const double vyn = vyi + (fpy + fy) * inv_m * dt;
        double sx, sy;
        transport_shift(ix, iy, h, ph.tc, sx, sy);
        double xo = xi + sx, yo = yi + sy;
        xo += 0.5 * dt * vxi;
        yo += 0.5 * dt * vyi;
        xo += 0.5 * dt * vxn;
        yo += 0.5 * dt * vyn;
        t.xn[i] = wrap_x(xo, ph.DL);
        t.yn[i] = yo;
        t.vxn[i] = vxn;
        t.vyn[i] = vyn;
        t.fpx[i] = fpx;
        t.fpy[i] = fpy;
        t.fx[i] = fx;
        t.fy[i] = fy;
    }
}

// ---------------------------------------------------------------------------------------------
// pass E: continuity rate with the kicked velocities (integration_2nd, sph_physics_mex.c:1076-1116),
// final half-step of rho and EOS (:1440-1450), per-block max |v|^2 for the next dt
// (SPH_Poiseuille.m:521) and the cell histogram of the end-of-step positions (neighbour rebuild,
// the K0 insert of mex/sph_neighbor_search_mex.c:269-296).
// ---------------------------------------------------------------------------------------------
template <int LPP>
__global__ __launch_bounds__(kBlock) void k_continuity(const Clock *clk, int q, Grid g, Phys ph,
                                                       int nf, FluidSet s, FluidTmp t, Walls w)
{
    if (!clk->run[q]) return;
    const int tid = blockIdx.x * kBlock + threadIdx.x;
    const int i = tid / LPP, sub = tid % LPP;
    const bool active = i < nf;
    double rate = 0.0, v2 = 0.0;
    double vxi = 0.0, vyi = 0.0;
    if (active) {
        const double xi = s.x[i], yi = s.y[i];
        vxi = t.vxn[i];
        vyi = t.vyn[i];
        int cx, cy;
        cell_of(g, xi, yi, cx, cy);
        sweep<LPP>(g, s.start, cx, cy, sub, xi, [&](int k, double xis) {
            const double dx = xis - s.x[k], dy = yi - s.y[k];
            const double r2 = dx * dx + dy * dy;
            if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                const double r = sqrt(r2), inv_r = 1.0 / r;
                const double ex = dx * inv_r, ey = dy * inv_r;
                const double u_jump = (vxi - t.vxn[k]) * ex + (vyi - t.vyn[k]) * ey;
                rate += u_jump * spline_dW(ph.kc, r) * t.Vol[k];
            }
        });
        if (w.row_any[cy]) {
            sweep<LPP>(g, w.start, cx, cy, sub, xi, [&](int k, double xis) {
                const double dx = xis - w.x[k], dy = yi - w.y[k];
                const double r2 = dx * dx + dy * dy;
                if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                    const double r = sqrt(r2), inv_r = 1.0 / r;
                    const double ex = dx * inv_r, ey = dy * inv_r;
                    const double vjx = 2.0 * w.vx[k] - vxi, vjy = 2.0 * w.vy[k] - vyi;
                    const double jump = (vxi - vjx) * ex + (vyi - vjy) * ey;
                    rate += jump * spline_dW(ph.kc, r) * w.Vol[k];
                }
            });
        }
    }
    rate = group_sum<LPP>(rate);
    if (active && sub == 0) {
        const double dt = clk->dt;
        const double rhoh = t.rhoh[i];
        const double drho_new = rate * rhoh;
        double rho = rhoh + drho_new * (0.5 * dt);
        if (rho < 1e-10) rho = ph.rho0;
        t.drhon[i] = drho_new;
        t.rho_out[i] = rho;
        t.p_out[i] = eos_pressure(rho, ph.rho0, ph.p0);
        v2 = vxi * vxi + vyi * vyi;
        int cx, cy;
        cell_of(g, t.xn[i], t.yn[i], cx, cy);
        const int c = cx * g.ncy + cy;
        t.cellid[i] = c;
        atomicAdd(&t.count[c], 1);
    }
    // block max of |v|^2 (NaN poisons the max on purpose: v2 != v2 -> +inf)
    if (v2 != v2) v2 = INFINITY;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v2 = fmax(v2, __shfl_xor(v2, off));
    __shared__ double s_max[kBlock / 64];
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = v2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = s_max[0];
        for (int k = 1; k < kBlock / 64; ++k) m = fmax(m, s_max[k]);
        t.vpart[blockIdx.x] = m;
    }
}

// standalone cell histogram (context creation / wall grid): same binning as pass E's epilogue
__global__ __launch_bounds__(kBlock) void k_bin(Grid g, int n, const double *x, const double *y,
                                                int *cellid, int *count)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int cx, cy;
    cell_of(g, x[i], y[i], cx, cy);
    const int c = cx * g.ncy + cy;
    cellid[i] = c;
    atomicAdd(&count[c], 1);
}

// block-wide exclusive scan of one int per thread (kScanBlock threads); returns the block total
__device__ __forceinline__ int block_exclusive_scan(int v, int &total, int *s_wave /*[16]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        int ws = lane < kScanBlock / 64 ? s_wave[lane] : 0;
        int winc = ws;
#pragma unroll
        for (int off = 1; off < kScanBlock / 64; off <<= 1) {
            const int o = __shfl_up(winc, off);
            if (lane >= off) winc += o;
        }
        if (lane < kScanBlock / 64) s_wave[lane] = winc - ws;  // exclusive wave offsets
        if (lane == kScanBlock / 64 - 1) s_wave[kScanBlock / 64] = winc;
    }
    __syncthreads();
    const int res = s_wave[wave] + inc - v;
    total = s_wave[kScanBlock / 64];
    __syncthreads();
    return res;
}

// exclusive scan of count[0..n) into start[0..n], single block (n_cells is small next to n_particles)
__device__ __forceinline__ void scan_counts(const int *count, int *start, int n)
{
    __shared__ int s_wave[kScanBlock / 64 + 1];
    int carry = 0;
    for (int base = 0; base < n; base += kScanBlock) {
        const int idx = base + (int)threadIdx.x;
        const int v = idx < n ? count[idx] : 0;
        int total;
        const int ex = block_exclusive_scan(v, total, s_wave);
        if (idx < n) start[idx] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) start[n] = carry;
}

// Step kernel 5: finish the clock of this step (vmax -> next dt, t += dt, stop test) and scan the
// cell histogram.  Single block.
__global__ __launch_bounds__(kScanBlock) void k_clock_scan(Clock *clk, int q, Phys ph, int n_vpart,
                                                           const double *vpart, const int *count,
                                                           int *start_next, int ncells)
{
    if (!clk->run[q]) {
        if (threadIdx.x == 0) clk->run[1 - q] = 0;
        return;
    }
    double m = 0.0;
    for (int k = threadIdx.x; k < n_vpart; k += kScanBlock) m = fmax(m, vpart[k]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    __shared__ double s_m[kScanBlock / 64];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kScanBlock / 64; ++k) m = fmax(m, s_m[k]);
        Clock c = *clk;
        c.vmax = sqrt(m);  // max of sqrt == sqrt of max (monotone, correctly rounded)
        c.t += c.dt;       // SPH_Poiseuille.m:267
        c.dt_last = c.dt;
        c.step += 1;
        if (c.steps_left > 0) c.steps_left -= 1;
        if (isinf(m)) c.status = SPHX_ERR_DIVERGED;
        c.dt = next_dt(c, ph);
        c.run[1 - q] = loop_continues(c) ? 1 : 0;
        *clk = c;
    }
    scan_counts(count, start_next, ncells);
}

__global__ __launch_bounds__(kScanBlock) void k_scan_only(const int *count, int *start, int n)
{
    scan_counts(count, start, n);
}

// Step kernel 6: place every particle index into its cell range (arrival order, made canonical by
// k_reorder).  atomicSub counts the histogram back down to zero, ready for the next step.
__global__ __launch_bounds__(kBlock) void k_scatter(const Clock *clk, int q, int n, const int *cellid,
                                                    int *count, const int *start_next, int *perm)
{
    if (clk && !clk->run[q]) return;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int c = cellid[i];
    const int k = atomicSub(&count[c], 1) - 1;
    perm[start_next[c] + k] = i;
}

struct ReorderArgs {
    int nd;
    const double *src[8];
    double *dst[8];
    const int *id_src;
    int *id_dst;
    int *src_of;
};

// Step kernel 7: canonical rank inside the cell (ascending previous slot -> deterministic order) and
// the gather of every persistent field into the new ordering.
__global__ __launch_bounds__(kBlock) void k_reorder(const Clock *clk, int q, int n, const int *cellid,
                                                    const int *start_next, const int *perm,
                                                    ReorderArgs a)
{
    if (clk && !clk->run[q]) return;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int c = cellid[i];
    const int lo = start_next[c], hi = start_next[c + 1];
    int rank = 0;
    for (int k = lo; k < hi; ++k) rank += (perm[k] < i) ? 1 : 0;
    const int dst = lo + rank;
#pragma unroll
    for (int f = 0; f < 8; ++f)
        if (f < a.nd) a.dst[f][dst] = a.src[f][i];
    a.id_dst[dst] = a.id_src[i];
    if (a.src_of) a.src_of[dst] = i;
}

__global__ __launch_bounds__(kBlock) void k_iota(int n, int *a, int base)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) a[i] = base + i;
}

__global__ __launch_bounds__(kBlock) void k_wrap_x(int n, double *x, double DL)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) x[i] = wrap_x(x[i], DL);
}

__global__ __launch_bounds__(kBlock) void k_wall_volume(int n, const double *mass, double rho0, double *Vol)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) Vol[i] = mass[i] / rho0;  // walls keep rho = rho0 (sph_physics_mex.c:214-216,233)
}

__global__ void k_row_any(Grid g, const int *wstart, int *row_any)
{
    const int cy = blockIdx.x * blockDim.x + threadIdx.x;
    if (cy >= g.ncy) return;
    int any = 0;
    for (int r = max(cy - 1, 0); r <= min(cy + 1, g.ncy - 1) && !any; ++r)
        for (int cx = 0; cx < g.ncx; ++cx) {
            const int c = cx * g.ncy + r;
            if (wstart[c + 1] > wstart[c]) { any = 1; break; }
        }
    row_any[cy] = any;
}

// initial max |v| (vecnorm over the fluid, SPH_Poiseuille.m:521); single block
__global__ __launch_bounds__(kScanBlock) void k_vmax_init(Clock *clk, int nf, const double *vx, const double *vy)
{
    double m = 0.0;
    for (int k = threadIdx.x; k < nf; k += kScanBlock) {
        double v2 = vx[k] * vx[k] + vy[k] * vy[k];
        if (v2 != v2) v2 = INFINITY;
        m = fmax(m, v2);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    __shared__ double s_m[kScanBlock / 64];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kScanBlock / 64; ++k) m = fmax(m, s_m[k]);
        clk->vmax = sqrt(m);
    }
}

// ---------------------------------------------------------------------------------------------
// monitors and pair-list emission on the current ordering
// ---------------------------------------------------------------------------------------------

// wall shear (sph_physics_mex.c:1713-1742): new neighbour structure, new pos/vel, Vol/B of the step
// that just finished (reached through src_of).  Per-block partial sums, reduced by k_tau_final.
__global__ __launch_bounds__(kBlock) void k_wall_shear(Grid g, Phys ph, int nf, FluidSet s, FluidTmp t,
                                                       Walls w, int have_src, double *part /*[2*grid]*/)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    double fb = 0.0, ft = 0.0;
    if (i < nf) {
        const double xi = s.x[i], yi = s.y[i];
        int cx, cy;
        cell_of(g, xi, yi, cx, cy);
        if (w.row_any[cy]) {
            const int o = have_src ? t.src_of[i] : i;
            const double Voli = t.Vol[o];
            const double b11 = t.b11[o], b12 = t.b12[o], b21 = t.b21[o], b22 = t.b22[o];
            const double vxi = s.vx[i];
            sweep<1>(g, w.start, cx, cy, 0, xi, [&](int k, double xis) {
                const double dx = xis - w.x[k], dy = yi - w.y[k];
                const double r2 = dx * dx + dy * dy;
                if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                    const double r = sqrt(r2);
                    const double ex = dx / r, ey = dy / r;
                    const double eBe = ex * (b11 * ex + b12 * ey) + ey * (b21 * ex + b22 * ey);
                    const double f = 4.0 * ph.mu * eBe * spline_dW(ph.kc, r) * w.Vol[k] * (vxi - w.vx[k]) /
                                     (r + 0.01 * ph.kc.h) * Voli;
                    const double yj = w.y[k];
                    if (yj <= 0.0) fb += f;
                    else if (yj >= ph.DH) ft += f;
                }
            });
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        fb += __shfl_xor(fb, off);
        ft += __shfl_xor(ft, off);
    }
    __shared__ double sb[kBlock / 64], st[kBlock / 64];
    if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = fb; st[threadIdx.x >> 6] = ft; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < kBlock / 64; ++k) { a += sb[k]; b += st[k]; }
        part[2 * blockIdx.x] = a;
        part[2 * blockIdx.x + 1] = b;
    }
}

__global__ __launch_bounds__(kScanBlock) void k_tau_final(int nblk, const double *part, double DL, double *out)
{
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nblk; k += kScanBlock) { a += part[2 * k]; b += part[2 * k + 1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    __shared__ double sa[kScanBlock / 64], sb[kScanBlock / 64];
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0.0; b = 0.0;
        for (int k = 0; k < kScanBlock / 64; ++k) { a += sa[k]; b += sb[k]; }
        out[0] = -a / DL;
        out[1] = -b / DL;
    }
}

// Pair emission in the MEX convention (sph_neighbor_search_mex.c:353-383): a fluid-fluid pair is
// produced once, from the particle with the smaller ORIGINAL index; fluid-wall pairs always.
// mode 0: count into cnt[orig]; mode 1: write at off[orig].
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_pairs(Grid g, Phys ph, int nf, FluidSet s, Walls w, int *cnt,
                                                  const int *off, double *o_i, double *o_j, double *o_dx,
                                                  double *o_dy, double *o_r, double *o_W, double *o_dW)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nf) return;
    const double xi = s.x[i], yi = s.y[i];
    const int a = s.id[i];
    int cx, cy;
    cell_of(g, xi, yi, cx, cy);
    int n = 0;
    const int base = MODE ? off[a] : 0;
    auto emit = [&](int b, double dx, double dy, double r2) {
        if (MODE) {
            const double r = sqrt(r2);
            double W, dW;
            spline(ph.kc, r, W, dW);
            const int p = base + n;
            o_i[p] = (double)(a + 1);
            o_j[p] = (double)(b + 1);
            o_dx[p] = dx; o_dy[p] = dy; o_r[p] = r; o_W[p] = W; o_dW[p] = dW;
        }
        ++n;
    };
    sweep<1>(g, s.start, cx, cy, 0, xi, [&](int k, double xis) {
        const double dx = xis - s.x[k], dy = yi - s.y[k];
        const double r2 = dx * dx + dy * dy;
        if (r2 > kR2Min && r2 < ph.kc.rcut2) {
            const int b = s.id[k];
            if (b > a) emit(b, dx, dy, r2);
        }
    });
    if (w.row_any[cy]) {
        sweep<1>(g, w.start, cx, cy, 0, xi, [&](int k, double xis) {
            const double dx = xis - w.x[k], dy = yi - w.y[k];
            const double r2 = dx * dx + dy * dy;
            if (r2 > kR2Min && r2 < ph.kc.rcut2) emit(w.id[k], dx, dy, r2);
        });
    }
    if (!MODE) cnt[a] = n;
}

// scatter a sorted field back to the caller's row numbering
__global__ __launch_bounds__(kBlock) void k_unsort(int n, const int *id, const double *src, double *dst)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[id[i]] = src[i];
}

__global__ __launch_bounds__(kBlock) void k_fill(int n, double *dst, double v)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = v;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct KernelTimer {
    std::vector<std::string> names;
    std::vector<double> total_ms;
    std::vector<int64_t> launches;
    struct Pending { int idx; hipEvent_t a, b; };
    std::vector<Pending> pending;
    int index_of(const char *name)
    {
        for (size_t k = 0; k < names.size(); ++k) if (names[k] == name) return (int)k;
        names.emplace_back(name); total_ms.push_back(0.0); launches.push_back(0);
        return (int)names.size() - 1;
    }
    void collect()
    {
        for (auto &p : pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { total_ms[p.idx] += ms; launches[p.idx] += 1; }
            (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b);
        }
        pending.clear();
    }
};

}  // namespace sphx

using namespace sphx;

struct sphx_ctx {
    sphx_params prm{};
    Grid grid{};
    Phys phys{};
    int nf = 0, nw = 0, nt = 0;
    int lpp = 1;
    int spg = 2;
    int cur = 0;  // which FluidSet holds the current state
    int64_t step_at_cur0 = 0;
    bool have_step_outputs = false;
    hipStream_t stream = nullptr;

    // storage
    DevBuf<double> fx_[2], fy_[2], fvx_[2], fvy_[2], fdrho_[2], fmass_[2];
    DevBuf<int> fid_[2], fstart_[2];
    DevBuf<double> xn, yn, vxn, vyn, drhon, rho, Vol, rhoh, ph, b11, b12, b21, b22, fpx, fpy, ffx, ffy, rho_out, p_out, vpart;
    DevBuf<int> cellid, count, perm, src_of;
    DevBuf<double> wx, wy, wVol, wvx, wvy;
    DevBuf<int> wid, wstart, wrow_any;
    DevBuf<Clock> clock;
    DevBuf<double> tau_part, tau_out;
    Clock *h_clock = nullptr;  // pinned

    FluidSet set[2]{};
    FluidTmp tmp{};
    Walls walls{};
    int n_blocks_particles = 0;  // grid of the LPP kernels

    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    bool profiling = false;
    KernelTimer timer;

    // pair list held for sphx_neighbor_fetch
    DevBuf<double> pl_i, pl_j, pl_dx, pl_dy, pl_r, pl_W, pl_dW;
    size_t pl_n = 0;
    bool pl_valid = false;

    ~sphx_ctx()
    {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (h_clock) (void)hipHostFree(h_clock);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

template <typename K, typename... Args>
void launch(sphx_ctx *c, const char *name, K kernel, dim3 grid, dim3 block, Args... args)
{
    if (c->profiling) {
        KernelTimer::Pending p;
        p.idx = c->timer.index_of(name);
        SPHX_HIP(hipEventCreate(&p.a));
        SPHX_HIP(hipEventCreate(&p.b));
        SPHX_HIP(hipEventRecord(p.a, c->stream));
        hipLaunchKernelGGL(kernel, grid, block, 0, c->stream, args...);
        SPHX_HIP(hipEventRecord(p.b, c->stream));
        c->timer.pending.push_back(p);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, 0, c->stream, args...);
    }
}

template <int LPP>
void launch_step_lpp(sphx_ctx *c, int q)
{
    const dim3 gp(c->n_blocks_particles), bp(kBlock);
    const dim3 g1(div_up(c->nf, kBlock));
    Clock *clk = c->clock.get();
    const FluidSet &s = c->set[q];
    const FluidSet &d = c->set[1 - q];
    launch(c, "k_density", k_density<LPP>, gp, bp, (const Clock *)clk, q, c->grid, c->phys, c->nf, s, c->tmp, c->walls);
    launch(c, "k_kgc", k_kgc<LPP>, gp, bp, (const Clock *)clk, q, c->grid, c->phys, c->nf, s, c->tmp, c->walls);
    launch(c, "k_forces", k_forces<LPP>, gp, bp, (const Clock *)clk, q, c->grid, c->phys, c->nf, s, c->tmp, c->walls);
    launch(c, "k_continuity", k_continuity<LPP>, gp, bp, (const Clock *)clk, q, c->grid, c->phys, c->nf, s, c->tmp, c->walls);
    launch(c, "k_clock_scan", k_clock_scan, dim3(1), dim3(kScanBlock), clk, q, c->phys, c->n_blocks_particles,
           (const double *)c->vpart.get(), (const int *)c->count.get(), d.start, c->grid.ncells);
    launch(c, "k_scatter", k_scatter, g1, bp, (const Clock *)clk, q, c->nf, (const int *)c->cellid.get(), c->count.get(),
           (const int *)d.start, c->perm.get());
    ReorderArgs ra{};
    ra.nd = 6;
    ra.src[0] = c->tmp.xn; ra.dst[0] = d.x;
    ra.src[1] = c->tmp.yn; ra.dst[1] = d.y;
    ra.src[2] = c->tmp.vxn; ra.dst[2] = d.vx;
    ra.src[3] = c->tmp.vyn; ra.dst[3] = d.vy;
    ra.src[4] = c->tmp.drhon; ra.dst[4] = d.drho;
    ra.src[5] = s.mass; ra.dst[5] = d.mass;
    ra.id_src = s.id; ra.id_dst = d.id; ra.src_of = c->tmp.src_of;
    launch(c, "k_reorder", k_reorder, g1, bp, (const Clock *)clk, q, c->nf, (const int *)c->cellid.get(),
           (const int *)d.start, (const int *)c->perm.get(), ra);
}

void launch_step(sphx_ctx *c, int q)
{
    switch (c->lpp) {
        case 1: launch_step_lpp<1>(c, q); break;
        case 2: launch_step_lpp<2>(c, q); break;
        case 4: launch_step_lpp<4>(c, q); break;
        case 8: launch_step_lpp<8>(c, q); break;
        case 16: launch_step_lpp<16>(c, q); break;
        case 32: launch_step_lpp<32>(c, q); break;
        default: throw Error(SPHX_ERR_ARG, "SPHX:Ctx:lpp", "lanes_per_particle must be 1,2,4,8,16 or 32");
    }
}

void build_graph(sphx_ctx *c)
{
    if (c->graph_exec) return;
    const bool prof = c->profiling;
    c->profiling = false;
    SPHX_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < c->spg; ++k) launch_step(c, k & 1);
    SPHX_HIP(hipStreamEndCapture(c->stream, &c->graph));
    SPHX_HIP(hipGraphInstantiate(&c->graph_exec, c->graph, nullptr, nullptr, 0));
    c->profiling = prof;
}

// enqueue `slots` step slots starting at parity c->cur (slots that find run[q]==0 are no-ops)
void enqueue_slots(sphx_ctx *c, int64_t slots)
{
    int q = c->cur;
    int64_t left = slots;
    if (left > 0 && q == 1) { launch_step(c, 1); q = 0; --left; }
    if (!c->profiling && left >= c->spg) {
        build_graph(c);
        while (left >= c->spg) { SPHX_HIP(hipGraphLaunch(c->graph_exec, c->stream)); left -= c->spg; }
    }
    while (left > 0) { launch_step(c, q); q ^= 1; --left; }
    SPHX_HIP(hipGetLastError());
}

void read_clock(sphx_ctx *c)
{
    SPHX_HIP(hipMemcpyAsync(c->h_clock, c->clock.get(), sizeof(Clock), hipMemcpyDeviceToHost, c->stream));
    SPHX_HIP(hipStreamSynchronize(c->stream));
    if (c->profiling) c->timer.collect();
    const int64_t executed = (int64_t)c->h_clock->step - c->step_at_cur0;
    const int new_cur = (int)(executed & 1);
    if (executed > 0) c->have_step_outputs = true;
    c->cur = new_cur;
}

void fill_status(sphx_ctx *c, sphx_status *st)
{
    if (!st) return;
    const Clock &k = *c->h_clock;
    st->t = k.t;
    st->dt_last = k.dt_last;
    st->dt_next = k.dt;
    st->vmax = k.vmax;
    st->step = k.step;
    st->done = (k.t < k.t_target - 1e-12) ? 0 : 1;
    st->device_status = k.status;
}

int pick_lpp(int nf)
{
    // enough lanes to put ~2 waves on each of the 1024 SIMDs, never more than 16 lanes per particle
    const long target = 256L * 4 * 2 * 64;
    int lpp = 1;
    while (lpp < 16 && (long)nf * lpp * 2 <= target) lpp *= 2;
    return lpp;
}

// sort `n` particles given in arbitrary order (x,y in tx,ty) into cell order; generic over the field list
void initial_sort(sphx_ctx *c, int n, const double *x, const double *y, int *cellid, int *count, int *start,
                  int *perm, const ReorderArgs &ra)
{
    if (n <= 0) {
        SPHX_HIP(hipMemsetAsync(start, 0, ((size_t)c->grid.ncells + 1) * sizeof(int), c->stream));
        return;
    }
    hipLaunchKernelGGL(k_bin, dim3(div_up(n, kBlock)), dim3(kBlock), 0, c->stream, c->grid, n, x, y, cellid, count);
    hipLaunchKernelGGL(k_scan_only, dim3(1), dim3(kScanBlock), 0, c->stream, (const int *)count, start, c->grid.ncells);
    hipLaunchKernelGGL(k_scatter, dim3(div_up(n, kBlock)), dim3(kBlock), 0, c->stream, (const Clock *)nullptr, 0, n,
                       (const int *)cellid, count, (const int *)start, perm);
    hipLaunchKernelGGL(k_reorder, dim3(div_up(n, kBlock)), dim3(kBlock), 0, c->stream, (const Clock *)nullptr, 0, n,
                       (const int *)cellid, (const int *)start, (const int *)perm, ra);
    SPHX_HIP(hipGetLastError());
}

void ctx_setup(sphx_ctx *c, const sphx_params *prm, int n_fluid, int n_total, const double *pos, const double *vel,
               const double *drho_dt, const double *mass, const double *wall_vel, double t0, int64_t step0)
{
    require(prm != nullptr, "SPHX:Ctx:params", "params must not be NULL");
    require(n_total > 0 && n_fluid > 0 && n_fluid <= n_total, "SPH:Neighbor:count",
            "Invalid n_fluid/n_total or inconsistent pos size.");
    require(prm->h > 0.0 && prm->DL > 0.0, "SPH:Neighbor:param", "h and DL must be positive.");
    ensure_device();
    c->prm = *prm;
    c->nf = n_fluid;
    c->nt = n_total;
    c->nw = n_total - n_fluid;
    const int nf = c->nf, nw = c->nw;
    const size_t ntz = (size_t)n_total;
    const double *px = pos, *py = pos + ntz;

    // grid: exact periodic tiling in x (cells >= 2h), 2h rows in y over fluid + wall extent
    double y_min = py[0], y_max = py[0];
    for (int i = 1; i < n_total; ++i) { y_min = std::min(y_min, py[i]); y_max = std::max(y_max, py[i]); }
    require(std::isfinite(y_min) && std::isfinite(y_max), "SPH:Neighbor:pos", "pos must be finite.");
    const double cs = 2.0 * prm->h;
    Grid g{};
    g.ncx = (int)std::floor(prm->DL / cs);
    require(g.ncx >= 3, "SPH:Neighbor:param", "device path needs DL >= 6h (three periodic cell columns).");
    g.ncy = (int)std::ceil((y_max - y_min + 1e-12) / cs) + 1;
    require((double)g.ncx * (double)g.ncy < 2.0e9, "SPH:Neighbor:param", "cell grid too large.");
    g.ncells = g.ncx * g.ncy;
    g.DL = prm->DL;
    g.y0 = y_min;
    g.inv_csx = (double)g.ncx / prm->DL;
    g.inv_csy = 1.0 / cs;
    c->grid = g;

    Phys ph{};
    ph.kc = make_kernel_const(prm->h);
    ph.rho0 = prm->rho0; ph.inv_sigma0 = prm->inv_sigma0; ph.mu = prm->mu; ph.p0 = prm->p0; ph.c_f = prm->c_f;
    ph.g = prm->gravity_g; ph.tc = prm->transport_coeff; ph.nu = prm->mu / prm->rho0; ph.DL = prm->DL; ph.DH = prm->DH;
    ph.w0 = ph.kc.sigma;
    c->phys = ph;

    c->lpp = prm->lanes_per_particle > 0 ? prm->lanes_per_particle : pick_lpp(nf);
    require(c->lpp == 1 || c->lpp == 2 || c->lpp == 4 || c->lpp == 8 || c->lpp == 16 || c->lpp == 32, "SPHX:Ctx:lpp",
            "lanes_per_particle must be 1,2,4,8,16 or 32");
    c->spg = prm->steps_per_graph > 0 ? prm->steps_per_graph : 16;
    if (c->spg & 1) c->spg += 1;
    c->n_blocks_particles = (int)div_up((size_t)nf * c->lpp, kBlock);

    SPHX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    SPHX_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_clock), sizeof(Clock), hipHostMallocDefault));

    for (int k = 0; k < 2; ++k) {
        c->fx_[k].alloc(nf); c->fy_[k].alloc(nf); c->fvx_[k].alloc(nf); c->fvy_[k].alloc(nf);
        c->fdrho_[k].alloc(nf); c->fmass_[k].alloc(nf); c->fid_[k].alloc(nf);
        c->fstart_[k].alloc((size_t)g.ncells + 1);
        c->set[k] = FluidSet{c->fx_[k].get(), c->fy_[k].get(), c->fvx_[k].get(), c->fvy_[k].get(), c->fdrho_[k].get(),
                             c->fmass_[k].get(), c->fid_[k].get(), c->fstart_[k].get()};
    }
    DevBuf<double> *dbl[] = {&c->xn, &c->yn, &c->vxn, &c->vyn, &c->drhon, &c->rho, &c->Vol, &c->rhoh, &c->ph, &c->b11,
                             &c->b12, &c->b21, &c->b22, &c->fpx, &c->fpy, &c->ffx, &c->ffy, &c->rho_out, &c->p_out};
    for (auto *b : dbl) { b->alloc(nf); b->zero(c->stream); }
    c->vpart.alloc(c->n_blocks_particles);
    c->cellid.alloc(nf); c->count.alloc((size_t)g.ncells + 1); c->perm.alloc(nf); c->src_of.alloc(nf);
    c->count.zero(c->stream);
    c->tmp = FluidTmp{c->xn.get(), c->yn.get(), c->vxn.get(), c->vyn.get(), c->drhon.get(), c->rho.get(), c->Vol.get(),
                      c->rhoh.get(), c->ph.get(), c->b11.get(), c->b12.get(), c->b21.get(), c->b22.get(), c->fpx.get(),
                      c->fpy.get(), c->ffx.get(), c->ffy.get(), c->rho_out.get(), c->p_out.get(), c->cellid.get(),
                      c->count.get(), c->perm.get(), c->src_of.get(), c->vpart.get()};

    // ---- fluid: upload in caller order into the "temp" arrays, wrap x, sort into set[0]
    hipStream_t s = c->stream;
    c->xn.upload(px, nf, s); c->yn.upload(py, nf, s);
    c->vxn.upload(vel, nf, s); c->vyn.upload(vel + ntz, nf, s);
    c->drhon.upload(drho_dt, nf, s);
    c->fmass_[1].upload(mass, nf, s);
    hipLaunchKernelGGL(k_iota, dim3(div_up(nf, kBlock)), dim3(kBlock), 0, s, nf, c->fid_[1].get(), 0);
    hipLaunchKernelGGL(k_wrap_x, dim3(div_up(nf, kBlock)), dim3(kBlock), 0, s, nf, c->xn.get(), prm->DL);
    {
        ReorderArgs ra{};
        ra.nd = 6;
        const double *src[6] = {c->xn.get(), c->yn.get(), c->vxn.get(), c->vyn.get(), c->drhon.get(), c->fmass_[1].get()};
        double *dst[6] = {c->set[0].x, c->set[0].y, c->set[0].vx, c->set[0].vy, c->set[0].drho, c->set[0].mass};
        for (int f = 0; f < 6; ++f) { ra.src[f] = src[f]; ra.dst[f] = dst[f]; }
        ra.id_src = c->fid_[1].get(); ra.id_dst = c->set[0].id; ra.src_of = nullptr;
        initial_sort(c, nf, c->xn.get(), c->yn.get(), c->cellid.get(), c->count.get(), c->set[0].start, c->perm.get(), ra);
    }

    // ---- walls: static, sorted once on the same grid
    const size_t nwz = nw > 0 ? (size_t)nw : 1;
    c->wx.alloc(nwz); c->wy.alloc(nwz); c->wVol.alloc(nwz); c->wvx.alloc(nwz); c->wvy.alloc(nwz); c->wid.alloc(nwz);
    c->wstart.alloc((size_t)g.ncells + 1); c->wrow_any.alloc(g.ncy);
    {
        DevBuf<double> tx(nwz), ty(nwz), tm(nwz), tV(nwz), tvx(nwz), tvy(nwz);
        DevBuf<int> tid(nwz), tcell(nwz), tperm(nwz), tcount((size_t)g.ncells + 1);
        tcount.zero(s);
        if (nw > 0) {
            tx.upload(px + nf, nw, s); ty.upload(py + nf, nw, s); tm.upload(mass + nf, nw, s);
            tvx.upload(wall_vel + nf, nw, s); tvy.upload(wall_vel + ntz + nf, nw, s);
            hipLaunchKernelGGL(k_iota, dim3(div_up(nw, kBlock)), dim3(kBlock), 0, s, nw, tid.get(), nf);
            hipLaunchKernelGGL(k_wrap_x, dim3(div_up(nw, kBlock)), dim3(kBlock), 0, s, nw, tx.get(), prm->DL);
            hipLaunchKernelGGL(k_wall_volume, dim3(div_up(nw, kBlock)), dim3(kBlock), 0, s, nw, (const double *)tm.get(),
                               prm->rho0, tV.get());
        }
        ReorderArgs ra{};
        ra.nd = 5;
        const double *src[5] = {tx.get(), ty.get(), tV.get(), tvx.get(), tvy.get()};
        double *dst[5] = {c->wx.get(), c->wy.get(), c->wVol.get(), c->wvx.get(), c->wvy.get()};
        for (int f = 0; f < 5; ++f) { ra.src[f] = src[f]; ra.dst[f] = dst[f]; }
        ra.id_src = tid.get(); ra.id_dst = c->wid.get(); ra.src_of = nullptr;
        initial_sort(c, nw, tx.get(), ty.get(), tcell.get(), tcount.get(), c->wstart.get(), tperm.get(), ra);
        hipLaunchKernelGGL(k_row_any, dim3(div_up(g.ncy, 64)), dim3(64), 0, s, g, (const int *)c->wstart.get(),
                           c->wrow_any.get());
        SPHX_HIP(hipGetLastError());
        SPHX_HIP(hipStreamSynchronize(s));  // temporaries die here
    }
    c->walls = Walls{c->wx.get(), c->wy.get(), c->wVol.get(), c->wvx.get(), c->wvy.get(), c->wid.get(),
                     c->wstart.get(), c->wrow_any.get(), nw};

    // ---- clock
    c->clock.alloc(1);
    Clock k{};
    k.t = t0; k.dt = 0.0; k.dt_last = 0.0; k.t_target = t0; k.t_end = prm->t_end; k.vmax = 0.0;
    k.step = step0; k.steps_left = -1; k.run[0] = 0; k.run[1] = 0; k.status = 0;
    *c->h_clock = k;
    SPHX_HIP(hipMemcpyAsync(c->clock.get(), c->h_clock, sizeof(Clock), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_vmax_init, dim3(1), dim3(kScanBlock), 0, s, c->clock.get(), nf, (const double *)c->set[0].vx,
                       (const double *)c->set[0].vy);
    SPHX_HIP(hipGetLastError());
    c->cur = 0;
    c->step_at_cur0 = step0;
    const int nblk = (int)div_up(nf, kBlock);
    c->tau_part.alloc((size_t)2 * nblk);
    c->tau_out.alloc(2);
    read_clock(c);
}

// emit the MEX-convention pair list of the current ordering into the ctx-held buffers
void emit_pairs(sphx_ctx *c)
{
    const int nf = c->nf;
    hipStream_t s = c->stream;
    DevBuf<int> cnt(nf), off((size_t)nf + 1);
    const FluidSet &fs = c->set[c->cur];
    const dim3 g1(div_up(nf, kBlock)), b1(kBlock);
    hipLaunchKernelGGL(k_pairs<0>, g1, b1, 0, s, c->grid, c->phys, nf, fs, c->walls, cnt.get(), (const int *)nullptr,
                       (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr,
                       (double *)nullptr, (double *)nullptr);
    hipLaunchKernelGGL(k_scan_only, dim3(1), dim3(kScanBlock), 0, s, (const int *)cnt.get(), off.get(), nf);
    int total = 0;
    SPHX_HIP(hipMemcpyAsync(&total, off.get() + nf, sizeof(int), hipMemcpyDeviceToHost, s));
    SPHX_HIP(hipStreamSynchronize(s));
    const size_t n = (size_t)total, m = n ? n : 1;
    c->pl_i.alloc(m); c->pl_j.alloc(m); c->pl_dx.alloc(m); c->pl_dy.alloc(m); c->pl_r.alloc(m); c->pl_W.alloc(m); c->pl_dW.alloc(m);
    hipLaunchKernelGGL(k_pairs<1>, g1, b1, 0, s, c->grid, c->phys, nf, fs, c->walls, (int *)nullptr, (const int *)off.get(),
                       c->pl_i.get(), c->pl_j.get(), c->pl_dx.get(), c->pl_dy.get(), c->pl_r.get(), c->pl_W.get(), c->pl_dW.get());
    SPHX_HIP(hipGetLastError());
    SPHX_HIP(hipStreamSynchronize(s));
    c->pl_n = n;
    c->pl_valid = true;
}

thread_local sphx_ctx *g_search_ctx = nullptr;  // owns the temporary context of sphx_neighbor_search until fetch
thread_local sphx_ctx *g_fetch_src = nullptr;   // context whose pair list the next sphx_neighbor_fetch copies out

}  // namespace

SPHX_EXPORT int sphx_ctx_create(sphx_ctx **out, const sphx_params *prm, int n_fluid, int n_total, const double *pos,
                                const double *vel, const double *drho_dt, const double *mass, const double *wall_vel,
                                double t0, int64_t step0)
{
    sphx_ctx *c = nullptr;
    try {
        require(out != nullptr, "SPHX:Ctx:out", "ctx output pointer must not be NULL");
        c = new sphx_ctx();
        ctx_setup(c, prm, n_fluid, n_total, pos, vel, drho_dt, mass, wall_vel, t0, step0);
        *out = c;
        return SPHX_OK;
    } catch (const Error &e) {
        delete c;
        return report(e);
    } catch (const std::exception &e) {
        delete c;
        return report_unknown(e);
    }
}

SPHX_EXPORT void sphx_ctx_destroy(sphx_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (g_fetch_src == ctx) g_fetch_src = nullptr;
    delete ctx;
}

SPHX_EXPORT int sphx_ctx_advance(sphx_ctx *c, double t_target, int64_t max_steps, sphx_status *status)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    for (int guard = 0; guard < 1000000; ++guard) {
        hipLaunchKernelGGL(k_prepare, dim3(1), dim3(1), 0, c->stream, c->clock.get(), c->phys, t_target,
                           (long long)max_steps, c->cur);
        // estimate how many slots this call needs from the current dt, then over-provision one graph
        const Clock &k = *c->h_clock;
        const double hh = c->phys.kc.h;
        const double dt_est = std::min(std::min(0.25 * hh / std::max(c->phys.c_f + k.vmax, 1e-12),
                                                0.125 * hh * hh / std::max(c->phys.nu, 1e-12)),
                                       0.25 * std::sqrt(hh / std::max(std::fabs(c->phys.g), 1e-12)));
        double want = std::ceil(std::max(0.0, std::min(t_target, k.t_end) - k.t) / std::max(dt_est, 1e-12)) + 1.0;
        if (max_steps > 0) want = std::min(want, (double)max_steps);
        want = std::min(want, 4096.0);
        int64_t slots = (int64_t)want;
        if (slots > c->spg) slots = ((slots + c->spg - 1) / c->spg) * c->spg;
        if (slots < 1) slots = 1;
        const int64_t step_before = c->h_clock->step;
        enqueue_slots(c, slots);
        read_clock(c);
        const int64_t executed = c->h_clock->step - step_before;
        if (max_steps > 0) {
            max_steps -= executed;
            if (max_steps <= 0) break;
        }
        if (c->h_clock->status != 0) break;
        if (!(c->h_clock->t < t_target - 1e-12)) break;
        if (!(c->h_clock->t < c->h_clock->t_end - 1e-12) && executed == 0) break;
    }
    fill_status(c, status);
    if (c->h_clock->status != 0)
        throw Error(c->h_clock->status, "SPHX:Ctx:diverged", "device step loop raised a status (non-finite velocity)");
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_enqueue_steps(sphx_ctx *c, int64_t n_steps)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    require(n_steps > 0, "SPHX:Ctx:steps", "n_steps must be positive");
    // parity of the first slot = parity of the state after everything already enqueued; callers of this
    // entry point run fixed step counts, so cur advances deterministically
    hipLaunchKernelGGL(k_prepare, dim3(1), dim3(1), 0, c->stream, c->clock.get(), c->phys, c->prm.t_end,
                       (long long)n_steps, c->cur);
    enqueue_slots(c, n_steps);
    c->cur = (int)((c->cur + n_steps) & 1);  // provisional; read_clock() recomputes from the device step count
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_sync(sphx_ctx *c, sphx_status *status)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    read_clock(c);
    fill_status(c, status);
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_download(sphx_ctx *c, double *pos, double *vel, double *rho, double *p, double *drho_dt,
                                  double *force, double *force_prior, double *Vol, double *B)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    read_clock(c);
    const bool need_outputs = rho || p || force || force_prior || Vol || B;
    if (need_outputs && !c->have_step_outputs)
        throw Error(SPHX_ERR_STATE, "SPHX:Ctx:download", "rho/p/force/Vol/B exist only after at least one step");
    const int nf = c->nf, nw = c->nw, nt = c->nt;
    hipStream_t s = c->stream;
    const FluidSet &fs = c->set[c->cur];
    const int *id_old = c->set[1 - c->cur].id;  // ordering the step outputs are stored in
    DevBuf<double> stage((size_t)4 * nt);
    const dim3 gf(div_up(nf, kBlock)), gw(div_up(std::max(nw, 1), kBlock)), gt(div_up(nt, kBlock)), b(kBlock);
    auto col = [&](int cidx) { return stage.get() + (size_t)cidx * nt; };
    auto unsort_f = [&](const int *id, const double *src, int cidx) {
        hipLaunchKernelGGL(k_unsort, gf, b, 0, s, nf, id, src, col(cidx));
    };
    auto unsort_w = [&](const double *src, int cidx) {
        if (nw > 0) hipLaunchKernelGGL(k_unsort, gw, b, 0, s, nw, (const int *)c->wid.get(), src, col(cidx));
    };
    auto fill_w = [&](int cidx, double v) {
        if (nw > 0) hipLaunchKernelGGL(k_fill, gw, b, 0, s, nw, col(cidx) + nf, v);
    };
    auto out = [&](double *host, int ncol) {
        SPHX_HIP(hipMemcpyAsync(host, stage.get(), (size_t)ncol * nt * sizeof(double), hipMemcpyDeviceToHost, s));
        SPHX_HIP(hipStreamSynchronize(s));
    };
    (void)gt;
    if (pos) { unsort_f(fs.id, fs.x, 0); unsort_f(fs.id, fs.y, 1); unsort_w(c->wx.get(), 0); unsort_w(c->wy.get(), 1); out(pos, 2); }
    if (vel) { unsort_f(fs.id, fs.vx, 0); unsort_f(fs.id, fs.vy, 1); fill_w(0, 0.0); fill_w(1, 0.0); out(vel, 2); }
    if (drho_dt) { unsort_f(fs.id, fs.drho, 0); fill_w(0, 0.0); out(drho_dt, 1); }
    if (rho) { unsort_f(id_old, c->rho_out.get(), 0); fill_w(0, c->prm.rho0); out(rho, 1); }
    if (p) { unsort_f(id_old, c->p_out.get(), 0); fill_w(0, 0.0); out(p, 1); }
    if (force) { unsort_f(id_old, c->ffx.get(), 0); unsort_f(id_old, c->ffy.get(), 1); fill_w(0, 0.0); fill_w(1, 0.0); out(force, 2); }
    if (force_prior) { unsort_f(id_old, c->fpx.get(), 0); unsort_f(id_old, c->fpy.get(), 1); fill_w(0, 0.0); fill_w(1, 0.0); out(force_prior, 2); }
    if (Vol) { unsort_f(id_old, c->Vol.get(), 0); unsort_w(c->wVol.get(), 0); out(Vol, 1); }
    if (B) {
        unsort_f(id_old, c->b11.get(), 0); unsort_f(id_old, c->b12.get(), 1); unsort_f(id_old, c->b21.get(), 2); unsort_f(id_old, c->b22.get(), 3);
        fill_w(0, 1.0); fill_w(1, 0.0); fill_w(2, 0.0); fill_w(3, 1.0);
        out(B, 4);
    }
    SPHX_HIP(hipGetLastError());
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_monitor(sphx_ctx *c, double *tau_bottom, double *tau_top, double *n_pairs)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    read_clock(c);
    hipStream_t s = c->stream;
    const int nf = c->nf;
    const FluidSet &fs = c->set[c->cur];
    if (tau_bottom || tau_top) {
        if (!c->have_step_outputs)
            throw Error(SPHX_ERR_STATE, "SPHX:Ctx:monitor", "wall shear needs Vol/B of a completed step");
        const int nblk = (int)div_up(nf, kBlock);
        hipLaunchKernelGGL(k_wall_shear, dim3(nblk), dim3(kBlock), 0, s, c->grid, c->phys, nf, fs, c->tmp, c->walls, 1,
                           c->tau_part.get());
        hipLaunchKernelGGL(k_tau_final, dim3(1), dim3(kScanBlock), 0, s, nblk, (const double *)c->tau_part.get(),
                           c->phys.DL, c->tau_out.get());
        double h[2];
        SPHX_HIP(hipMemcpyAsync(h, c->tau_out.get(), sizeof(h), hipMemcpyDeviceToHost, s));
        SPHX_HIP(hipStreamSynchronize(s));
        if (tau_bottom) *tau_bottom = h[0];
        if (tau_top) *tau_top = h[1];
    }
    if (n_pairs) {
        DevBuf<int> cnt(nf), off((size_t)nf + 1);
        hipLaunchKernelGGL(k_pairs<0>, dim3(div_up(nf, kBlock)), dim3(kBlock), 0, s, c->grid, c->phys, nf, fs, c->walls,
                           cnt.get(), (const int *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr,
                           (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr);
        hipLaunchKernelGGL(k_scan_only, dim3(1), dim3(kScanBlock), 0, s, (const int *)cnt.get(), off.get(), nf);
        int total = 0;
        SPHX_HIP(hipMemcpyAsync(&total, off.get() + nf, sizeof(int), hipMemcpyDeviceToHost, s));
        SPHX_HIP(hipStreamSynchronize(s));
        *n_pairs = (double)total;
    }
    SPHX_HIP(hipGetLastError());
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_neighbor_list(sphx_ctx *c, size_t *n_pairs)
{
    SPHX_TRY
    require(c != nullptr && n_pairs != nullptr, "SPHX:Ctx:null", "ctx / n_pairs must not be NULL");
    read_clock(c);
    emit_pairs(c);
    *n_pairs = c->pl_n;
    g_fetch_src = c;
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_neighbor_search(const double *pos, int n_fluid, int n_total, double h, double DL, size_t *n_pairs)
{
    sphx_ctx *c = nullptr;
    try {
        require(n_pairs != nullptr && pos != nullptr, "SPH:Neighbor:pos", "pos must be a double matrix of size [n_total x 2].");
        require(n_total > 0 && n_fluid > 0 && n_fluid <= n_total, "SPH:Neighbor:count",
                "Invalid n_fluid/n_total or inconsistent pos size.");
        require(h > 0.0 && DL > 0.0, "SPH:Neighbor:param", "h and DL must be positive.");
        sphx_params prm{};
        prm.DL = DL; prm.DH = 1.0; prm.dp = h / 1.3; prm.h = h; prm.rho0 = 1.0; prm.mu = 1.0; prm.c_f = 1.0; prm.p0 = 1.0;
        prm.inv_sigma0 = 1.0; prm.gravity_g = 0.0; prm.transport_coeff = 0.0; prm.t_end = 0.0; prm.lanes_per_particle = 1;
        const size_t nt = (size_t)n_total;
        std::vector<double> zeros2(2 * nt, 0.0), ones(nt, 1.0);
        c = new sphx_ctx();
        ctx_setup(c, &prm, n_fluid, n_total, pos, zeros2.data(), zeros2.data(), ones.data(), zeros2.data(), 0.0, 0);
        emit_pairs(c);
        *n_pairs = c->pl_n;
        if (g_search_ctx) delete g_search_ctx;
        g_search_ctx = c;
        g_fetch_src = c;
        return SPHX_OK;
    } catch (const Error &e) {
        delete c;
        return report(e);
    } catch (const std::exception &e) {
        delete c;
        return report_unknown(e);
    }
}

SPHX_EXPORT int sphx_neighbor_fetch(double *pair_i, double *pair_j, double *dx, double *dy, double *r, double *W,
                                    double *dW, size_t capacity)
{
    SPHX_TRY
    sphx_ctx *c = g_fetch_src;
    if (!c || !c->pl_valid) throw Error(SPHX_ERR_STATE, "SPHX:Neighbor:fetch", "no pair list pending (call a search first)");
    require(capacity >= c->pl_n, "SPHX:Neighbor:capacity", "output arrays are shorter than the pair list");
    const size_t n = c->pl_n;
    hipStream_t s = c->stream;
    if (n) {
        if (pair_i) c->pl_i.download(pair_i, n, s);
        if (pair_j) c->pl_j.download(pair_j, n, s);
        if (dx) c->pl_dx.download(dx, n, s);
        if (dy) c->pl_dy.download(dy, n, s);
        if (r) c->pl_r.download(r, n, s);
        if (W) c->pl_W.download(W, n, s);
        if (dW) c->pl_dW.download(dW, n, s);
        SPHX_HIP(hipStreamSynchronize(s));
    }
    c->pl_valid = false;
    c->pl_i.release(); c->pl_j.release(); c->pl_dx.release(); c->pl_dy.release(); c->pl_r.release(); c->pl_W.release(); c->pl_dW.release();
    g_fetch_src = nullptr;
    if (g_search_ctx == c) { delete g_search_ctx; g_search_ctx = nullptr; }
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_profile_enable(sphx_ctx *c, int on)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    SPHX_HIP(hipStreamSynchronize(c->stream));
    c->timer.collect();
    c->profiling = on != 0;
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_profile_read(sphx_ctx *c, int capacity, const char **names, double *avg_ms, int64_t *launches,
                                      int *n_kernels)
{
    SPHX_TRY
    require(c != nullptr && n_kernels != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    SPHX_HIP(hipStreamSynchronize(c->stream));
    c->timer.collect();
    const int n = (int)c->timer.names.size();
    *n_kernels = n;
    for (int k = 0; k < n && k < capacity; ++k) {
        if (names) names[k] = c->timer.names[k].c_str();
        if (avg_ms) avg_ms[k] = c->timer.launches[k] ? c->timer.total_ms[k] / (double)c->timer.launches[k] : 0.0;
        if (launches) launches[k] = c->timer.launches[k];
    }
    for (auto &v : c->timer.total_ms) v = 0.0;
    for (auto &v : c->timer.launches) v = 0;
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_info(sphx_ctx *c, int *n_fluid, int *n_wall, int *n_cell_x, int *n_cell_y)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    if (n_fluid) *n_fluid = c->nf;
    if (n_wall) *n_wall = c->nw;
    if (n_cell_x) *n_cell_x = c->grid.ncx;
    if (n_cell_y) *n_cell_y = c->grid.ncy;
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_tuning(sphx_ctx *c, int *lanes_per_particle, int *steps_per_graph)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    if (lanes_per_particle) *lanes_per_particle = c->lpp;
    if (steps_per_graph) *steps_per_graph = c->spg;
    return SPHX_OK;
    SPHX_CATCH
}
