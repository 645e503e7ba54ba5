// sphx_kernels_tiled.hpp -- LDS-tiled variants of the four neighbour passes (large particle counts).
//
// PMC profile of the list-walking kernels at 6 M particles (profiles/r01_pmc_c5_before_xcd_remap_summary.txt): the
// texture-address unit is busy 83-87 % of the kernel, VALU 22 % -- every neighbour costs up to 13
// scattered 8-byte gathers through L1.  Here a workgroup owns a TILE = up to CT consecutive cells of one
// cell column (cells are column-major, so the tile's particles and each of its three halo column
// segments are contiguous index ranges).  The fields a pass needs for tile + halo are staged into LDS
// with coalesced loads once; candidate sweeps and neighbour gathers then read LDS.  The neighbour list
// holds 16-bit halo-local indices (all four passes use the identical tiling, so an index written by
// k_density_t means the same particle in the later passes).
//
// Formulas, guards and summation structure are those of sphx_kernels.hpp (same reference lines); only
// where the operands come from differs.  Walls stay a global sweep (3 % of particles touch them at the
// sizes this path is used for).
#pragma once
#include "sphx_kernels.hpp"

namespace sphx {

struct TileCfg {
    int ct;     // cells per tile (rows of one column)
    int nseg;   // tiles per column = ceil(ncy / ct)
    int hcap;   // LDS capacity in particles (tile + halo)
};

struct Tile {  // scalar members only: a runtime-indexed array here lands in scratch memory
    int cx, r0, r1, rlo, rhi;   // own rows [r0,r1), halo rows [rlo,rhi]
    int g0a, g0b, g0c;          // global start of the three halo column segments (cx-1, cx, cx+1)
    int offb, offc;             // LDS offsets of segments b and c (segment a starts at 0)
    int col_a, col_b, col_c;    // wrapped column indices, -1 = outside an open window
    int n_halo;
    int p0, pn;                 // own particles: global slots [p0, p0+pn)
    int own;                    // LDS index of the first own particle
};

__device__ __forceinline__ int tile_col(const Grid &g, int col)
{
    if (g.periodic) return col < 0 ? col + g.ncx : (col >= g.ncx ? col - g.ncx : col);
    return (col < 0 || col >= g.ncx) ? -1 : col;
}

// uniform per block: every thread computes the same tile from the cell-start table
__device__ __forceinline__ Tile make_tile(const Grid &g, const int *__restrict__ start, int tile_id, const TileCfg &tc)
{
    Tile t;
    t.cx = tile_id / tc.nseg;
    const int seg = tile_id - t.cx * tc.nseg;
    t.r0 = seg * tc.ct;
    t.r1 = min(t.r0 + tc.ct, g.ncy);
    t.rlo = max(t.r0 - 1, 0);
    t.rhi = min(t.r1, g.ncy - 1);
    t.col_a = tile_col(g, t.cx - 1);
    t.col_b = t.cx;
    t.col_c = tile_col(g, t.cx + 1);
    const int ca = max(t.col_a, 0), cc = max(t.col_c, 0);
    t.g0a = start[ca * g.ncy + t.rlo];
    t.g0b = start[t.cx * g.ncy + t.rlo];
    t.g0c = start[cc * g.ncy + t.rlo];
    const int na = t.col_a >= 0 ? start[ca * g.ncy + t.rhi + 1] - t.g0a : 0;
    const int nb = start[t.cx * g.ncy + t.rhi + 1] - t.g0b;
    const int nc = t.col_c >= 0 ? start[cc * g.ncy + t.rhi + 1] - t.g0c : 0;
    t.offb = na;
    t.offc = na + nb;
    t.n_halo = na + nb + nc;
    t.p0 = start[t.cx * g.ncy + t.r0];
    t.pn = start[t.cx * g.ncy + t.r1] - t.p0;
    t.own = t.offb + (t.p0 - t.g0b);
    return t;
}

__device__ __forceinline__ int halo_to_global(const Tile &t, int j)
{
    return j < t.offb ? t.g0a + j : (j < t.offc ? t.g0b + (j - t.offb) : t.g0c + (j - t.offc));
}

// stage one global field into LDS for the whole halo
__device__ __forceinline__ void stage(const Tile &t, const double *__restrict__ src, double *dst)
{
    for (int j = threadIdx.x; j < t.n_halo; j += kBlock) dst[j] = src[halo_to_global(t, j)];
}

// LDS-local start of every halo cell: ls[c*(ct+4) + (r - rlo)] for r in [rlo, rhi+1]
__device__ __forceinline__ void stage_cell_starts(const Grid &g, const Tile &t, const TileCfg &tc,
                                                  const int *__restrict__ start, int *ls)
{
    const int rows = t.rhi - t.rlo + 2;
    for (int e = threadIdx.x; e < 3 * rows; e += kBlock) {
        const int c = e / rows, r = t.rlo + (e - c * rows);
        const int col = c == 0 ? t.col_a : (c == 1 ? t.col_b : t.col_c);
        const int g0 = c == 0 ? t.g0a : (c == 1 ? t.g0b : t.g0c);
        const int off = c == 0 ? 0 : (c == 1 ? t.offb : t.offc);
        ls[c * (tc.ct + 4) + (r - t.rlo)] = col >= 0 ? (start[col * g.ncy + r] - g0 + off) : off;
    }
}

// ---------------------------------------------------------------------------------------------
// pass A (tiled): candidate sweep in LDS -> 16-bit neighbour list; sigma sum -> rho, Vol, rho_half, p_half
// ---------------------------------------------------------------------------------------------
template <int LPP>
__global__ __launch_bounds__(kBlock) void k_density_t(const Clock *clk, int q, Grid g, Phys ph, FluidSet s,
                                                      FluidTmp t, Walls w, TileCfg tc, unsigned short *nl16)
{
    if (!clk->run[q]) return;
    extern __shared__ double lds[];
    double *lx = lds, *ly = lds + tc.hcap;
    int *ls = reinterpret_cast<int *>(lds + 2 * (size_t)tc.hcap);
    const Tile tl = make_tile(g, s.start, xcd_block(blockIdx.x, gridDim.x), tc);
    if (tl.n_halo > tc.hcap) { if (threadIdx.x == 0) atomicOr(t.flags, 4); return; }
    stage(tl, s.x, lx);
    stage(tl, s.y, ly);
    stage_cell_starts(g, tl, tc, s.start, ls);
    __syncthreads();
    const int sub = threadIdx.x % LPP;
    const int cst = tc.ct + 4;
    for (int base = 0; base < tl.pn; base += kBlock / LPP) {
        const int pp = base + threadIdx.x / LPP;
        const bool active = pp < tl.pn;
        double s_in = 0.0, s_ct = 0.0;
        int cnt = 0;
        const int i = tl.p0 + pp;
        const size_t lane = (size_t)i * LPP + sub;
        if (active) {
            const double xi = lx[tl.own + pp], yi = ly[tl.own + pp];
            int cx, cy;
            cell_of(g, xi, yi, cx, cy);
            cy = min(max(cy, tl.r0), tl.r1 - 1);  // the particle IS in this tile
            const int a = max(cy - 1, 0) - tl.rlo, b = min(cy + 1, g.ncy - 1) + 1 - tl.rlo;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int lo = ls[c * cst + a], hi = ls[c * cst + b];
                for (int k = lo + sub; k < hi; k += LPP) {
                    const double dx = min_image(g, xi - lx[k]), dy = yi - ly[k];
                    const double r2 = dx * dx + dy * dy;
                    if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                        s_in += spline_W(ph.kc, r2 * rsqrt(r2));
                        if (cnt < t.nl_cap) nl16[(size_t)cnt * t.nl_stride + lane] = (unsigned short)k;
                        ++cnt;
                    }
                }
            }
            if (w.row_any[cy]) {
                sweep<LPP>(g, w.start, tl.cx, cy, sub, [&](int k) {
                    const double dx = min_image(g, xi - w.x[k]), dy = yi - w.y[k];
                    const double r2 = dx * dx + dy * dy;
                    if (r2 > kR2Min && r2 < ph.kc.rcut2) s_ct += spline_W(ph.kc, r2 * rsqrt(r2)) * w.Vol[k];
                });
            }
            if (cnt > t.nl_cap) { atomicOr(t.flags, 1); cnt = t.nl_cap; }
            t.nl_cnt[lane] = cnt;
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
}

// ---------------------------------------------------------------------------------------------
// pass B (tiled): KGC matrix
// ---------------------------------------------------------------------------------------------
template <int LPP>
__global__ __launch_bounds__(kBlock) void k_kgc_t(const Clock *clk, int q, Grid g, Phys ph, FluidSet s, FluidTmp t,
                                                  Walls w, TileCfg tc, const unsigned short *nl16)
{
    if (!clk->run[q]) return;
    extern __shared__ double lds[];
    double *lx = lds, *ly = lds + tc.hcap, *lV = lds + 2 * (size_t)tc.hcap;
    const Tile tl = make_tile(g, s.start, xcd_block(blockIdx.x, gridDim.x), tc);
    if (tl.n_halo > tc.hcap) return;
    stage(tl, s.x, lx);
    stage(tl, s.y, ly);
    stage(tl, t.Vol, lV);
    __syncthreads();
    const int sub = threadIdx.x % LPP;
    for (int base = 0; base < tl.pn; base += kBlock / LPP) {
        const int pp = base + threadIdx.x / LPP;
        const bool active = pp < tl.pn;
        const int i = tl.p0 + pp;
        const size_t lane = (size_t)i * LPP + sub;
        double a11 = 0.0, a12 = 0.0, a21 = 0.0, a22 = 0.0;
        if (active) {
            const double xi = lx[tl.own + pp], yi = ly[tl.own + pp];
            auto term = [&](double dx, double dy, double Volj) {
                const double r2 = dx * dx + dy * dy, inv_r = rsqrt(r2), r = r2 * inv_r;
                const double ex = dx * inv_r, ey = dy * inv_r;
                const double fxj = spline_dW(ph.kc, r) * Volj;
                a11 -= dx * (fxj * ex);
                a12 -= dx * (fxj * ey);
                a21 -= dy * (fxj * ex);
                a22 -= dy * (fxj * ey);
            };
            const int nn = t.nl_cnt[lane];
            for (int m = 0; m < nn; ++m) {
                const int k = nl16[(size_t)m * t.nl_stride + lane];
                term(min_image(g, xi - lx[k]), yi - ly[k], lV[k]);
            }
            int cx, cy;
            cell_of(g, xi, yi, cx, cy);
            if (w.row_any[cy]) {
                sweep<LPP>(g, w.start, cx, cy, sub, [&](int k) {
                    const double dx = min_image(g, xi - w.x[k]), dy = yi - w.y[k];
                    const double r2 = dx * dx + dy * dy;
                    if (r2 > kR2Min && r2 < ph.kc.rcut2) term(dx, dy, w.Vol[k]);
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
}

// ---------------------------------------------------------------------------------------------
// pass CD (tiled): viscous + transport + pressure, kick, drift, wrap
// ---------------------------------------------------------------------------------------------
template <int LPP>
__global__ __launch_bounds__(kBlock) void k_forces_t(const Clock *clk, int q, Grid g, Phys ph, FluidSet s, FluidTmp t,
                                                     Walls w, TileCfg tc, const unsigned short *nl16)
{
    if (!clk->run[q]) return;
    extern __shared__ double lds[];
    const size_t H = tc.hcap;
    double *lx = lds, *ly = lds + H, *lvx = lds + 2 * H, *lvy = lds + 3 * H, *lV = lds + 4 * H, *lp = lds + 5 * H,
           *lr = lds + 6 * H, *l11 = lds + 7 * H, *l12 = lds + 8 * H, *l21 = lds + 9 * H, *l22 = lds + 10 * H;
    const Tile tl = make_tile(g, s.start, xcd_block(blockIdx.x, gridDim.x), tc);
    if (tl.n_halo > tc.hcap) return;
    stage(tl, s.x, lx); stage(tl, s.y, ly); stage(tl, s.vx, lvx); stage(tl, s.vy, lvy);
    stage(tl, t.Vol, lV); stage(tl, t.ph, lp); stage(tl, t.rhoh, lr);
    stage(tl, t.b11, l11); stage(tl, t.b12, l12); stage(tl, t.b21, l21); stage(tl, t.b22, l22);
    __syncthreads();
    const int sub = threadIdx.x % LPP;
    const double h = ph.kc.h;
    for (int base = 0; base < tl.pn; base += kBlock / LPP) {
        const int pp = base + threadIdx.x / LPP;
        const bool active = pp < tl.pn;
        const int i = tl.p0 + pp;
        const size_t lane = (size_t)i * LPP + sub;
        double ax = 0.0, ay = 0.0, ix = 0.0, iy = 0.0, px = 0.0, py = 0.0;
        double xi = 0.0, yi = 0.0, vxi = 0.0, vyi = 0.0, Voli = 0.0, mi = 1.0, p_i = 0.0, rhoh_i = 0.0;
        double b11i = 1.0, b12i = 0.0, b21i = 0.0, b22i = 1.0;
        int cx = 0, cy = 0;
        bool near_wall = false;
        if (active) {
            const int o = tl.own + pp;
            xi = lx[o]; yi = ly[o]; vxi = lvx[o]; vyi = lvy[o];
            Voli = lV[o]; mi = s.mass[i]; p_i = lp[o]; rhoh_i = lr[o];
            b11i = l11[o]; b12i = l12[o]; b21i = l21[o]; b22i = l22[o];
            const int nn = t.nl_cnt[lane];
            for (int m = 0; m < nn; ++m) {
                const int k = nl16[(size_t)m * t.nl_stride + lane];
                const double dx = min_image(g, xi - lx[k]), dy = yi - ly[k];
                const double r2 = dx * dx + dy * dy, inv_r = rsqrt(r2), r = r2 * inv_r;
                const double ex = dx * inv_r, ey = dy * inv_r;
                const double dW = spline_dW(ph.kc, r);
                const double Volj = lV[k];
                const double tx = (b11i + l11[k]) * ex + (b12i + l12[k]) * ey;
                const double ty = (b21i + l21[k]) * ex + (b22i + l22[k]) * ey;
                const double eBe = ex * tx + ey * ty;
                const double vxj = lvx[k], vyj = lvy[k];
                const double dWVj = dW * Volj;
                const double coeff = eBe * ph.mu * dWVj / (r + 0.01 * h);
                ax += coeff * (vxi - vxj);
                ay += coeff * (vyi - vyj);
                ix -= dWVj * tx;
                iy -= dWVj * ty;
                const double p_j = lp[k];
                const double rho_bar = 0.5 * (rhoh_i + lr[k]);
                const double un_l = vxi * ex + vyi * ey, un_r = vxj * ex + vyj * ey;
                const double beta = riemann_beta(un_l, un_r, ph.c_f);
                const double p_avg = 0.5 * (p_i + p_j);
                const double p_star = p_avg + 0.5 * beta * rho_bar * (un_l - un_r);
                const double p_face = 0.5 * (p_avg + p_star);
                px -= (p_face * tx) * dWVj;
                py -= (p_face * ty) * dWVj;
            }
            cell_of(g, xi, yi, cx, cy);
            near_wall = w.row_any[cy] != 0;
            if (near_wall) {
                sweep<LPP>(g, w.start, cx, cy, sub, [&](int k) {
                    const double dx = min_image(g, xi - w.x[k]), dy = yi - w.y[k];
                    const double r2 = dx * dx + dy * dy;
                    if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                        const double inv_r = rsqrt(r2), r = r2 * inv_r;
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
        const double fpx = ax * Voli + mi * ph.g;
        const double fpy = ay * Voli;
        if (active && near_wall) {
            const double acx = fpx / mi, acy = fpy / mi;
            sweep<LPP>(g, w.start, cx, cy, sub, [&](int k) {
                const double dx = min_image(g, xi - w.x[k]), dy = yi - w.y[k];
                const double r2 = dx * dx + dy * dy;
                if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                    const double inv_r = rsqrt(r2), r = r2 * inv_r;
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
            t.xn[i] = g.periodic ? wrap_x(xo, ph.DL) : xo;
            t.yn[i] = yo;
            t.vxn[i] = vxn;
            t.vyn[i] = vyn;
            t.fpx[i] = fpx;
            t.fpy[i] = fpy;
            t.fx[i] = fx;
            t.fy[i] = fy;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pass E (tiled): continuity, final rho/p, max |v|^2, cell histogram
// ---------------------------------------------------------------------------------------------
template <int LPP>
__global__ __launch_bounds__(kBlock) void k_continuity_t(const Clock *clk, int q, Grid g, Phys ph, FluidSet s,
                                                         FluidTmp t, Walls w, int do_hist, TileCfg tc,
                                                         const unsigned short *nl16)
{
    if (!clk->run[q]) return;
    extern __shared__ double lds[];
    const size_t H = tc.hcap;
    double *lx = lds, *ly = lds + H, *lvx = lds + 2 * H, *lvy = lds + 3 * H, *lV = lds + 4 * H;
    const int tile_id = xcd_block(blockIdx.x, gridDim.x);
    const Tile tl = make_tile(g, s.start, tile_id, tc);
    double v2max = 0.0;
    if (tl.n_halo <= tc.hcap) {
        stage(tl, s.x, lx); stage(tl, s.y, ly); stage(tl, t.vxn, lvx); stage(tl, t.vyn, lvy); stage(tl, t.Vol, lV);
        __syncthreads();
        const int sub = threadIdx.x % LPP;
        for (int base = 0; base < tl.pn; base += kBlock / LPP) {
            const int pp = base + threadIdx.x / LPP;
            const bool active = pp < tl.pn;
            const int i = tl.p0 + pp;
            const size_t lane = (size_t)i * LPP + sub;
            double rate = 0.0, vxi = 0.0, vyi = 0.0, xi = 0.0;
            if (active) {
                const int o = tl.own + pp;
                xi = lx[o];
                const double yi = ly[o];
                vxi = lvx[o];
                vyi = lvy[o];
                const int nn = t.nl_cnt[lane];
                for (int m = 0; m < nn; ++m) {
                    const int k = nl16[(size_t)m * t.nl_stride + lane];
                    const double dx = min_image(g, xi - lx[k]), dy = yi - ly[k];
                    const double r2 = dx * dx + dy * dy, inv_r = rsqrt(r2), r = r2 * inv_r;
                    const double ex = dx * inv_r, ey = dy * inv_r;
                    const double u_jump = (vxi - lvx[k]) * ex + (vyi - lvy[k]) * ey;
                    rate += u_jump * spline_dW(ph.kc, r) * lV[k];
                }
                int cx, cy;
                cell_of(g, xi, yi, cx, cy);
                if (w.row_any[cy]) {
                    sweep<LPP>(g, w.start, cx, cy, sub, [&](int k) {
                        const double dx = min_image(g, xi - w.x[k]), dy = yi - w.y[k];
                        const double r2 = dx * dx + dy * dy;
                        if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                            const double inv_r = rsqrt(r2), r = r2 * inv_r;
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
                if (xi >= g.own_lo && xi < g.own_hi) {
                    double v2 = vxi * vxi + vyi * vyi;
                    if (v2 != v2) v2 = INFINITY;
                    v2max = fmax(v2max, v2);
                }
                if (do_hist) {
                    int cx, cy;
                    cell_of(g, t.xn[i], t.yn[i], cx, cy);
                    const int c = cx * g.ncy + cy;
                    t.cellid[i] = c;
                    atomicAdd(&t.count[c], 1);
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v2max = fmax(v2max, __shfl_xor(v2max, off));
    __shared__ double s_max[kBlock / 64];
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = v2max;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = s_max[0];
        for (int k = 1; k < kBlock / 64; ++k) m = fmax(m, s_max[k]);
        t.vpart[tile_id] = m;
    }
}

}  // namespace sphx
