// sphx_device.hpp -- per-pair and per-particle SPH formulas shared by the pair-list kernels
// (stateless MEX surface) and the cell-grid gather kernels (resident step).  gfx950 only.
//
// Formula sources (reference file:line):
//   cubic spline W, dW/dr .......... mex/sph_neighbor_search_mex.c:116-133, mex/sph_physics_mex.c:33-38
//   density from sigma sums ........ mex/sph_physics_mex.c:218-234
//   KGC pseudo-inverse blend ....... mex/sph_physics_mex.c:321-366
//   viscous pair term .............. mex/sph_physics_mex.c:489-535
//   transport pair term / limiter .. mex/sph_physics_mex.c:656-710
//   Riemann pressure pair term ..... mex/sph_physics_mex.c:884-950, 1121-1129
//   continuity pair term ........... mex/sph_physics_mex.c:1090-1108
//   wall shear pair term ........... mex/sph_physics_mex.c:1726-1738
#pragma once
#include <hip/hip_runtime.h>

namespace sphx {

constexpr double kPi = 3.14159265358979323846;
constexpr double kEpsReg = 1e-8;  // EPS_REG, sph_physics_mex.c:30
constexpr double kRMin = 1e-12;   // pair filter r <= 1e-12 (sph_physics_mex.c:246,477,...)
constexpr double kR2Min = 1e-24;  // neighbour filter r^2 > 1e-24 (sph_neighbor_search_mex.c:368)

struct KernelConst {
    double h, inv_h, sigma, sigma_over_h, rcut2;
};

__host__ __device__ inline KernelConst make_kernel_const(double h)
{
    KernelConst k;
    k.h = h;
    k.inv_h = 1.0 / h;
    k.sigma = 10.0 / (7.0 * kPi * h * h);
    k.sigma_over_h = k.sigma / h;
    k.rcut2 = (2.0 * h) * (2.0 * h);
    return k;
}

// W and dW/dr for r < 2h (callers have already applied the cut-off).
__device__ __forceinline__ void spline(const KernelConst &kc, double r, double &W, double &dW)
{
    const double q = r * kc.inv_h;
    if (q < 1.0) {
        W = kc.sigma * (1.0 - 1.5 * q * q + 0.75 * q * q * q);
        dW = kc.sigma_over_h * (-3.0 * q + 2.25 * q * q);
    } else if (q < 2.0) {
        const double tq = 2.0 - q;
        W = kc.sigma * 0.25 * tq * tq * tq;
        dW = -kc.sigma_over_h * 0.75 * tq * tq;
    } else {
        W = 0.0;
        dW = 0.0;
    }
}

__device__ __forceinline__ double spline_dW(const KernelConst &kc, double r)
{
    const double q = r * kc.inv_h;
    if (q < 1.0) return kc.sigma_over_h * (-3.0 * q + 2.25 * q * q);
    if (q < 2.0) { const double tq = 2.0 - q; return -kc.sigma_over_h * 0.75 * tq * tq; }
    return 0.0;
}

__device__ __forceinline__ double spline_W(const KernelConst &kc, double r)
{
    const double q = r * kc.inv_h;
    if (q < 1.0) return kc.sigma * (1.0 - 1.5 * q * q + 0.75 * q * q * q);
    if (q < 2.0) { const double tq = 2.0 - q; return kc.sigma * 0.25 * tq * tq * tq; }
    return 0.0;
}

// dW/dr and W without branches or selects, for the large-channel walks (VALU-bound: the two-piece form cost two compares, four
// v_cndmask and -- the compiler turned the selects back into branches -- six scalar instructions per neighbour on top of
// both polynomials).  With a = (2 - q)+ and b = (1 - q)+ the cubic spline is ONE expression on [0, inf):
//     W(q)  = sigma/4       [a^3 - 4 b^3]        q < 1: 1 - 1.5 q^2 + 0.75 q^3,   1 <= q < 2: 0.25 (2 - q)^3
//     W'(q) = -0.75 sigma/h [a^2 - 4 b^2]        q < 1: -3 q + 2.25 q^2,          1 <= q < 2: -0.75 (2 - q)^2
// -- the same polynomials as spline_dW / spline_W (sph_physics_mex.c:76-105), differently rounded (a few 1e-16 of sigma).
__device__ __forceinline__ double spline_dW_sel(const KernelConst &kc, double r)
{
    const double q = r * kc.inv_h;
    const double a = fmax(2.0 - q, 0.0), b = fmax(1.0 - q, 0.0);
    return (-0.75 * kc.sigma_over_h) * fma(-4.0 * b, b, a * a);
}

// ... for a neighbour known to lie inside the support (the step's list: pass A accepted it at these positions): no clamp of a
// (q can exceed 2 by a rounding error: a^2 ~ 1e-31)
__device__ __forceinline__ double spline_dW_in(const KernelConst &kc, double r)
{
    const double q = r * kc.inv_h;
    const double a = 2.0 - q, b = fmax(1.0 - q, 0.0);
    return (-0.75 * kc.sigma_over_h) * fma(-4.0 * b, b, a * a);
}

// (callers discard the value where r >= 2h: no clamp of a)
__device__ __forceinline__ double spline_W_sel(const KernelConst &kc, double r)
{
    const double q = r * kc.inv_h;
    const double a = 2.0 - q, b = fmax(1.0 - q, 0.0);
    return (0.25 * kc.sigma) * fma(-4.0 * (b * b), b, (a * a) * a);
}

// 1/x for a positive normal x of moderate magnitude (here: lengths of order h): hardware estimate + two Newton steps,
// ~1 ulp, 5 instructions against the 11 of an IEEE division with its scaling and fix-up
__device__ __forceinline__ double rcp_nr(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}

// 1/sqrt(x) for 1e-24 < x < huge (squared pair distances): hardware estimate + one third-order correction step
__device__ __forceinline__ double rsqrt_nr(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}

// rho from the two sigma sums; sigma_inner already includes W(0).
__device__ __forceinline__ double density_from_sigma(double sigma_inner, double sigma_contact,
                                                     double mass_i, double rho0, double inv_sigma0)
{
    double rhoi = sigma_inner * rho0 * inv_sigma0;
    rhoi += sigma_contact * rho0 * rho0 * inv_sigma0 / mass_i;
    if (rhoi <= 1e-12) rhoi = rho0;
    return rhoi;
}

struct Mat2 {
    double m11, m12, m21, m22;
};

// Kernel-gradient-correction matrix from the moment matrix A (sph_physics_mex.c:321-366):
//   P = (A^T A + eps I)^-1 A^T      Tikhonov-regularised pseudo-inverse (identity when A^T A + eps I is singular)
//   B = lambda P + (1 - lambda) I   with lambda = det A / (det A + max(1 - det A, 0))  (0 when that denominator vanishes)
// i.e. the full correction where the particle's support is complete (det A -> 1) and a fade to "no correction" where
// it is truncated.  The thresholds (1e-20, 1e-12) and the order of the operations are the reference's: they decide
// which branch a borderline particle takes.
__device__ __forceinline__ Mat2 kgc_from_A(double a11, double a12, double a21, double a22)
{
    // normal matrix N = A^T A + eps I (symmetric: n11, n12, n22)
    const double n11 = a11 * a11 + a21 * a21 + kEpsReg;
    const double n12 = a11 * a12 + a21 * a22;
    const double n22 = a12 * a12 + a22 * a22 + kEpsReg;
    const double det_n = n11 * n22 - n12 * n12;
    Mat2 P{1.0, 0.0, 0.0, 1.0};
    if (!(fabs(det_n) < 1e-20)) {
        // N^-1 = [n22 -n12; -n12 n11] / det_n, then P = N^-1 A^T
        const double i11 = n22 / det_n, i12 = -n12 / det_n, i22 = n11 / det_n;
        P.m11 = i11 * a11 + i12 * a12;
        P.m12 = i11 * a21 + i12 * a22;
        P.m21 = i12 * a11 + i22 * a12;
        P.m22 = i12 * a21 + i22 * a22;
    }
    const double det_a = a11 * a22 - a12 * a21;
    const double deficit = fmax(1.0 - det_a, 0.0);
    const double total = det_a + deficit;
    double lambda = 0.0, rest = 1.0;
    if (!(fabs(total) < 1e-12)) { lambda = det_a / total; rest = deficit / total; }
    Mat2 B;
    B.m11 = lambda * P.m11 + rest;
    B.m12 = lambda * P.m12;
    B.m21 = lambda * P.m21;
    B.m22 = lambda * P.m22 + rest;
    return B;
}

// dissipation speed of the low-dissipation Riemann pressure (sph_physics_mex.c:1121-1129): three times the approach
// speed of the pair along its axis, capped by the sound speed; zero for a separating pair
__device__ __forceinline__ double riemann_beta(double un_l, double un_r, double c_f)
{
    double approach = un_l - un_r;
    if (approach < 0.0) approach = 0.0;
    return fmin(3.0 * approach, c_f);
}

// transport-velocity shift (sph_physics_mex.c:702-710): coeff h^2 inc, faded in by min(1, 100 |inc|^2 / h^2)
__device__ __forceinline__ void transport_shift(double inc_x, double inc_y, double h, double coeff,
                                                double &sx, double &sy)
{
    double fade = 100.0 * (inc_x * inc_x + inc_y * inc_y) / (h * h);
    const double gain = coeff * h * h;
    if (fade > 1.0) fade = 1.0;
    if (fade < 0.0) fade = 0.0;
    sx = gain * fade * inc_x;
    sy = gain * fade * inc_y;
}

__device__ __forceinline__ double eos_pressure(double rho, double rho0, double p0)
{
    return p0 * (rho / rho0 - 1.0);
}

// order-preserving bits of a non-negative double, for atomicMax
__device__ __forceinline__ unsigned long long nonneg_bits(double v)
{
    return (unsigned long long)__double_as_longlong(v);
}

}  // namespace sphx
