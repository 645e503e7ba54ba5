// sphx_pairlist.hip -- the eight sph_physics_shell_mex modes on a caller-supplied flat pair list
// (the stateless MEX surface, include/sphx.h section 1).  One thread per pair, FP64 hardware
// atomics (global_atomic_add_f64, -munsafe-fp-atomics) for the scatter -- the same "parallel-for
// over pairs + atomic add" structure the reference runs under OpenMP
// (mex/sph_physics_mex.c:186-212,237-312,467-536,634-700,868-951,1074-1109), so any pair list a
// MATLAB caller builds gives the reference's result up to summation order.  The fast path (the
// device-resident step in sphx_resident.hip) does not use pair lists at all.
#include "sphx_common.hpp"
#include "sphx_device.hpp"

namespace sphx {
namespace {

constexpr int kBlock = 256;

struct PairView {
    const double *pi, *pj, *dx, *dy, *r, *W, *dW;
    long n;
};

// decode one pair; returns false when the reference's loop would `continue`
__device__ __forceinline__ bool pair_ids(const PairView &pv, long k, int nf, int nt, int &ii, int &jj)
{
    ii = (int)pv.pi[k] - 1;
    jj = (int)pv.pj[k] - 1;
    return !(ii < 0 || ii >= nf || jj < 0 || jj >= nt);
}

// ---- density_correction ---------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pl_sigma(PairView pv, const double *mass, int nf, int nt,
                                                   double rho0, double *sig_in, double *sig_ct)
{
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= pv.n) return;
    int ii, jj;
    if (!pair_ids(pv, k, nf, nt, ii, jj)) return;
    const double wk = pv.W[k];
    if (jj < nf) {
        atomicAdd(&sig_in[ii], wk);
        atomicAdd(&sig_in[jj], wk);
    } else {
        atomicAdd(&sig_ct[ii], wk * (mass[jj] / rho0));
    }
}

__global__ __launch_bounds__(kBlock) void pp_density(int nf, int nt, const double *sig_in,
                                                     const double *sig_ct, const double *mass,
                                                     double rho0, double inv_sigma0, double w0,
                                                     double *rho, double *Vol)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    double rhoi = rho0;
    if (i < nf) rhoi = density_from_sigma(w0 + sig_in[i], sig_ct[i], mass[i], rho0, inv_sigma0);
    if (rhoi <= 1e-12) rhoi = rho0;
    rho[i] = rhoi;
    Vol[i] = mass[i] / rhoi;
}

__global__ __launch_bounds__(kBlock) void pl_kgc(PairView pv, const double *Vol, int nf, int nt,
                                                 double *A /* 4 x nf */)
{
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= pv.n) return;
    int ii, jj;
    if (!pair_ids(pv, k, nf, nt, ii, jj)) return;
    const double rk = pv.r[k];
    if (rk <= kRMin) return;
    const double dWk = pv.dW[k], dxk = pv.dx[k], dyk = pv.dy[k];
    const double ex = dxk / rk, ey = dyk / rk;
    const double fxj = dWk * Vol[jj];
    atomicAdd(&A[ii], -(dxk * (fxj * ex)));
    atomicAdd(&A[ii + nf], -(dxk * (fxj * ey)));
    atomicAdd(&A[ii + 2 * (size_t)nf], -(dyk * (fxj * ex)));
    atomicAdd(&A[ii + 3 * (size_t)nf], -(dyk * (fxj * ey)));
    if (jj < nf) {
        const double fxi = dWk * Vol[ii];
        atomicAdd(&A[jj], -(dxk * (fxi * ex)));
        atomicAdd(&A[jj + nf], -(dxk * (fxi * ey)));
        atomicAdd(&A[jj + 2 * (size_t)nf], -(dyk * (fxi * ex)));
        atomicAdd(&A[jj + 3 * (size_t)nf], -(dyk * (fxi * ey)));
    }
}

__global__ __launch_bounds__(kBlock) void pp_kgc(int nf, int nt, const double *A, double *B)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    Mat2 b{1.0, 0.0, 0.0, 1.0};
    if (i < nf) b = kgc_from_A(A[i], A[i + nf], A[i + 2 * (size_t)nf], A[i + 3 * (size_t)nf]);
    B[i] = b.m11;
    B[i + nt] = b.m12;
    B[i + 2 * (size_t)nt] = b.m21;
    B[i + 3 * (size_t)nt] = b.m22;
}

// ---- viscous_force --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pl_viscous(PairView pv, const double *vel,
                                                     const double *Vol, const double *B, double mu,
                                                     double h, int nf, int nt,
                                                     const double *wall_vel, double *acc)
{
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= pv.n) return;
    int ii, jj;
    if (!pair_ids(pv, k, nf, nt, ii, jj)) return;
    const double rk = pv.r[k];
    if (rk <= kRMin) return;
    const size_t n = (size_t)nt;
    const double dWk = pv.dW[k];
    const double ex = pv.dx[k] / rk, ey = pv.dy[k] / rk;
    const double b11i = B[ii], b12i = B[ii + n], b21i = B[ii + 2 * n], b22i = B[ii + 3 * n];
    const double denom = rk + 0.01 * h;
    if (jj < nf) {
        const double bs11 = b11i + B[jj], bs12 = b12i + B[jj + n];
        const double bs21 = b21i + B[jj + 2 * n], bs22 = b22i + B[jj + 3 * n];
        const double eBe = ex * (bs11 * ex + bs12 * ey) + ey * (bs21 * ex + bs22 * ey);
        const double dvx = vel[ii] - vel[jj], dvy = vel[ii + n] - vel[jj + n];
        const double coeff_i = eBe * mu * dWk * Vol[jj] / denom;
        const double coeff_j = eBe * mu * dWk * Vol[ii] / denom;
        atomicAdd(&acc[ii], coeff_i * dvx);
        atomicAdd(&acc[ii + n], coeff_i * dvy);
        atomicAdd(&acc[jj], -(coeff_j * dvx));
        atomicAdd(&acc[jj + n], -(coeff_j * dvy));
    } else {
        const double eBe = ex * (b11i * ex + b12i * ey) + ey * (b21i * ex + b22i * ey);
        const double dvx = vel[ii] - wall_vel[jj], dvy = vel[ii + n] - wall_vel[jj + n];
        const double coeff = 4.0 * eBe * mu * dWk * Vol[jj] / denom;
        atomicAdd(&acc[ii], coeff * dvx);
        atomicAdd(&acc[ii + n], coeff * dvy);
    }
}

// force = acc*Vol for fluid, 0 for walls; optionally + mass*g on x (sph_physics_mex.c:1575-1580)
__global__ __launch_bounds__(kBlock) void pp_viscous(int nf, int nt, const double *acc,
                                                     const double *Vol, const double *mass,
                                                     double gravity_g, int add_gravity,
                                                     double *force)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    double fx = 0.0, fy = 0.0;
    if (i < nf) {
        fx = acc[i] * Vol[i];
        fy = acc[i + (size_t)nt] * Vol[i];
        if (add_gravity) fx += mass[i] * gravity_g;
    }
    force[i] = fx;
    force[i + (size_t)nt] = fy;
}

// ---- transport_correction -------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pl_transport(PairView pv, const double *Vol,
                                                       const double *B, int nf, int nt, double *inc)
{
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= pv.n) return;
    int ii, jj;
    if (!pair_ids(pv, k, nf, nt, ii, jj)) return;
    const double rk = pv.r[k];
    if (rk <= kRMin) return;
    const size_t n = (size_t)nt;
    const double dWk = pv.dW[k];
    const double ex = pv.dx[k] / rk, ey = pv.dy[k] / rk;
    const double b11i = B[ii], b12i = B[ii + n], b21i = B[ii + 2 * n], b22i = B[ii + 3 * n];
    if (jj < nf) {
        const double bs11 = b11i + B[jj], bs12 = b12i + B[jj + n];
        const double bs21 = b21i + B[jj + 2 * n], bs22 = b22i + B[jj + 3 * n];
        const double tx = bs11 * ex + bs12 * ey, ty = bs21 * ex + bs22 * ey;
        const double coeff_i = -dWk * Vol[jj], coeff_j = dWk * Vol[ii];
        atomicAdd(&inc[ii], coeff_i * tx);
        atomicAdd(&inc[ii + n], coeff_i * ty);
        atomicAdd(&inc[jj], coeff_j * tx);
        atomicAdd(&inc[jj + n], coeff_j * ty);
    } else {
        const double tx = b11i * ex + b12i * ey, ty = b21i * ex + b22i * ey;
        const double coeff = -2.0 * dWk * Vol[jj];
        atomicAdd(&inc[ii], coeff * tx);
        atomicAdd(&inc[ii + n], coeff * ty);
    }
}

__global__ __launch_bounds__(kBlock) void pp_transport(int nf, int nt, const double *inc,
                                                       const double *pos, double h, double coeff,
                                                       double *pos_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    const size_t n = (size_t)nt;
    double x = pos[i], y = pos[i + n];
    if (i < nf) {
        double sx, sy;
        transport_shift(inc[i], inc[i + n], h, coeff, sx, sy);
        x += sx;
        y += sy;
    }
    pos_out[i] = x;
    pos_out[i + n] = y;
}

// ---- integration_1st ------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pp_int1_pre(int nf, int nt, const double *rho_in,
                                                      const double *drho_in, const double *pos_in,
                                                      const double *vel, double dt, double rho0,
                                                      double p0, double *rho_out, double *p_out,
                                                      double *pos_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    const size_t n = (size_t)nt;
    double rho = rho_in[i], p = 0.0, x = pos_in[i], y = pos_in[i + n];
    if (i < nf) {
        rho = rho + 0.5 * dt * drho_in[i];
        if (rho < 1e-10) rho = rho0;
        p = eos_pressure(rho, rho0, p0);
        x += 0.5 * dt * vel[i];
        y += 0.5 * dt * vel[i + n];
    }
    rho_out[i] = rho;
    p_out[i] = p;
    pos_out[i] = x;
    pos_out[i + n] = y;
}

__global__ __launch_bounds__(kBlock) void pl_int1(PairView pv, const double *Vol, const double *B,
                                                  const double *rho_h, const double *p_h,
                                                  const double *vel, const double *force_prior,
                                                  const double *mass, int nf, int nt, double rho0,
                                                  double c_f, double *F, double *diss)
{
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= pv.n) return;
    int ii, jj;
    if (!pair_ids(pv, k, nf, nt, ii, jj)) return;
    const double rk = pv.r[k];
    if (rk <= kRMin) return;
    const size_t n = (size_t)nt;
    const double dWk = pv.dW[k];
    const double ex = pv.dx[k] / rk, ey = pv.dy[k] / rk;
    const double b11i = B[ii], b12i = B[ii + n], b21i = B[ii + 2 * n], b22i = B[ii + 3 * n];
    if (jj < nf) {
        const double p_i = p_h[ii], p_j = p_h[jj];
        const double rho_bar = 0.5 * (rho_h[ii] + rho_h[jj]);
        const double un_l = vel[ii] * ex + vel[ii + n] * ey;
        const double un_r = vel[jj] * ex + vel[jj + n] * ey;
        const double beta = riemann_beta(un_l, un_r, c_f);
        const double p_star = 0.5 * (p_i + p_j) + 0.5 * beta * rho_bar * (un_l - un_r);
        const double p_face = 0.5 * (0.5 * (p_i + p_j) + p_star);
        const double tx = p_face * ((b11i + B[jj]) * ex + (b12i + B[jj + n]) * ey);
        const double ty = p_face * ((b21i + B[jj + 2 * n]) * ex + (b22i + B[jj + 3 * n]) * ey);
        const double dWVj = dWk * Vol[jj], dWVi = dWk * Vol[ii];
        const double p_diff = p_i - p_j;
        atomicAdd(&F[ii], -(tx * dWVj));
        atomicAdd(&F[ii + n], -(ty * dWVj));
        atomicAdd(&F[jj], tx * dWVi);
        atomicAdd(&F[jj + n], ty * dWVi);
        atomicAdd(&diss[ii], (p_diff / (rho0 * c_f)) * dWVj);
        atomicAdd(&diss[jj], (-p_diff / (rho0 * c_f)) * dWVi);
    } else {
        const double p_i = p_h[ii], rho_i = rho_h[ii];
        const double dWVj = dWk * Vol[jj];
        const double ax = force_prior[ii] / mass[ii], ay = force_prior[ii + n] / mass[ii];
        const double face_wall_ext_acc = -(ax * ex + ay * ey);
        const double p_wall = p_i + rho_i * rk * fmax(0.0, face_wall_ext_acc);
        const double tx = b11i * ex + b12i * ey, ty = b21i * ex + b22i * ey;
        atomicAdd(&F[ii], -((p_i + p_wall) * dWVj * tx));
        atomicAdd(&F[ii + n], -((p_i + p_wall) * dWVj * ty));
        atomicAdd(&diss[ii], ((p_i - p_wall) / (rho0 * c_f)) * dWVj);
    }
}

__global__ __launch_bounds__(kBlock) void pp_int1_post(int nf, int nt, const double *F,
                                                       const double *diss, const double *Vol,
                                                       const double *rho_h, double *force_out,
                                                       double *drho_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    const size_t n = (size_t)nt;
    double fx = 0.0, fy = 0.0, d = 0.0;
    if (i < nf) {
        fx = F[i] * Vol[i];
        fy = F[i + n] * Vol[i];
        d = diss[i] * rho_h[i];
    }
    force_out[i] = fx;
    force_out[i + n] = fy;
    drho_out[i] = d;
}

// velocity kick of integration_verlet (sph_physics_mex.c:1400-1408)
__global__ __launch_bounds__(kBlock) void pp_kick(int nf, int nt, const double *vel_in,
                                                  const double *force_prior, const double *force,
                                                  const double *mass, double dt, double *vel_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    const size_t n = (size_t)nt;
    double vx = 0.0, vy = 0.0;
    if (i < nf) {
        const double inv_mass = 1.0 / mass[i];
        vx = vel_in[i] + (force_prior[i] + force[i]) * inv_mass * dt;
        vy = vel_in[i + n] + (force_prior[i + n] + force[i + n]) * inv_mass * dt;
    }
    vel_out[i] = vx;
    vel_out[i + n] = vy;
}

// ---- integration_2nd ------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pp_int2_pre(int nf, int nt, const double *pos_in,
                                                      const double *vel, double dt, double *pos_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    const size_t n = (size_t)nt;
    double x = pos_in[i], y = pos_in[i + n];
    if (i < nf) {
        x += 0.5 * dt * vel[i];
        y += 0.5 * dt * vel[i + n];
    }
    pos_out[i] = x;
    pos_out[i + n] = y;
}

__global__ __launch_bounds__(kBlock) void pl_int2(PairView pv, const double *Vol, const double *vel,
                                                  const double *wall_vel, int nf, int nt,
                                                  double *rate)
{
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= pv.n) return;
    int ii, jj;
    if (!pair_ids(pv, k, nf, nt, ii, jj)) return;
    const double rk = pv.r[k];
    if (rk <= kRMin) return;
    const size_t n = (size_t)nt;
    const double dWk = pv.dW[k];
    const double ex = pv.dx[k] / rk, ey = pv.dy[k] / rk;
    if (jj < nf) {
        const double u_jump = (vel[ii] - vel[jj]) * ex + (vel[ii + n] - vel[jj + n]) * ey;
        atomicAdd(&rate[ii], u_jump * dWk * Vol[jj]);
        atomicAdd(&rate[jj], u_jump * dWk * Vol[ii]);
    } else {
        const double vjx = 2.0 * wall_vel[jj] - vel[ii], vjy = 2.0 * wall_vel[jj + n] - vel[ii + n];
        const double jump = (vel[ii] - vjx) * ex + (vel[ii + n] - vjy) * ey;
        atomicAdd(&rate[ii], jump * dWk * Vol[jj]);
    }
}

__global__ __launch_bounds__(kBlock) void pp_int2_post(int nf, int nt, const double *rate,
                                                       const double *rho, double *drho_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    drho_out[i] = (i < nf) ? rate[i] * rho[i] : 0.0;
}

// final half-step of integration_verlet (sph_physics_mex.c:1440-1450)
__global__ __launch_bounds__(kBlock) void pp_verlet_final(int nf, int nt, const double *rho_h,
                                                          const double *drho_new, double dt,
                                                          double rho0, double p0, double *rho_out,
                                                          double *p_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    double rho = rho_h[i], p = 0.0;
    if (i < nf) {
        rho += drho_new[i] * (0.5 * dt);
        if (rho < 1e-10) rho = rho0;
        p = eos_pressure(rho, rho0, p0);
    }
    rho_out[i] = rho;
    p_out[i] = p;
}

// ---- wall_shear_monitor ---------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pl_wallshear(PairView pv, const double *pos,
                                                       const double *vel, const double *wall_vel,
                                                       const double *Vol, const double *B, int nf,
                                                       int nt, double DH, double mu, double h,
                                                       double *sums /* [2] */)
{
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    double fb = 0.0, ft = 0.0;
    if (k < pv.n) {
        const int ii = (int)pv.pi[k] - 1, jj = (int)pv.pj[k] - 1;
        const double rk = pv.r[k];
        if (!(ii < 0 || ii >= nf || jj < nf || jj >= nt || rk <= kRMin)) {
            const size_t n = (size_t)nt;
            const double ex = pv.dx[k] / rk, ey = pv.dy[k] / rk;
            const double eBe = ex * (B[ii] * ex + B[ii + n] * ey) + ey * (B[ii + 2 * n] * ex + B[ii + 3 * n] * ey);
            const double dv_x = vel[ii] - wall_vel[jj];
            const double f_pair = 4.0 * mu * eBe * pv.dW[k] * Vol[jj] * dv_x / (rk + 0.01 * h) * Vol[ii];
            const double yj = pos[jj + n];
            if (yj <= 0.0) fb = f_pair;
            else if (yj >= DH) ft = f_pair;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        fb += __shfl_xor(fb, off);
        ft += __shfl_xor(ft, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (fb != 0.0) atomicAdd(&sums[0], fb);
        if (ft != 0.0) atomicAdd(&sums[1], ft);
    }
}

// ---------------------------------------------------------------------------------------------
// host-side composition on device pointers
// ---------------------------------------------------------------------------------------------
struct PairListDev {
    DevBuf<double> pi, pj, dx, dy, r, W, dW;
    long n = 0;
    void upload(size_t n_pairs, const double *hpi, const double *hpj, const double *hdx,
                const double *hdy, const double *hr, const double *hW, const double *hdW)
    {
        n = (long)n_pairs;
        const size_t m = n_pairs ? n_pairs : 1;
        auto up = [&](DevBuf<double> &b, const double *h) {
            b.alloc(m);
            if (h && n_pairs) b.upload(h, n_pairs);
        };
        up(pi, hpi); up(pj, hpj); up(dx, hdx); up(dy, hdy); up(r, hr);
        if (hW) up(W, hW);
        up(dW, hdW);
    }
    PairView view() const { return PairView{pi.get(), pj.get(), dx.get(), dy.get(), r.get(), W.get(), dW.get(), n}; }
};

inline unsigned gp(long n) { return n > 0 ? div_up((size_t)n, kBlock) : 1; }

struct Dev {
    // device-side versions of the modes; all pointers device, column-major like the host arrays
    static void density(const PairView &pv, const double *mass, int nf, int nt, double rho0, double h,
                        double inv_sigma0, double *rho, double *Vol, double *B)
    {
        DevBuf<double> sig_in(nf), sig_ct(nf), A((size_t)4 * nf);
        sig_in.zero(); sig_ct.zero(); A.zero();
        hipLaunchKernelGGL(pl_sigma, dim3(gp(pv.n)), dim3(kBlock), 0, 0, pv, mass, nf, nt, rho0, sig_in.get(), sig_ct.get());
        const double w0 = 10.0 / (7.0 * kPi * h * h);
        hipLaunchKernelGGL(pp_density, dim3(gp(nt)), dim3(kBlock), 0, 0, nf, nt, sig_in.get(), sig_ct.get(), mass, rho0, inv_sigma0, w0, rho, Vol);
        hipLaunchKernelGGL(pl_kgc, dim3(gp(pv.n)), dim3(kBlock), 0, 0, pv, (const double *)Vol, nf, nt, A.get());
        hipLaunchKernelGGL(pp_kgc, dim3(gp(nt)), dim3(kBlock), 0, 0, nf, nt, (const double *)A.get(), B);
        SPHX_HIP(hipGetLastError());
        SPHX_HIP(hipDeviceSynchronize());  // scratch buffers die at scope exit
    }
    static void viscous(const PairView &pv, const double *vel, const double *Vol, const double *B, double mu,
                        double h, int nf, int nt, const double *mass, const double *wall_vel, double gravity_g,
                        int add_gravity, double *force)
    {
        DevBuf<double> acc((size_t)2 * nt);
        acc.zero();
        hipLaunchKernelGGL(pl_viscous, dim3(gp(pv.n)), dim3(kBlock), 0, 0, pv, vel, Vol, B, mu, h, nf, nt, wall_vel, acc.get());
        hipLaunchKernelGGL(pp_viscous, dim3(gp(nt)), dim3(kBlock), 0, 0, nf, nt, (const double *)acc.get(), Vol, mass, gravity_g, add_gravity, force);
        SPHX_HIP(hipGetLastError());
        SPHX_HIP(hipDeviceSynchronize());
    }
    static void transport(const PairView &pv, const double *Vol, const double *B, const double *pos, double h,
                          int nf, int nt, double coeff, double *pos_out)
    {
        DevBuf<double> inc((size_t)2 * nt);
        inc.zero();
        hipLaunchKernelGGL(pl_transport, dim3(gp(pv.n)), dim3(kBlock), 0, 0, pv, Vol, B, nf, nt, inc.get());
        hipLaunchKernelGGL(pp_transport, dim3(gp(nt)), dim3(kBlock), 0, 0, nf, nt, (const double *)inc.get(), pos, h, coeff, pos_out);
        SPHX_HIP(hipGetLastError());
        SPHX_HIP(hipDeviceSynchronize());
    }
    static void int1(const PairView &pv, const double *Vol, const double *B, const double *rho, const double *mass,
                     const double *pos, const double *vel, const double *drho, const double *force_prior,
                     double dt, int nf, int nt, double rho0, double p0, double c_f, double *rho_out,
                     double *p_out, double *pos_out, double *force_out, double *drho_out)
    {
        DevBuf<double> F((size_t)2 * nt), diss(nt);
        F.zero(); diss.zero();
        hipLaunchKernelGGL(pp_int1_pre, dim3(gp(nt)), dim3(kBlock), 0, 0, nf, nt, rho, drho, pos, vel, dt, rho0, p0, rho_out, p_out, pos_out);
        hipLaunchKernelGGL(pl_int1, dim3(gp(pv.n)), dim3(kBlock), 0, 0, pv, Vol, B, (const double *)rho_out, (const double *)p_out, vel, force_prior, mass, nf, nt, rho0, c_f, F.get(), diss.get());
        hipLaunchKernelGGL(pp_int1_post, dim3(gp(nt)), dim3(kBlock), 0, 0, nf, nt, (const double *)F.get(), (const double *)diss.get(), Vol, (const double *)rho_out, force_out, drho_out);
        SPHX_HIP(hipGetLastError());
        SPHX_HIP(hipDeviceSynchronize());
    }
    static void int2(const PairView &pv, const double *Vol, const double *rho, const double *pos, const double *vel,
                     double dt, int nf, int nt, const double *wall_vel, double *pos_out, double *drho_out)
    {
        DevBuf<double> rate(nt);
        rate.zero();
        hipLaunchKernelGGL(pp_int2_pre, dim3(gp(nt)), dim3(kBlock), 0, 0, nf, nt, pos, vel, dt, pos_out);
        hipLaunchKernelGGL(pl_int2, dim3(gp(pv.n)), dim3(kBlock), 0, 0, pv, Vol, vel, wall_vel, nf, nt, rate.get());
        hipLaunchKernelGGL(pp_int2_post, dim3(gp(nt)), dim3(kBlock), 0, 0, nf, nt, (const double *)rate.get(), rho, drho_out);
        SPHX_HIP(hipGetLastError());
        SPHX_HIP(hipDeviceSynchronize());
    }
    static void verlet(const PairView &pv, const double *Vol, const double *B, const double *rho, const double *mass,
                       const double *pos, const double *vel, const double *drho, const double *force_prior,
                       double dt, int nf, int nt, double rho0, double p0, double c_f, const double *wall_vel,
                       double *rho_out, double *p_out, double *pos_out, double *vel_out, double *drho_out,
                       double *force_out)
    {
        DevBuf<double> rho_h(nt), p_h(nt), pos_h((size_t)2 * nt), diss(nt);
        int1(pv, Vol, B, rho, mass, pos, vel, drho, force_prior, dt, nf, nt, rho0, p0, c_f, rho_h.get(), p_h.get(),
             pos_h.get(), force_out, diss.get());
        hipLaunchKernelGGL(pp_kick, dim3(gp(nt)), dim3(kBlock), 0, 0, nf, nt, vel, force_prior, (const double *)force_out, mass, dt, vel_out);
        int2(pv, Vol, rho_h.get(), pos_h.get(), vel_out, dt, nf, nt, wall_vel, pos_out, drho_out);
        hipLaunchKernelGGL(pp_verlet_final, dim3(gp(nt)), dim3(kBlock), 0, 0, nf, nt, (const double *)rho_h.get(), (const double *)drho_out, dt, rho0, p0, rho_out, p_out);
        SPHX_HIP(hipGetLastError());
        SPHX_HIP(hipDeviceSynchronize());
    }
};

struct Up {  // host array -> device copy with MEX [n x c] size
    DevBuf<double> b;
    Up(const double *h, size_t n) : b(n ? n : 1) { if (n) b.upload(h, n); }
    const double *get() const { return b.get(); }
};

void check_counts(int nf, int nt, const char *id)
{
    require(nf > 0 && nt >= nf, id, "Invalid n_fluid/n_total.");
}

void check_pairs(size_t n_pairs, const char *id)
{
    require(n_pairs <= (size_t)2147483647, id, "Pair count exceeds INT_MAX.");
}

}  // namespace
}  // namespace sphx

using namespace sphx;

SPHX_EXPORT int sphx_density_correction(size_t n_pairs, const double *pair_i, const double *pair_j,
                                        const double *dx, const double *dy, const double *r,
                                        const double *W, const double *dW, const double *mass,
                                        int n_fluid, int n_total, double rho0, double h,
                                        double inv_sigma0, double *rho, double *Vol, double *B)
{
    SPHX_TRY
    check_pairs(n_pairs, "SPH:Physics:density:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:density:count");
    require(rho0 > 0.0 && h > 0.0, "SPH:Physics:density:param", "rho0 and h must be positive.");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairListDev pl;
    pl.upload(n_pairs, pair_i, pair_j, dx, dy, r, W, dW);
    Up d_mass(mass, nt);
    DevBuf<double> d_rho(nt), d_Vol(nt), d_B(4 * nt);
    Dev::density(pl.view(), d_mass.get(), n_fluid, n_total, rho0, h, inv_sigma0, d_rho.get(), d_Vol.get(), d_B.get());
    d_rho.download(rho, nt); d_Vol.download(Vol, nt); d_B.download(B, 4 * nt);
    SPHX_HIP(hipDeviceSynchronize());
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_viscous_force(size_t n_pairs, const double *pair_i, const double *pair_j,
                                   const double *dx, const double *dy, const double *r,
                                   const double *dW, const double *vel, const double *Vol,
                                   const double *B, double mu, double h, int n_fluid, int n_total,
                                   const double *mass, const double *wall_vel, double *force)
{
    SPHX_TRY
    check_pairs(n_pairs, "SPH:Physics:viscous:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:viscous:count");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairListDev pl;
    pl.upload(n_pairs, pair_i, pair_j, dx, dy, r, nullptr, dW);
    Up d_vel(vel, 2 * nt), d_Vol(Vol, nt), d_B(B, 4 * nt), d_mass(mass, nt), d_wv(wall_vel, 2 * nt);
    DevBuf<double> d_force(2 * nt);
    Dev::viscous(pl.view(), d_vel.get(), d_Vol.get(), d_B.get(), mu, h, n_fluid, n_total, d_mass.get(), d_wv.get(), 0.0, 0, d_force.get());
    d_force.download(force, 2 * nt);
    SPHX_HIP(hipDeviceSynchronize());
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_transport_correction(size_t n_pairs, const double *pair_i,
                                          const double *pair_j, const double *dx, const double *dy,
                                          const double *r, const double *dW, const double *Vol,
                                          const double *B, const double *pos, double h, int n_fluid,
                                          int n_total, double transport_coeff, double *pos_out)
{
    SPHX_TRY
    check_pairs(n_pairs, "SPH:Physics:transport:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:transport:count");
    require(transport_coeff >= 0.0, "SPH:Physics:transport:coeff", "transport_coeff must be non-negative.");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairListDev pl;
    pl.upload(n_pairs, pair_i, pair_j, dx, dy, r, nullptr, dW);
    Up d_Vol(Vol, nt), d_B(B, 4 * nt), d_pos(pos, 2 * nt);
    DevBuf<double> d_out(2 * nt);
    Dev::transport(pl.view(), d_Vol.get(), d_B.get(), d_pos.get(), h, n_fluid, n_total, transport_coeff, d_out.get());
    d_out.download(pos_out, 2 * nt);
    SPHX_HIP(hipDeviceSynchronize());
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_integration_1st(size_t n_pairs, const double *pair_i, const double *pair_j,
                                     const double *dx, const double *dy, const double *r,
                                     const double *dW, const double *Vol, const double *B,
                                     const double *rho, const double *mass, const double *pos,
                                     const double *vel, const double *drho_dt,
                                     const double *force_prior, double dt, int n_fluid, int n_total,
                                     double rho0, double p0, double c_f, const double *wall_vel,
                                     double *rho_out, double *p_out, double *pos_out,
                                     double *force_out, double *drho_out)
{
    SPHX_TRY
    (void)wall_vel;
    check_pairs(n_pairs, "SPH:Physics:int1:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:int1:count");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairListDev pl;
    pl.upload(n_pairs, pair_i, pair_j, dx, dy, r, nullptr, dW);
    Up d_Vol(Vol, nt), d_B(B, 4 * nt), d_rho(rho, nt), d_mass(mass, nt), d_pos(pos, 2 * nt), d_vel(vel, 2 * nt),
        d_drho(drho_dt, nt), d_fp(force_prior, 2 * nt);
    DevBuf<double> o_rho(nt), o_p(nt), o_pos(2 * nt), o_f(2 * nt), o_d(nt);
    Dev::int1(pl.view(), d_Vol.get(), d_B.get(), d_rho.get(), d_mass.get(), d_pos.get(), d_vel.get(), d_drho.get(),
              d_fp.get(), dt, n_fluid, n_total, rho0, p0, c_f, o_rho.get(), o_p.get(), o_pos.get(), o_f.get(), o_d.get());
    o_rho.download(rho_out, nt); o_p.download(p_out, nt); o_pos.download(pos_out, 2 * nt);
    o_f.download(force_out, 2 * nt); o_d.download(drho_out, nt);
    SPHX_HIP(hipDeviceSynchronize());
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_integration_2nd(size_t n_pairs, const double *pair_i, const double *pair_j,
                                     const double *dx, const double *dy, const double *r,
                                     const double *dW, const double *Vol, const double *rho,
                                     const double *pos, const double *vel, double dt, int n_fluid,
                                     int n_total, const double *wall_vel, double *pos_out,
                                     double *drho_out, double *zeros_out)
{
    SPHX_TRY
    check_pairs(n_pairs, "SPH:Physics:int2:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:int2:count");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairListDev pl;
    pl.upload(n_pairs, pair_i, pair_j, dx, dy, r, nullptr, dW);
    Up d_Vol(Vol, nt), d_rho(rho, nt), d_pos(pos, 2 * nt), d_vel(vel, 2 * nt), d_wv(wall_vel, 2 * nt);
    DevBuf<double> o_pos(2 * nt), o_d(nt);
    Dev::int2(pl.view(), d_Vol.get(), d_rho.get(), d_pos.get(), d_vel.get(), dt, n_fluid, n_total, d_wv.get(), o_pos.get(), o_d.get());
    o_pos.download(pos_out, 2 * nt); o_d.download(drho_out, nt);
    SPHX_HIP(hipDeviceSynchronize());
    if (zeros_out) std::memset(zeros_out, 0, 2 * nt * sizeof(double));  // sph_physics_mex.c:1064
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_integration_verlet(size_t n_pairs, const double *pair_i, const double *pair_j,
                                        const double *dx, const double *dy, const double *r,
                                        const double *dW, const double *Vol, const double *B,
                                        const double *rho, const double *mass, const double *pos,
                                        const double *vel, const double *drho_dt,
                                        const double *force_prior, double dt, int n_fluid,
                                        int n_total, double rho0, double p0, double c_f,
                                        const double *wall_vel, double *rho_out, double *p_out,
                                        double *pos_out, double *vel_out, double *drho_out,
                                        double *force_out)
{
    SPHX_TRY
    check_pairs(n_pairs, "SPH:Physics:verlet:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:verlet:count");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairListDev pl;
    pl.upload(n_pairs, pair_i, pair_j, dx, dy, r, nullptr, dW);
    Up d_Vol(Vol, nt), d_B(B, 4 * nt), d_rho(rho, nt), d_mass(mass, nt), d_pos(pos, 2 * nt), d_vel(vel, 2 * nt),
        d_drho(drho_dt, nt), d_fp(force_prior, 2 * nt), d_wv(wall_vel, 2 * nt);
    DevBuf<double> o_rho(nt), o_p(nt), o_pos(2 * nt), o_vel(2 * nt), o_d(nt), o_f(2 * nt);
    Dev::verlet(pl.view(), d_Vol.get(), d_B.get(), d_rho.get(), d_mass.get(), d_pos.get(), d_vel.get(), d_drho.get(),
                d_fp.get(), dt, n_fluid, n_total, rho0, p0, c_f, d_wv.get(), o_rho.get(), o_p.get(), o_pos.get(),
                o_vel.get(), o_d.get(), o_f.get());
    o_rho.download(rho_out, nt); o_p.download(p_out, nt); o_pos.download(pos_out, 2 * nt);
    o_vel.download(vel_out, 2 * nt); o_d.download(drho_out, nt); o_f.download(force_out, 2 * nt);
    SPHX_HIP(hipDeviceSynchronize());
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_advance_shell_step(size_t n_pairs, const double *pair_i, const double *pair_j,
                                        const double *dx, const double *dy, const double *r,
                                        const double *W, const double *dW, const double *mass,
                                        const double *pos, const double *vel,
                                        const double *wall_vel, const double *rho,
                                        const double *drho_dt, double dt, int n_fluid, int n_total,
                                        double rho0, double p0, double c_f, double mu, double h,
                                        double inv_sigma0, double gravity_g, double *rho_out,
                                        double *p_out, double *pos_out, double *vel_out,
                                        double *drho_out, double *force_out,
                                        double *force_prior_out, double *Vol_out, double *B_out)
{
    SPHX_TRY
    (void)rho;  // size-checked only in the reference (sph_physics_mex.c:1532); density is re-summed
    check_pairs(n_pairs, "SPH:Physics:advance:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:advance:count");
    require(rho0 > 0.0 && h > 0.0, "SPH:Physics:density:param", "rho0 and h must be positive.");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairListDev pl;
    pl.upload(n_pairs, pair_i, pair_j, dx, dy, r, W, dW);
    Up d_mass(mass, nt), d_pos(pos, 2 * nt), d_vel(vel, 2 * nt), d_wv(wall_vel, 2 * nt), d_drho(drho_dt, nt);
    DevBuf<double> rho_d(nt), Vol(nt), B(4 * nt), fp(2 * nt), pos_t(2 * nt);
    DevBuf<double> o_rho(nt), o_p(nt), o_pos(2 * nt), o_vel(2 * nt), o_d(nt), o_f(2 * nt);
    const PairView pv = pl.view();
    Dev::density(pv, d_mass.get(), n_fluid, n_total, rho0, h, inv_sigma0, rho_d.get(), Vol.get(), B.get());
    Dev::viscous(pv, d_vel.get(), Vol.get(), B.get(), mu, h, n_fluid, n_total, d_mass.get(), d_wv.get(), gravity_g, 1, fp.get());
    Dev::transport(pv, Vol.get(), B.get(), d_pos.get(), h, n_fluid, n_total, 0.2, pos_t.get());  // :584,:1596
    Dev::verlet(pv, Vol.get(), B.get(), rho_d.get(), d_mass.get(), pos_t.get(), d_vel.get(), d_drho.get(), fp.get(), dt,
                n_fluid, n_total, rho0, p0, c_f, d_wv.get(), o_rho.get(), o_p.get(), o_pos.get(), o_vel.get(),
                o_d.get(), o_f.get());
    o_rho.download(rho_out, nt); o_p.download(p_out, nt); o_pos.download(pos_out, 2 * nt);
    o_vel.download(vel_out, 2 * nt); o_d.download(drho_out, nt); o_f.download(force_out, 2 * nt);
    fp.download(force_prior_out, 2 * nt); Vol.download(Vol_out, nt); B.download(B_out, 4 * nt);
    SPHX_HIP(hipDeviceSynchronize());
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_wall_shear_monitor(size_t n_pairs, const double *pair_i, const double *pair_j,
                                        const double *dx, const double *dy, const double *r,
                                        const double *dW, const double *pos, const double *vel,
                                        const double *wall_vel, const double *Vol, const double *B,
                                        int n_fluid, int n_total, double DL, double DH, double mu,
                                        double h, double *tau_bottom, double *tau_top)
{
    SPHX_TRY
    check_pairs(n_pairs, "SPH:Physics:wallshear:pairsize");
    require(DL > 0.0 && h > 0.0, "SPH:Physics:wallshear:param", "DL and h must be positive.");
    require(n_total > 0 && n_fluid >= 0 && n_fluid <= n_total, "SPH:Physics:wallshear:count", "Invalid n_fluid/n_total.");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairListDev pl;
    pl.upload(n_pairs, pair_i, pair_j, dx, dy, r, nullptr, dW);
    Up d_pos(pos, 2 * nt), d_vel(vel, 2 * nt), d_wv(wall_vel, 2 * nt), d_Vol(Vol, nt), d_B(B, 4 * nt);
    DevBuf<double> sums(2);
    sums.zero();
    hipLaunchKernelGGL(pl_wallshear, dim3(gp(pl.n)), dim3(kBlock), 0, 0, pl.view(), d_pos.get(), d_vel.get(), d_wv.get(),
                       d_Vol.get(), d_B.get(), n_fluid, n_total, DH, mu, h, sums.get());
    SPHX_HIP(hipGetLastError());
    double hs[2];
    sums.download(hs, 2);
    SPHX_HIP(hipDeviceSynchronize());
    *tau_bottom = -hs[0] / DL;
    *tau_top = -hs[1] / DL;
    return SPHX_OK;
    SPHX_CATCH
}
