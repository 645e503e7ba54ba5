// sphx_pairlist.hip -- the eight sph_physics_shell_mex modes on a caller-supplied flat pair list (the stateless
// MEX surface, include/sphx.h section 1), as per-particle GATHERS.
//
// The caller's list names every fluid-fluid pair once and every fluid-wall pair once, in any order.  Instead of
// walking the pairs and scattering into both partners (what mex/sph_physics_mex.c does under OpenMP atomics), each
// call first turns the list into an incidence structure on the device --
//     row[i] .. row[i+1]  : the pairs fluid particle i takes part in, as  (pair index << 1) | side
//                           side 0: i is the pair's first particle, side 1: its second (fluid-fluid only)
// sorted by pair index inside every row (count -> scan -> fill -> rank), and then every mode is a sum over a
// particle's own row: kLanes lanes of a wavefront stride over the row and combine with a fixed shuffle tree.  No
// floating-point atomics anywhere, the summation order is a function of the list alone, so results are bitwise
// repeatable; the pair geometry (dx, dy, r, W, dW) is the caller's, never recomputed.  Every fluid-fluid term of
// the reference is symmetric under exchanging the partners, so the second particle of a
// pair evaluates the same one-sided formula with the separation vector reversed.
//
// Formula sources: density / KGC sph_physics_mex.c:188-234,239-366; viscous :469-545; transport :636-710;
// integration_1st :857-957 (+ riemann_beta :1121-1129); integration_2nd :1066-1116; verlet :1386-1450;
// advance_shell_step :1560-1631; wall shear :1713-1742.  Pairs the reference skips (index out of range, r <= 1e-12)
// are skipped here as well.
#include "sphx_common.hpp"
#include "sphx_device.hpp"

namespace sphx {
namespace {

constexpr int kBlock = 256;
constexpr int kLanes = 8;        // lanes sharing one particle's row (rows hold ~25-45 pairs)
constexpr int kScanTile = 1024;  // row-length scan: one tile per workgroup

// the caller's pair list on the device plus the incidence rows built from it
struct Incidence {
    const double *first, *second, *dx, *dy, *r, *W, *dW;
    long n_pairs;
    const int *row;        // [nf + 1]
    const unsigned *ent;   // [row[nf]]
    int nf, nt;
};

// what a lane sees of one incident pair: the partner, and the unit vector pointing from the partner to `me`
struct Incident {
    int other;
    bool other_is_wall;
    double ex, ey, r, W, dW;
    double dxs, dys;  // separation me - other (the caller's dx, dy, reversed for the pair's second particle)
};

__device__ __forceinline__ bool pair_in_range(const Incidence &I, long k, int &a, int &b)
{
    a = (int)I.first[k] - 1;
    b = (int)I.second[k] - 1;
    return a >= 0 && a < I.nf && b >= 0 && b < I.nt;
}

__device__ __forceinline__ Incident load_incident(const Incidence &I, unsigned e)
{
    const long k = (long)(e >> 1);
    const bool second = (e & 1u) != 0;
    Incident q;
    q.other = (int)(second ? I.first[k] : I.second[k]) - 1;
    q.other_is_wall = q.other >= I.nf;
    q.r = I.r[k];
    q.W = I.W ? I.W[k] : 0.0;
    q.dW = I.dW[k];
    const double sgn = second ? -1.0 : 1.0;
    q.dxs = sgn * I.dx[k];
    q.dys = sgn * I.dy[k];
    q.ex = q.dxs / q.r;
    q.ey = q.dys / q.r;
    return q;
}

template <typename T>
__device__ __forceinline__ T lanes_sum(T v)
{
#pragma unroll
    for (int off = kLanes / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// thread layout of every gather kernel: kLanes consecutive lanes own fluid particle `i`
#define SPHX_ROW_THREAD()                                                   \
    const long gt = (long)blockIdx.x * kBlock + threadIdx.x;                 \
    const int i = (int)(gt / kLanes), sub = (int)(gt % kLanes);              \
    const bool live = i < I.nf;                                              \
    const int m0 = live ? I.row[i] : 0, m1 = live ? I.row[i + 1] : 0

// ---------------------------------------------------------------------------------------------------------------
// incidence build
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void inc_degree(Incidence I, int *deg)
{
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= I.n_pairs) return;
    int a, b;
    if (!pair_in_range(I, k, a, b)) return;
    atomicAdd(&deg[a], 1);
    if (b < I.nf) atomicAdd(&deg[b], 1);
}

// exclusive scan of deg[0..n) in tiles of kScanTile: tile-local scan + tile totals, scan of the totals (one
// workgroup), then the tile offsets are added back
__device__ __forceinline__ int tile_exclusive_scan(int v, int &total, int *s_w /*[kScanTile/64 + 1]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        const int ws = lane < kScanTile / 64 ? s_w[lane] : 0;
        int winc = ws;
#pragma unroll
        for (int off = 1; off < kScanTile / 64; off <<= 1) {
            const int o = __shfl_up(winc, off);
            if (lane >= off) winc += o;
        }
        if (lane < kScanTile / 64) s_w[lane] = winc - ws;
        if (lane == kScanTile / 64 - 1) s_w[kScanTile / 64] = winc;
    }
    __syncthreads();
    const int res = s_w[wave] + inc - v;
    total = s_w[kScanTile / 64];
    __syncthreads();
    return res;
}

__global__ __launch_bounds__(kScanTile) void inc_scan_tiles(const int *deg, int *row, int *tile_total, int n)
{
    __shared__ int s_w[kScanTile / 64 + 1];
    const int idx = blockIdx.x * kScanTile + (int)threadIdx.x;
    int total;
    const int ex = tile_exclusive_scan(idx < n ? deg[idx] : 0, total, s_w);
    if (idx < n) row[idx] = ex;
    if (threadIdx.x == 0) tile_total[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanTile) void inc_scan_totals(int *tile_total, int n_tiles)
{  // in place: tile_total[t] becomes the offset of tile t, tile_total[n_tiles] the grand total
    __shared__ int s_w[kScanTile / 64 + 1];
    int carry = 0;
    for (int base = 0; base < n_tiles; base += kScanTile) {
        const int idx = base + (int)threadIdx.x;
        int total;
        const int ex = tile_exclusive_scan(idx < n_tiles ? tile_total[idx] : 0, total, s_w);
        if (idx < n_tiles) tile_total[idx] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) tile_total[n_tiles] = carry;
}

__global__ __launch_bounds__(kScanTile) void inc_scan_add(int *row, const int *tile_off, int n, int n_tiles)
{
    const int idx = blockIdx.x * kScanTile + (int)threadIdx.x;
    if (idx < n) row[idx] += tile_off[blockIdx.x];
    if (idx == 0) row[n] = tile_off[n_tiles];
}

// arrival order (integer atomics); inc_rank makes it canonical
__global__ __launch_bounds__(kBlock) void inc_fill(Incidence I, const int *row, int *cursor, unsigned *ent)
{
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= I.n_pairs) return;
    int a, b;
    if (!pair_in_range(I, k, a, b)) return;
    ent[row[a] + atomicAdd(&cursor[a], 1)] = (unsigned)k << 1;
    if (b < I.nf) ent[row[b] + atomicAdd(&cursor[b], 1)] = ((unsigned)k << 1) | 1u;
}

// every entry finds its rank among the entries of its row (ascending pair index, first-particle side before
// second) and moves there: the row order no longer depends on which thread arrived first
__global__ __launch_bounds__(kBlock) void inc_rank(Incidence I, const int *row, const unsigned *ent_in, unsigned *ent_out)
{
    SPHX_ROW_THREAD();
    (void)live;
    for (int m = m0 + sub; m < m1; m += kLanes) {
        const unsigned mine = ent_in[m];
        int rank = 0;
        for (int o = m0; o < m1; ++o) rank += ent_in[o] < mine ? 1 : 0;
        ent_out[m0 + rank] = mine;
    }
    (void)row;
}

// ---------------------------------------------------------------------------------------------------------------
// density_correction: number density -> rho, Vol; then the gradient-correction matrix
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void g_density(Incidence I, const double *mass, double rho0, double inv_sigma0,
                                                    double w_self, double *rho, double *Vol)
{
    SPHX_ROW_THREAD();
    double same = 0.0, contact = 0.0;  // kernel sums over fluid / wall partners
    for (int m = m0 + sub; m < m1; m += kLanes) {
        const unsigned e = I.ent[m];
        const long k = (long)(e >> 1);
        const int other = (int)((e & 1u) ? I.first[k] : I.second[k]) - 1;
        const double w = I.W[k];
        if (other < I.nf) same += w;
        else contact += w * (mass[other] / rho0);
    }
    same = lanes_sum(same);
    contact = lanes_sum(contact);
    if (live && sub == 0) {
        const double d = density_from_sigma(w_self + same, contact, mass[i], rho0, inv_sigma0);
        rho[i] = d;
        Vol[i] = mass[i] / d;
    }
}

__global__ __launch_bounds__(kBlock) void g_wall_defaults(int nf, int nt, const double *mass, double rho0, double *rho,
                                                          double *Vol, double *B)
{  // wall rows: rho0, m/rho0, identity (sph_physics_mex.c:214-216,233,314-319)
    const int i = nf + blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    if (rho) rho[i] = rho0;
    if (Vol) Vol[i] = mass[i] / rho0;
    if (B) {
        B[i] = 1.0;
        B[i + (size_t)nt] = 0.0;
        B[i + 2 * (size_t)nt] = 0.0;
        B[i + 3 * (size_t)nt] = 1.0;
    }
}

__global__ __launch_bounds__(kBlock) void g_kgc(Incidence I, const double *Vol, double *B)
{
    SPHX_ROW_THREAD();
    double a11 = 0.0, a12 = 0.0, a21 = 0.0, a22 = 0.0;
    for (int m = m0 + sub; m < m1; m += kLanes) {
        const Incident q = load_incident(I, I.ent[m]);
        if (q.r <= kRMin) continue;
        const double g = q.dW * Vol[q.other];  // same term for fluid and wall partners
        a11 -= q.dxs * (g * q.ex);
        a12 -= q.dxs * (g * q.ey);
        a21 -= q.dys * (g * q.ex);
        a22 -= q.dys * (g * q.ey);
    }
    a11 = lanes_sum(a11); a12 = lanes_sum(a12); a21 = lanes_sum(a21); a22 = lanes_sum(a22);
    if (live && sub == 0) {
        const Mat2 b = kgc_from_A(a11, a12, a21, a22);
        const size_t n = (size_t)I.nt;
        B[i] = b.m11; B[i + n] = b.m12; B[i + 2 * n] = b.m21; B[i + 3 * n] = b.m22;
    }
}

struct Bmat {
    double b11, b12, b21, b22;
};
__device__ __forceinline__ Bmat load_B(const double *B, int i, size_t n)
{
    return Bmat{B[i], B[i + n], B[i + 2 * n], B[i + 3 * n]};
}

// ---------------------------------------------------------------------------------------------------------------
// viscous_force (+ optional gravity, advance_shell_step) and transport_correction
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void g_viscous(Incidence I, const double *vel, const double *Vol, const double *B,
                                                    const double *wall_vel, const double *mass, double mu, double h,
                                                    double gravity_g, int add_gravity, double *force)
{
    SPHX_ROW_THREAD();
    const size_t n = (size_t)I.nt;
    double ax = 0.0, ay = 0.0;
    if (live) {
        const Bmat Bi = load_B(B, i, n);
        const double vxi = vel[i], vyi = vel[i + n];
        for (int m = m0 + sub; m < m1; m += kLanes) {
            const Incident q = load_incident(I, I.ent[m]);
            if (q.r <= kRMin) continue;
            const int j = q.other;
            const double soft = q.r + 0.01 * h;
            if (!q.other_is_wall) {
                const Bmat Bj = load_B(B, j, n);
                const double tx = (Bi.b11 + Bj.b11) * q.ex + (Bi.b12 + Bj.b12) * q.ey;
                const double ty = (Bi.b21 + Bj.b21) * q.ex + (Bi.b22 + Bj.b22) * q.ey;
                const double c = (q.ex * tx + q.ey * ty) * mu * q.dW * Vol[j] / soft;
                ax += c * (vxi - vel[j]);
                ay += c * (vyi - vel[j + n]);
            } else {
                const double tx = Bi.b11 * q.ex + Bi.b12 * q.ey, ty = Bi.b21 * q.ex + Bi.b22 * q.ey;
                const double c = 4.0 * (q.ex * tx + q.ey * ty) * mu * q.dW * Vol[j] / soft;
                ax += c * (vxi - wall_vel[j]);
                ay += c * (vyi - wall_vel[j + n]);
            }
        }
    }
    ax = lanes_sum(ax);
    ay = lanes_sum(ay);
    if (live && sub == 0) {
        double fx = ax * Vol[i];
        if (add_gravity) fx += mass[i] * gravity_g;
        force[i] = fx;
        force[i + n] = ay * Vol[i];
    }
}

__global__ __launch_bounds__(kBlock) void g_transport(Incidence I, const double *Vol, const double *B, const double *pos,
                                                      double h, double coeff, double *pos_out)
{
    SPHX_ROW_THREAD();
    const size_t n = (size_t)I.nt;
    double sx = 0.0, sy = 0.0;
    if (live) {
        const Bmat Bi = load_B(B, i, n);
        for (int m = m0 + sub; m < m1; m += kLanes) {
            const Incident q = load_incident(I, I.ent[m]);
            if (q.r <= kRMin) continue;
            const int j = q.other;
            if (!q.other_is_wall) {
                const Bmat Bj = load_B(B, j, n);
                const double g = q.dW * Vol[j];
                sx -= g * ((Bi.b11 + Bj.b11) * q.ex + (Bi.b12 + Bj.b12) * q.ey);
                sy -= g * ((Bi.b21 + Bj.b21) * q.ex + (Bi.b22 + Bj.b22) * q.ey);
            } else {
                const double g = 2.0 * q.dW * Vol[j];
                sx -= g * (Bi.b11 * q.ex + Bi.b12 * q.ey);
                sy -= g * (Bi.b21 * q.ex + Bi.b22 * q.ey);
            }
        }
    }
    sx = lanes_sum(sx);
    sy = lanes_sum(sy);
    if (live && sub == 0) {
        double mx, my;
        transport_shift(sx, sy, h, coeff, mx, my);
        pos_out[i] = pos[i] + mx;
        pos_out[i + n] = pos[i + n] + my;
    }
}

// rows nf..nt-1 of a two-column output: copy of `src` (positions) or zeros (forces, velocities)
__global__ __launch_bounds__(kBlock) void g_wall_rows2(int nf, int nt, const double *src, double *dst)
{
    const int i = nf + blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    dst[i] = src ? src[i] : 0.0;
    dst[i + (size_t)nt] = src ? src[i + (size_t)nt] : 0.0;
}

__global__ __launch_bounds__(kBlock) void g_wall_rows1(int nf, int nt, const double *src, double *dst)
{
    const int i = nf + blockIdx.x * kBlock + threadIdx.x;
    if (i < nt) dst[i] = src ? src[i] : 0.0;
}

// ---------------------------------------------------------------------------------------------------------------
// integration_1st: half-step density / pressure / position, then the pressure force and the dissipative rate
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void g_half_state(int nf, int nt, const double *rho_in, const double *drho_in,
                                                       const double *pos_in, const double *vel, double dt, double rho0,
                                                       double p0, double *rho_h, double *p_h, double *pos_h)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    const size_t n = (size_t)nt;
    double d = rho_in[i], p = 0.0, x = pos_in[i], y = pos_in[i + n];
    if (i < nf) {
        d += 0.5 * dt * drho_in[i];
        if (d < 1e-10) d = rho0;
        p = eos_pressure(d, rho0, p0);
        x += 0.5 * dt * vel[i];
        y += 0.5 * dt * vel[i + n];
    }
    rho_h[i] = d;
    p_h[i] = p;
    pos_h[i] = x;
    pos_h[i + n] = y;
}

__global__ __launch_bounds__(kBlock) void g_pressure(Incidence I, const double *Vol, const double *B, const double *rho_h,
                                                     const double *p_h, const double *vel, const double *force_prior,
                                                     const double *mass, double rho0, double c_f, double *force,
                                                     double *diss_out)
{
    SPHX_ROW_THREAD();
    const size_t n = (size_t)I.nt;
    double fx = 0.0, fy = 0.0, ds = 0.0;
    if (live) {
        const Bmat Bi = load_B(B, i, n);
        const double vxi = vel[i], vyi = vel[i + n], pi_ = p_h[i], di = rho_h[i];
        const double gx = force_prior[i] / mass[i], gy = force_prior[i + n] / mass[i];  // acceleration seen by the wall
        const double inv_imp = 1.0 / (rho0 * c_f);
        for (int m = m0 + sub; m < m1; m += kLanes) {
            const Incident q = load_incident(I, I.ent[m]);
            if (q.r <= kRMin) continue;
            const int j = q.other;
            const double g = q.dW * Vol[j];
            if (!q.other_is_wall) {
                const Bmat Bj = load_B(B, j, n);
                const double pj = p_h[j];
                const double ui = vxi * q.ex + vyi * q.ey, uj = vel[j] * q.ex + vel[j + n] * q.ey;
                const double mean = 0.5 * (pi_ + pj);
                const double riem = mean + 0.5 * riemann_beta(ui, uj, c_f) * (0.5 * (di + rho_h[j])) * (ui - uj);
                const double face = 0.5 * (mean + riem);
                fx -= (face * ((Bi.b11 + Bj.b11) * q.ex + (Bi.b12 + Bj.b12) * q.ey)) * g;
                fy -= (face * ((Bi.b21 + Bj.b21) * q.ex + (Bi.b22 + Bj.b22) * q.ey)) * g;
                ds += ((pi_ - pj) * inv_imp) * g;
            } else {
                const double pw = pi_ + di * q.r * fmax(0.0, -(gx * q.ex + gy * q.ey));
                fx -= (pi_ + pw) * g * (Bi.b11 * q.ex + Bi.b12 * q.ey);
                fy -= (pi_ + pw) * g * (Bi.b21 * q.ex + Bi.b22 * q.ey);
                ds += ((pi_ - pw) * inv_imp) * g;
            }
        }
    }
    fx = lanes_sum(fx);
    fy = lanes_sum(fy);
    ds = lanes_sum(ds);
    if (live && sub == 0) {
        force[i] = fx * Vol[i];
        force[i + n] = fy * Vol[i];
        diss_out[i] = ds * rho_h[i];
    }
}

// velocity kick between the two halves of integration_verlet (sph_physics_mex.c:1400-1408); walls come out zero
__global__ __launch_bounds__(kBlock) void g_kick(int nf, int nt, const double *vel, const double *fa, const double *fb,
                                                 const double *mass, double dt, double *vel_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    const size_t n = (size_t)nt;
    double vx = 0.0, vy = 0.0;
    if (i < nf) {
        const double w = 1.0 / mass[i];
        vx = vel[i] + (fa[i] + fb[i]) * w * dt;
        vy = vel[i + n] + (fa[i + n] + fb[i + n]) * w * dt;
    }
    vel_out[i] = vx;
    vel_out[i + n] = vy;
}

// ---------------------------------------------------------------------------------------------------------------
// integration_2nd: second position half-drift and the continuity rate
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void g_drift(int nf, int nt, const double *pos_in, const double *vel, double dt,
                                                  double *pos_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nt) return;
    const size_t n = (size_t)nt;
    const double s = i < nf ? 0.5 * dt : 0.0;
    pos_out[i] = pos_in[i] + s * vel[i];
    pos_out[i + n] = pos_in[i + n] + s * vel[i + n];
}

// final: 0 = drho_dt only (integration_2nd); 1 = also the closing half-step of density and the EOS (verlet :1440-1450)
__global__ __launch_bounds__(kBlock) void g_continuity(Incidence I, const double *Vol, const double *vel,
                                                       const double *wall_vel, const double *rho_h, double dt, double rho0,
                                                       double p0, int final, double *drho_out, double *rho_out,
                                                       double *p_out)
{
    SPHX_ROW_THREAD();
    const size_t n = (size_t)I.nt;
    double rate = 0.0;
    if (live) {
        const double vxi = vel[i], vyi = vel[i + n];
        for (int m = m0 + sub; m < m1; m += kLanes) {
            const Incident q = load_incident(I, I.ent[m]);
            if (q.r <= kRMin) continue;
            const int j = q.other;
            double ux, uy;  // velocity the partner presents: its own, or the wall's mirrored through the particle's
            if (!q.other_is_wall) { ux = vel[j]; uy = vel[j + n]; }
            else { ux = 2.0 * wall_vel[j] - vxi; uy = 2.0 * wall_vel[j + n] - vyi; }
            rate += ((vxi - ux) * q.ex + (vyi - uy) * q.ey) * q.dW * Vol[j];
        }
    }
    rate = lanes_sum(rate);
    if (live && sub == 0) {
        const double d_new = rate * rho_h[i];
        drho_out[i] = d_new;
        if (final) {
            double d = rho_h[i] + d_new * (0.5 * dt);
            if (d < 1e-10) d = rho0;
            rho_out[i] = d;
            p_out[i] = eos_pressure(d, rho0, p0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wall_shear_monitor: per-workgroup partial sums over the fluid-wall pairs, then one workgroup adds them up
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void g_wall_shear(Incidence I, const double *pos, const double *vel,
                                                       const double *wall_vel, const double *Vol, const double *B, double DH,
                                                       double mu, double h, double *part /*[2 * gridDim.x]*/)
{
    SPHX_ROW_THREAD();
    const size_t n = (size_t)I.nt;
    double lo = 0.0, hi = 0.0;
    if (live) {
        const Bmat Bi = load_B(B, i, n);
        const double vxi = vel[i], Vi = Vol[i];
        for (int m = m0 + sub; m < m1; m += kLanes) {
            const unsigned e = I.ent[m];
            if (e & 1u) continue;  // second particle of a fluid-fluid pair
            const Incident q = load_incident(I, e);
            if (!q.other_is_wall || q.r <= kRMin) continue;
            const int j = q.other;
            const double ebe = q.ex * (Bi.b11 * q.ex + Bi.b12 * q.ey) + q.ey * (Bi.b21 * q.ex + Bi.b22 * q.ey);
            const double f = 4.0 * mu * ebe * q.dW * Vol[j] * (vxi - wall_vel[j]) / (q.r + 0.01 * h) * Vi;
            const double yw = pos[j + n];
            if (yw <= 0.0) lo += f;
            else if (yw >= DH) hi += f;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo += __shfl_xor(lo, off);
        hi += __shfl_xor(hi, off);
    }
    __shared__ double s_lo[kBlock / 64], s_hi[kBlock / 64];
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) { a += s_lo[w]; b += s_hi[w]; }
        part[2 * (size_t)blockIdx.x] = a;
        part[2 * (size_t)blockIdx.x + 1] = b;
    }
}

__global__ __launch_bounds__(kScanTile) void g_shear_total(int n_part, const double *part, double *out)
{
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < n_part; k += kScanTile) { a += part[2 * (size_t)k]; b += part[2 * (size_t)k + 1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    __shared__ double sa[kScanTile / 64], sb[kScanTile / 64];
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0.0; b = 0.0;
        for (int w = 0; w < kScanTile / 64; ++w) { a += sa[w]; b += sb[w]; }
        out[0] = a;
        out[1] = b;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
inline unsigned blocks_for(long n) { return n > 0 ? div_up((size_t)n, kBlock) : 1; }

// the caller's pair list uploaded once per call + the incidence rows
class PairRows {
public:
    PairRows(size_t n_pairs, const double *h_first, const double *h_second, const double *h_dx, const double *h_dy,
             const double *h_r, const double *h_W, const double *h_dW, int nf, int nt)
    {
        const size_t m = n_pairs ? n_pairs : 1;
        auto up = [&](DevBuf<double> &b, const double *h) {
            b.alloc(m);
            if (n_pairs) b.upload(h, n_pairs);
        };
        up(first_, h_first); up(second_, h_second); up(dx_, h_dx); up(dy_, h_dy); up(r_, h_r);
        if (h_W) up(W_, h_W);
        up(dW_, h_dW);
        row_.alloc((size_t)nf + 1);
        I_ = Incidence{first_.get(), second_.get(), dx_.get(), dy_.get(), r_.get(), h_W ? W_.get() : nullptr, dW_.get(),
                       (long)n_pairs, row_.get(), nullptr, nf, nt};
        build();
    }
    const Incidence &view() const { return I_; }
    unsigned row_blocks() const { return blocks_for((long)I_.nf * kLanes); }

private:
    void build()
    {
        const int nf = I_.nf;
        const int n_tiles = (int)div_up((size_t)nf, kScanTile);
        DevBuf<int> deg((size_t)nf), tiles((size_t)n_tiles + 1);
        deg.zero();
        hipLaunchKernelGGL(inc_degree, dim3(blocks_for(I_.n_pairs)), dim3(kBlock), 0, 0, I_, deg.get());
        hipLaunchKernelGGL(inc_scan_tiles, dim3(n_tiles), dim3(kScanTile), 0, 0, (const int *)deg.get(), row_.get(), tiles.get(), nf);
        hipLaunchKernelGGL(inc_scan_totals, dim3(1), dim3(kScanTile), 0, 0, tiles.get(), n_tiles);
        hipLaunchKernelGGL(inc_scan_add, dim3(n_tiles), dim3(kScanTile), 0, 0, row_.get(), (const int *)tiles.get(), nf, n_tiles);
        const size_t cap = 2 * (size_t)(I_.n_pairs ? I_.n_pairs : 1);  // every pair appears in at most two rows
        DevBuf<unsigned> arrival(cap);
        ent_.alloc(cap);
        deg.zero();  // reused as the per-row fill cursor
        hipLaunchKernelGGL(inc_fill, dim3(blocks_for(I_.n_pairs)), dim3(kBlock), 0, 0, I_, (const int *)row_.get(), deg.get(), arrival.get());
        I_.ent = ent_.get();
        hipLaunchKernelGGL(inc_rank, dim3(row_blocks()), dim3(kBlock), 0, 0, I_, (const int *)row_.get(),
                           (const unsigned *)arrival.get(), ent_.get());
        SPHX_HIP(hipGetLastError());
        SPHX_HIP(hipDeviceSynchronize());  // the build's scratch arrays go back to the pool here
    }
    DevBuf<double> first_, second_, dx_, dy_, r_, W_, dW_;
    DevBuf<int> row_;
    DevBuf<unsigned> ent_;
    Incidence I_{};
};

// the modes on device pointers (column-major like the host arrays); composed by the entry points below
struct Modes {
    const PairRows &rows;
    int nf, nt;
    unsigned gp() const { return rows.row_blocks(); }
    unsigned ga() const { return blocks_for(nt); }
    unsigned gw() const { return blocks_for(nt - nf); }

    void density(const double *mass, double rho0, double h, double inv_sigma0, double *rho, double *Vol, double *B) const
    {
        const Incidence &I = rows.view();
        hipLaunchKernelGGL(g_wall_defaults, dim3(gw()), dim3(kBlock), 0, 0, nf, nt, mass, rho0, rho, Vol, B);
        hipLaunchKernelGGL(g_density, dim3(gp()), dim3(kBlock), 0, 0, I, mass, rho0, inv_sigma0, 10.0 / (7.0 * kPi * h * h), rho, Vol);
        hipLaunchKernelGGL(g_kgc, dim3(gp()), dim3(kBlock), 0, 0, I, (const double *)Vol, B);
    }
    void viscous(const double *vel, const double *Vol, const double *B, double mu, double h, const double *mass,
                 const double *wall_vel, double gravity_g, int add_gravity, double *force) const
    {
        hipLaunchKernelGGL(g_wall_rows2, dim3(gw()), dim3(kBlock), 0, 0, nf, nt, (const double *)nullptr, force);
        hipLaunchKernelGGL(g_viscous, dim3(gp()), dim3(kBlock), 0, 0, rows.view(), vel, Vol, B, wall_vel, mass, mu, h, gravity_g,
                           add_gravity, force);
    }
    void transport(const double *Vol, const double *B, const double *pos, double h, double coeff, double *pos_out) const
    {
        hipLaunchKernelGGL(g_wall_rows2, dim3(gw()), dim3(kBlock), 0, 0, nf, nt, pos, pos_out);
        hipLaunchKernelGGL(g_transport, dim3(gp()), dim3(kBlock), 0, 0, rows.view(), Vol, B, pos, h, coeff, pos_out);
    }
    void first_half(const double *Vol, const double *B, const double *rho, const double *mass, const double *pos,
                    const double *vel, const double *drho, const double *force_prior, double dt, double rho0, double p0,
                    double c_f, double *rho_h, double *p_h, double *pos_h, double *force, double *diss) const
    {
        hipLaunchKernelGGL(g_half_state, dim3(ga()), dim3(kBlock), 0, 0, nf, nt, rho, drho, pos, vel, dt, rho0, p0, rho_h, p_h, pos_h);
        hipLaunchKernelGGL(g_wall_rows2, dim3(gw()), dim3(kBlock), 0, 0, nf, nt, (const double *)nullptr, force);
        hipLaunchKernelGGL(g_wall_rows1, dim3(gw()), dim3(kBlock), 0, 0, nf, nt, (const double *)nullptr, diss);
        hipLaunchKernelGGL(g_pressure, dim3(gp()), dim3(kBlock), 0, 0, rows.view(), Vol, B, (const double *)rho_h,
                           (const double *)p_h, vel, force_prior, mass, rho0, c_f, force, diss);
    }
    // final: see g_continuity; rho_out / p_out may be null when final == 0
    void second_half(const double *Vol, const double *rho_h, const double *pos_h, const double *vel, double dt,
                     const double *wall_vel, double rho0, double p0, int final, double *pos_out, double *drho_out,
                     double *rho_out, double *p_out) const
    {
        hipLaunchKernelGGL(g_drift, dim3(ga()), dim3(kBlock), 0, 0, nf, nt, pos_h, vel, dt, pos_out);
        hipLaunchKernelGGL(g_wall_rows1, dim3(gw()), dim3(kBlock), 0, 0, nf, nt, (const double *)nullptr, drho_out);
        if (final) {
            hipLaunchKernelGGL(g_wall_rows1, dim3(gw()), dim3(kBlock), 0, 0, nf, nt, rho_h, rho_out);
            hipLaunchKernelGGL(g_wall_rows1, dim3(gw()), dim3(kBlock), 0, 0, nf, nt, (const double *)nullptr, p_out);
        }
        hipLaunchKernelGGL(g_continuity, dim3(gp()), dim3(kBlock), 0, 0, rows.view(), Vol, vel, wall_vel, rho_h, dt, rho0, p0,
                           final, drho_out, rho_out, p_out);
    }
    // integration_verlet: first half -> kick -> second half with the closing density half-step
    void verlet(const double *Vol, const double *B, const double *rho, const double *mass, const double *pos,
                const double *vel, const double *drho, const double *force_prior, double dt, double rho0, double p0,
                double c_f, const double *wall_vel, double *rho_out, double *p_out, double *pos_out, double *vel_out,
                double *drho_out, double *force_out) const
    {
        const size_t n = (size_t)nt;
        DevBuf<double> rho_h(n), p_h(n), pos_h(2 * n), diss(n);
        first_half(Vol, B, rho, mass, pos, vel, drho, force_prior, dt, rho0, p0, c_f, rho_h.get(), p_h.get(), pos_h.get(),
                   force_out, diss.get());
        hipLaunchKernelGGL(g_kick, dim3(ga()), dim3(kBlock), 0, 0, nf, nt, vel, force_prior, (const double *)force_out, mass, dt, vel_out);
        second_half(Vol, rho_h.get(), pos_h.get(), vel_out, dt, wall_vel, rho0, p0, 1, pos_out, drho_out, rho_out, p_out);
        SPHX_HIP(hipGetLastError());
        SPHX_HIP(hipDeviceSynchronize());  // the half-step arrays die here
    }
};

struct Up {  // host array -> device copy
    DevBuf<double> b;
    Up(const double *h, size_t n) : b(n ? n : 1) { if (n) b.upload(h, n); }
    const double *get() const { return b.get(); }
};

void check_counts(int nf, int nt, const char *id) { require(nf > 0 && nt >= nf, id, "Invalid n_fluid/n_total."); }
void check_pairs(size_t n_pairs, const char *id) { require(n_pairs <= (size_t)2147483647, id, "Pair count exceeds INT_MAX."); }

void finish()
{
    SPHX_HIP(hipGetLastError());
    SPHX_HIP(hipDeviceSynchronize());
}

}  // namespace
}  // namespace sphx

using namespace sphx;

SPHX_EXPORT int sphx_density_correction(size_t n_pairs, const double *pair_i, const double *pair_j, const double *dx,
                                        const double *dy, const double *r, const double *W, const double *dW,
                                        const double *mass, int n_fluid, int n_total, double rho0, double h,
                                        double inv_sigma0, double *rho, double *Vol, double *B)
{
    SPHX_TRY
    check_pairs(n_pairs, "SPH:Physics:density:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:density:count");
    require(rho0 > 0.0 && h > 0.0, "SPH:Physics:density:param", "rho0 and h must be positive.");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairRows rows(n_pairs, pair_i, pair_j, dx, dy, r, W, dW, n_fluid, n_total);
    Up d_mass(mass, nt);
    DevBuf<double> d_rho(nt), d_Vol(nt), d_B(4 * nt);
    Modes{rows, n_fluid, n_total}.density(d_mass.get(), rho0, h, inv_sigma0, d_rho.get(), d_Vol.get(), d_B.get());
    d_rho.download(rho, nt); d_Vol.download(Vol, nt); d_B.download(B, 4 * nt);
    finish();
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_viscous_force(size_t n_pairs, const double *pair_i, const double *pair_j, const double *dx,
                                   const double *dy, const double *r, const double *dW, const double *vel,
                                   const double *Vol, const double *B, double mu, double h, int n_fluid, int n_total,
                                   const double *mass, const double *wall_vel, double *force)
{
    SPHX_TRY
    check_pairs(n_pairs, "SPH:Physics:viscous:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:viscous:count");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairRows rows(n_pairs, pair_i, pair_j, dx, dy, r, nullptr, dW, n_fluid, n_total);
    Up d_vel(vel, 2 * nt), d_Vol(Vol, nt), d_B(B, 4 * nt), d_mass(mass, nt), d_wv(wall_vel, 2 * nt);
    DevBuf<double> d_force(2 * nt);
    Modes{rows, n_fluid, n_total}.viscous(d_vel.get(), d_Vol.get(), d_B.get(), mu, h, d_mass.get(), d_wv.get(), 0.0, 0, d_force.get());
    d_force.download(force, 2 * nt);
    finish();
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_transport_correction(size_t n_pairs, const double *pair_i, const double *pair_j, const double *dx,
                                          const double *dy, const double *r, const double *dW, const double *Vol,
                                          const double *B, const double *pos, double h, int n_fluid, int n_total,
                                          double transport_coeff, double *pos_out)
{
    SPHX_TRY
    check_pairs(n_pairs, "SPH:Physics:transport:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:transport:count");
    require(transport_coeff >= 0.0, "SPH:Physics:transport:coeff", "transport_coeff must be non-negative.");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairRows rows(n_pairs, pair_i, pair_j, dx, dy, r, nullptr, dW, n_fluid, n_total);
    Up d_Vol(Vol, nt), d_B(B, 4 * nt), d_pos(pos, 2 * nt);
    DevBuf<double> d_out(2 * nt);
    Modes{rows, n_fluid, n_total}.transport(d_Vol.get(), d_B.get(), d_pos.get(), h, transport_coeff, d_out.get());
    d_out.download(pos_out, 2 * nt);
    finish();
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_integration_1st(size_t n_pairs, const double *pair_i, const double *pair_j, const double *dx,
                                     const double *dy, const double *r, const double *dW, const double *Vol,
                                     const double *B, const double *rho, const double *mass, const double *pos,
                                     const double *vel, const double *drho_dt, const double *force_prior, double dt,
                                     int n_fluid, int n_total, double rho0, double p0, double c_f, const double *wall_vel,
                                     double *rho_out, double *p_out, double *pos_out, double *force_out, double *drho_out)
{
    SPHX_TRY
    (void)wall_vel;  // size-checked only by the gateway (sph_physics_mex.c:802); the mode does not read it
    check_pairs(n_pairs, "SPH:Physics:int1:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:int1:count");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairRows rows(n_pairs, pair_i, pair_j, dx, dy, r, nullptr, dW, n_fluid, n_total);
    Up d_Vol(Vol, nt), d_B(B, 4 * nt), d_rho(rho, nt), d_mass(mass, nt), d_pos(pos, 2 * nt), d_vel(vel, 2 * nt),
        d_drho(drho_dt, nt), d_fp(force_prior, 2 * nt);
    DevBuf<double> o_rho(nt), o_p(nt), o_pos(2 * nt), o_f(2 * nt), o_d(nt);
    Modes{rows, n_fluid, n_total}.first_half(d_Vol.get(), d_B.get(), d_rho.get(), d_mass.get(), d_pos.get(), d_vel.get(),
                                             d_drho.get(), d_fp.get(), dt, rho0, p0, c_f, o_rho.get(), o_p.get(), o_pos.get(),
                                             o_f.get(), o_d.get());
    o_rho.download(rho_out, nt); o_p.download(p_out, nt); o_pos.download(pos_out, 2 * nt);
    o_f.download(force_out, 2 * nt); o_d.download(drho_out, nt);
    finish();
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_integration_2nd(size_t n_pairs, const double *pair_i, const double *pair_j, const double *dx,
                                     const double *dy, const double *r, const double *dW, const double *Vol,
                                     const double *rho, const double *pos, const double *vel, double dt, int n_fluid,
                                     int n_total, const double *wall_vel, double *pos_out, double *drho_out,
                                     double *zeros_out)
{
    SPHX_TRY
    check_pairs(n_pairs, "SPH:Physics:int2:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:int2:count");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairRows rows(n_pairs, pair_i, pair_j, dx, dy, r, nullptr, dW, n_fluid, n_total);
    Up d_Vol(Vol, nt), d_rho(rho, nt), d_pos(pos, 2 * nt), d_vel(vel, 2 * nt), d_wv(wall_vel, 2 * nt);
    DevBuf<double> o_pos(2 * nt), o_d(nt);
    Modes{rows, n_fluid, n_total}.second_half(d_Vol.get(), d_rho.get(), d_pos.get(), d_vel.get(), dt, d_wv.get(), 0.0, 0.0, 0,
                                              o_pos.get(), o_d.get(), nullptr, nullptr);
    o_pos.download(pos_out, 2 * nt); o_d.download(drho_out, nt);
    finish();
    if (zeros_out) std::memset(zeros_out, 0, 2 * nt * sizeof(double));  // the mode's third output (sph_physics_mex.c:1064)
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_integration_verlet(size_t n_pairs, const double *pair_i, const double *pair_j, const double *dx,
                                        const double *dy, const double *r, const double *dW, const double *Vol,
                                        const double *B, const double *rho, const double *mass, const double *pos,
                                        const double *vel, const double *drho_dt, const double *force_prior, double dt,
                                        int n_fluid, int n_total, double rho0, double p0, double c_f,
                                        const double *wall_vel, double *rho_out, double *p_out, double *pos_out,
                                        double *vel_out, double *drho_out, double *force_out)
{
    SPHX_TRY
    check_pairs(n_pairs, "SPH:Physics:verlet:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:verlet:count");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairRows rows(n_pairs, pair_i, pair_j, dx, dy, r, nullptr, dW, n_fluid, n_total);
    Up d_Vol(Vol, nt), d_B(B, 4 * nt), d_rho(rho, nt), d_mass(mass, nt), d_pos(pos, 2 * nt), d_vel(vel, 2 * nt),
        d_drho(drho_dt, nt), d_fp(force_prior, 2 * nt), d_wv(wall_vel, 2 * nt);
    DevBuf<double> o_rho(nt), o_p(nt), o_pos(2 * nt), o_vel(2 * nt), o_d(nt), o_f(2 * nt);
    Modes{rows, n_fluid, n_total}.verlet(d_Vol.get(), d_B.get(), d_rho.get(), d_mass.get(), d_pos.get(), d_vel.get(),
                                         d_drho.get(), d_fp.get(), dt, rho0, p0, c_f, d_wv.get(), o_rho.get(), o_p.get(),
                                         o_pos.get(), o_vel.get(), o_d.get(), o_f.get());
    o_rho.download(rho_out, nt); o_p.download(p_out, nt); o_pos.download(pos_out, 2 * nt);
    o_vel.download(vel_out, 2 * nt); o_d.download(drho_out, nt); o_f.download(force_out, 2 * nt);
    finish();
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_advance_shell_step(size_t n_pairs, const double *pair_i, const double *pair_j, const double *dx,
                                        const double *dy, const double *r, const double *W, const double *dW,
                                        const double *mass, const double *pos, const double *vel, const double *wall_vel,
                                        const double *rho, const double *drho_dt, double dt, int n_fluid, int n_total,
                                        double rho0, double p0, double c_f, double mu, double h, double inv_sigma0,
                                        double gravity_g, double *rho_out, double *p_out, double *pos_out, double *vel_out,
                                        double *drho_out, double *force_out, double *force_prior_out, double *Vol_out,
                                        double *B_out)
{
    SPHX_TRY
    (void)rho;  // size-checked only in the reference (sph_physics_mex.c:1532); density is re-summed
    check_pairs(n_pairs, "SPH:Physics:advance:pairsize");
    check_counts(n_fluid, n_total, "SPH:Physics:advance:count");
    require(rho0 > 0.0 && h > 0.0, "SPH:Physics:density:param", "rho0 and h must be positive.");
    ensure_device();
    const size_t nt = (size_t)n_total;
    PairRows rows(n_pairs, pair_i, pair_j, dx, dy, r, W, dW, n_fluid, n_total);
    Up d_mass(mass, nt), d_pos(pos, 2 * nt), d_vel(vel, 2 * nt), d_wv(wall_vel, 2 * nt), d_drho(drho_dt, nt);
    DevBuf<double> rho_d(nt), Vol(nt), B(4 * nt), fp(2 * nt), pos_t(2 * nt);
    DevBuf<double> o_rho(nt), o_p(nt), o_pos(2 * nt), o_vel(2 * nt), o_d(nt), o_f(2 * nt);
    const Modes md{rows, n_fluid, n_total};
    // density -> viscous (+ mass*g on x, :1575-1580) -> transport with the 13-argument default 0.2 (:584,:1596) -> verlet
    md.density(d_mass.get(), rho0, h, inv_sigma0, rho_d.get(), Vol.get(), B.get());
    md.viscous(d_vel.get(), Vol.get(), B.get(), mu, h, d_mass.get(), d_wv.get(), gravity_g, 1, fp.get());
    md.transport(Vol.get(), B.get(), d_pos.get(), h, 0.2, pos_t.get());
    md.verlet(Vol.get(), B.get(), rho_d.get(), d_mass.get(), pos_t.get(), d_vel.get(), d_drho.get(), fp.get(), dt, rho0, p0,
              c_f, d_wv.get(), o_rho.get(), o_p.get(), o_pos.get(), o_vel.get(), o_d.get(), o_f.get());
    o_rho.download(rho_out, nt); o_p.download(p_out, nt); o_pos.download(pos_out, 2 * nt);
    o_vel.download(vel_out, 2 * nt); o_d.download(drho_out, nt); o_f.download(force_out, 2 * nt);
    fp.download(force_prior_out, 2 * nt); Vol.download(Vol_out, nt); B.download(B_out, 4 * nt);
    finish();
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_wall_shear_monitor(size_t n_pairs, const double *pair_i, const double *pair_j, const double *dx,
                                        const double *dy, const double *r, const double *dW, const double *pos,
                                        const double *vel, const double *wall_vel, const double *Vol, const double *B,
                                        int n_fluid, int n_total, double DL, double DH, double mu, double h,
                                        double *tau_bottom, double *tau_top)
{
    SPHX_TRY
    check_pairs(n_pairs, "SPH:Physics:wallshear:pairsize");
    require(DL > 0.0 && h > 0.0, "SPH:Physics:wallshear:param", "DL and h must be positive.");
    require(n_total > 0 && n_fluid >= 0 && n_fluid <= n_total, "SPH:Physics:wallshear:count", "Invalid n_fluid/n_total.");
    double sums[2] = {0.0, 0.0};
    if (n_fluid > 0) {
        ensure_device();
        const size_t nt = (size_t)n_total;
        PairRows rows(n_pairs, pair_i, pair_j, dx, dy, r, nullptr, dW, n_fluid, n_total);
        Up d_pos(pos, 2 * nt), d_vel(vel, 2 * nt), d_wv(wall_vel, 2 * nt), d_Vol(Vol, nt), d_B(B, 4 * nt);
        const unsigned nb = rows.row_blocks();
        DevBuf<double> part(2 * (size_t)nb), total(2);
        hipLaunchKernelGGL(g_wall_shear, dim3(nb), dim3(kBlock), 0, 0, rows.view(), d_pos.get(), d_vel.get(), d_wv.get(),
                           d_Vol.get(), d_B.get(), DH, mu, h, part.get());
        hipLaunchKernelGGL(g_shear_total, dim3(1), dim3(kScanTile), 0, 0, (int)nb, (const double *)part.get(), total.get());
        total.download(sums, 2);
        finish();
    }
    *tau_bottom = -sums[0] / DL;  // sph_physics_mex.c:1741-1742
    *tau_top = -sums[1] / DL;
    return SPHX_OK;
    SPHX_CATCH
}
